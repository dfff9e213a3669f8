"""Drop-in for the reference's ``seqvec_search/create_index.py`` (console script
``seqvec_search_create_index``, pyproject.toml:30).

seqvec_search/create_index.py:16-47: ``--dir`` (directory holding ``train.npy``, default
"."), ``--index`` (output file, required), ``--param`` (LSH bits, default 1024); builds
``IndexLSH(d, param)``, trains, adds, and writes the index file.  The reference's test
only asserts that the file exists (tests/test_utils.py:17-21).
"""
import argparse
import logging
from pathlib import Path
from typing import Optional, Sequence

import numpy

from .. import faiss

logger = logging.getLogger(__name__)


def _parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser()
    p.add_argument("--dir", type=Path, default=Path(), help="The name of the directory containing the database")
    p.add_argument("--index", type=Path, required=True, help="The location to write the index to")
    p.add_argument("--param", type=int, default=1024,
                   help="The tuning parameter of the index. Higher means higher precision")
    return p


def main(args: Optional[Sequence[str]] = None):
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
    opts = _parser().parse_args(args)
    train_file = opts.dir / "train.npy"
    logger.info(f"Loading database from {train_file}")
    embeddings = numpy.load(str(train_file))
    if embeddings.dtype != numpy.float32:
        embeddings = embeddings.astype(numpy.float32)
    logger.info(f"Training LSH index with {opts.param} bits on {embeddings.shape}")
    lsh_index = faiss.IndexLSH(embeddings.shape[1], opts.param)
    lsh_index.train(embeddings)
    lsh_index.add(embeddings)
    logger.info("Writing out the LSH index")
    faiss.write_index(lsh_index, str(opts.index))


if __name__ == "__main__":
    main()
