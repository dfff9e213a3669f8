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

import numpy

from .. import faiss

logger = logging.getLogger(__name__)

# option -> argparse keywords (same names, defaults and help texts as the reference's parser)
_OPTIONS = {
    "--dir": dict(type=Path, default=Path(), help="The name of the directory containing the database"),
    "--index": dict(type=Path, required=True, help="The location to write the index to"),
    "--param": dict(type=int, default=1024, help="The tuning parameter of the index. Higher means higher precision"),
}


def parse_options(argv=None):
    parser = argparse.ArgumentParser()
    for flag, spec in _OPTIONS.items():
        parser.add_argument(flag, **spec)
    return parser.parse_args(argv)


def build_lsh_index(database, nbits):
    """IndexLSH over ``database`` (any float dtype: faiss wants float32), trained and filled."""
    vectors = numpy.ascontiguousarray(database, dtype=numpy.float32)
    logger.info(f"Training LSH index with {nbits} bits on {vectors.shape}")
    index = faiss.IndexLSH(vectors.shape[1], nbits)
    index.train(vectors)
    index.add(vectors)
    return index


def main(args=None):
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
    options = parse_options(args)
    source = options.dir / "train.npy"
    logger.info(f"Loading database from {source}")
    index = build_lsh_index(numpy.load(str(source)), options.param)
    logger.info("Writing out the LSH index")
    faiss.write_index(index, str(options.index))


if __name__ == "__main__":
    main()
