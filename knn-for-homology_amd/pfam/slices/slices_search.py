"""Drop-in for the reference's ``pfam/slices/slices_search.py``: exhaustive cosine
self-search (k = 1000) over the ``slices`` and ``full_sequences`` embedding sets.

pfam/slices/slices_search.py:9-31: a set whose ``<set>_scores.npy`` already exists is
skipped; embeddings are cast to float32 and normalised in place; only the search call
is timed and the elapsed seconds are printed; ``<set>_scores.npy`` / ``<set>_hits.npy``
are written with ``numpy.save``.  (The reference notes 2540 s for this on one CPU core.)
"""
from pathlib import Path
from time import time

import numpy

from ... import faiss, ranks
from ...paths import slices_data as _default_dir

K = 1000
SETS = ("slices", "full_sequences")


def self_search(vectors, k):
    """Normalises ``vectors`` in place, indexes them and searches them against themselves.
    Returns (scores, hits, seconds spent in the search call alone)."""
    ranks.launched_group()  # (multi-rank launch: this rank's GPU is chosen before the first device call)
    faiss.normalize_L2(vectors)
    flat = ranks.flat_index(vectors.shape[1], faiss.METRIC_INNER_PRODUCT)  # (multi-rank launch: replicated rows, split queries)
    flat.train(vectors)
    flat.add(vectors)
    began = time()
    scores, hits = flat.search_self(k)  # the rows just added are the queries (reference: index.search(embeddings, 1000))
    return scores, hits, time() - began


def main(data_dir=None, k=K):
    folder = _default_dir() if data_dir is None else Path(data_dir)
    todo = ranks.same_everywhere([name for name in SETS if not (folder / f"{name}_scores.npy").is_file()])
    for name in todo:
        vectors = numpy.load(folder / f"{name}.npy").astype(numpy.float32)
        scores, hits, seconds = self_search(vectors, k)
        if ranks.writer():
            print(name, vectors.shape)
            print(seconds)
            numpy.save(folder / f"{name}_scores.npy", scores)
            numpy.save(folder / f"{name}_hits.npy", hits)
        ranks.barrier()


if __name__ == "__main__":
    main()
