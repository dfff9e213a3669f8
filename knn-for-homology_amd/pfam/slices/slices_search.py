"""Drop-in for the reference's ``pfam/slices/slices_search.py``: exhaustive cosine
self-search (k = 1000) over the ``slices`` and ``full_sequences`` embedding sets.

pfam/slices/slices_search.py:9-31: a set whose ``<set>_scores.npy`` already exists is
skipped; embeddings are cast to float32 and normalised in place; only the search call
is timed and the elapsed seconds are printed; ``<set>_scores.npy`` / ``<set>_hits.npy``
are written with ``numpy.save``.  (The reference notes 2540 s for this on one CPU core.)
"""
from pathlib import Path
from time import time
from typing import Optional

import numpy

from ... import faiss
from ...paths import slices_data as _default_dir

K = 1000


def main(data_dir: Optional[Path] = None, k: int = K):
    data_dir = Path(data_dir) if data_dir is not None else _default_dir()
    for sequence_set in ("slices", "full_sequences"):
        scores_file = data_dir / f"{sequence_set}_scores.npy"
        if scores_file.is_file():
            continue
        embeddings = numpy.load(data_dir / f"{sequence_set}.npy").astype(numpy.float32)
        print(sequence_set, embeddings.shape)
        faiss.normalize_L2(embeddings)
        index = faiss.IndexFlat(embeddings.shape[1], faiss.METRIC_INNER_PRODUCT)
        index.train(embeddings)
        index.add(embeddings)
        t0 = time()
        scores, hits = index.search(embeddings, k)
        print(time() - t0)
        numpy.save(scores_file, scores)
        numpy.save(data_dir / f"{sequence_set}_hits.npy", hits)


if __name__ == "__main__":
    main()
