"""Drop-in for the reference's ``cath/search.py`` (all-vs-all CATH20 search).

``search`` keeps the reference contract (cath/search.py:13-26):
  * one extra neighbour is requested and column 0 (the self hit) is dropped,
  * the return order is ``(hits int64 [N,hits], scores float32 [N,hits])`` -- ids first,
    i.e. swapped relative to ``index.search``,
  * for the inner-product metric the rows are L2-normalised, not the caller's array (the
    reference normalises a copy, here the rows are normalised after the upload); for L2
    nothing is normalised.
``search_and_save`` keeps the file protocol (cath/search.py:29-53): for "Cosine" and
"Euclidean", every ``*.npy`` in the data directory (fp16 files are cast to fp32) is
searched, ``<stem>.<metric>-search-time.txt`` records the wall time of
copy+normalise+add+search, and ``hits_<metric>.npz`` / ``scores_<metric>.npz`` hold one
array per file stem.
"""
import time
from pathlib import Path
from typing import Optional, Tuple

import numpy
from numpy import ndarray

from .. import faiss, ranks
from ..paths import cath_data as _default_cath_data

_METRICS = (("Cosine", faiss.METRIC_INNER_PRODUCT), ("Euclidean", faiss.METRIC_L2))


def search(embeddings: ndarray, hits: int = 10, metric=faiss.METRIC_INNER_PRODUCT) -> Tuple[ndarray, ndarray]:
    # One upload: the rows are added as they are, normalised in HBM (the reference normalises
    # a private copy on the host, so the caller's array is left alone either way) and then
    # serve as their own queries.  Same bits as normalize_L2 + add + search on the host arrays
    # (tests/test_flat_gpu.py::test_search_self_equals_host_path).
    # (under a torch.distributed launch: every rank holds the rows and answers its slice of them, see ranks.py)
    index = ranks.flat_index(embeddings.shape[1], metric)
    if metric == faiss.METRIC_INNER_PRODUCT:
        index.add(numpy.ascontiguousarray(embeddings))  # the reference's .copy() also makes it contiguous
        index.normalize_rows()
    else:
        index.add(embeddings)
    scores, neighbours = index.search_self(hits + 1)
    return neighbours[:, 1:], scores[:, 1:]


def search_and_save(cath_data: Optional[Path] = None):
    data_dir = Path(cath_data) if cath_data is not None else _default_cath_data()
    say = print if ranks.writer() else (lambda *a, **k: None)  # (multi-rank launch: rank 0 prints and writes)
    for label, metric in _METRICS:
        say(f"Searching with {label}")
        all_hits, all_scores = {}, {}
        for npy in sorted(data_dir.glob("*.npy")):
            embeddings = numpy.load(npy).astype(numpy.float32)
            say(npy.stem, embeddings.shape)
            t0 = time.time()
            all_hits[npy.stem], all_scores[npy.stem] = search(embeddings, metric=metric)
            elapsed = time.time() - t0
            say(elapsed)
            if ranks.writer():
                npy.with_suffix(f".{label.lower()}-search-time.txt").write_text(str(elapsed))
        if ranks.writer():
            numpy.savez(data_dir / f"hits_{label.lower()}.npz", **all_hits)
            numpy.savez(data_dir / f"scores_{label.lower()}.npz", **all_scores)
        ranks.barrier()  # (the files exist when any rank returns)


if __name__ == "__main__":
    search_and_save()
