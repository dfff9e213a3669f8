"""Drop-in for the reference's ``pfam/proteins_search.py`` (all-vs-all search over the
Pfam full-sequence embeddings, k = 1000).

Which index for which size (one MI355X, d = 1024): at this script's size -- 200 k rows -- the EXACT ``flat``
mode is also the fastest: 0.42 s for the all-vs-all k = 1000 search against 0.88 s for ``hnsw`` (M = 42, efSearch 256:
recall@300 0.95, recall@1000 0.81 -- a sequential CPU construction with the same graph parameters reaches 0.88 / 0.74 on
these rows, tests/golden/hnsw_refshape_200k.npz) and 0.21 s for ``lsh``.  HNSW pays from about 2 M rows on (10 M rows, k = 100:
440-590 k queries/s at recall 0.984-0.988 against 6.6 k for the exact scan); a graph search of fewer than ~1000 queries is
latency-bound (1.2 ms for one query against 0.30 ms for the exact scan of 200 k rows).

Protocol kept from pfam/proteins_search.py:11-57:
  argv[1] in {flat, lsh, hnsw} (anything else: ValueError(argv[1]));
  ``full_sequences.npy`` is loaded, cast to float32 and L2-normalised in place;
  flat  -> IndexFlat(d, METRIC_INNER_PRODUCT)
  lsh   -> IndexLSH(d, 2048)
  hnsw  -> IndexHNSWFlat(d, 42, METRIC_INNER_PRODUCT) with hnsw.efSearch = 256;
  train + add, "Index creation took ..s", the index is written to
  ``full_sequences_<mode>.index`` and the size line is printed, then a self-search with
  k = 1000 is timed ("Search took ..s") and ``full_sequences_<mode>_{scores,hits}.npy``
  are saved (float32 / int64, C order, plain ``numpy.save``).
"""
import sys
from pathlib import Path
from time import time
from typing import Optional, Sequence

import numpy

from .. import faiss, ranks
from ..paths import full_sequences_data as _default_dir

K = 1000
HNSW_M = 42
HNSW_EF_SEARCH = 256
LSH_BITS = 2048


def naturalsize(nbytes: int) -> str:
    """Decimal size string in the style the reference prints via ``humanize.naturalsize``
    (pfam/proteins_search.py:41-45): 819200128 -> '819.2 MB'."""
    sign = "-" if nbytes < 0 else ""
    n = abs(int(nbytes))
    if n == 1:
        return f"{sign}1 Byte"
    if n < 1000:
        return f"{sign}{n} Bytes"
    value = float(n)
    for unit in ("kB", "MB", "GB", "TB", "PB", "EB", "ZB", "YB"):
        value /= 1000.0
        if value < 1000.0 or unit == "YB":
            return f"{sign}{value:.1f} {unit}"


def build_index(index_mode: str, d: int):
    if index_mode == "flat":
        return ranks.flat_index(d, faiss.METRIC_INNER_PRODUCT)  # (one GPU: faiss.IndexFlat; multi-rank launch: query-sharded)
    if index_mode == "lsh":
        return faiss.IndexLSH(d, LSH_BITS)
    if index_mode == "hnsw":
        index = faiss.IndexHNSWFlat(d, HNSW_M, faiss.METRIC_INNER_PRODUCT)
        index.hnsw.efSearch = HNSW_EF_SEARCH
        return index
    raise ValueError(index_mode)


def run(embeddings: numpy.ndarray, index_mode: str, data_dir: Path, npy_size: int, k: int = K):
    """pfam/proteins_search.py:21-57 on an array already loaded and cast: normalises ``embeddings`` IN PLACE
    (:22 ``faiss.normalize_L2(embeddings)``), builds / writes the index, searches, saves.  ``embeddings`` may be a
    callable that loads the array: the ranks that only wait (HNSW / LSH under a multi-rank launch) never call it."""
    if index_mode != "flat" and ranks.launched_group()[1] > 1:  # (also: this rank's GPU is chosen before the first device call)
        # HNSW and LSH do not shard (replicas only): under a multi-rank launch rank 0 runs them on its
        # GPU, the other ranks wait for its files
        # (ranks.rank0_only: they wait for its outcome -- an error on rank 0 raises everywhere)
        return ranks.rank0_only(lambda: _run(embeddings() if callable(embeddings) else embeddings, index_mode, data_dir, npy_size, k, print, lambda: None))
    if callable(embeddings):
        embeddings = embeddings()
    return _run(embeddings, index_mode, data_dir, npy_size, k, print if ranks.writer() else (lambda *a, **kw: None), ranks.barrier)


def _run(embeddings, index_mode, data_dir, npy_size, k, say, barrier):
    started = time()
    index = build_index(index_mode, embeddings.shape[1])
    if index_mode == "flat":
        # one upload: the raw rows go to the GPU, are normalised there (same kernel, same bits as
        # faiss.normalize_L2) and come back into the caller's array -- :22's in-place contract with two
        # PCIe crossings instead of normalise (up + down) + add (up)
        index.train(embeddings)
        index.add(embeddings)
        index.normalize_rows()
        index.reconstruct_into(embeddings)
    else:
        faiss.normalize_L2(embeddings)
        index.train(embeddings)
        index.add(embeddings)
    say(f"Index creation took {int(time() - started)}s")
    index_file = data_dir / f"full_sequences_{index_mode}.index"
    if ranks.writer():
        if index_mode in ("flat", "hnsw"):
            faiss.write_index(getattr(index, "replica", index), str(index_file), rows=embeddings)  # (the rows are still here: no second download)
        else:
            faiss.write_index(index, str(index_file))
    barrier()
    index_size = index_file.stat().st_size
    say(f"Embeddings: {naturalsize(npy_size)} Index: {naturalsize(index_size)} "
          f"Difference: {naturalsize(index_size - npy_size)}")

    started = time()
    if index_mode == "flat":
        # the queries are the rows just added (pfam/proteins_search.py:37,49): no second upload
        scores, hits = index.search_self(k)
    else:
        scores, hits = index.search(embeddings, k)
    say(f"Search took {int(time() - started)}s")
    if ranks.writer():
        numpy.save(data_dir / f"full_sequences_{index_mode}_scores.npy", scores)
        numpy.save(data_dir / f"full_sequences_{index_mode}_hits.npy", hits)
    barrier()
    return scores, hits


def main(argv: Optional[Sequence[str]] = None, data_dir: Optional[Path] = None, k: int = K):
    argv = sys.argv if argv is None else argv
    index_mode = argv[1]
    data_dir = Path(data_dir) if data_dir is not None else _default_dir()
    npy = data_dir / "full_sequences.npy"

    def load():
        embeddings = numpy.load(npy).astype(numpy.float32)
        if ranks.writer():
            print("full_sequences", embeddings.shape)
        return embeddings

    # (HNSW / LSH under a multi-rank launch: the ranks that only wait for rank 0 do not load 0.8 GB to sit beside it)
    waits = index_mode in ("lsh", "hnsw") and ranks.launched_group()[1] > 1 and not ranks.writer()
    run(load if waits else load(), index_mode, data_dir, npy.stat().st_size, k)


if __name__ == "__main__":
    main()
