"""Drop-in for the reference's ``pfam/search.py`` (train/test Pfam-domain search).

pfam/search.py:14-21  load_embeddings: train.npy / test.npy -> float32, L2-normalised
pfam/search.py:24-39  search_index: LSH-1024 index cached as ``index_lsh_1024.bin``,
                      k = 1000 -> ``index_scores.npy`` / ``index_hits.npy``
pfam/search.py:42-53  search_flat: IndexFlat inner product, k = 1000 ->
                      ``flat_scores.npy`` / ``flat_hits.npy``
pfam/search.py:56-61  main: both searches for subset10_t5 and subset10
"""
from pathlib import Path

import numpy

from .. import faiss, ranks
from ..paths import subset10, subset10_t5

K = 1000
LSH_BITS = 1024


def _normalised(npy: Path):
    ranks.launched_group()  # (multi-rank launch: this rank's GPU is chosen before the first device call)
    vectors = numpy.load(npy).astype(numpy.float32)
    faiss.normalize_L2(vectors)
    return vectors


def load_embeddings(embedding_set: Path):
    """(train, test), float32 and L2-normalised."""
    folder = Path(embedding_set)
    return _normalised(folder / "train.npy"), _normalised(folder / "test.npy")


def _filled(index, rows):
    index.train(rows)
    index.add(rows)
    return index


def _search_and_store(folder: Path, stem: str, index, queries, k: int, sync: bool = True):
    """``<stem>_scores.npy`` / ``<stem>_hits.npy`` in the embedding set's directory."""
    scores, hits = index.search(queries, k)
    if ranks.writer():  # (multi-rank launch: every rank has the gathered arrays, rank 0 writes them)
        for suffix, array in (("scores", scores), ("hits", hits)):
            numpy.save(folder / f"{stem}_{suffix}.npy", array)
    if sync:
        ranks.barrier()


def search_index(embedding_set: Path, k: int = K):
    # (LSH does not shard: rank 0 of a multi-rank launch runs it alone, the others wait for its outcome without loading anything)
    ranks.rank0_only(lambda: _search_index(embedding_set, k))


def _search_index(embedding_set: Path, k: int):
    folder = Path(embedding_set)
    train, test = load_embeddings(folder)
    cached = folder / f"index_lsh_{LSH_BITS}.bin"
    if cached.is_file():
        lsh = faiss.read_index(str(cached))
    else:
        lsh = _filled(faiss.IndexLSH(train.shape[1], LSH_BITS), train)
        faiss.write_index(lsh, str(cached))
    _search_and_store(folder, "index", lsh, test, k, sync=False)


def search_flat(embedding_set: Path, k: int = K):
    folder = Path(embedding_set)
    train, test = load_embeddings(folder)
    flat = _filled(ranks.flat_index(train.shape[1], faiss.METRIC_INNER_PRODUCT), train)  # (multi-rank launch: the test rows are split over the ranks)
    _search_and_store(folder, "flat", flat, test, k)


def main():
    for embedding_set in (subset10_t5(), subset10()):
        for label, run in (("index", search_index), ("flat", search_flat)):
            if ranks.writer():
                print(embedding_set, label)
            run(embedding_set)


if __name__ == "__main__":
    main()
