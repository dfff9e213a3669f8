"""Drop-in for the reference's ``pfam/search.py`` (train/test Pfam-domain search).

pfam/search.py:14-21  load_embeddings: train.npy / test.npy -> float32, L2-normalised
pfam/search.py:24-39  search_index: LSH-1024 index cached as ``index_lsh_1024.bin``,
                      k = 1000 -> ``index_scores.npy`` / ``index_hits.npy``
pfam/search.py:42-53  search_flat: IndexFlat inner product, k = 1000 ->
                      ``flat_scores.npy`` / ``flat_hits.npy``
pfam/search.py:56-61  main: both searches for subset10_t5 and subset10
"""
from pathlib import Path
from typing import Tuple

import numpy
from numpy import ndarray

from .. import faiss
from ..paths import subset10, subset10_t5

K = 1000
LSH_BITS = 1024


def load_embeddings(embedding_set: Path) -> Tuple[ndarray, ndarray]:
    out = []
    for name in ("train.npy", "test.npy"):
        x = numpy.load(Path(embedding_set) / name).astype(numpy.float32)
        faiss.normalize_L2(x)
        out.append(x)
    return out[0], out[1]


def search_index(embedding_set: Path, k: int = K):
    embedding_set = Path(embedding_set)
    train, test = load_embeddings(embedding_set)
    cache = embedding_set / f"index_lsh_{LSH_BITS}.bin"
    if cache.is_file():
        lsh_index = faiss.read_index(str(cache))
    else:
        lsh_index = faiss.IndexLSH(train.shape[1], LSH_BITS)
        lsh_index.train(train)
        lsh_index.add(train)
        faiss.write_index(lsh_index, str(cache))
    scores, hits = lsh_index.search(test, k)
    numpy.save(embedding_set / "index_scores.npy", scores)
    numpy.save(embedding_set / "index_hits.npy", hits)


def search_flat(embedding_set: Path, k: int = K):
    embedding_set = Path(embedding_set)
    train, test = load_embeddings(embedding_set)
    index = faiss.IndexFlat(train.shape[1], faiss.METRIC_INNER_PRODUCT)
    index.train(train)
    index.add(train)
    scores, hits = index.search(test, k)
    numpy.save(embedding_set / "flat_scores.npy", scores)
    numpy.save(embedding_set / "flat_hits.npy", hits)


def main():
    for embedding_set in (subset10_t5(), subset10()):
        print(embedding_set, "index")
        search_index(embedding_set)
        print(embedding_set, "flat")
        search_flat(embedding_set)


if __name__ == "__main__":
    main()
