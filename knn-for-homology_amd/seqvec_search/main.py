"""Drop-in for the kNN part of the reference's ``seqvec_search/main.py``.

``faiss_search`` (seqvec_search/main.py:22-50) keeps every observable behaviour:
  * with the inner-product metric BOTH ``queries`` and an ndarray ``haystack`` are
    L2-normalised IN PLACE (the caller's arrays change),
  * ``haystack`` may also be a prebuilt index (``read_index`` result), used as is,
  * returns ``(result int64 [nq,hits], scores float32 [nq,hits], search_time seconds)``,
  * logs "Preprocessing took ..." and "Searching ... took ...".
``evaluate_faiss`` / ``evaluate`` (seqvec_search/main.py:53-82) are the AUC1 / TP
metrics the reference's tests assert on (tests/test_main.py:10-27); they are host-side
bookkeeping over the returned ids and are restated here so the parity tests read like
the reference's own.
"""
import argparse
import logging
import time
from collections import Counter
from pathlib import Path
from typing import Iterable, List, Tuple, Union

import numpy
from numpy import ndarray

from .. import faiss, ranks
from .constants import default_hits
from .data import LoadedData

logger = logging.getLogger(__name__)


def faiss_search(haystack: Union[ndarray, "faiss.Index"], queries: ndarray, hits: int = default_hits,
                 metric=faiss.METRIC_INNER_PRODUCT) -> Tuple[ndarray, ndarray, float]:
    """Searches the haystack for queries and returns the specified number of hits for each."""
    t0 = time.time()
    ranks.launched_group()  # (multi-rank launch: this rank's GPU is chosen before the first device call)
    cosine = metric == faiss.METRIC_INNER_PRODUCT
    if cosine:
        faiss.normalize_L2(queries)
    if isinstance(haystack, ndarray):
        if cosine:
            faiss.normalize_L2(haystack)
        index = ranks.flat_index(haystack.shape[1], metric)  # (one GPU: faiss.IndexFlat; multi-rank launch: the queries are split, ranks.py)
        index.train(haystack)
        index.add(haystack)
    else:
        index = haystack
    logging.info(f"Preprocessing took {time.time() - t0}s")
    t0 = time.time()
    scores, result = index.search(queries, hits)
    search_time = time.time() - t0
    logging.info(f"Searching {len(queries)} samples and {hits} hits took {search_time}s")
    return result, scores, search_time


def evaluate_faiss(data: LoadedData, results: ndarray) -> Tuple[List[float], List[float]]:
    """Maps neighbour row numbers to string ids, then scores them with ``evaluate``."""
    named = ((data.test_ids[qi], [data.train_ids[j] for j in row]) for qi, row in enumerate(results))
    return evaluate(data, named)


def evaluate(data: LoadedData, results: Iterable[Tuple[str, Iterable[str]]]) -> Tuple[List[float], List[float]]:
    """AUC1 (hits of the query's family before the first foreign hit) and TP (hits of
    the query's family anywhere in the list), each divided by the family's size in the
    training set."""
    family_of = data.ids_to_family
    family_size = Counter(family_of[i] for i in data.train_ids)
    auc1s, tps = [], []
    for query, matches in results:
        matches = list(matches)
        own = family_of[query]
        same = [family_of[m] == own for m in matches]
        leading = same.index(False) if False in same else len(same)
        auc1s.append(leading / family_size[own])
        tps.append(sum(same) / family_size[own])
    return auc1s, tps


def main(argv=None):
    """kNN leg of the reference CLI (seqvec_search/main.py:115-143): dataset directory,
    optional ``--knn-index`` file, ``--hits``.  The MMseqs2 alignment and figure legs of
    the reference need the external ``mmseqs`` binary and are out of scope."""
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
    parser = argparse.ArgumentParser(description="Nearest neighbour search over per-protein embeddings on an MI355X")
    parser.add_argument("dataset", type=Path)
    parser.add_argument("--knn-index", type=Path)
    parser.add_argument("--hits", type=int, default=default_hits)
    args = parser.parse_args(argv)
    data = LoadedData.from_options(args.dataset, args.hits, args.knn_index)
    queries = numpy.load(str(data.test))
    knn_index = faiss.read_index(str(args.knn_index)) if args.knn_index else numpy.load(str(data.train))
    results, scores, search_time = faiss_search(knn_index, queries, data.hits)
    auc1s, tps = evaluate_faiss(data, results)
    logger.info(f"Mean AUC1 for k-NN: {numpy.mean(auc1s):f}, Mean TP: {numpy.mean(tps):f}, Time: {int(search_time)}s")
    return results, scores, auc1s, tps


if __name__ == "__main__":
    main()
