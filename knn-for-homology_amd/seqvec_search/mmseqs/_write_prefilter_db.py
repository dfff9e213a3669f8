"""Drop-in for ``seqvec_search/mmseqs/_write_prefilter_db.py``: turns kNN hits into an
MMseqs2 prefilter database (the reference's Python double loop over nq x k, lines 79-97, is
the next bottleneck after the search at k = 300, N = 200k).

Format (reference lines 66-97): ``<db>.dbtype`` = 07 00 00 00; ``<db>.0`` holds per query the
lines ``<target mmseqs id>\\t<int(score*100)>\\t0\\n`` (hits equal to -1 skipped) followed by a
NUL byte; ``<db>.index`` holds ``<query mmseqs id>\\t<offset>\\t<length>\\n`` per query.
Formatting runs natively (libknn355, OpenMP) and writes the same bytes -- the bytes the reference
writes under the caller's numpy: ``numpy.clip(scores, -(10**30), 10**30) * 100`` (line 75) is
evaluated in double by numpy 1.x (the reference pins 1.22.2) and in float32 by numpy >= 2.
"""
import logging
from pathlib import Path
from typing import Dict, List

import numpy
from numpy import ndarray

from ... import _lib

logger = logging.getLogger(__name__)


def make_id_map(ids: List[str], mmseqs_db: Path) -> ndarray:
    """faiss row -> MMseqs2 internal id, from ``<db>.lookup`` (reference lines 20-31)."""
    mmseqs_map: Dict[str, int] = {}
    with Path(mmseqs_db).with_suffix(".lookup").open() as fp:
        for line in fp:
            seq_mmseqs_id, seq_name, _ = line.split("\t")
            mmseqs_map[seq_name] = int(seq_mmseqs_id)
    return numpy.asarray([mmseqs_map[entry_id] for entry_id in ids], dtype=numpy.int64)


def write_prefilter_db(hits: ndarray, prefilter_db: Path, queries: ndarray, scores: ndarray,
                       test_faiss_to_mmseqs: ndarray, train_faiss_to_mmseqs: ndarray, clip: bool = True):
    logger.info("Writing prefilter")
    missing = int(numpy.sum(hits == -1))
    if missing > 0:
        logger.warning(f"There are {missing} missing hits")
    prefilter_db = Path(prefilter_db)
    prefilter_db.with_suffix(".dbtype").write_bytes(b"\x07\x00\x00\x00")
    hits = numpy.ascontiguousarray(hits, dtype=numpy.int64)
    scores = numpy.ascontiguousarray(scores, dtype=numpy.float32)
    queries = numpy.ascontiguousarray(queries, dtype=numpy.int64)
    tmap = numpy.ascontiguousarray(test_faiss_to_mmseqs, dtype=numpy.int64)
    rmap = numpy.ascontiguousarray(train_faiss_to_mmseqs, dtype=numpy.int64)
    if hits.shape != scores.shape or hits.ndim != 2 or queries.shape[0] != hits.shape[0]:
        raise ValueError("write_prefilter_db: hits, scores and queries do not line up")
    _lib.check(_lib.lib().knn_write_prefilter_db(
        str(prefilter_db.with_suffix(".0")).encode(), str(prefilter_db.with_suffix(".index")).encode(),
        hits.ctypes.data, scores.ctypes.data, hits.shape[0], hits.shape[1], queries.ctypes.data,
        tmap.ctypes.data, tmap.shape[0], rmap.ctypes.data, rmap.shape[0], _clip_mode(clip)))


def _clip_mode(clip: bool) -> int:
    if not clip:
        return 0
    return 2 if int(numpy.__version__.split(".")[0]) >= 2 else 1
