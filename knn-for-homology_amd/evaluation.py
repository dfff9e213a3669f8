"""Consumers of the (hits, scores) arrays as single GPU passes.

Each function keeps the semantics of the reference's per-row Python loop it replaces:
  remove_self_hit         pfam/proteins.py:85-122
  evaluate_faiss          seqvec_search/main.py:53-82 (AUC1 / TP with one family label per id)
  compute_auc1            pfam/proteins_shared.py:139-157 (sets of homologous proteins)
  compute_is_correct      cath/cath.py:76-84 (C/A/T/H level matrix)
  compute_tps_comulative  seqvec_search/tp_cumulative.py:15-34
"""
from collections import Counter
from typing import Dict, List, Sequence, Set, Tuple

import numpy as np
from numpy import ndarray

from . import _lib


def _hits(h):
    h = np.ascontiguousarray(h, dtype=np.int64)
    if h.ndim != 2:
        raise ValueError("hits must be 2-D")
    return h


def remove_self_hit(hits: ndarray, scores: ndarray, self_ids: ndarray = None) -> Tuple[ndarray, ndarray]:
    """Removes the self hit from every row even when an approximate search did not put it
    first; rows that do not contain their own id lose their last hit instead."""
    hits = _hits(hits)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    nq, k = hits.shape
    self_ids = np.arange(nq, dtype=np.int64) if self_ids is None else np.ascontiguousarray(self_ids, np.int64)
    print(f"Fixing {int((hits[:, 0] != self_ids).sum())} misplaced self hits")
    ho = np.empty((nq, k - 1), np.int64)
    so = np.empty((nq, k - 1), np.float32)
    missing = np.empty(nq, np.int32)
    _lib.check(_lib.lib().knn_eval_remove_self_hit(hits.ctypes.data, scores.ctypes.data, nq, k, self_ids.ctypes.data,
                                                   ho.ctypes.data, so.ctypes.data, missing.ctypes.data))
    print(f"There are {int(missing.sum())} missing self hits")
    return ho, so


def label_matches(hits: ndarray, labels_q: ndarray, labels_db: ndarray, want_matrix=True):
    """(is_correct bool [nq,k] or None, leading-run length int32 [nq], match count int32 [nq])"""
    hits = _hits(hits)
    nq, k = hits.shape
    lq = np.ascontiguousarray(labels_q, np.int32)
    ldb = np.ascontiguousarray(labels_db, np.int32)
    ic = np.empty((nq, k), np.uint8) if want_matrix else None
    lead = np.empty(nq, np.int32)
    tp = np.empty(nq, np.int32)
    _lib.check(_lib.lib().knn_eval_labels(hits.ctypes.data, nq, k, lq.ctypes.data, ldb.ctypes.data, ldb.shape[0],
                                          ic.ctypes.data if want_matrix else None, lead.ctypes.data, tp.ctypes.data))
    return (ic.astype(bool) if want_matrix else None), lead, tp


def _family_codes(data):
    fams = sorted(set(data.ids_to_family.values()))
    code = {f: i for i, f in enumerate(fams)}
    lq = np.asarray([code[data.ids_to_family[i]] for i in data.test_ids], np.int32)
    ldb = np.asarray([code[data.ids_to_family[i]] for i in data.train_ids], np.int32)
    sizes = Counter(data.ids_to_family[i] for i in data.train_ids)
    fam_size = np.asarray([sizes[data.ids_to_family[i]] for i in data.test_ids], np.int64)
    return lq, ldb, fam_size


def evaluate_faiss(data, results: ndarray) -> Tuple[List[float], List[float]]:
    """Same lists as seqvec_search.main.evaluate_faiss (AUC1, TP per query)."""
    lq, ldb, fam_size = _family_codes(data)
    _, lead, tp = label_matches(results, lq, ldb, want_matrix=False)
    return [int(a) / int(s) for a, s in zip(lead, fam_size)], [int(t) / int(s) for t, s in zip(tp, fam_size)]


def compute_tps_comulative(data, results: ndarray) -> ndarray:
    lq, ldb, fam_size = _family_codes(data)
    is_correct, _, _ = label_matches(results, lq, ldb)
    max_tp_expanded = fam_size.repeat(is_correct.shape[1]).reshape(is_correct.shape)
    return (is_correct.cumsum(axis=1) / max_tp_expanded).mean(axis=0)


def compute_auc1(hits: ndarray, homologous_proteins: Dict[str, Set[str]], queries: Sequence[str],
                 target_ids: Sequence[str]) -> ndarray:
    hits = _hits(hits)
    nq, k = hits.shape
    pos = {t: i for i, t in enumerate(target_ids)}
    offsets = np.zeros(nq + 1, np.int64)
    members = []
    sizes = np.empty(nq, np.int64)
    for i in range(nq):
        allc = homologous_proteins[queries[i]]
        sizes[i] = max(len(allc), 1)
        rows = sorted(pos[t] for t in allc if t in pos)
        members.extend(rows)
        offsets[i + 1] = len(members)
    members = np.asarray(members, np.int64)
    lead = np.empty(nq, np.int32)
    tp = np.empty(nq, np.int32)
    _lib.check(_lib.lib().knn_eval_sets(hits.ctypes.data, nq, k, offsets.ctypes.data,
                                        members.ctypes.data if members.size else None, lead.ctypes.data, tp.ctypes.data))
    return lead / sizes


def compute_is_correct(results: ndarray, mapping_array: ndarray, query_rows: ndarray = None) -> ndarray:
    """bool [nq, levels, hits]: does hit j share query q's label at level l (cath: C, A, T, H)."""
    results = _hits(results)
    nq, k = results.shape
    mapping = np.ascontiguousarray(mapping_array)
    if mapping.ndim != 2:
        raise ValueError("mapping_array must be [n, levels]")
    # arbitrary label values -> dense int32 codes per level (equality is all that matters)
    codes = np.empty(mapping.shape, np.int32)
    for l in range(mapping.shape[1]):
        _, codes[:, l] = np.unique(mapping[:, l], return_inverse=True)
    qrows = np.arange(nq, dtype=np.int64) if query_rows is None else np.ascontiguousarray(query_rows, np.int64)
    out = np.empty((nq, mapping.shape[1], k), np.uint8)
    _lib.check(_lib.lib().knn_eval_levels(results.ctypes.data, nq, k, qrows.ctypes.data, codes.ctypes.data,
                                          mapping.shape[0], mapping.shape[1], out.ctypes.data))
    return out.astype(bool)
