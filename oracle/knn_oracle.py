"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product package never does.

Three things live here:

1. ``Oracle``: ctypes binding of ``oracle/libknn_oracle.so`` (``knn_oracle.c``),
   the fp32 restatement of the FAISS calls the reference makes
   (``faiss.normalize_L2`` / ``IndexFlat.add`` / ``IndexFlat.search``; call sites
   ``cath/search.py:13-26``, ``seqvec_search/main.py:22-50``,
   ``pfam/proteins_search.py:21-50``) in the *knn355 arithmetic contract*
   (see the header of ``knn_oracle.c``).  The HIP path is compared with it
   bit for bit.
2. ``exact_knn_f64`` + ``compare_tie_tolerant``: fp64 ground truth and the
   comparator that says where an fp32 result is *allowed* to differ from it
   (FAISS's own fp32 order depends on its BLAS).
3. ``faiss_flat_blas_restated``: FAISS 1.7.2's flat CPU algorithm restated with
   numpy's BLAS (4096-query x 1024-row sgemm blocks + per-row top-k); this is
   what ``bench.py`` times as ``cpu_baseline`` (kind "port").

``OracleFaiss`` is a faiss-shaped facade over (1) used by
``tests/golden/make_golden.py`` to drive the reference's own
``faiss_search``/``evaluate_faiss`` in the build container.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

METRIC_INNER_PRODUCT = 0
METRIC_L2 = 1

_HERE = Path(__file__).resolve().parent
_SO = Path(os.environ["KNN_ORACLE_SO"]) if os.environ.get("KNN_ORACLE_SO") else _HERE / "libknn_oracle.so"  # (sanitizer builds: tools/run_asan_cpu_tests.sh)


def build(force: bool = False) -> Path:
    """Compile knn_oracle.c with the committed Makefile (gcc, no GPU needed)."""
    newest = max((_HERE / f).stat().st_mtime for f in ("knn_oracle.c", "hnsw_oracle.c", "Makefile"))
    if force or not _SO.exists() or _SO.stat().st_mtime < newest:
        subprocess.check_call(["make", "-C", str(_HERE), "-s", "-B"])
    return _SO


class Oracle:
    def __init__(self):
        if not _SO.exists():
            build()
        self.lib = ctypes.CDLL(str(_SO))
        L = self.lib
        f32p = ctypes.POINTER(ctypes.c_float)
        i64p = ctypes.POINTER(ctypes.c_int64)
        L.orc_dot.restype = ctypes.c_float
        L.orc_dot.argtypes = [f32p, f32p, ctypes.c_int]
        L.orc_norm_l2sqr.restype = None
        L.orc_norm_l2sqr.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, f32p]
        L.orc_normalize_l2.restype = None
        L.orc_normalize_l2.argtypes = [f32p, ctypes.c_int64, ctypes.c_int]
        L.orc_flat_search.restype = ctypes.c_int
        L.orc_flat_search.argtypes = [f32p, ctypes.c_int64, f32p, ctypes.c_int64, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_int64, f32p, i64p]
        L.orc_set_l2_mode.restype = None
        L.orc_set_l2_mode.argtypes = [ctypes.c_int]
        L.orc_pair_distances.restype = None
        L.orc_pair_distances.argtypes = [f32p, f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                         i64p, i64p, f32p]

    @staticmethod
    def _f32(a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))

    def normalize_l2(self, x: np.ndarray) -> None:
        assert x.dtype == np.float32 and x.flags.c_contiguous and x.ndim == 2
        self.lib.orc_normalize_l2(x.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), x.shape[0], x.shape[1])

    def norm_l2sqr(self, x: np.ndarray) -> np.ndarray:
        x, xp = self._f32(x)
        out = np.empty(x.shape[0], np.float32)
        self.lib.orc_norm_l2sqr(xp, x.shape[0], x.shape[1], out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
        return out

    def set_l2_mode(self, mode: int) -> None:
        """0: FAISS's rule (fewer than 20 queries -> sum of squared differences, else the norm formula); 1: the norm
        formula whatever the batch; 2: differences whatever the batch"""
        self.lib.orc_set_l2_mode(int(mode))

    def flat_search(self, xb: np.ndarray, xq: np.ndarray, k: int, metric: int, l2_mode: int = 0):
        """l2_mode: see set_l2_mode (applies to this call only)"""
        self.lib.orc_set_l2_mode(int(l2_mode))
        try:
            return self._flat_search(xb, xq, k, metric)
        finally:
            self.lib.orc_set_l2_mode(0)

    def _flat_search(self, xb: np.ndarray, xq: np.ndarray, k: int, metric: int):
        xb, bp = self._f32(xb)
        xq, qp = self._f32(xq)
        assert xb.ndim == 2 and xq.ndim == 2 and xb.shape[1] == xq.shape[1]
        D = np.empty((xq.shape[0], k), np.float32)
        I = np.empty((xq.shape[0], k), np.int64)
        rc = self.lib.orc_flat_search(bp, xb.shape[0], qp, xq.shape[0], xb.shape[1], metric, k,
                                      D.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                      I.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
        if rc != 0:
            raise RuntimeError("orc_flat_search failed")
        return D, I

    def pair_distances(self, xb, xq, qidx, ridx, metric):
        xb, bp = self._f32(xb)
        xq, qp = self._f32(xq)
        qidx = np.ascontiguousarray(qidx, np.int64)
        ridx = np.ascontiguousarray(ridx, np.int64)
        out = np.empty(len(qidx), np.float32)
        self.lib.orc_pair_distances(bp, qp, xb.shape[1], metric, len(qidx),
                                    qidx.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                    ridx.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                    out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
        return out


class OracleHNSW:
    """Sequential-insertion HNSW restated on the CPU (oracle/hnsw_oracle.c): the quality
    reference for the GPU-offloaded, batch-synchronous build."""

    def __init__(self, d, M=32, metric=METRIC_L2, ef_construction=40):
        self.lib = oracle().lib
        L = self.lib
        L.hnsw_orc_new.restype = ctypes.c_void_p
        L.hnsw_orc_new.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.hnsw_orc_free.argtypes = [ctypes.c_void_p]
        L.hnsw_orc_set_ef.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.hnsw_orc_add.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
        L.hnsw_orc_search.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        self.d = d
        self.h = L.hnsw_orc_new(d, M, metric, ef_construction)

    def add(self, x):
        x = np.ascontiguousarray(x, np.float32)
        self.lib.hnsw_orc_add(self.h, x.ctypes.data, x.shape[0])

    def search(self, q, k, ef_search):
        q = np.ascontiguousarray(q, np.float32)
        self.lib.hnsw_orc_set_ef(self.h, ef_search)
        D = np.empty((q.shape[0], k), np.float32)
        I = np.empty((q.shape[0], k), np.int64)
        self.lib.hnsw_orc_search(self.h, q.ctypes.data, q.shape[0], k, D.ctypes.data, I.ctypes.data)
        return D, I

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.hnsw_orc_free(self.h)
            self.h = None


_ORACLE = None


def oracle() -> Oracle:
    global _ORACLE
    if _ORACLE is None:
        _ORACLE = Oracle()
    return _ORACLE


# --------------------------------------------------------------------------
# fp64 ground truth + comparator
# --------------------------------------------------------------------------
def exact_scores_f64(xb: np.ndarray, xq: np.ndarray, metric: int) -> np.ndarray:
    b = np.asarray(xb, np.float64)
    q = np.asarray(xq, np.float64)
    ip = q @ b.T
    if metric == METRIC_INNER_PRODUCT:
        return ip
    return np.maximum((q * q).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2.0 * ip, 0.0)


def exact_knn_f64(xb, xq, k, metric):
    """(D float64 [nq,k], I int64 [nq,k]) best first, ties -> lower id (stable sort)."""
    s = exact_scores_f64(xb, xq, metric)
    order = np.argsort(-s if metric == METRIC_INNER_PRODUCT else s, axis=1, kind="stable")[:, :k]
    return np.take_along_axis(s, order, 1), order.astype(np.int64)


def compare_tie_tolerant(I_got, D_got, xb, xq, metric, tau_rel=8e-6, dist_rtol=1e-5):
    """Check an fp32 (D, I) result against the fp64 truth.

    An fp32 result is accepted iff, per query,
      * every returned id's TRUE (fp64) score is within ``tau`` of the true score
        at the same rank (so ids may only be permuted inside a cluster of
        near-ties, or swapped across the k-boundary with a near-tied candidate),
      * returned distances are within ``dist_rtol * scale`` of the true score of the
        returned id, scale = max(1, |x||y|) for inner product and
        max(1, |x|^2 + |y|^2) for squared L2 (the magnitudes the fp32 sum is formed
        from; for unit-norm cosine this is the north-star's absolute 1e-5),
      * no id is returned twice.
    tau = tau_rel * sqrt(d) * scale bounds fp32 summation-order noise (SURVEY.md 7.3
    H1, probabilistic form).  Returns a dict of counts; raises AssertionError on
    violation.
    """
    I_got = np.asarray(I_got)
    D_got = np.asarray(D_got, np.float64)
    nq, k = I_got.shape
    s = exact_scores_f64(xb, xq, metric)
    nb = s.shape[1]
    kk = min(k, nb)
    D_true, I_true = exact_knn_f64(xb, xq, kk, metric)
    d = np.asarray(xb).shape[1]
    got = I_got[:, :kk]
    assert (got >= 0).all() and (got < nb).all(), "id out of range"
    bn2 = (np.asarray(xb, np.float64) ** 2).sum(1)
    qn2 = (np.asarray(xq, np.float64) ** 2).sum(1)
    if metric == METRIC_INNER_PRODUCT:
        scale = np.maximum(1.0, np.sqrt(qn2)[:, None] * np.sqrt(bn2)[got])
        scale_max = np.maximum(1.0, np.sqrt(qn2)[:, None] * np.sqrt(bn2.max(initial=0.0)))
    else:
        scale = np.maximum(1.0, qn2[:, None] + bn2[got])
        scale_max = np.maximum(1.0, qn2[:, None] + bn2.max(initial=0.0))
    tau = tau_rel * np.sqrt(d) * scale_max
    srt = np.sort(got, axis=1)
    assert (srt[:, 1:] != srt[:, :-1]).all(), "duplicate ids in a result row"
    true_of_got = np.take_along_axis(s, got, 1)
    rank_err = np.abs(true_of_got - D_true)
    assert (rank_err <= tau).all(), f"rank error {rank_err.max()} exceeds tau {tau.min()}"
    derr = np.abs(D_got[:, :kk] - true_of_got)
    assert (derr <= dist_rtol * scale).all(), f"distance error {(derr / scale).max()} (relative to scale)"
    if k > kk:
        assert (I_got[:, kk:] == -1).all(), "unfilled slots must be id -1"
    return {"permuted": int((got != I_true).sum()), "max_rank_err": float(rank_err.max(initial=0.0)),
            "max_dist_err_rel": float((derr / scale).max(initial=0.0))}


def recall_at_k(I_got, I_true):
    k = I_true.shape[1]
    hit = 0
    for a, b in zip(I_got, I_true):
        hit += len(np.intersect1d(a[a >= 0], b))
    return hit / (I_true.shape[0] * k)


# --------------------------------------------------------------------------
# FAISS 1.7.2 flat CPU algorithm restated with numpy's BLAS (cpu_baseline)
# --------------------------------------------------------------------------
def faiss_flat_blas_restated(xb, xq, k, metric, bs_x=4096, bs_y=1024):
    """knn_inner_product_blas / knn_L2sqr_blas restated: query blocks of 4096,
    database blocks of 1024, one sgemm per block pair, running top-k per row
    (FAISS: heap for k<100, reservoir otherwise; here argpartition over
    [current top-k | block scores], same result set)."""
    xb = np.ascontiguousarray(xb, np.float32)
    xq = np.ascontiguousarray(xq, np.float32)
    nq, nb = xq.shape[0], xb.shape[0]
    ip_metric = metric == METRIC_INNER_PRODUCT
    D = np.full((nq, k), -np.finfo(np.float32).max if ip_metric else np.finfo(np.float32).max, np.float32)
    I = np.full((nq, k), -1, np.int64)
    if not ip_metric:
        xn = (xq * xq).sum(1)
        yn = (xb * xb).sum(1)
    for i0 in range(0, nq, bs_x):
        i1 = min(nq, i0 + bs_x)
        bd = D[i0:i1]
        bi = I[i0:i1]
        for j0 in range(0, nb, bs_y):
            j1 = min(nb, j0 + bs_y)
            blk = xq[i0:i1] @ xb[j0:j1].T
            if not ip_metric:
                blk = xn[i0:i1, None] + yn[None, j0:j1] - 2 * blk
                np.maximum(blk, 0, out=blk)
            cand_d = np.concatenate([bd, blk], 1)
            cand_i = np.concatenate([bi, np.broadcast_to(np.arange(j0, j1, dtype=np.int64), blk.shape)], 1)
            key = -cand_d if ip_metric else cand_d
            if cand_d.shape[1] > k:
                sel = np.argpartition(key, k - 1, axis=1)[:, :k]
            else:
                sel = np.broadcast_to(np.arange(cand_d.shape[1]), (i1 - i0, cand_d.shape[1]))
            bd = np.take_along_axis(cand_d, sel, 1)
            bi = np.take_along_axis(cand_i, sel, 1)
            if bd.shape[1] < k:
                pad = k - bd.shape[1]
                bd = np.concatenate([bd, np.full((i1 - i0, pad), -3.4028235e38 if ip_metric else 3.4028235e38, np.float32)], 1)
                bi = np.concatenate([bi, np.full((i1 - i0, pad), -1, np.int64)], 1)
        order = np.argsort(-bd if ip_metric else bd, axis=1, kind="stable")
        D[i0:i1] = np.take_along_axis(bd, order, 1)
        I[i0:i1] = np.take_along_axis(bi, order, 1)
    return D, I


# --------------------------------------------------------------------------
# faiss-shaped facade over the oracle (for tests/golden/make_golden.py only)
# --------------------------------------------------------------------------
class _OracleIndexFlat:
    def __init__(self, d, metric=METRIC_L2):
        self.d = d
        self.metric_type = metric
        self.ntotal = 0
        self._xb = np.empty((0, d), np.float32)
        self.is_trained = True

    def train(self, x):
        pass

    def add(self, x):
        assert x.dtype == np.float32 and x.shape[1] == self.d
        self._xb = np.concatenate([self._xb, x], 0)
        self.ntotal = self._xb.shape[0]

    def search(self, x, k):
        return oracle().flat_search(self._xb, x, k, self.metric_type)


class OracleFaiss:
    """Module-like object exposing the faiss symbols the reference uses."""
    METRIC_INNER_PRODUCT = METRIC_INNER_PRODUCT
    METRIC_L2 = METRIC_L2
    IndexFlat = _OracleIndexFlat

    class IndexLSH:  # only referenced in a type annotation (seqvec_search/main.py:23)
        pass

    @staticmethod
    def normalize_L2(x):
        oracle().normalize_l2(x)
