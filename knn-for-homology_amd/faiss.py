"""The slice of the ``faiss`` Python module that the reference calls, served by
libknn355.so on an MI355X.

Reference call sites (konstin/knn-for-homology):
  METRIC_INNER_PRODUCT / METRIC_L2  cath/search.py:14,31-32; seqvec_search/main.py:26
  normalize_L2                      cath/search.py:19; pfam/proteins_search.py:22;
                                    seqvec_search/main.py:31,34; pfam/search.py:18,20
  IndexFlat(d, metric)              cath/search.py:20; pfam/proteins_search.py:24;
                                    seqvec_search/main.py:35; pfam/search.py:44
  .train / .add / .search           cath/search.py:22-24; seqvec_search/main.py:37-45
  IndexHNSWFlat(d, M, metric)       pfam/proteins_search.py:30-31 (hnsw.efSearch)
  IndexLSH(d, nbits)                seqvec_search/create_index.py:41; pfam/search.py:27
  write_index / read_index          pfam/proteins_search.py:40; seqvec_search/main.py:132

Semantics kept: float32 C-contiguous 2-D inputs only (anything else raises, as the
faiss SWIG wrapper does); ``search`` returns freshly allocated ``(D float32 [nq,k],
I int64 [nq,k])`` best first; unfilled slots are id -1 with -FLT_MAX (IP) / +FLT_MAX
(L2); ``normalize_L2`` works in place on the caller's array.
"""
import ctypes

import numpy as np

from . import _lib

METRIC_INNER_PRODUCT = 0
METRIC_L2 = 1


def _check_matrix(x, d=None, what="x"):
    if not isinstance(x, np.ndarray):
        raise TypeError(f"{what}: expected numpy.ndarray, got {type(x).__name__}")
    if x.dtype != np.float32:
        raise TypeError(f"{what}: expected float32, got {x.dtype} (cast with .astype(numpy.float32))")
    if x.ndim != 2:
        raise ValueError(f"{what}: expected a 2-D array, got {x.ndim}-D")
    if not x.flags.c_contiguous:
        raise ValueError(f"{what}: array must be C-contiguous")
    if d is not None and x.shape[1] != d:
        raise AssertionError(f"{what}: dimension {x.shape[1]} does not match index dimension {d}")
    return x


def normalize_L2(x):
    """faiss.normalize_L2: row-wise x /= ||x||_2 in place; zero rows untouched."""
    _check_matrix(x)
    if not x.flags.writeable:
        raise ValueError("normalize_L2: array is read-only")
    _lib.check(_lib.lib().knn_normalize_l2(x.ctypes.data, x.shape[0], x.shape[1]))


class Index:
    """Common surface: d, ntotal, metric_type, is_trained, train/add/search/reset."""

    is_trained = True

    def train(self, x):
        _check_matrix(x, self.d)

    def __len__(self):
        return self.ntotal


class IndexFlat(Index):
    """Exhaustive search over device-resident rows (faiss.IndexFlat)."""

    def __init__(self, d, metric=METRIC_L2):
        self._h = ctypes.c_void_p()
        self._d = int(d)
        self._metric = int(metric)
        L = _lib.lib()
        _lib.check(L.knn_flat_create(self._d, self._metric, ctypes.byref(self._h)))

    # -- properties mirroring the SWIG object --
    @property
    def d(self):
        return self._d

    @property
    def metric_type(self):
        return self._metric

    @property
    def ntotal(self):
        return int(_lib.lib().knn_ntotal(self._h)) if self._h else 0

    def add(self, x):
        _check_matrix(x, self._d)
        _lib.check(_lib.lib().knn_flat_add(self._h, x.ctypes.data, x.shape[0]))

    def search(self, x, k):
        _check_matrix(x, self._d)
        k = int(k)
        if k < 1:
            raise AssertionError("k must be positive")
        D = np.empty((x.shape[0], k), np.float32)
        I = np.empty((x.shape[0], k), np.int64)
        _lib.check(_lib.lib().knn_flat_search(self._h, x.ctypes.data, x.shape[0], k, D.ctypes.data, I.ctypes.data))
        return D, I

    def reset(self):
        _lib.check(_lib.lib().knn_reset(self._h))

    def reconstruct_n(self, i0=0, n=None):
        n = self.ntotal - i0 if n is None else n
        out = np.empty((n, self._d), np.float32)
        _lib.check(_lib.lib().knn_flat_reconstruct(self._h, i0, n, out.ctypes.data))
        return out

    def reconstruct(self, i):
        return self.reconstruct_n(int(i), 1)[0]

    # -- knn355 extras (not in faiss) --
    def set_tuning(self, query_tile=0, nchunks=0, flags=0):
        _lib.check(_lib.lib().knn_set_tuning(self._h, query_tile, nchunks, flags))

    def last_scan(self):
        L = _lib.lib()
        name = ctypes.create_string_buffer(64)
        qt, dt, nc, grid = (ctypes.c_int32() for _ in range(4))
        _lib.check(L.knn_last_scan_info(self._h, name, 64, ctypes.byref(qt), ctypes.byref(dt), ctypes.byref(nc), ctypes.byref(grid)))
        return {"kernel": name.value.decode(), "query_tile": qt.value, "db_tile": dt.value, "nchunks": nc.value,
                "grid": grid.value, "ms": float(L.knn_last_scan_ms(self._h))}

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().knn_free(h)
            except Exception:
                pass
            self._h = None


class IndexFlatIP(IndexFlat):
    def __init__(self, d):
        super().__init__(d, METRIC_INNER_PRODUCT)


class IndexFlatL2(IndexFlat):
    def __init__(self, d):
        super().__init__(d, METRIC_L2)
