"""ctypes binding of libknn355.so (C ABI: include/knn355.h).

Nothing touches HIP at import time: the shared object is dlopen'ed on first use and
the library itself creates its HIP context lazily, so forked workers
(cath/compare_seqvec_layer.py:58-64 in the reference) are safe.

If ``torch`` is going to be used in the same process (multi-GPU path, bench.py),
import it BEFORE the first call here so that both share one HIP runtime
(torch ships its own libamdhip64 with the same SONAME).
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int32, c_int64, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# KNN355_LIB: another build of the same library (timing-ablation builds, tools/ only)
LIB_PATH = os.environ.get("KNN355_LIB") or os.path.join(_HERE, "libknn355.so")

_lib = None


class Knn355Error(RuntimeError):
    """Raised for every non-zero return code of the C ABI (FAISS surfaces its C++
    exceptions as RuntimeError too)."""


def _declare(L):
    f32p, i64p, u64p = POINTER(c_float), POINTER(c_int64), POINTER(c_uint64)
    H = c_void_p
    sig = {
        "knn_last_error": (c_char_p, []),
        "knn_version": (c_char_p, []),
        "knn_device_count": (c_int32, []),
        "knn_init": (c_int32, [c_int32]),
        "knn_trim": (c_int64, []),
        "knn_host_alloc": (c_void_p, [c_int64]),
        "knn_host_free": (None, [c_void_p]),
        "knn_normalize_l2": (c_int32, [c_void_p, c_int64, c_int32]),
        "knn_normalize_l2_dev": (c_int32, [c_void_p, c_int64, c_int32, c_void_p]),
        "knn_flat_create": (c_int32, [c_int32, c_int32, POINTER(H)]),
        "knn_flat_add": (c_int32, [H, c_void_p, c_int64]),
        "knn_flat_add_dev": (c_int32, [H, c_void_p, c_int64, c_void_p]),
        "knn_flat_search": (c_int32, [H, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
        "knn_flat_search_dev": (c_int32, [H, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
        "knn_flat_search_self": (c_int32, [H, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
        "knn_flat_search_self_dev": (c_int32, [H, c_int64, c_void_p, c_void_p]),
        "knn_flat_normalize_rows": (c_int32, [H]),
        "knn_flat_view": (c_int32, [H, POINTER(H)]),
        "knn_flat_search_keys_dev": (c_int32, [H, c_void_p, c_int64, c_int64, c_uint32, c_void_p, c_void_p]),
        "knn_merge_keys_dev": (c_int32, [H, c_void_p, c_int32, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
        "knn_comm_unique_id": (c_int32, [c_void_p]),
        "knn_comm_create": (c_int32, [c_void_p, c_int32, c_int32, c_int32, POINTER(H)]),
        "knn_comm_free": (None, [H]),
        "knn_sharded_search_dev": (c_int32, [H, H, c_void_p, c_int64, c_int64, c_uint32, c_void_p, c_void_p, c_void_p]),
        "knn_ntotal": (c_int64, [H]),
        "knn_dim": (c_int32, [H]),
        "knn_metric": (c_int32, [H]),
        "knn_device_of": (c_int32, [H]),
        "knn_reset": (c_int32, [H]),
        "knn_flat_reconstruct": (c_int32, [H, c_int64, c_int64, c_void_p]),
        "knn_free": (None, [H]),
        "knn_gather_distances": (c_int32, [H, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
        "knn_last_scan_info": (c_int32, [H, c_char_p, c_int32, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(c_int32)]),
        "knn_last_scan_ms": (c_float, [H]),
        "knn_set_tuning": (c_int32, [H, c_int32, c_int32, c_int32]),
        "knn_flat_set_batch": (c_int32, [H, c_int64]),
        "knn_flat_read_rate": (c_int32, [H, c_int32, POINTER(c_float), POINTER(c_int64)]),
        "knn_mfma_rate": (c_int32, [c_int32, POINTER(c_float), POINTER(c_float)]),
        "knn_hnsw_create": (c_int32, [c_int32, c_int32, c_int32, POINTER(H)]),
        "knn_hnsw_set_ef": (c_int32, [H, c_int32, c_int32]),
        "knn_hnsw_set_walk": (c_int32, [H, c_int32, c_int32]),
        "knn_hnsw_set_entry": (c_int32, [H, c_int32]),
        "knn_hnsw_get_params": (c_int32, [H, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(c_int64)]),
        "knn_hnsw_add": (c_int32, [H, c_void_p, c_int64]),
        "knn_hnsw_add_dev": (c_int32, [H, c_void_p, c_int64, c_void_p]),
        "knn_hnsw_search": (c_int32, [H, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
        "knn_hnsw_ntotal": (c_int64, [H]),
        "knn_hnsw_storage": (H, [H]),
        "knn_hnsw_free": (None, [H]),
        "knn_hnsw_graph_sizes": (c_int32, [H, POINTER(c_int64), POINTER(c_int64), POINTER(c_int32)]),
        "knn_hnsw_graph_export": (c_int32, [H, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
        "knn_hnsw_graph_import": (c_int32, [H, c_int64, c_void_p, c_void_p, c_int64, c_int32, c_int64]),
        "knn_hnsw_stats": (c_int32, [H, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64), POINTER(ctypes.c_double), POINTER(ctypes.c_double), c_int32]),
        "knn_lsh_create": (c_int32, [c_int32, c_int32, c_void_p, POINTER(H)]),
        "knn_lsh_add": (c_int32, [H, c_void_p, c_int64]),
        "knn_lsh_search": (c_int32, [H, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
        "knn_lsh_ntotal": (c_int64, [H]),
        "knn_lsh_code_words": (c_int32, [H]),
        "knn_lsh_get_codes": (c_int32, [H, c_void_p, c_int32]),
        "knn_lsh_add_codes": (c_int32, [H, c_void_p, c_int64, c_int32]),
        "knn_lsh_free": (None, [H]),
        "knn_eval_remove_self_hit": (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
        "knn_eval_labels": (c_int32, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
        "knn_eval_sets": (c_int32, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
        "knn_eval_levels": (c_int32, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
        "knn_write_prefilter_db": (c_int32, [c_char_p, c_char_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int32]),
        "knn_scan_times": (c_int32, [H, c_void_p, c_int32]),
        "knn_last_seed_info": (c_int32, [H, POINTER(c_int32), POINTER(c_int32), POINTER(c_int64), POINTER(c_int64)]),
        "knn_flat_reserve": (c_int32, [H, c_int64]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError here == the library does not export the ABI
        fn.restype = res
        fn.argtypes = args
    return sig


EXPORTS = None


def lib():
    """Returns the loaded library; raises Knn355Error if it has not been built."""
    global _lib, EXPORTS
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Knn355Error(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C knn-for-homology_amd/csrc` (there is no CPU fallback)")
        L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL if hasattr(ctypes, "RTLD_GLOBAL") else 0)
        EXPORTS = _declare(L)
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().knn_last_error()
        raise Knn355Error((msg or b"unknown error").decode("utf-8", "replace") + f" (code {rc})")


def device_count():
    return int(lib().knn_device_count())


# ---- result arrays in page-locked memory ------------------------------------------------------
# Large (D, I) arrays are numpy arrays over hipHostMalloc'ed blocks: the result download is then a single DMA
# at PCIe line rate instead of a staged copy (CATH20-sized k=300 result: 52 MB).  Blocks come from a small
# size-classed free list and return to it when the last view of the array dies; at most PINNED_LIMIT bytes
# are ever outstanding (beyond that: plain numpy.empty).  Page-locking is not free -- hipHostMalloc pins about
# 4 GiB/s (3 GiB for a Pfam-sized k=1000 result: 0.7 s), a staged download into pageable memory loses about 0.08 s per
# GB -- so a block larger than PINNED_FIRST_MAX is only allocated when its size class is asked for the SECOND time:
# a script that searches once (pfam/proteins_search.py) never pays, a loop over files (cath/search.py) pays once.
# A NEW block is only page-locked while fewer than PINNED_LIVE_PER_CLASS arrays of its size class are alive: a caller that
# drops each result before the next search (or holds one while the next is computed) recycles two blocks for ever; a caller
# that KEEPS every result -- cath/search.py:37-50 collects the hits of every file of a metric before it saves them -- would
# otherwise page-lock a fresh block per search (52 MB of CATH-sized results: 13-23 ms, five times the search) and gets plain
# arrays from its third live result on (a staged download: +1 ms).
PINNED_MIN = 4 << 20
PINNED_FIRST_MAX = 64 << 20
PINNED_LIMIT = 4 << 30
PINNED_LIVE_PER_CLASS = 2
_pinned_free = {}    # size class -> [pointers]
_pinned_seen = {}    # size class -> requests so far
_pinned_live = {}    # size class -> arrays alive (handed out, not yet returned)
_pinned_out = 0      # bytes handed out or cached


class _PinnedBlock:
    """Owner of one page-locked block; numpy views keep it alive through ``__array_interface__``."""

    def __init__(self, ptr, nbytes, cls, shape, typestr):
        self.ptr, self.nbytes, self.cls = ptr, nbytes, cls
        self.__array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 3}

    def __del__(self):
        try:
            _pinned_free.setdefault(self.cls, []).append(self.ptr)
            _pinned_live[self.cls] = _pinned_live.get(self.cls, 1) - 1
        except Exception:  # interpreter shutdown
            pass


def result_array(shape, dtype):
    """numpy.empty(shape, dtype) -- in page-locked memory when the array is large and the pool has room."""
    import numpy as np
    global _pinned_out
    dt = np.dtype(dtype)
    nbytes = int(np.prod(shape)) * dt.itemsize
    if nbytes < PINNED_MIN:
        return np.empty(shape, dt)
    cls = 1 << (nbytes - 1).bit_length()
    free = _pinned_free.get(cls)
    ptr = free.pop() if free else None
    if ptr is None:
        seen = _pinned_seen.get(cls, 0)
        _pinned_seen[cls] = seen + 1
        if (_pinned_out + cls > PINNED_LIMIT or (cls > PINNED_FIRST_MAX and seen == 0)
                or _pinned_live.get(cls, 0) >= PINNED_LIVE_PER_CLASS):
            return np.empty(shape, dt)
        ptr = lib().knn_host_alloc(cls)
        if not ptr:
            return np.empty(shape, dt)
        _pinned_out += cls
    _pinned_live[cls] = _pinned_live.get(cls, 0) + 1
    return np.asarray(_PinnedBlock(ptr, nbytes, cls, tuple(int(v) for v in shape), dt.str))
