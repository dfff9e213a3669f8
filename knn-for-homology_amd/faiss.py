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
        D = _lib.result_array((x.shape[0], k), np.float32)
        I = _lib.result_array((x.shape[0], k), np.int64)
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
    def reconstruct_into(self, out, i0=0):
        """downloads rows [i0, i0 + len(out)) into the caller's float32 array"""
        _check_matrix(out, self._d)
        if not out.flags.writeable:
            raise ValueError("reconstruct_into: array is read-only")
        _lib.check(_lib.lib().knn_flat_reconstruct(self._h, int(i0), out.shape[0], out.ctypes.data))

    def search_self(self, k, row0=0, nrows=None):
        """``index.search(x, k)`` for x = the index's own rows [row0, row0+nrows): the all-vs-all
        the reference runs (cath/search.py:22-24), without uploading the queries again."""
        k = int(k)
        if k < 1:
            raise AssertionError("k must be positive")
        nrows = self.ntotal - row0 if nrows is None else int(nrows)
        D = _lib.result_array((nrows, k), np.float32)
        I = _lib.result_array((nrows, k), np.int64)
        _lib.check(_lib.lib().knn_flat_search_self(self._h, int(row0), nrows, k, D.ctypes.data, I.ctypes.data))
        return D, I

    def view(self):
        """A read-only second handle on the same device rows with its own stream and scratch
        memory: searches on an index and its view overlap on the GPU (sharded.py alternates
        between the two).  Keeps its parent alive."""
        v = object.__new__(IndexFlat)
        v._h = ctypes.c_void_p()
        v._d, v._metric, v._parent = self._d, self._metric, self
        _lib.check(_lib.lib().knn_flat_view(self._h, ctypes.byref(v._h)))
        return v

    def normalize_rows(self):
        """L2-normalises the stored rows in place on the device (``faiss.normalize_L2`` applied
        to what was added, bit for bit, without the host round trip)."""
        _lib.check(_lib.lib().knn_flat_normalize_rows(self._h))

    def set_tuning(self, query_tile=0, nchunks=0, flags=0):
        _lib.check(_lib.lib().knn_set_tuning(self._h, query_tile, nchunks, flags))

    def set_batch(self, nq_whole=0):
        """The searches that follow are pieces of one batch of ``nq_whole`` queries (0: each call is its own batch):
        FAISS picks the squared-L2 formula by the size of the batch handed to ``index.search``."""
        _lib.check(_lib.lib().knn_flat_set_batch(self._h, int(nq_whole)))

    def last_scan(self):
        L = _lib.lib()
        name = ctypes.create_string_buffer(64)
        qt, dt, nc, grid = (ctypes.c_int32() for _ in range(4))
        _lib.check(L.knn_last_scan_info(self._h, name, 64, ctypes.byref(qt), ctypes.byref(dt), ctypes.byref(nc), ctypes.byref(grid)))
        return {"kernel": name.value.decode(), "query_tile": qt.value, "db_tile": dt.value, "nchunks": nc.value,
                "grid": grid.value, "ms": float(L.knn_last_scan_ms(self._h))}

    def last_seed(self):
        """{"stride": seed-sample stride of the last search (0: none), "stat_rank": j of a statistical seed
        (0: exact bound), "stat_redo": searches repeated because a statistical threshold failed verification,
        "sample_rows": rows the sample pass scanned instead of the main scan kernel}"""
        st, j, redo, rows = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int64()
        _lib.check(_lib.lib().knn_last_seed_info(self._h, ctypes.byref(st), ctypes.byref(j), ctypes.byref(redo), ctypes.byref(rows)))
        return {"stride": st.value, "stat_rank": j.value, "stat_redo": redo.value, "sample_rows": rows.value}

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


class _HNSW:
    """``index.hnsw``: the graph parameters the reference touches (``efSearch``,
    pfam/proteins_search.py:31) plus read-only structure."""

    def __init__(self, owner):
        object.__setattr__(self, "_owner", owner)

    def _params(self):
        M, efs, efc, ml = (ctypes.c_int32() for _ in range(4))
        ep = ctypes.c_int64()
        _lib.check(_lib.lib().knn_hnsw_get_params(self._owner._h, ctypes.byref(M), ctypes.byref(efs), ctypes.byref(efc),
                                                  ctypes.byref(ml), ctypes.byref(ep)))
        return {"M": M.value, "efSearch": efs.value, "efConstruction": efc.value, "max_level": ml.value,
                "entry_point": ep.value}

    def __getattr__(self, name):
        p = self._params()
        if name in p:
            return p[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name == "efSearch":
            _lib.check(_lib.lib().knn_hnsw_set_ef(self._owner._h, int(value), -1))
        elif name == "efConstruction":
            _lib.check(_lib.lib().knn_hnsw_set_ef(self._owner._h, -1, int(value)))
        else:
            raise AttributeError(f"hnsw.{name} is read-only")


class IndexHNSWFlat(Index):
    """faiss.IndexHNSWFlat(d, M, metric): HNSW graph over flat float32 storage.  The graph
    is built and walked on the host (libknn355's C++), every distance is computed on the
    GPU in lock-step batches (pfam/proteins_search.py:27-31)."""

    def __init__(self, d, M=32, metric=METRIC_L2):
        self._h = ctypes.c_void_p()
        self._d = int(d)
        self._metric = int(metric)
        _lib.check(_lib.lib().knn_hnsw_create(self._d, int(M), self._metric, ctypes.byref(self._h)))
        self.hnsw = _HNSW(self)

    @property
    def d(self):
        return self._d

    @property
    def metric_type(self):
        return self._metric

    @property
    def ntotal(self):
        return int(_lib.lib().knn_hnsw_ntotal(self._h)) if self._h else 0

    def add(self, x):
        _check_matrix(x, self._d)
        _lib.check(_lib.lib().knn_hnsw_add(self._h, x.ctypes.data, x.shape[0]))

    def search(self, x, k):
        _check_matrix(x, self._d)
        k = int(k)
        if k < 1:
            raise AssertionError("k must be positive")
        D = np.empty((x.shape[0], k), np.float32)
        I = np.empty((x.shape[0], k), np.int64)
        _lib.check(_lib.lib().knn_hnsw_search(self._h, x.ctypes.data, x.shape[0], k, D.ctypes.data, I.ctypes.data))
        return D, I

    def add_dev(self, x):
        """knn355 extra: rows from a float32 CUDA tensor [n, d] on the index's device (no host copy)."""
        if not (x.is_cuda and x.is_contiguous() and x.dim() == 2 and x.shape[1] == self._d and x.element_size() == 4):
            raise ValueError("add_dev: expected a contiguous float32 CUDA tensor [n, d]")
        _lib.check(_lib.lib().knn_hnsw_add_dev(self._h, x.data_ptr(), x.shape[0], None))

    def set_entry(self, coarse_entries=4):
        """knn355 extra: level-0 entry points per query from an exact scan of the nodes above level 0 (0: FAISS's greedy descent)."""
        _lib.check(_lib.lib().knn_hnsw_set_entry(self._h, int(coarse_entries)))

    def set_walk(self, expand=0, max_batch=0):
        """knn355 extra: candidates expanded per walker per lock-step round / walkers per batch."""
        _lib.check(_lib.lib().knn_hnsw_set_walk(self._h, int(expand), int(max_batch)))

    def reconstruct_n(self, i0=0, n=None):
        n = self.ntotal - i0 if n is None else n
        out = np.empty((n, self._d), np.float32)
        _lib.check(_lib.lib().knn_flat_reconstruct(_lib.lib().knn_hnsw_storage(self._h), i0, n, out.ctypes.data))
        return out

    def graph(self):
        """(levels int32 [n], offsets int64 [n+1], neighbors int32 [nslots] with -1 = empty,
        cum_nneighbor_per_level int32, assign_probas float64)"""
        L = _lib.lib()
        n, ns = ctypes.c_int64(), ctypes.c_int64()
        nl = ctypes.c_int32()
        _lib.check(L.knn_hnsw_graph_sizes(self._h, ctypes.byref(n), ctypes.byref(ns), ctypes.byref(nl)))
        levels = np.empty(n.value, np.int32)
        offsets = np.empty(n.value + 1, np.int64)
        nbrs = np.empty(ns.value, np.int32)
        cum = np.empty(nl.value, np.int32)
        probas = np.empty(max(nl.value - 1, 0), np.float64)
        _lib.check(L.knn_hnsw_graph_export(self._h, levels.ctypes.data, offsets.ctypes.data, nbrs.ctypes.data,
                                           cum.ctypes.data, probas.ctypes.data))
        return levels, offsets, nbrs, cum, probas

    def stats(self, reset=False):
        pairs, rounds, shrinks = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        g, h = ctypes.c_double(), ctypes.c_double()
        _lib.check(_lib.lib().knn_hnsw_stats(self._h, ctypes.byref(pairs), ctypes.byref(rounds), ctypes.byref(shrinks),
                                             ctypes.byref(g), ctypes.byref(h), 1 if reset else 0))
        return {"pairs": pairs.value, "rounds": rounds.value, "shrinks": shrinks.value, "gpu_s": g.value, "host_s": h.value}

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().knn_hnsw_free(h)
            except Exception:
                pass
            self._h = None


# ---------------------------------------------------------------------------
# write_index / read_index -- FAISS 1.7.2 binary layout (impl/index_write.cpp,
# impl/index_read.cpp; restated from the published format, the FAISS sources are not
# part of the reference tree):
#   fourcc, header {int d; int64 ntotal; int64 dummy=1<<20 (x2); bool is_trained;
#   int metric_type}, then per type
#     "IxFI"/"IxF2"  IndexFlat : vector<float> xb (size_t count + data)
#     "IHNf"         IndexHNSWFlat : HNSW {vector<double> assign_probas; vector<int>
#                    cum_nneighbor_per_level; vector<int> levels (top level + 1);
#                    vector<size_t> offsets; vector<int> neighbors (-1 = empty);
#                    int entry_point, max_level, efConstruction, efSearch, upper_beam}
#                    followed by the storage index ("IxFI"/"IxF2")
#     "IxHe"         IndexLSH (see lsh.py)
# Reference call sites: pfam/proteins_search.py:39-40, seqvec_search/create_index.py:47,
# pfam/search.py:32,34, seqvec_search/main.py:132.
# ---------------------------------------------------------------------------
import struct as _struct


def _w_vec(f, arr, dtype):
    a = np.ascontiguousarray(arr, dtype=dtype)
    f.write(_struct.pack("<Q", a.size))
    f.write(a.tobytes())


def _r_exact(f, nbytes):
    b = f.read(nbytes)
    if len(b) != nbytes:
        raise RuntimeError("read_index: truncated file")
    return b


def _r_vec(f, dtype):
    (n,) = _struct.unpack("<Q", _r_exact(f, 8))
    item = np.dtype(dtype).itemsize
    # (a corrupted count must not turn into a multi-terabyte read request: the payload cannot be longer than the file)
    here = f.tell()
    f.seek(0, 2)
    left = f.tell() - here
    f.seek(here)
    if n * item > left:
        raise RuntimeError("read_index: truncated or corrupted file (vector of %d items, %d bytes left)" % (n, left))
    return np.frombuffer(_r_exact(f, n * item), dtype=dtype)


def _w_header(f, fourcc, d, ntotal, metric):
    f.write(fourcc)
    f.write(_struct.pack("<iqqq?i", d, ntotal, 1 << 20, 1 << 20, True, metric))


def _r_header(f):
    d, ntotal, _, _, trained, metric = _struct.unpack("<iqqq?i", _r_exact(f, 4 + 8 * 3 + 1 + 4))
    if metric > 1:
        _r_exact(f, 4)  # metric_arg
    if d <= 0 or d > (1 << 20) or ntotal < 0 or metric not in (METRIC_INNER_PRODUCT, METRIC_L2):
        raise RuntimeError("read_index: corrupted header (d=%d, ntotal=%d, metric=%d)" % (d, ntotal, metric))
    return d, ntotal, metric


def _write_flat(f, d, metric, rows):
    _w_header(f, b"IxFI" if metric == METRIC_INNER_PRODUCT else b"IxF2", d, rows.shape[0], metric)
    _w_vec(f, rows.reshape(-1), np.float32)


def write_index(index, fname, rows=None):
    """faiss.write_index(index, str(path)).  rows (not in faiss): the float32 rows the index holds, if the caller
    still has them on the host -- saves downloading them again (pfam/proteins_search.py writes the index right
    after adding the array it still holds)."""
    if rows is not None and (rows.dtype != np.float32 or rows.shape != (index.ntotal, index.d) or not rows.flags.c_contiguous):
        raise RuntimeError("write_index: rows must be the index's float32 [ntotal, d] contents")
    with open(str(fname), "wb") as f:
        if isinstance(index, IndexHNSWFlat):
            levels, offsets, nbrs, cum, probas = index.graph()
            p = index.hnsw._params()
            _w_header(f, b"IHNf", index.d, index.ntotal, index.metric_type)
            _w_vec(f, probas, np.float64)
            _w_vec(f, cum, np.int32)
            _w_vec(f, levels + 1, np.int32)
            _w_vec(f, offsets, np.uint64)
            _w_vec(f, nbrs, np.int32)
            f.write(_struct.pack("<iiiii", int(p["entry_point"]), int(p["max_level"]), int(p["efConstruction"]),
                                 int(p["efSearch"]), 1))
            _write_flat(f, index.d, index.metric_type, rows if rows is not None else index.reconstruct_n(0, index.ntotal))
        elif isinstance(index, IndexFlat):
            _write_flat(f, index.d, index.metric_type, rows if rows is not None else index.reconstruct_n(0, index.ntotal))
        elif hasattr(index, "_write"):
            index._write(f)
        else:
            raise RuntimeError(f"write_index: unsupported index type {type(index).__name__}")


def read_index(fname):
    """faiss.read_index(str(path)) -> index (rows go back to the GPU)"""
    with open(str(fname), "rb") as f:
        return _read_index(f)


def _read_index(f):
    fourcc = f.read(4)
    if fourcc in (b"IxFI", b"IxF2"):
        d, ntotal, metric = _r_header(f)
        xb = _r_vec(f, np.float32)
        if xb.size != ntotal * d:
            raise RuntimeError("read_index: IndexFlat payload size mismatch")
        idx = IndexFlat(d, metric)
        if ntotal:
            idx.add(np.ascontiguousarray(xb.reshape(ntotal, d)))
        return idx
    if fourcc == b"IHNf":
        d, ntotal, metric = _r_header(f)
        probas = _r_vec(f, np.float64)
        cum = _r_vec(f, np.int32)
        levels = _r_vec(f, np.int32)
        offsets = _r_vec(f, np.uint64)
        nbrs = _r_vec(f, np.int32)
        entry, max_level, efc, efs, _upper = _struct.unpack("<iiiii", _r_exact(f, 20))
        if levels.size != ntotal or offsets.size != ntotal + 1 or cum.size < 2 or (ntotal and int(offsets[-1]) != nbrs.size):
            raise RuntimeError("read_index: IHNf tables do not fit ntotal = %d" % ntotal)
        if ntotal and (levels.min() < 1 or levels.max() >= cum.size):
            raise RuntimeError("read_index: IHNf level table out of range")
        storage = _read_index(f)
        if not isinstance(storage, IndexFlat) or storage.ntotal != ntotal or storage.d != d:
            raise RuntimeError("read_index: IHNf storage does not match the graph")
        M = int(cum[2] - cum[1]) if cum.size > 2 else int(cum[1] // 2)
        if M < 2 or M > 512 or int(cum[1]) != 2 * M:
            raise RuntimeError("read_index: IHNf neighbour table does not describe an HNSW graph (M = %d)" % M)
        idx = IndexHNSWFlat(d, M, metric)
        L = _lib.lib()
        # the neighbour table is addressed through the per-level slot counts: the file's table must
        # be the one this build derives from M (HnswGraph::init follows FAISS's set_default_probas:
        # float levelMult, float proba, cut at 1e-9)
        own_cum = idx.graph()[3]
        if cum.size != own_cum.size or not np.array_equal(cum, own_cum):
            raise RuntimeError("read_index: IHNf cum_nneighbor_per_level does not match M = %d" % M)
        rows = storage.reconstruct_n(0, ntotal) if ntotal else np.empty((0, d), np.float32)
        if ntotal:
            _lib.check(L.knn_flat_add(L.knn_hnsw_storage(idx._h), rows.ctypes.data, ntotal))
        lv = np.ascontiguousarray(levels - 1, np.int32)
        nb = np.ascontiguousarray(nbrs, np.int32)
        _lib.check(L.knn_hnsw_graph_import(idx._h, ntotal, lv.ctypes.data, nb.ctypes.data, nb.size, max_level, entry))
        _lib.check(L.knn_hnsw_set_ef(idx._h, efs, efc))
        return idx
    if fourcc == b"IxHe":
        from .lsh import IndexLSH
        return IndexLSH._read(f)
    raise RuntimeError(f"read_index: unsupported index type {fourcc!r}")


def __getattr__(name):  # IndexLSH lives in lsh.py (it imports this module)
    if name == "IndexLSH":
        from .lsh import IndexLSH
        return IndexLSH
    raise AttributeError(name)
