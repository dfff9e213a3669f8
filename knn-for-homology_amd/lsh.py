"""faiss.IndexLSH(d, nbits) on the GPU (sign bits of a random orthonormal projection,
Hamming top-k).  Reference call sites: seqvec_search/create_index.py:41-47 (build + write),
pfam/search.py:27-37 (1024 bits, k = 1000), pfam/proteins_search.py:25-26 (2048 bits).

FAISS seeds its RandomRotationMatrix from its own generator, so its exact bits cannot be
reproduced without FAISS; here the matrix is the Q factor of a ``numpy.random.default_rng(5)``
Gaussian matrix (FAISS also uses seed 5).  Parity for this index is therefore stated as
recall against the exact flat search, not code equality.  The file format written by
``write_index`` is FAISS's "IxHe" layout, so FAISS can load the index.
"""
import ctypes
import struct

import numpy as np

from . import _lib
from .faiss import Index, METRIC_L2, _check_matrix, _r_header, _r_vec, _w_header, _w_vec


def random_rotation(d, nbits, seed=5):
    """[nbits, d] float32 with orthonormal rows (nbits <= d) or orthonormal columns (nbits > d)."""
    n = max(d, nbits)
    g = np.random.default_rng(seed).standard_normal((n, n))
    q, r = np.linalg.qr(g)
    q = q * np.sign(np.diag(r))  # unique factorisation
    return np.ascontiguousarray(q[:nbits, :d], dtype=np.float32)


class IndexLSH(Index):
    def __init__(self, d, nbits, rotate_data=True, train_thresholds=False, _rotation=None):
        if train_thresholds:
            raise RuntimeError("IndexLSH: train_thresholds=True is not supported")
        self._d, self.nbits = int(d), int(nbits)
        self.rotate_data = bool(rotate_data)
        if _rotation is not None:
            self._rot = np.ascontiguousarray(_rotation, np.float32)
        elif rotate_data:
            self._rot = random_rotation(self._d, self.nbits)
        else:
            if nbits > d:
                raise RuntimeError("IndexLSH: nbits > d requires rotate_data")
            self._rot = np.ascontiguousarray(np.eye(self._d, dtype=np.float32)[: self.nbits])
        self._h = ctypes.c_void_p()
        _lib.check(_lib.lib().knn_lsh_create(self._d, self.nbits, self._rot.ctypes.data, ctypes.byref(self._h)))
        self.metric_type = METRIC_L2
        self.is_trained = True

    @property
    def d(self):
        return self._d

    @property
    def ntotal(self):
        return int(_lib.lib().knn_lsh_ntotal(self._h)) if self._h else 0

    @property
    def code_size(self):
        return (self.nbits + 7) // 8

    def add(self, x):
        _check_matrix(x, self._d)
        _lib.check(_lib.lib().knn_lsh_add(self._h, x.ctypes.data, x.shape[0]))

    def search(self, x, k):
        _check_matrix(x, self._d)
        k = int(k)
        D = np.empty((x.shape[0], k), np.float32)
        I = np.empty((x.shape[0], k), np.int64)
        _lib.check(_lib.lib().knn_lsh_search(self._h, x.ctypes.data, x.shape[0], k, D.ctypes.data, I.ctypes.data))
        return D, I

    def codes(self):
        out = np.empty((self.ntotal, self.code_size), np.uint8)
        _lib.check(_lib.lib().knn_lsh_get_codes(self._h, out.ctypes.data, self.code_size))
        return out

    # ---- FAISS "IxHe" layout: header, nbits, rotate_data, train_thresholds, thresholds,
    # code_size, VectorTransform "rrot" {have_bias, A, b, d_in, d_out, is_trained}, codes
    def _write(self, f):
        _w_header(f, b"IxHe", self._d, self.ntotal, METRIC_L2)
        f.write(struct.pack("<i??", self.nbits, self.rotate_data, False))
        _w_vec(f, np.empty(0, np.float32), np.float32)
        f.write(struct.pack("<i", self.code_size))
        f.write(b"rrot")
        f.write(struct.pack("<?", False))
        _w_vec(f, self._rot.reshape(-1), np.float32)
        _w_vec(f, np.empty(0, np.float32), np.float32)
        f.write(struct.pack("<ii?", self._d, self.nbits, True))
        _w_vec(f, self.codes().reshape(-1), np.uint8)

    @classmethod
    def _read(cls, f):
        d, ntotal, _metric = _r_header(f)
        from .faiss import _r_exact
        nbits, rotate, thr = struct.unpack("<i??", _r_exact(f, 6))
        _r_vec(f, np.float32)
        (code_size,) = struct.unpack("<i", _r_exact(f, 4))
        if f.read(4) != b"rrot":
            raise RuntimeError("read_index: IndexLSH without a random rotation block")
        _r_exact(f, 1)
        A = _r_vec(f, np.float32)
        _r_vec(f, np.float32)
        d_in, d_out, _tr = struct.unpack("<ii?", _r_exact(f, 9))
        codes = _r_vec(f, np.uint8)
        if nbits <= 0 or code_size != (nbits + 7) // 8 or codes.size != ntotal * code_size or (A.size and (d_in != d or d_out < nbits or A.size != d_in * d_out)):
            raise RuntimeError("read_index: IxHe tables do not fit the header")
        idx = cls(d, nbits, rotate_data=rotate, _rotation=A.reshape(d_out, d_in)[:nbits] if A.size else None)
        if ntotal:
            c = np.ascontiguousarray(codes.reshape(ntotal, code_size))
            _lib.check(_lib.lib().knn_lsh_add_codes(idx._h, c.ctypes.data, ntotal, code_size))
        return idx

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().knn_lsh_free(h)
            except Exception:
                pass
            self._h = None
