"""CPU BASELINE -- TEST / MEASUREMENT INFRASTRUCTURE ONLY.

ctypes binding of ``oracle/libcpu_scan.so`` (``cpu_scan.c``): the OpenMP + AVX-512 flat scan that
``bench.py``'s ``cpu_baseline`` leg times beside the GPU (the reference's CPU path is
``/root/reference/seqvec_search/main.py:45`` ``index.search`` on FAISS's OpenMP + BLAS).  Only
``bench.py`` and ``tests/`` may import this module; the product package never does.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SO = Path(os.environ["CPU_SCAN_SO"]) if os.environ.get("CPU_SCAN_SO") else _HERE / "libcpu_scan.so"  # (sanitizer builds)
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not _SO.exists() or _SO.stat().st_mtime < (_HERE / "cpu_scan.c").stat().st_mtime:
            subprocess.check_call(["make", "-C", str(_HERE), "-s", "libcpu_scan.so"])
        L = ctypes.CDLL(str(_SO))
        f32p = ctypes.POINTER(ctypes.c_float)
        L.cpu_scan_alloc.restype = ctypes.c_void_p
        L.cpu_scan_alloc.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32]
        L.cpu_scan_free.restype = None
        L.cpu_scan_free.argtypes = [ctypes.c_void_p]
        L.cpu_scan_copy.restype = None
        L.cpu_scan_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32]
        L.cpu_scan_has_avx512.restype = ctypes.c_int
        L.cpu_scan_max_threads.restype = ctypes.c_int
        L.cpu_scan_read_seconds.restype = ctypes.c_double
        L.cpu_scan_read_seconds.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                            ctypes.POINTER(ctypes.c_double)]
        L.cpu_scan_fma_gflops.restype = ctypes.c_double
        L.cpu_scan_fma_gflops.argtypes = [ctypes.c_int32, ctypes.c_double, ctypes.POINTER(ctypes.c_double)]
        L.cpu_scan_norms.restype = None
        L.cpu_scan_norms.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, f32p, ctypes.c_int32]
        L.cpu_scan_search.restype = ctypes.c_int
        L.cpu_scan_search.argtypes = [ctypes.c_void_p, f32p, ctypes.c_int64, f32p, ctypes.c_int64, ctypes.c_int32,
                                      ctypes.c_int32, ctypes.c_int64, f32p, ctypes.POINTER(ctypes.c_int64), ctypes.c_int32]
        _lib = L
    return _lib


class Rows:
    """n x d float32 rows in memory whose pages were first touched by the `threads` OpenMP threads that will
    scan them (each thread owns a contiguous range: NUMA-local reads)."""

    def __init__(self, n: int, d: int, threads: int):
        self.n, self.d, self.threads = int(n), int(d), int(threads)
        self.ptr = lib().cpu_scan_alloc(self.n, self.d, self.threads)
        if not self.ptr:
            raise MemoryError(f"cpu_scan_alloc({n}, {d})")
        buf = (ctypes.c_float * (self.n * self.d)).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=np.float32).reshape(self.n, self.d)
        self.norms = None

    def fill(self, src: np.ndarray, row0: int = 0):
        """copies src into rows [row0, row0 + len(src)) (plain stores into pages that already have a home)"""
        self.array[row0:row0 + src.shape[0]] = src
        self.norms = None

    def search(self, xq: np.ndarray, k: int, metric: int, threads: int | None = None):
        L = lib()
        xq = np.ascontiguousarray(xq, dtype=np.float32)
        assert xq.ndim == 2 and xq.shape[1] == self.d
        f32p = ctypes.POINTER(ctypes.c_float)
        if metric == 1 and self.norms is None:
            self.norms = np.empty(self.n, np.float32)
            L.cpu_scan_norms(self.ptr, self.n, self.d, self.norms.ctypes.data_as(f32p), self.threads)
        D = np.empty((xq.shape[0], k), np.float32)
        I = np.empty((xq.shape[0], k), np.int64)
        yn = self.norms.ctypes.data_as(f32p) if metric == 1 else None
        rc = L.cpu_scan_search(self.ptr, yn, self.n, xq.ctypes.data_as(f32p), xq.shape[0], self.d, metric, k,
                               D.ctypes.data_as(f32p), I.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                               threads or self.threads)
        if rc != 0:
            raise RuntimeError("cpu_scan_search failed")
        return D, I

    def read_seconds(self, threads: int | None = None) -> float:
        sink = ctypes.c_double()
        return lib().cpu_scan_read_seconds(self.ptr, self.n, self.d, threads or self.threads, ctypes.byref(sink))

    def close(self):
        if self.ptr:
            self.array = None
            lib().cpu_scan_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def flat_search(xb: np.ndarray, xq: np.ndarray, k: int, metric: int, threads: int = 0):
    """one-shot convenience (tests): copies xb into first-touched memory and searches"""
    threads = threads or lib().cpu_scan_max_threads()
    xb = np.ascontiguousarray(xb, dtype=np.float32)
    rows = Rows(xb.shape[0], xb.shape[1], threads)
    rows.fill(xb)
    try:
        return rows.search(xq, k, metric)
    finally:
        rows.close()


def fma_gflops(threads: int, seconds: float = 0.5) -> float:
    """what `threads` threads reach running nothing but vector FMAs out of registers: the compute floor of a scan"""
    sink = ctypes.c_double()
    return lib().cpu_scan_fma_gflops(int(threads), float(seconds), ctypes.byref(sink))
