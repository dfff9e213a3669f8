#!/usr/bin/env python3
"""Developer tool: per-workgroup stamps of a launch shaped like the statistical seed's SAMPLE PASS of a CATH-sized search (14433
queries x 451 rows, k = 28, no bound: every score is kept; 113 query tiles x 4 one-tile chunks).  Needs the trace build
(KNN355_LIB=<repo>/knn-for-homology_amd/libknn355_trace.so)."""
import ctypes, sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss, _lib  # noqa: E402
L = _lib.lib()
nq, nb, d, k = 14433, 451, 1024, 28
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(21)
x = torch.randn((nb, d), generator=g, device=dev)
q = torch.randn((nq, d), generator=g, device=dev)
L.knn_dev_trace_read.restype = ctypes.c_int
for metric in (1, 0):
    idx = faiss.IndexFlat(d, metric)
    _lib.check(L.knn_flat_add_dev(idx._h, x.data_ptr(), nb, None))
    D = torch.empty((nq, k), device=dev); I = torch.empty((nq, k), device=dev, dtype=torch.int64)
    idx.set_tuning(0, 0, 8)
    for _ in range(4):
        _lib.check(L.knn_flat_search_dev(idx._h, q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), None))
    torch.cuda.synchronize()
    buf = np.zeros((4096, 128), np.uint64)
    grid = L.knn_dev_trace_read(buf.ctypes.data_as(ctypes.c_void_p), 4096)
    t = buf[:grid].astype(np.float64) / 100.0
    t0 = t[:, 0].min()
    info = idx.last_scan()
    print(f"== metric {metric}: grid {grid} kernel {info['kernel']} chunks {info['nchunks']} scan_ms {info['ms']:.4f}")
    def col(c): return t[:, c] - t0
    for name, a, b in (("start", None, 0), ("prologue + K loop", 0, 1), ("filter (all scores kept)", 1, 64), ("barrier + compaction", 64, 2), ("flush", 2, 63)):
        v = col(b) - (col(a) if a is not None else 0)
        print(f"   {name:28s} med {np.median(v):7.1f}  p10 {np.percentile(v, 10):7.1f}  p90 {np.percentile(v, 90):7.1f}  max {v.max():7.1f} us")
    print(f"   end of the last workgroup {col(63).max():.1f} us")
