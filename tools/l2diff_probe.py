#!/usr/bin/env python3
"""The small-batch squared-L2 kernel (sum of squared differences, fewer than 20 queries) beside the norm-formula scan
(flags 32) on a 1.25 M and a 10 M-row database, d = 1024, k = 100, device-resident."""
import sys, time
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss, _lib
L = _lib.lib(); dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(5)
for nb in (1_250_000, 10_000_000):
    idx = faiss.IndexFlat(1024, 1)
    _lib.check(L.knn_flat_reserve(idx._h, nb))
    for i0 in range(0, nb, 500_000):
        m = min(500_000, nb - i0)
        x = torch.randn((m, 1024), generator=g, device=dev)
        _lib.check(L.knn_flat_add_dev(idx._h, x.data_ptr(), m, None))
        del x
    for nq in (1, 4, 8, 12, 16, 19):
        q = torch.randn((nq, 1024), generator=g, device=dev)
        D = torch.empty((nq, 100), device=dev); I = torch.empty((nq, 100), device=dev, dtype=torch.int64)
        for flags in (0, 32):
            idx.set_tuning(0, 0, flags)
            ts = []
            for _ in range(8):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                _lib.check(L.knn_flat_search_dev(idx._h, q.data_ptr(), nq, 100, D.data_ptr(), I.data_ptr(), None))
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            t = float(np.median(ts[2:]))
            print(f"rows {nb:9d} nq {nq:2d} flags {flags:2d}: {idx.last_scan()['kernel']:28s} search {1e3*t:8.3f} ms  scan {idx.last_scan()['ms']:8.3f} ms  {nb*4096/t/1e12:5.2f} TB/s", flush=True)
    del idx
    L.knn_trim()
