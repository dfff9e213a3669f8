#!/usr/bin/env python3
"""Whole-index symmetric self-search (cosine, clustered rows, device-resident) at several sizes: the statistical sample's stride
(developer build: KNN355_STAT_STRIDE).  usage: sym_stride_probe.py k n [n ...]"""
import sys, time
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss, _lib
L = _lib.lib(); dev = torch.device("cuda:0")
k = int(sys.argv[1]); d = 1024
for n in [int(a) for a in sys.argv[2:]]:
    g = torch.Generator(device=dev); g.manual_seed(20)
    cent = torch.randn((max(20, n // 100), d), generator=g, device=dev)
    x = cent[torch.randint(0, cent.shape[0], (n,), generator=g, device=dev)] + 0.35 * torch.randn((n, d), generator=g, device=dev)
    _lib.check(L.knn_normalize_l2_dev(x.data_ptr(), n, d, None))
    idx = faiss.IndexFlat(d, 0)
    _lib.check(L.knn_flat_add_dev(idx._h, x.data_ptr(), n, None))
    D = torch.empty((n, k), device=dev, dtype=torch.float32); I = torch.empty((n, k), device=dev, dtype=torch.int64)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _lib.check(L.knn_flat_search_self_dev(idx._h, k, D.data_ptr(), I.data_ptr()))
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"n {n} k {k}: {1e3 * float(np.median(ts[1:])):9.3f} ms  {idx.last_scan()['kernel']}  seed {idx.last_seed()}", flush=True)
    del idx, x, D, I
