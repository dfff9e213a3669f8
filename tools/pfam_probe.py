#!/usr/bin/env python3
"""Pfam-sized all-vs-all (200 k x 1024 clustered, cosine): plain search in 16384-query launches and the symmetric self-search,
device-resident, for the tuning flags given (A/B on one box).  usage: pfam_probe.py [k] [flags ...]"""
import sys, time
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss, _lib
L = _lib.lib(); dev = torch.device("cuda:0")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 100
FLAGS = [int(a) for a in sys.argv[2:]] or [0, 256]
n, d = 200_000, 1024
g = torch.Generator(device=dev); g.manual_seed(21)
cent = torch.randn((2000, d), generator=g, device=dev)
x = cent[torch.randint(0, 2000, (n,), generator=g, device=dev)] + 0.35 * torch.randn((n, d), generator=g, device=dev)
_lib.check(L.knn_normalize_l2_dev(x.data_ptr(), n, d, None))
idx = faiss.IndexFlat(d, 0)
_lib.check(L.knn_flat_add_dev(idx._h, x.data_ptr(), n, None))
D = torch.empty((n, k), device=dev, dtype=torch.float32); I = torch.empty((n, k), device=dev, dtype=torch.int64)
for rep in range(2):
    for flags in FLAGS:
        idx.set_tuning(0, 0, flags)
        out = []
        for name, call in (("plain", lambda: L.knn_flat_search_dev(idx._h, x.data_ptr(), n, k, D.data_ptr(), I.data_ptr(), None)),
                           ("self ", lambda: L.knn_flat_search_self_dev(idx._h, k, D.data_ptr(), I.data_ptr()))):
            ts = []
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                _lib.check(call())
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            info = idx.last_scan()
            out.append(f"{name} {1e3*min(ts[1:]):8.1f} ms ({2.0*n*n*d/min(ts[1:])/1e12:6.1f} TFLOP/s of the full matrix; last launch {info['kernel']} {info['ms']:.2f} ms grid {info['grid']} seed {idx.last_seed()['stat_rank']})")
        print(f"k {k} flags {flags:4d}: " + "   ".join(out), flush=True)
