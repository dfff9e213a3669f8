#!/usr/bin/env python3
"""CATH20-sized all-vs-all (14433 x 1024, L2 k=301 and cosine k=11): plain search and whole-index self-search (symmetric
launch), device-resident, for the tuning flags given (A/B on one box).  usage: cath_probe.py [flags ...]"""
import sys, time
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss, _lib
L = _lib.lib(); dev = torch.device("cuda:0")
FLAGS = [int(a) for a in sys.argv[1:]] or [0, 1]
n, d = 14433, 1024
g = torch.Generator(device=dev); g.manual_seed(20)
x = torch.randn((n, d), generator=g, device=dev)
for metric, k in ((1, 301), (0, 11)):
    idx = faiss.IndexFlat(d, metric)
    _lib.check(L.knn_flat_add_dev(idx._h, x.data_ptr(), n, None))
    D = torch.empty((n, k), device=dev, dtype=torch.float32); I = torch.empty((n, k), device=dev, dtype=torch.int64)
    for rep in range(2):
        for flags in FLAGS:
            idx.set_tuning(0, 0, flags)
            res = {}
            for name, call in (("plain", lambda: L.knn_flat_search_dev(idx._h, x.data_ptr(), n, k, D.data_ptr(), I.data_ptr(), None)),
                               ("self ", lambda: L.knn_flat_search_self_dev(idx._h, k, D.data_ptr(), I.data_ptr()))):
                ts = []
                for _ in range(12):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    _lib.check(call())
                    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
                res[name] = (1e3 * float(np.median(ts[3:])), idx.last_scan()["ms"], idx.last_scan()["kernel"])
            print(f"metric {metric} k {k} flags {flags:5d}: " + "   ".join(f"{nm} search {v[0]:6.3f} ms (scan launch {v[1]:6.3f}, {v[2]})" for nm, v in res.items()), flush=True)
