#!/usr/bin/env python3
"""Whole-index self-search: symmetric launch vs plain launch, CATH20-sized (L2 k=301) and Pfam-sized (cosine k=100 / 1000)."""
import sys, time
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss, _lib
L = _lib.lib(); dev = torch.device("cuda:0")
def run(n, d, metric, k, label, clustered=False):
    g = torch.Generator(device=dev); g.manual_seed(20)
    if clustered:
        cent = torch.randn((2000, d), generator=g, device=dev)
        x = cent[torch.randint(0, 2000, (n,), generator=g, device=dev)] + 0.35 * torch.randn((n, d), generator=g, device=dev)
        _lib.check(L.knn_normalize_l2_dev(x.data_ptr(), n, d, None))
    else:
        x = torch.randn((n, d), generator=g, device=dev)
    idx = faiss.IndexFlat(d, metric)
    _lib.check(L.knn_flat_add_dev(idx._h, x.data_ptr(), n, None))
    D = torch.empty((n, k), device=dev, dtype=torch.float32); I = torch.empty((n, k), device=dev, dtype=torch.int64)
    for flags in (0, 1024):
        idx.set_tuning(0, 0, flags)
        ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            _lib.check(L.knn_flat_search_self_dev(idx._h, k, D.data_ptr(), I.data_ptr()))
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        info = idx.last_scan(); t = float(np.median(ts[1:]))
        fl = 2.0 * n * n * d
        print(f"{label} flags={flags}: {info['kernel']:26s} search {1e3*t:9.2f} ms  (last scan launch(es) {info['ms']:8.2f} ms)  {n/t:10.0f} q/s  "
              f"{fl/t/1e12:6.1f} 'TFLOP/s' of the full matrix; run length {info['nchunks']} grid {info['grid']}; seed {idx.last_seed()}", flush=True)
run(14433, 1024, 1, 301, "cath L2 k=301 ")
if len(sys.argv) > 1 and sys.argv[1] == "cath":
    sys.exit(0)
run(14433, 1024, 0, 11, "cath IP k=11  ")
run(200000, 1024, 0, 100, "pfam IP k=100 ", clustered=True)
run(200000, 1024, 0, 1000, "pfam IP k=1000", clustered=True)
