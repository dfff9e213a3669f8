#!/usr/bin/env python3
"""IndexFlat.search(x, 1000) from host arrays at Pfam size (200 k x 1024 queries against themselves): where the time beyond the
device-resident search goes -- pageable or page-locked queries, fresh or recycled (page-locked) result arrays."""
import sys, time
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss, _lib
n, d, k = 200_000, 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rng = np.random.default_rng(21)
cent = rng.standard_normal((2000, d), dtype=np.float32)
x = cent[rng.integers(0, 2000, n)] + 0.35 * rng.standard_normal((n, d), dtype=np.float32)
faiss.normalize_L2(x)
idx = faiss.IndexFlat(d, 0)
idx.add(x)
xp = torch.from_numpy(x).pin_memory().numpy()
for name, q in (("pageable queries", x), ("page-locked queries", xp)):
    for it in range(3):
        t0 = time.perf_counter()
        D, I = idx.search(q, k)
        t = time.perf_counter() - t0
        print(f"{name}, call {it}: {t:.3f} s  (results page-locked: {not D.flags.owndata})", flush=True)
        del D, I
L = _lib.lib(); dev = torch.device("cuda:0")
xd = torch.from_numpy(x).to(dev)
Dd = torch.empty((n, k), device=dev, dtype=torch.float32); Id = torch.empty((n, k), device=dev, dtype=torch.int64)
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _lib.check(L.knn_flat_search_dev(idx._h, xd.data_ptr(), n, k, Dd.data_ptr(), Id.data_ptr(), None))
    torch.cuda.synchronize(); print(f"device-resident: {time.perf_counter() - t0:.3f} s", flush=True)
