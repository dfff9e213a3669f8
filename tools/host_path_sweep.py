#!/usr/bin/env python3
"""One-GPU sweep of the HOST entry points (numpy in, numpy out -- what the reference's scripts call): IndexFlat.search and
IndexHNSWFlat.search over batch sizes and k on a 200 k x 1024 database; the device-resident time of the same flat search
beside it.  What is left between the two columns is PCIe + staging + allocation."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss, _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
n, d = 200_000, 1024
rng = np.random.default_rng(21)
cent = rng.standard_normal((2000, d), dtype=np.float32)
x = cent[rng.integers(0, 2000, n)] + 0.35 * rng.standard_normal((n, d), dtype=np.float32)
faiss.normalize_L2(x)
flat = faiss.IndexFlat(d, faiss.METRIC_INNER_PRODUCT)
flat.add(x)
hnsw = faiss.IndexHNSWFlat(d, 32, faiss.METRIC_INNER_PRODUCT)
hnsw.hnsw.efSearch = 256
t0 = time.perf_counter()
hnsw.add(x)
print(f"hnsw build {time.perf_counter() - t0:.2f} s", flush=True)


def best(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


for k in (10, 100, 1000):
    for nq in (1, 8, 32, 100, 1000, 10_000, 50_000):
        q = x[:nq].copy()
        t_host = best(lambda: flat.search(q, k), reps=3 if nq >= 10_000 else 7)
        qd = torch.from_numpy(q).to(dev)
        D = torch.empty((nq, k), device=dev, dtype=torch.float32)
        I = torch.empty((nq, k), device=dev, dtype=torch.int64)

        def dev_search():
            _lib.check(L.knn_flat_search_dev(flat._h, qd.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), None))
        t_dev = best(dev_search, reps=3 if nq >= 10_000 else 7)
        t_h = best(lambda: hnsw.search(q, k), reps=3 if nq >= 10_000 else 7)
        print(f"k={k:>4} nq={nq:>6}: flat host {1e3 * t_host:9.3f} ms  device-resident {1e3 * t_dev:9.3f} ms  (x{t_host / t_dev:4.1f})   "
              f"hnsw host {1e3 * t_h:9.3f} ms ({nq / t_h:10.0f} q/s)", flush=True)
