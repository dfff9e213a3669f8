#!/usr/bin/env python3
"""IndexFlat.search_self into host arrays (whole index, symmetric launch): wall time per call.  KNN355_SELF_STREAM=0 in the
environment keeps the result in one piece behind the search (A/B of the grouped, streamed form).
usage: self_host_probe.py [n] [k] [reps]"""
import sys, time
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss
n = int(sys.argv[1]) if len(sys.argv) > 1 else 14433
k = int(sys.argv[2]) if len(sys.argv) > 2 else 301
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 7
d = 1024
rng = np.random.default_rng(20)
cent = rng.standard_normal((max(20, n // 100), d), dtype=np.float32)
x = cent[rng.integers(0, cent.shape[0], n)] + 0.35 * rng.standard_normal((n, d), dtype=np.float32)
idx = faiss.IndexFlat(d, 0)
idx.add(x)
idx.normalize_rows()
idx.search_self(k)
ts = []
for _ in range(reps):
    t0 = time.perf_counter(); D, I = idx.search_self(k); ts.append(time.perf_counter() - t0)
print(f"n {n} k {k}: search_self to host {1e3 * np.median(ts):.2f} ms (min {1e3 * min(ts):.2f})  {idx.last_scan()['kernel']}  "
      f"checksum {int(I[:, 1].sum())}", flush=True)
