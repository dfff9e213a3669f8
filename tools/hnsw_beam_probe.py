#!/usr/bin/env python3
"""Rows scored per query by the level-0 beam (S-pfam data, M=32) and the bytes that moves."""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
d, k, nq = 1024, 100, 4096
rng = np.random.default_rng(21)
cent = rng.standard_normal((max(2000, n // 100), d), dtype=np.float32)
x = cent[rng.integers(0, len(cent), n)] + 0.35 * rng.standard_normal((n, d), dtype=np.float32)
faiss.normalize_L2(x)
idx = faiss.IndexHNSWFlat(d, 32, faiss.METRIC_INNER_PRODUCT)
idx.add(x)
q = x[rng.integers(0, n, nq)]
idx.search(q[:64], k)
for efs in (128, 256, 512):
    idx.hnsw.efSearch = efs
    idx.search(q, k)
    s0 = idx.stats()
    t0 = time.perf_counter()
    idx.search(q, k)
    t = time.perf_counter() - t0
    s1 = idx.stats()
    rows = (s1["pairs"] - s0["pairs"]) / nq
    print(f"efSearch={efs}: {nq / t:.0f} q/s, {rows:.0f} rows scored per query = {rows * 4096 / 1e6:.1f} MB; whole batch {rows * 4096 * nq / t / 1e12:.2f} TB/s of row reads", flush=True)
