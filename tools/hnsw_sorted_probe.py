#!/usr/bin/env python3
"""HNSW on rows that arrive GROUPED BY CLUSTER (a Pfam FASTA lists its families one after the other) against the same rows
shuffled: a batch-synchronous build links a batch against the frozen graph only, so a family that arrives inside one
insertion batch must still end up connected.  usage: hnsw_sorted_probe.py [rows] [rows per cluster] [max_batch]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
per = int(sys.argv[2]) if len(sys.argv) > 2 else 100
mb = int(sys.argv[3]) if len(sys.argv) > 3 else 0
d, k = 256, 100
rng = np.random.default_rng(5)
ncl = n // per
cent = rng.standard_normal((ncl, d), dtype=np.float32)
lab = np.repeat(np.arange(ncl), per)[:n]
x = cent[lab] + 0.35 * rng.standard_normal((n, d), dtype=np.float32)
faiss.normalize_L2(x)
perm = rng.permutation(n)
for name, rows in (("grouped by cluster", x), ("shuffled", np.ascontiguousarray(x[perm]))):
    idx = faiss.IndexHNSWFlat(d, 32, faiss.METRIC_INNER_PRODUCT)
    if mb:
        idx.set_walk(0, mb)
    t0 = time.perf_counter()
    idx.add(rows)
    tb = time.perf_counter() - t0
    flat = faiss.IndexFlat(d, faiss.METRIC_INNER_PRODUCT)
    flat.add(rows)
    q = np.ascontiguousarray(rows[:: max(1, n // 4096)][:4096])
    _, It = flat.search(q, k)
    out = []
    for efs in (128, 512):
        idx.hnsw.efSearch = efs
        _, I = idx.search(q, k)
        out.append(f"efSearch {efs}: recall@{k} {np.mean([len(np.intersect1d(a[a >= 0], b)) for a, b in zip(I, It)]) / k:.4f}")
    print(f"{name:20s} build {tb:.2f} s  " + "  ".join(out), flush=True)
