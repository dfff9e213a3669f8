#!/usr/bin/env python3
"""Recall@100/300/1000 of the reference-shaped HNSW (200 k x 1024, M = 42, efSearch 256, k = 1000) for an insertion batch size
(argument; 0 = default) and insertion order (KNN355_HNSW_ORDER=sequential: the rows' own order)."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from knn_for_homology_amd import faiss
n, k = 200000, 1000
rng = np.random.default_rng(21)
cent = rng.standard_normal((2000, 1024), dtype=np.float32)
x = cent[rng.integers(0, 2000, n)] + 0.35 * rng.standard_normal((n, 1024), dtype=np.float32)
faiss.normalize_L2(x)
flat = faiss.IndexFlat(1024, 0); flat.add(x)
q = np.ascontiguousarray(x[:: n // 4096][:4096])  # (spread over the whole file: the first rows are the nodes a sequential build inserts first -- the best connected ones)
_, It = flat.search(q, k)
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 0
idx = faiss.IndexHNSWFlat(1024, 42, 0)
if mb: idx.set_walk(0, mb)
idx.hnsw.efSearch = 256
t0 = time.time(); idx.add(x); tb = time.time() - t0
_, I = idx.search(q, k)
rec = lambda m: sum(len(np.intersect1d(a[:m][a[:m] >= 0], b[:m])) for a, b in zip(I, It)) / (len(I) * m)
print(f"order={os.environ.get('KNN355_HNSW_ORDER','shuffled')} max_batch={mb or 'default'} build {tb:.2f}s recall@100 {rec(100):.4f} @300 {rec(300):.4f} @1000 {rec(1000):.4f}", flush=True)
