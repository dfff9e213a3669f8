#!/usr/bin/env python3
"""The reference's HNSW run (pfam/proteins_search.py hnsw: M=42, inner product, efSearch=256,
all-vs-all k=1000; source comments: build 15 s, search 77 s, hardware unstated) on synthetic
Pfam-sized data: N clustered rows x 1024."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rng = np.random.default_rng(21)
cent = rng.standard_normal((2000, 1024), dtype=np.float32)
x = cent[rng.integers(0, 2000, n)] + 0.35 * rng.standard_normal((n, 1024), dtype=np.float32)
t0 = time.time(); faiss.normalize_L2(x); print(f"normalize {time.time()-t0:.2f}s", flush=True)
idx = faiss.IndexHNSWFlat(1024, 42, faiss.METRIC_INNER_PRODUCT)
idx.hnsw.efSearch = 256
t0 = time.time(); idx.train(x); idx.add(x); tb = time.time() - t0
print(f"Index creation took {tb:.1f}s  {idx.stats(reset=True)}", flush=True)
t0 = time.time(); D, I = idx.search(x, k); ts = time.time() - t0
st = idx.stats(reset=True)
print(f"Search took {ts:.1f}s ({n/ts:.0f} q/s)  {st}", flush=True)
flat = faiss.IndexFlat(1024, faiss.METRIC_INNER_PRODUCT)
flat.add(x)
t0 = time.time(); Dt, It = flat.search(x[:4096], k); tf = time.time() - t0
hit = sum(len(np.intersect1d(a[a >= 0], b)) for a, b in zip(I[:4096], It)) / It.size
print(f"recall@{k} vs flat (first 4096 queries) = {hit:.4f}; self hit first: {(I[:, 0] == np.arange(n)).mean():.4f}; flat 4096 queries {tf:.2f}s", flush=True)
t0 = time.time(); Dt, It = flat.search(x, k); print(f"flat all-vs-all k={k}: {time.time()-t0:.1f}s", flush=True)
