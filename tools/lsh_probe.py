#!/usr/bin/env python3
"""LSH build/search timing (pfam/search.py search_index shape: 1024 bits, k = 1000)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
nbits = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
k = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
rng = np.random.default_rng(21)
cent = rng.standard_normal((2000, 1024), dtype=np.float32)
x = cent[rng.integers(0, 2000, n)] + 0.35 * rng.standard_normal((n, 1024), dtype=np.float32)
faiss.normalize_L2(x)
t0 = time.time(); idx = faiss.IndexLSH(1024, nbits); idx.train(x); idx.add(x); print(f"LSH build {n}x1024 -> {nbits} bits: {time.time()-t0:.2f}s", flush=True)
for nq in (4096, n):
    t0 = time.time(); D, I = idx.search(x[:nq], k); t = time.time() - t0
    print(f"LSH search nq={nq} k={k}: {t:.2f}s ({nq/t:.0f} q/s)", flush=True)
flat = faiss.IndexFlat(1024, 0); flat.add(x)
Dt, It = flat.search(x[:2048], 100)
hit = sum(len(np.intersect1d(a, b)) for a, b in zip(I[:2048, :100], It)) / It.size
print(f"recall@100 of LSH top-100 vs flat: {hit:.3f}; self first: {(I[:, 0] == np.arange(len(I))).mean():.3f}")
