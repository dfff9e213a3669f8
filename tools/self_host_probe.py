#!/usr/bin/env python3
"""Host-level search_self at Pfam size (200 k x 1024, cosine): k = 100 and 1000, repeated calls (the first one pays for
the page-locked result arrays)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss
n, d = 200000, 1024
rng = np.random.default_rng(21)
cent = rng.standard_normal((2000, d), dtype=np.float32)
x = cent[rng.integers(0, 2000, n)] + 0.35 * rng.standard_normal((n, d), dtype=np.float32)
faiss.normalize_L2(x)
idx = faiss.IndexFlat(d, 0)
idx.add(x)
for k in (100, 1000):
    for it in range(3):
        t0 = time.perf_counter()
        D, I = idx.search_self(k)
        t = time.perf_counter() - t0
        print(f"k={k} call {it}: {t:.3f} s, pinned result: {not D.flags.owndata}", flush=True)
        del D, I
