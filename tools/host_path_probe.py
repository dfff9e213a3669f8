#!/usr/bin/env python3
"""End-to-end (host numpy buffers in and out, PCIe included) timings of the drop-in entry points."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss  # noqa: E402
from knn_for_homology_amd.cath.search import search  # noqa: E402

rng = np.random.default_rng(20)
x = rng.standard_normal((14433, 1024), dtype=np.float32)
for metric, name in ((faiss.METRIC_INNER_PRODUCT, "cosine"), (faiss.METRIC_L2, "euclidean")):
    for hits in (10, 300):
        ts = []
        for _ in range(4):
            t0 = time.time(); h, s = search(x, hits=hits, metric=metric); ts.append(time.time() - t0)
        print(f"cath.search.search 14433x1024 {name} hits={hits}: {1e3*min(ts):.1f} ms end-to-end (copy+normalise+add+search) -> {14433/min(ts):.0f} q/s", flush=True)
n = 200000
x = rng.standard_normal((n, 1024), dtype=np.float32)
t0 = time.time(); faiss.normalize_L2(x); print(f"normalize_L2 200000x1024 host: {time.time()-t0:.3f}s")
idx = faiss.IndexFlat(1024, faiss.METRIC_INNER_PRODUCT)
t0 = time.time(); idx.add(x); print(f"add: {time.time()-t0:.3f}s")
for k in (100, 1000):
    t0 = time.time(); D, I = idx.search(x, k); t = time.time() - t0
    print(f"IndexFlat.search 200000 x 200000 k={k}: {t:.3f}s end-to-end -> {n/t:.0f} q/s", flush=True)
    t0 = time.time(); D2, I2 = idx.search_self(k); t = time.time() - t0
    assert (I2 == I).all()
    print(f"IndexFlat.search_self 200000 k={k}: {t:.3f}s end-to-end -> {n/t:.0f} q/s", flush=True)
