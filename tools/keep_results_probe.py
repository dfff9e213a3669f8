#!/usr/bin/env python3
"""cath/search.py:37-50 keeps the hits of every file of a metric until it saves them: the cost per cath.search.search call when
the results of 8 CATH-sized files are kept alive, with the pinned-block cap of _lib.result_array and without it (the round-4
behaviour: a fresh page-locked block per search)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss, _lib
from knn_for_homology_amd.cath.search import search as cath_search
x = np.random.default_rng(20).standard_normal((14433, 1024), dtype=np.float32)
cath_search(x, hits=300, metric=faiss.METRIC_L2)
for cap in (_lib.PINNED_LIVE_PER_CLASS, 1000):
    _lib.PINNED_LIVE_PER_CLASS = cap
    for rep in range(2):
        kept, ts = {}, []
        for f in range(8):
            t0 = time.perf_counter()
            kept[f] = cath_search(x, hits=300, metric=faiss.METRIC_L2)
            ts.append(1e3 * (time.perf_counter() - t0))
        print(f"live blocks per size class <= {cap}: per call " + " ".join(f"{t:.1f}" for t in ts) + f" ms; all eight {sum(ts):.1f} ms", flush=True)
        del kept
