#!/usr/bin/env python3
"""Where the wall time of cath.search.search(numpy [14433,1024], hits=300, L2) goes."""
import sys, time
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss, _lib
from knn_for_homology_amd.cath.search import search as cath_search
x = np.random.default_rng(20).standard_normal((14433, 1024), dtype=np.float32)
for metric in (faiss.METRIC_L2, faiss.METRIC_INNER_PRODUCT):
    cath_search(x, hits=300, metric=metric)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); cath_search(x, hits=300, metric=metric); ts.append(time.perf_counter() - t0)
    print(f"metric {metric}: cath.search end to end {1e3*np.median(ts):.2f} ms")
    def stage(name, fn, reps=7):
        v = []
        r = None
        for _ in range(reps):
            t0 = time.perf_counter(); r = fn(); v.append(time.perf_counter() - t0)
        print(f"   {name:34s} {1e3*np.median(v):7.2f} ms")
        return r
    idx = stage("IndexFlat()", lambda: faiss.IndexFlat(1024, metric))
    def add():
        i = faiss.IndexFlat(1024, metric); i.add(x); return i
    idx = stage("IndexFlat() + add", add)
    if metric == faiss.METRIC_INNER_PRODUCT:
        stage("normalize_rows", idx.normalize_rows)
    stage("search_self(301)", lambda: idx.search_self(301))
    stage("result_array x2 (alloc only)", lambda: (_lib.result_array((14433, 301), np.float32), _lib.result_array((14433, 301), np.int64)))
    D, I = idx.search_self(301)
    stage("slice [:, 1:] (views)", lambda: (I[:, 1:], D[:, 1:]))
