#!/usr/bin/env python3
"""A/B of library builds on the streaming step (32 / 8 / 1 queries, 10 M and 1.25 M rows): KNN355_LIB picks the build.
usage: KNN355_LIB=... stream_ab.py"""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss, _lib
from knn_for_homology_amd.sharded import ShardedFlatIndex
L = _lib.lib(); dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(23)
d, k = 1024, 100
out = []
for nb in (10_000_000, 1_250_000):
    index = ShardedFlatIndex(d, 0, rank=0, world=1, row_offset=0)
    index.reserve(nb)
    for i0 in range(0, nb, 500_000):
        m = min(500_000, nb - i0)
        x = torch.randn((m, d), generator=g, device=dev)
        _lib.check(L.knn_normalize_l2_dev(x.data_ptr(), m, d, None))
        index.add_dev(x); del x
    for nq in (32, 8, 1):
        q = torch.randn((nq, d), generator=g, device=dev)
        best = None
        for rep in range(3):
            for _ in range(5): index.backend._turn = 0; index.submit(q, k)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            steps = 30
            for _ in range(steps): index.backend._turn = 0; p = index.submit(q, k)
            torch.cuda.synchronize(); t = (time.perf_counter() - t0) / steps
            best = t if best is None else min(best, t)
        out.append(f"{nb//1000}k/{nq}: {1e3*best:.4f}")
    del index; L.knn_trim()
print("  ".join(out))
