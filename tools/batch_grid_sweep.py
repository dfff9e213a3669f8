#!/usr/bin/env python3
"""One-GPU sweep of the batch regime over (database rows) x (queries): d = 1024, IP, k = 100 (and 1000), device-resident; and of
the whole-index self-search over n.  TFLOP/s of the full product per cell: low cells are plans worth a look."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss, _lib  # noqa: E402
from knn_for_homology_amd.sharded import ShardedFlatIndex  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
g = torch.Generator(device=dev)
g.manual_seed(13)
d = 1024
what = [a for a in sys.argv[1:] if "=" not in a] or ["grid", "self"]
opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
NQS = [int(v) for v in opts.get("nq", "256,512,1024,2048,4096,10000,16384,20000").split(",")]
NS = [int(v) for v in opts.get("n", "2000,5000,8191,8192,10000,14433,20000,30000,50000,100000").split(",")]


def timed(fn, flop):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    best = None
    reps = 3
    for rep in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        best = t if best is None else min(best, t)
    return best, flop / best / 1e12


if "grid" in what:
    for nb in (50_000, 200_000, 1_000_000):
        index = ShardedFlatIndex(d, faiss.METRIC_INNER_PRODUCT, rank=0, world=1, row_offset=0)
        x = torch.randn((nb, d), generator=g, device=dev)
        _lib.check(L.knn_normalize_l2_dev(x.data_ptr(), nb, d, None))
        index.add_dev(x)
        del x
        for k in (100, 1000):
            for nq in NQS:
                q = torch.randn((nq, d), generator=g, device=dev)
                D = torch.empty((nq, k), device=dev, dtype=torch.float32)
                I = torch.empty((nq, k), device=dev, dtype=torch.int64)
                t, tf = timed(lambda: _lib.check(L.knn_flat_search_dev(index.local._h, q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), None)),
                              2.0 * nq * nb * d)
                info, seed = index.local.last_scan(), index.local.last_seed()
                print(f"nb={nb:>8} nq={nq:>6} k={k:>4}: {1e3 * t:9.3f} ms {tf:6.1f} TFLOP/s  grid {info['grid']:>5} chunks {info['nchunks']:>4} seed {seed['stride']}/{seed['stat_rank']}", flush=True)
        del index
        torch.cuda.empty_cache()
        L.knn_trim()

if "self" in what:
    for n in NS:
        index = faiss.IndexFlat(d, faiss.METRIC_INNER_PRODUCT)
        x = torch.randn((n, d), generator=g, device=dev)
        _lib.check(L.knn_normalize_l2_dev(x.data_ptr(), n, d, None))
        _lib.check(L.knn_flat_add_dev(index._h, x.data_ptr(), n, None))
        for k in (11, 101, 301, 1001):
            if k >= n:
                continue
            D = torch.empty((n, k), device=dev, dtype=torch.float32)
            I = torch.empty((n, k), device=dev, dtype=torch.int64)
            t, tf = timed(lambda: _lib.check(L.knn_flat_search_self_dev(index._h, k, D.data_ptr(), I.data_ptr())), 2.0 * n * n * d)
            info, seed = index.last_scan(), index.last_seed()
            print(f"self n={n:>7} k={k:>4}: {1e3 * t:9.3f} ms {tf:6.1f} TFLOP/s of the full product  {info['kernel']} grid {info['grid']} seed {seed['stride']}/{seed['stat_rank']}", flush=True)
        del index, x
        torch.cuda.empty_cache()
        L.knn_trim()
