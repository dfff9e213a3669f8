#!/usr/bin/env python3
"""Developer probe for rocprofv3 --kernel-trace --stats: a CATH20-sized L2 k=301 search and a streaming
search (32 queries x 2.5 M rows), a few repetitions each, so that the per-kernel table shows what the sample
pass, the scan and the final selection cost.  Usage: select_probe.py [cath] [stream]"""
import sys
import time
from pathlib import Path

import torch
import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss, _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
which = sys.argv[1:] or ["cath", "stream"]


def timed(idx, q, k, reps=6):
    nq = q.shape[0]
    D = torch.empty((nq, k), device=dev, dtype=torch.float32)
    I = torch.empty((nq, k), device=dev, dtype=torch.int64)
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(L.knn_flat_search_dev(idx._h, q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), None))
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts[1:])), idx.last_scan(), idx.last_seed()


if "cath" in which:
    x = torch.from_numpy(np.random.default_rng(20).standard_normal((14433, 1024), dtype=np.float32)).to(dev)
    idx = faiss.IndexFlat(1024, faiss.METRIC_L2)
    _lib.check(L.knn_flat_add_dev(idx._h, x.data_ptr(), 14433, None))
    for flags in (0, 512):
        idx.set_tuning(0, 0, flags)
        ms, info, seed = timed(idx, x, 301)
        print(f"cath L2 k=301 flags={flags}: search {ms:.3f} ms, scan {info['ms']:.3f} ms, chunks {info['nchunks']}, seed {seed}", flush=True)
if "stream" in which:
    nb = 2_500_000
    idx = faiss.IndexFlat(1024, 0)
    _lib.check(L.knn_flat_reserve(idx._h, nb))
    g = torch.Generator(device=dev)
    g.manual_seed(23)
    for i0 in range(0, nb, 500_000):
        xx = torch.randn((500_000, 1024), generator=g, device=dev)
        _lib.check(L.knn_normalize_l2_dev(xx.data_ptr(), 500_000, 1024, None))
        _lib.check(L.knn_flat_add_dev(idx._h, xx.data_ptr(), 500_000, None))
        del xx
    q = torch.randn((32, 1024), generator=g, device=dev)
    _lib.check(L.knn_normalize_l2_dev(q.data_ptr(), 32, 1024, None))
    ms, info, seed = timed(idx, q, 100, reps=10)
    print(f"stream 32 x {nb} k=100: search {ms:.3f} ms, scan {info['ms']:.3f} ms, chunks {info['nchunks']}, seed {seed}", flush=True)
