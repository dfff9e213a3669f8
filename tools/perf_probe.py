#!/usr/bin/env python3
"""Developer perf probe: scan-kernel time, GB/s and TFLOP/s for the regimes of SURVEY 8(d).
Usage: perf_probe.py [stream] [cath] [pfam] [flags=N]"""
import ctypes
import sys
import time
from pathlib import Path

import torch  # before the library: one shared HIP runtime
import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss, _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
FLAGS = 0
for a in sys.argv[1:]:
    if a.startswith("flags="):
        FLAGS = int(a.split("=")[1])
which = [a for a in sys.argv[1:] if "=" not in a] or ["stream", "cath", "pfam"]


def make_index(nb, d, metric, normalize=True, seed=23, chunk=1 << 20):
    idx = faiss.IndexFlat(d, metric)
    _lib.check(L.knn_flat_reserve(idx._h, nb))
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    for i0 in range(0, nb, chunk):
        m = min(chunk, nb - i0)
        x = torch.randn((m, d), generator=g, device=dev, dtype=torch.float32)
        if normalize:
            _lib.check(L.knn_normalize_l2_dev(x.data_ptr(), m, d, None))
        torch.cuda.synchronize()
        _lib.check(L.knn_flat_add_dev(idx._h, x.data_ptr(), m, None))
        del x
    return idx


def run(idx, q, k, reps=5, qt=0, nch=0):
    nq, d = q.shape
    idx.set_tuning(qt, nch, FLAGS)
    D = torch.empty((nq, k), device=dev, dtype=torch.float32)
    I = torch.empty((nq, k), device=dev, dtype=torch.int64)
    best = 1e9
    walls = []
    for r in range(reps):
        torch.cuda.synchronize()
        t0 = time.time()
        _lib.check(L.knn_flat_search_dev(idx._h, q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), None))
        torch.cuda.synchronize()
        walls.append(time.time() - t0)
        info = idx.last_scan()
        best = min(best, info["ms"])
    nb = idx.ntotal
    flops = 2.0 * nq * nb * d
    passes = (nq + info["query_tile"] - 1) // info["query_tile"]
    byts = passes * nb * d * 4 + nq * d * 4 + nq * k * 12
    print(f"  nq={nq:6d} nb={nb:8d} k={k:4d} {info['kernel']} chunks={info['nchunks']} grid={info['grid']} "
          f"scan={best:9.3f}ms wall={1e3*min(walls):9.3f}ms  {flops/best/1e9:8.1f} TFLOP/s  {byts/best/1e6:8.1f} GB/s(alg, P={passes})  qps={nq/min(walls):10.1f}",
          flush=True)
    return D, I


if "stream" in which:
    print("== streaming regime: IP, k=100, nb=2M x 1024 (8.2 GB)", flush=True)
    idx = make_index(2_000_000, 1024, 0)
    g = torch.Generator(device=dev); g.manual_seed(24)
    q = torch.randn((1024, 1024), generator=g, device=dev)
    _lib.check(L.knn_normalize_l2_dev(q.data_ptr(), 1024, 1024, None))
    for nq in (1, 8, 16, 32, 64, 128, 1024):
        run(idx, q[:nq].contiguous(), 100)
    for nch in (256, 512):
        print(f"  forced nchunks={nch}")
        run(idx, q[:32].contiguous(), 100, nch=nch)
    del idx
if "cath" in which:
    print("== CATH20-like: 14433 x 1024 all-vs-all", flush=True)
    for metric, norm in ((1, False), (0, True)):
        idx = make_index(14433, 1024, metric, normalize=norm, seed=20)
        xb = torch.from_numpy(idx.reconstruct_n(0, 14433)).to(dev)
        for k in (11, 301):
            run(idx, xb, k)
        del idx
if "pfam" in which:
    print("== Pfam-like: nb=200000 x 1024, IP, nq=16384 batch", flush=True)
    idx = make_index(200_000, 1024, 0, seed=21)
    xb = torch.from_numpy(idx.reconstruct_n(0, 16384)).to(dev)
    for k in (100, 1000):
        run(idx, xb, k, reps=3)
    del idx
