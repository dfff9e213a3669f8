#!/usr/bin/env python3
"""One-GPU sweeps over the shapes the headline numbers do not cover: k (1 ... 2048), d (64 ... 2048) and small
databases (the latency floor of IndexFlat.search through host pointers).  Prints JSON (profiles/rNN_shape_sweep.json).
usage: shape_sweep.py [k] [d] [small]"""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss, _lib  # noqa: E402
from knn_for_homology_amd.sharded import ShardedFlatIndex  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
what = [a for a in sys.argv[1:]] or ["k", "d", "small"]
g = torch.Generator(device=dev)
g.manual_seed(7)
out = {}


def build(nb, d, metric):
    index = ShardedFlatIndex(d, metric, rank=0, world=1, row_offset=0)
    index.reserve(nb)
    step = max(1, min(nb, (1 << 29) // (4 * d)))
    for i0 in range(0, nb, step):
        m = min(step, nb - i0)
        x = torch.randn((m, d), generator=g, device=dev)
        _lib.check(L.knn_normalize_l2_dev(x.data_ptr(), m, d, None))
        index.add_dev(x)
        del x
    torch.cuda.synchronize()
    return index


def timed(index, q, k, target_s=0.3):
    for _ in range(2):
        index.backend._turn = 0
        index.submit(q, k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    index.backend._turn = 0
    index.submit(q, k)
    torch.cuda.synchronize()
    one = time.perf_counter() - t0
    steps = max(3, min(50, int(target_s / max(one, 1e-5))))
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            index.backend._turn = 0
            pend = index.submit(q, k)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / steps
        best = t if best is None else min(best, t)
    pend.result()
    return best


if "k" in what:
    rows = []
    d = 1024
    for nb, nqs in ((2_000_000, (32, 1024)), (14433, (14433,))):
        for metric in (faiss.METRIC_INNER_PRODUCT, faiss.METRIC_L2):
            index = build(nb, d, metric)
            for nq in nqs:
                q = torch.randn((nq, d), generator=g, device=dev)
                _lib.check(L.knn_normalize_l2_dev(q.data_ptr(), nq, d, None))
                for k in (1, 10, 100, 301, 1000, 1536, 1537, 2048):
                    t = timed(index, q, k)
                    info = index.local.last_scan()
                    rec = {"nb": nb, "nq": nq, "metric": "IP" if metric == 0 else "L2", "k": k, "ms": 1e3 * t, "kernel": info["kernel"],
                           "grid": info["grid"], "hbm_frac": nb * d * 4 / t / 8e12, "mfma_frac": 2.0 * nq * nb * d / t / 157.3e12,
                           "seed": index.local.last_seed()}
                    rows.append(rec)
                    print(json.dumps(rec), file=sys.stderr, flush=True)
            del index
            torch.cuda.empty_cache()
            L.knn_trim()
    out["k"] = rows

if "d" in what:
    rows = []
    for d in (64, 100, 128, 256, 512, 768, 1024, 1280, 2048):
        nb = (8 << 30) // (4 * ((d + 31) // 32 * 32))  # 8 GB of rows
        index = build(nb, d, faiss.METRIC_INNER_PRODUCT)
        for nq in (32, 1024):
            q = torch.randn((nq, d), generator=g, device=dev)
            t = timed(index, q, 100)
            info = index.local.last_scan()
            dp = (d + 31) // 32 * 32
            rec = {"d": d, "nb": nb, "nq": nq, "k": 100, "ms": 1e3 * t, "kernel": info["kernel"], "grid": info["grid"],
                   "hbm_frac": nb * dp * 4 / t / 8e12, "mfma_frac": 2.0 * nq * nb * dp / t / 157.3e12}
            rows.append(rec)
            print(json.dumps(rec), file=sys.stderr, flush=True)
        del index
        torch.cuda.empty_cache()
        L.knn_trim()
    out["d"] = rows

if "small" in what:
    rows = []
    rng = np.random.default_rng(3)
    for nb, nq, d, k in ((200, 200, 1024, 10), (11, 6, 1024, 5), (1000, 1, 1024, 10), (1000, 100, 1024, 100), (10_000, 10, 1024, 100),
                         (10_000, 1000, 1024, 100), (100_000, 1, 1024, 100), (100_000, 200, 1024, 100)):
        for metric in (0, 1):
            xb = rng.standard_normal((nb, d), dtype=np.float32)
            xq = rng.standard_normal((nq, d), dtype=np.float32)
            t0 = time.perf_counter()
            idx = faiss.IndexFlat(d, metric)
            idx.add(xb)
            t_add = time.perf_counter() - t0
            for _ in range(3):
                idx.search(xq, k)
            ts = []
            for _ in range(20):
                t0 = time.perf_counter()
                idx.search(xq, k)
                ts.append(time.perf_counter() - t0)
            rec = {"nb": nb, "nq": nq, "d": d, "k": k, "metric": "IP" if metric == 0 else "L2", "create_add_ms": 1e3 * t_add,
                   "search_ms_median": 1e3 * float(np.median(ts)), "search_ms_min": 1e3 * min(ts), "kernel": idx.last_scan()["kernel"],
                   "scan_ms": idx.last_scan()["ms"]}
            rows.append(rec)
            print(json.dumps(rec), file=sys.stderr, flush=True)
            del idx
    out["small"] = rows
print(json.dumps(out))
