#!/usr/bin/env python3
"""One-GPU shard sweep: the step of BASELINE configs[3] (32 queries, k=100, IP, d=1024) on the shard a 10 M-row
database leaves on each of N = 1, 2, 4, 8 GPUs, one search in flight and two (index + view on streams of different
priority), with the tile-minimum seed and with the sample pass (flags 2048).  Prints JSON (profiles/rNN_shard_sweep.json):
the predicted strong-scaling efficiency is t(10 M) / (N t(10 M / N)) before any all-gather.
usage: shard_sweep.py [flags=F ...] [N ...]"""
import ctypes
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
Ns = [int(a) for a in sys.argv[1:] if "=" not in a] or [8, 4, 2, 1]
FLAGS = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("flags=")] or [0, 2048]
d, k, nq, total = 1024, 100, 32, 10_000_000
g = torch.Generator(device=dev)
g.manual_seed(23)
q = torch.randn((nq, d), generator=g, device=dev)
_lib.check(L.knn_normalize_l2_dev(q.data_ptr(), nq, d, None))
out = {"workload": f"{total}x{d} IP k={k}, {nq} queries per step; shard = total / N rows on ONE GPU", "rows": []}
for N in sorted(Ns, reverse=True):
    nb = (total + N - 1) // N
    index = ShardedFlatIndex(d, faiss.METRIC_INNER_PRODUCT, rank=0, world=1, row_offset=0)
    index.reserve(nb)
    for i0 in range(0, nb, 500_000):
        m = min(500_000, nb - i0)
        x = torch.randn((m, d), generator=g, device=dev)
        _lib.check(L.knn_normalize_l2_dev(x.data_ptr(), m, d, None))
        index.add_dev(x)
        del x
    torch.cuda.synchronize()
    for flags in FLAGS:
        index.backend.next_lane(); index.backend.next_lane()  # (creates the two lanes)
        for lane_index, _stream in index.backend._lanes:
            lane_index.set_tuning(0, 0, flags)
        rec = {"N": N, "rows": nb, "flags": flags}
        for lanes in (1, 2):
            best = None
            for rep in range(4):
                steps = 40 if nb < 6_000_000 else 15
                for _ in range(3):
                    if lanes == 1:
                        index.backend._turn = 0
                    index.submit(q, k)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    if lanes == 1:
                        index.backend._turn = 0
                    pend = index.submit(q, k)
                torch.cuda.synchronize()
                t = (time.perf_counter() - t0) / steps
                best = t if best is None else min(best, t)
            pend.result()
            rec[f"ms_lanes{lanes}"] = 1e3 * best
            rec[f"hbm_frac_lanes{lanes}"] = nb * d * 4 / best / 8e12
        info, seed = index.local.last_scan(), index.local.last_seed()
        rec.update({"kernel": info["kernel"], "grid": info["grid"], "scan_ms": info["ms"], "seed": seed})
        out["rows"].append(rec)
        print(json.dumps(rec), file=sys.stderr, flush=True)
    del index
    torch.cuda.empty_cache()
    L.knn_trim()
for flags in FLAGS:
    rows = {r["N"]: r for r in out["rows"] if r["flags"] == flags}
    if 1 in rows:
        for lanes in (1, 2):
            out[f"predicted_strong_scaling_efficiency_flags{flags}_lanes{lanes}"] = {
                str(N): rows[1][f"ms_lanes{lanes}"] / (N * rows[N][f"ms_lanes{lanes}"]) for N in sorted(rows)}
print(json.dumps(out))
