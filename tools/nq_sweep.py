#!/usr/bin/env python3
"""One-GPU sweep over the batch size between the streaming regime (<= 32 queries: HBM-bound) and the batch regime
(>= 128 queries: MFMA-bound) on a resident database: which kernel serves each batch, what the search costs and how far it
is from max(HBM time, MFMA time).  Prints JSON (profiles/rNN_nq_sweep.json).
Batches of more than 128 queries go through the synchronous entry (what IndexFlat.search runs on device buffers: it may take the
statistical seed and the 256 x 256 tile), smaller ones through the sharded index's lanes as in the bench's timed step
(entry=lanes: everything through the lanes, the sweeps of rounds 3-4).
usage: nq_sweep.py [rows=N] [k=K] [metric=ip|l2] [entry=auto|lanes] [nq ...]"""
import json
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
opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
nb = int(opts.get("rows", 10_000_000))
k = int(opts.get("k", 100))
metric = faiss.METRIC_L2 if opts.get("metric", "ip") == "l2" else faiss.METRIC_INNER_PRODUCT
flags = int(opts.get("flags", 0))
chunks = int(opts.get("chunks", 0))
entry = opts.get("entry", "auto")
NQ = [int(a) for a in sys.argv[1:] if "=" not in a] or [1, 4, 8, 16, 24, 32, 33, 36, 40, 48, 49, 64, 65, 72, 80, 96, 128, 129, 192, 256, 320, 384, 512, 640, 768, 1024, 1536, 2048]
d = 1024
HBM, MFMA = 8e12, 157.3e12
g = torch.Generator(device=dev)
g.manual_seed(23)
index = ShardedFlatIndex(d, metric, rank=0, world=1, row_offset=0)
index.reserve(nb)
for i0 in range(0, nb, 500_000):
    m = min(500_000, nb - i0)
    x = torch.randn((m, d), generator=g, device=dev)
    _lib.check(L.knn_normalize_l2_dev(x.data_ptr(), m, d, None))
    index.add_dev(x)
    del x
torch.cuda.synchronize()
index.local.set_tuning(0, chunks, flags)
out = {"workload": f"{nb}x{d} {'L2' if metric == faiss.METRIC_L2 else 'IP'} k={k}, resident database, device queries", "rows": []}
for nq in NQ:
    q = torch.randn((nq, d), generator=g, device=dev)
    _lib.check(L.knn_normalize_l2_dev(q.data_ptr(), nq, d, None))
    best = None
    steps = max(2, min(20, int(0.4 / (7e-3 * max(1.0, nq / 40.0)))))
    sync = entry == "auto" and nq > 128
    for rep in range(3):
        if sync:
            for _ in range(2):
                index.backend.search(q, k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                index.backend.search(q, k)
            torch.cuda.synchronize()
        else:
            for _ in range(2):
                index.backend._turn = 0
                index.submit(q, k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                index.backend._turn = 0
                pend = index.submit(q, k)
            torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / steps
        best = t if best is None else min(best, t)
    if not sync:
        pend.result()
    info = index.local.last_scan()
    t_hbm = nb * d * 4 / HBM
    t_mfma = 2.0 * nq * nb * d / MFMA
    rec = {"nq": nq, "ms": 1e3 * best, "queries_per_s": nq / best, "kernel": info["kernel"], "grid": info["grid"],
           "query_tile": info["query_tile"], "db_passes": -(-nq // info["query_tile"]), "last_scan_ms": info["ms"],
           "seed": index.local.last_seed(), "entry": "synchronous" if sync else "lanes",
           "floor_ms": 1e3 * max(t_hbm, t_mfma), "bound": "hbm" if t_hbm >= t_mfma else "mfma",
           "frac_of_floor": max(t_hbm, t_mfma) / best}
    out["rows"].append(rec)
    print(json.dumps(rec), file=sys.stderr, flush=True)
print(json.dumps(out))
