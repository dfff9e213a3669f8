#!/usr/bin/env python3
"""Two-lane submit throughput on one GPU for an 8-GPU shard (1.25 M rows) and the full 10 M database,
for several forced chunk counts (0 = planner): does a multi-round scan let the other lane's small
launches in?  usage: lane_probe.py [rows ...]"""
import os
import sys
import time
from pathlib import Path

import torch
import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss, _lib  # noqa: E402
from knn_for_homology_amd.sharded import ShardedFlatIndex  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
sizes = [int(a) for a in sys.argv[1:]] or [1_250_000, 10_000_000]
g = torch.Generator(device=dev); g.manual_seed(3)
q = torch.randn((32, 1024), generator=g, device=dev)
_lib.check(L.knn_normalize_l2_dev(q.data_ptr(), 32, 1024, None))
for nb in sizes:
    idx = ShardedFlatIndex(1024, faiss.METRIC_INNER_PRODUCT)
    idx.reserve(nb)
    for i0 in range(0, nb, 1 << 20):
        m = min(1 << 20, nb - i0)
        x = torch.randn((m, 1024), generator=g, device=dev)
        _lib.check(L.knn_normalize_l2_dev(x.data_ptr(), m, 1024, None))
        torch.cuda.synchronize()
        idx.add_dev(x)
        del x
    idx.backend.next_lane(); idx.backend.next_lane()
    for nch in ([int(a) for a in os.environ.get('LANE_CHUNKS','').split(',') if a] or (0, 768, 1024, 1536)):
        for lane_index, _ in idx.backend._lanes:
            lane_index.set_tuning(0, nch, 0)
        for mode in ("two lanes", "one lane"):
            res = []
            for rep in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(40):
                    if mode == "one lane":
                        idx.backend._turn = 0
                    p = idx.submit(q, 100)
                torch.cuda.synchronize()
                res.append((time.perf_counter() - t0) / 40)
            info = idx.local.last_scan()
            print(f"nb={nb:9d} forced chunks={nch:5d} (-> {info['nchunks']}) {mode}: {1e3*min(res):7.3f} ms/step  {32/min(res):8.0f} q/s", flush=True)
    del idx
