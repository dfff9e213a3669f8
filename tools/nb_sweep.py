#!/usr/bin/env python3
"""One-GPU sweep over the database size at 32 and 1024 queries (d = 1024, IP, k = 100): where the plans change (static
chunks -> paired walk at 2048 tiles, seeds, tile-minimum seed) the time per row should not jump.  Prints one line per size."""
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
g.manual_seed(9)
d, k = 1024, 100
NQS = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("nq=")] or [32, 1024]
sizes = [int(a) for a in sys.argv[1:] if "=" not in a] or [10_000, 30_000, 65_536, 100_000, 200_000, 300_000, 400_000, 500_000, 524_288, 600_000, 800_000,
                                          1_000_000, 1_250_000, 2_000_000, 3_000_000, 5_000_000]
METRIC = faiss.METRIC_L2 if "metric=l2" in sys.argv else faiss.METRIC_INNER_PRODUCT
index = ShardedFlatIndex(d, METRIC, rank=0, world=1, row_offset=0)
index.reserve(max(sizes))
have = 0
for nb in sizes:
    while have < nb:
        m = min(500_000, nb - have)
        x = torch.randn((m, d), generator=g, device=dev)
        _lib.check(L.knn_normalize_l2_dev(x.data_ptr(), m, d, None))
        index.add_dev(x)
        have += m
        del x
    torch.cuda.synchronize()
    for nq in NQS:
        q = torch.randn((nq, d), generator=g, device=dev)
        best = None
        steps = max(3, min(60, int(0.15 / max(1e-4, nb * (7e-10 if nq == 32 else 1.6e-8)))))
        for rep in range(3):
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
        pend.result()
        info, seed = index.local.last_scan(), index.local.last_seed()
        print(f"nb={nb:>8} nq={nq:>4}: {1e3 * best:8.3f} ms  {1e9 * best / nb:7.3f} ns/row  {nb * d * 4 / best / 1e12:5.2f} TB/s  "
              f"{2.0 * nq * nb * d / best / 1e12:6.1f} TFLOP/s  {info['kernel']} grid {info['grid']} chunks {info['nchunks']} seed {seed['stride']}/{seed['stat_rank']}", flush=True)
