#!/usr/bin/env python3
"""Per-shard step time (32 queries, k=100, IP) for shard sizes a 10 M database gets on 2/4/8
GPUs, with and without threshold seeding; searches are enqueued back to back on a side stream.
usage: shard_probe.py [rows ...]"""
import ctypes
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
sizes = [int(a) for a in sys.argv[1:] if "=" not in a] or [1_250_000, 2_500_000, 5_000_000]
FLAGS = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("flags=")] or [0, 8, 16]
g = torch.Generator(device=dev); g.manual_seed(3)
q = torch.randn((32, 1024), generator=g, device=dev)
_lib.check(L.knn_normalize_l2_dev(q.data_ptr(), 32, 1024, None))
side = torch.cuda.Stream(dev)
for nb in sizes:
    idx = faiss.IndexFlat(1024, 0)
    _lib.check(L.knn_flat_reserve(idx._h, nb))
    for i0 in range(0, nb, 1 << 20):
        m = min(1 << 20, nb - i0)
        x = torch.randn((m, 1024), generator=g, device=dev)
        _lib.check(L.knn_normalize_l2_dev(x.data_ptr(), m, 1024, None))
        torch.cuda.synchronize()
        _lib.check(L.knn_flat_add_dev(idx._h, x.data_ptr(), m, None))
        del x
    keys = torch.empty((32, 100), dtype=torch.int64, device=dev)
    for flags in FLAGS:
        idx.set_tuning(0, 0, flags)
        res = []
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                _lib.check(L.knn_flat_search_keys_dev(idx._h, q.data_ptr(), 32, 100, 0, keys.data_ptr(), ctypes.c_void_p(side.cuda_stream)))
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / 30)
        info = idx.last_scan()
        print(f"nb={nb:9d} flags={flags:2d} step={1e3*min(res):7.3f} ms  scan={info['ms']:.3f} ms chunks={info['nchunks']} -> {nb*4096/min(res)/1e12:.2f} TB/s whole step", flush=True)
    del idx
