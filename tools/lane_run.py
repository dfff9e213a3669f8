#!/usr/bin/env python3
"""Two-lane submit loop on an 8-GPU-sized shard (1.25 M rows), for a kernel trace: do the two lanes' launches overlap?"""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss, _lib
from knn_for_homology_amd.sharded import ShardedFlatIndex
L = _lib.lib(); dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
q = torch.randn((32, 1024), generator=g, device=dev)
_lib.check(L.knn_normalize_l2_dev(q.data_ptr(), 32, 1024, None))
nb = 1_250_000
idx = ShardedFlatIndex(1024, faiss.METRIC_INNER_PRODUCT); idx.reserve(nb)
for i0 in range(0, nb, 1 << 20):
    m = min(1 << 20, nb - i0)
    x = torch.randn((m, 1024), generator=g, device=dev)
    _lib.check(L.knn_normalize_l2_dev(x.data_ptr(), m, 1024, None)); torch.cuda.synchronize()
    idx.add_dev(x); del x
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30):
        p = idx.submit(q, 100)
    torch.cuda.synchronize()
    print(f"two lanes: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms/step", flush=True)
