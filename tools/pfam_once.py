#!/usr/bin/env python3
"""One Pfam-sized all-vs-all (200 k x 1024 clustered, cosine, plain search in 16384-query launches), device-resident, twice --
for a kernel trace (rocprofv3 --kernel-trace -- python3 tools/pfam_once.py [k] [flags]; tools/kernel_avg.py reads it)."""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss, _lib
L = _lib.lib(); dev = torch.device("cuda:0")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 100
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n, d = 200_000, 1024
g = torch.Generator(device=dev); g.manual_seed(21)
cent = torch.randn((2000, d), generator=g, device=dev)
x = cent[torch.randint(0, 2000, (n,), generator=g, device=dev)] + 0.35 * torch.randn((n, d), generator=g, device=dev)
_lib.check(L.knn_normalize_l2_dev(x.data_ptr(), n, d, None))
idx = faiss.IndexFlat(d, 0)
_lib.check(L.knn_flat_add_dev(idx._h, x.data_ptr(), n, None))
D = torch.empty((n, k), device=dev, dtype=torch.float32); I = torch.empty((n, k), device=dev, dtype=torch.int64)
idx.set_tuning(0, 0, flags)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _lib.check(L.knn_flat_search_dev(idx._h, x.data_ptr(), n, k, D.data_ptr(), I.data_ptr(), None))
    torch.cuda.synchronize(); print(f"k {k} flags {flags}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
