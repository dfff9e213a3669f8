#!/usr/bin/env python3
"""The flat all-vs-all on rows GROUPED BY CLUSTER (the order of a Pfam FASTA) against the same rows shuffled: the results are
exact either way -- what could differ is the time (thresholds seeded from a strided sample, candidate arrays sized for an
expected count, the verification that sends a search to the plain path).  Pfam-sized 200 k x 1024, cosine, k = 100 / 1000,
device-resident; plain search in 16384-query launches and the symmetric self-search.  usage: grouped_rows_probe.py [rows per cluster]"""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss, _lib
L = _lib.lib(); dev = torch.device("cuda:0")
per = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n, d = 200_000, 1024
g = torch.Generator(device=dev); g.manual_seed(21)
cent = torch.randn((n // per, d), generator=g, device=dev)
lab = torch.arange(n, device=dev) // per
x = cent[lab] + 0.35 * torch.randn((n, d), generator=g, device=dev)
_lib.check(L.knn_normalize_l2_dev(x.data_ptr(), n, d, None))
perm = torch.randperm(n, generator=g, device=dev)
for name, rows in (("grouped ", x), ("shuffled", x[perm].contiguous())):
    idx = faiss.IndexFlat(d, 0)
    _lib.check(L.knn_flat_add_dev(idx._h, rows.data_ptr(), n, None))
    for k in (100, 1000):
        D = torch.empty((n, k), device=dev, dtype=torch.float32); I = torch.empty((n, k), device=dev, dtype=torch.int64)
        out = []
        for what, call in (("plain", lambda: L.knn_flat_search_dev(idx._h, rows.data_ptr(), n, k, D.data_ptr(), I.data_ptr(), None)),
                           ("self ", lambda: L.knn_flat_search_self_dev(idx._h, k, D.data_ptr(), I.data_ptr()))):
            ts = []
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                _lib.check(call())
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            out.append(f"{what} {1e3 * min(ts[1:]):7.1f} ms")
        ok = bool((I[:, 0] == torch.arange(n, device=dev)).all())
        print(f"{name} clusters of {per}: k {k:4d}: " + "   ".join(out) + f"   self first {ok}   stats {idx.stats() if hasattr(idx, 'stats') else ''}", flush=True)
        del D, I
    del idx
