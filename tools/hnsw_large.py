#!/usr/bin/env python3
"""HNSW where it should pay: N clustered rows (default 10 M x 1024, generated on the device, never in host
memory), M = 32, efSearch = 256, k = 100 -- build time, queries/s and recall@100 against the exact flat search of
the same rows, and the flat search's own queries/s on the same queries.
Usage: hnsw_large.py [rows] [queries] [max_batch]"""
import json
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
max_batch = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
efc = int(sys.argv[4]) if len(sys.argv) > 4 else 40
d, k, M = 1024, 100, 32
ncent = max(2000, n // 100)  # SURVEY 8(d) S-pfam: ~100 rows per centre
g = torch.Generator(device=dev)
g.manual_seed(21)
cent = torch.randn((ncent, d), generator=g, device=dev)
idx = faiss.IndexHNSWFlat(d, M, faiss.METRIC_INNER_PRODUCT)
idx.set_walk(0, max_batch)
idx.hnsw.efConstruction = efc
flat = faiss.IndexFlat(d, faiss.METRIC_INNER_PRODUCT)
_lib.check(L.knn_flat_reserve(flat._h, n))
_lib.check(L.knn_flat_reserve(L.knn_hnsw_storage(idx._h), n))
queries = []
t_build = 0.0
chunk = 500_000
for i0 in range(0, n, chunk):
    m = min(chunk, n - i0)
    which = torch.randint(0, ncent, (m,), generator=g, device=dev)
    x = cent[which] + 0.35 * torch.randn((m, d), generator=g, device=dev)
    _lib.check(L.knn_normalize_l2_dev(x.data_ptr(), m, d, None))
    _lib.check(L.knn_flat_add_dev(flat._h, x.data_ptr(), m, None))
    take = max(1, nq * m // n)
    queries.append(x[torch.randint(0, m, (take,), generator=g, device=dev)].cpu().numpy())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    idx.add_dev(x)
    t_build += time.perf_counter() - t0
    del x, which
    print(f"  linked {i0 + m} rows, {t_build:.1f}s so far", flush=True)
q = np.ascontiguousarray(np.concatenate(queries)[:nq])
nq = q.shape[0]
flat.search(q[:64], k)
t0 = time.perf_counter()
Dt, It = flat.search(q, k)
t_flat = time.perf_counter() - t0
idx.hnsw.efSearch = 256
idx.search(q[:256], k)  # uploads the level-0 lists
res = {}
for efs in (128, 256, 512, 1024):
    idx.hnsw.efSearch = efs
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        D, I = idx.search(q, k)
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    rec = float(np.mean([len(np.intersect1d(a[a >= 0], b)) for a, b in zip(I, It)])) / k
    res[efs] = {"queries_per_s": nq / t, "recall_at_100": rec}
    print(f"  efSearch={efs}: {nq / t:.0f} q/s, recall@100 {rec:.4f}", flush=True)
out = {"max_batch": max_batch, "efConstruction": efc, "rows": n, "d": d, "M": M, "k": k, "nq": nq, "centres": ncent, "build_s": t_build, "build_rows_per_s": n / t_build,
       "flat_queries_per_s": nq / t_flat, "hnsw": res, "stats": idx.stats()}
if hasattr(L, "knn_dev_hnsw_profile"):
    import ctypes
    prof = (ctypes.c_double * 8)()
    L.knn_dev_hnsw_profile(prof, 0)
    names = ["host walkers (upper levels)", "level-0 candidates (device)", "forward selection", "forward links + sort of reverse requests",
             "reverse links appended", "reverse pruning", "mirror + coarse update", "mirror rebuild in front of a batch"]
    out["build_profile_s"] = {n: round(prof[i], 3) for i, n in enumerate(names) if n != "-"}
print(json.dumps(out))
