#!/usr/bin/env python3
"""ROWS (default 10 M) x 1024 rows, one synchronous search of nq queries (k = 100): which tile the plan takes and what it costs.
Developer build + KNN355_BIG_MIN_NQ=<n> moves the 256 x 256 tile's lower bound; FLAGS=<tuning flags> (524288: the 256 x 256
tile wherever a batch holds more than 128 queries, 262144: never).  usage: [FLAGS=n] big_min_nq_probe.py nq [nq ...]"""
import os, sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss, _lib
L = _lib.lib(); dev = torch.device("cuda:0")
d, k, nb = 1024, int(os.environ.get('K', '100')), int(os.environ.get('ROWS', '10000000'))
idx = faiss.IndexFlat(d, 0)
_lib.check(L.knn_flat_reserve(idx._h, nb))
g = torch.Generator(device=dev); g.manual_seed(23)
for i0 in range(0, nb, 500_000):
    m = min(500_000, nb - i0)
    x = torch.randn((m, d), generator=g, device=dev)
    _lib.check(L.knn_normalize_l2_dev(x.data_ptr(), m, d, None))
    _lib.check(L.knn_flat_add_dev(idx._h, x.data_ptr(), m, None))
    del x
idx.set_tuning(0, 0, int(os.environ.get('FLAGS', '0')))
for nq in [int(a) for a in sys.argv[1:]] or [1024]:
    q = torch.randn((nq, d), generator=g, device=dev)
    _lib.check(L.knn_normalize_l2_dev(q.data_ptr(), nq, d, None))
    D = torch.empty((nq, k), device=dev); I = torch.empty((nq, k), device=dev, dtype=torch.int64)
    ts = []
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _lib.check(L.knn_flat_search_dev(idx._h, q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), None))
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[1]
    print(f"nq {nq}: {1e3 * t:.2f} ms  {2.0 * nq * nb * d / t / 1e12:.1f} TFLOP/s = {2.0 * nq * nb * d / t / 157.3e12:.4f}  {idx.last_scan()['kernel']} seed {idx.last_seed()}", flush=True)
