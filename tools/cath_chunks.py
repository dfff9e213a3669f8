import sys, time
sys.path.insert(0, '/root/repo')
import torch, numpy as np
from knn_for_homology_amd import faiss, _lib
L = _lib.lib(); dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randn((14433, 1024), generator=g, device=dev)
for metric in (1, 0):
    idx = faiss.IndexFlat(1024, metric)
    xx = x.clone()
    if metric == 0: _lib.check(L.knn_normalize_l2_dev(xx.data_ptr(), 14433, 1024, None))
    _lib.check(L.knn_flat_add_dev(idx._h, xx.data_ptr(), 14433, None))
    for k in (11, 301):
        D = torch.empty((14433, k), device=dev); I = torch.empty((14433, k), device=dev, dtype=torch.int64)
        for nch in (0, 4, 5, 6, 7, 8, 9, 13, 18):
            idx.set_tuning(0, nch, 0)
            best = 1e9; wall = 1e9
            for r in range(4):
                torch.cuda.synchronize(); t0 = time.time()
                _lib.check(L.knn_flat_search_dev(idx._h, xx.data_ptr(), 14433, k, D.data_ptr(), I.data_ptr(), None))
                torch.cuda.synchronize(); wall = min(wall, time.time() - t0)
                best = min(best, idx.last_scan()["ms"])
            info = idx.last_scan()
            print(f"metric={metric} k={k:4d} nch={nch:2d} -> chunks={info['nchunks']:3d} grid={info['grid']:5d} scan={best:.3f} ms wall={1e3*wall:.3f} ms", flush=True)
