import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from knn_for_homology_amd import faiss
from oracle import knn_oracle as ko
orc = ko.oracle()
rng = np.random.default_rng(3)
bad = 0
for metric in (0, 1):
    for nb, d, nq, k in ((70_000, 32, 100, 50), (200_000, 24, 128, 100), (150_001, 32, 65, 300), (300_000, 16, 97, 1000), (40_000, 48, 128, 10)):
        xb = rng.standard_normal((nb, d), dtype=np.float32); xb[nb//2:nb//2+100] = xb[:100]
        xq = rng.standard_normal((nq, d), dtype=np.float32); xq[-1] = xb[5]
        idx = faiss.IndexFlat(d, metric); idx.add(xb)
        D, I = idx.search(xq, k)
        Do, Io = orc.flat_search(xb, xq, k, metric)
        ok = np.array_equal(I, Io) and np.array_equal(D.view(np.uint32), Do.view(np.uint32))
        print(metric, nb, d, nq, k, idx.last_scan()["kernel"], idx.last_scan()["grid"], idx.last_seed()["stride"], "OK" if ok else "FAIL", flush=True)
        bad += not ok
print("FAILS", bad)
