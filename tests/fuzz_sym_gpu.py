#!/usr/bin/env python3
"""Developer fuzz: whole-index self-searches (symmetric launch) against the plain launch of the same handle and the CPU
oracle on sampled rows, bit for bit.  usage: fuzz_sym_gpu.py [ncases] [seed]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss  # noqa: E402
from oracle import knn_oracle as ko  # noqa: E402


def run(ncases=40, seed=1, budget_s=None):
    """-> (failures, cases run); budget_s: stop starting new cases after that many seconds (the -m gpu tests run a bounded batch)"""
    rng = np.random.default_rng(seed)
    orc = ko.oracle()
    fails = 0
    t0 = time.time()
    ran = 0
    for case in range(ncases):
        if budget_s is not None and time.time() - t0 > budget_s:
            break
        ran = case + 1
        d = int(rng.choice([8, 31, 32, 64, 100, 128, 256, 1024]))
        n = int(rng.choice([3000, 3001, 4096, 5000, 6143, 8192, 8193, 9000, 12799, 12800, 14433, 20000, 33333])) if d < 1024 else int(rng.choice([3000, 8192, 14433]))
        k = 0
        k = min(n - 1, int(rng.choice([1, 2, 10, 11, 100, 101, 301, 512, 1000, 1400, 1536])))
        metric = int(rng.integers(0, 2))
        kind = int(rng.integers(0, 5))
        if kind == 0:
            x = rng.standard_normal((n, d), dtype=np.float32)
        elif kind == 1:  # few distinct values: ties everywhere
            x = rng.integers(-2, 3, (n, d)).astype(np.float32)
        elif kind == 2:  # duplicated rows
            base = rng.standard_normal((max(1, n // 7), d), dtype=np.float32)
            x = base[rng.integers(0, base.shape[0], n)]
        elif kind == 3:  # clustered, families contiguous
            cent = rng.standard_normal((max(2, n // 150), d), dtype=np.float32)
            x = cent[np.sort(rng.integers(0, cent.shape[0], n))] + 0.3 * rng.standard_normal((n, d), dtype=np.float32)
        else:  # sorted columns
            x = np.sort(rng.standard_normal((n, d), dtype=np.float32), axis=0)
        x = np.ascontiguousarray(x)
        idx = faiss.IndexFlat(d, metric)
        idx.add(x)
        if rng.integers(0, 4) == 0:
            idx.set_tuning(0, 0, 524288)  # (the symmetric launch on 256-row tiles)
        D, I = idx.search_self(k)
        kern, seedinfo = idx.last_scan()["kernel"], idx.last_seed()
        idx.set_tuning(0, 0, 1024)
        Dp, Ip = idx.search_self(k)
        sample = rng.choice(n, 16, replace=False)
        Do, Io = orc.flat_search(x, x[sample], k, metric, l2_mode=1)  # (the search was one big batch: the norm formula)
        ok = (np.array_equal(I, Ip) and np.array_equal(D.view(np.uint32), Dp.view(np.uint32))
              and np.array_equal(I[sample], Io) and np.array_equal(D[sample].view(np.uint32), Do.view(np.uint32)))
        if not ok:
            fails += 1
            print(f"FAIL case {case}: d={d} n={n} k={k} metric={metric} kind={kind} kernel={kern} seed={seedinfo} "
                  f"diff_vs_plain={int((I != Ip).sum())} diff_vs_oracle={int((I[sample] != Io).sum())}", flush=True)
        elif case % 5 == 0:
            print(f"case {case} ok ({time.time()-t0:.0f}s): d={d} n={n} k={k} m={metric} kind={kind} {kern} redo={seedinfo['stat_redo']}", flush=True)
    print(f"FUZZ FAILS: {fails} of {ran}")
    return fails, ran


if __name__ == "__main__":
    _n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    _s = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    sys.exit(1 if run(_n, _s)[0] else 0)
