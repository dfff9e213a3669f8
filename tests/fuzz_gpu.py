#!/usr/bin/env python3
"""Developer fuzz: random shapes / metrics / k / tuning flags of the flat search against the CPU
oracle, bit for bit.  Data mixes: gaussian, few distinct values (massive ties), duplicated rows,
sorted columns, tiny norms, constant rows.  usage: fuzz_gpu.py [ncases] [seed] [budget seconds]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss  # noqa: E402
from oracle import knn_oracle as ko  # noqa: E402


def run(ncases=200, seed=1, budget_s=None):
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
        d = int(rng.choice([1, 3, 8, 31, 32, 33, 64, 100, 128, 257, 512, 1024]))
        nb = int(rng.choice([1, 2, 7, 63, 64, 65, 255, 256, 257, 1000, 4097, 8193, 20000, 70001]))
        nq = int(rng.choice([1, 2, 31, 32, 33, 40, 48, 49, 64, 65, 80, 96, 97, 127, 128, 129, 170, 300, 1000]))
        if nb * nq * d > 3e10:
            nq = max(1, int(3e10 / (nb * d)))
        k = int(rng.choice([1, 2, 10, 11, 64, 100, 101, 301, 512, 1000, 1537, 2048]))
        metric = int(rng.integers(0, 2))
        flags = int(rng.choice([0, 0, 0, 8, 16, 128, 128, 524288, 524288 | 128, 262144]))  # (524288: the 256 x 256 tile wherever a batch holds > 128 queries)
        qt = int(rng.choice([0, 0, 0, 0, 32, 48, 64, 96, 128, 256]))
        nch = int(rng.choice([0, 0, 0, 1, 3, 17]))
        kind = int(rng.integers(0, 6))
        if kind == 0:
            xb = rng.standard_normal((nb, d), dtype=np.float32)
        elif kind == 1:  # few distinct values: ties everywhere
            xb = rng.integers(-2, 3, (nb, d)).astype(np.float32)
        elif kind == 2:  # duplicated rows
            base = rng.standard_normal((max(1, nb // 7), d), dtype=np.float32)
            xb = base[rng.integers(0, base.shape[0], nb)]
        elif kind == 3:  # sorted columns: every tile beats the previous one
            xb = np.sort(rng.standard_normal((nb, d), dtype=np.float32), axis=0)
        elif kind == 4:  # tiny magnitudes
            xb = (1e-20 * rng.standard_normal((nb, d))).astype(np.float32)
        else:  # constant rows
            xb = np.full((nb, d), 0.5, np.float32)
        xq = rng.standard_normal((nq, d), dtype=np.float32) if kind not in (1, 5) else rng.integers(-2, 3, (nq, d)).astype(np.float32)
        if rng.integers(0, 3) == 0 and nb >= 1:
            xq[: min(nq, nb)] = xb[: min(nq, nb)]
        xb, xq = np.ascontiguousarray(xb), np.ascontiguousarray(xq)
        idx = faiss.IndexFlat(d, metric)
        idx.set_tuning(qt, nch, flags)
        idx.add(xb)
        D, I = idx.search(xq, k)
        Do, Io = orc.flat_search(xb, xq, k, metric)
        ok = np.array_equal(I, Io) and np.array_equal(D.view(np.uint32), Do.view(np.uint32))
        if not ok:
            fails += 1
            bad = np.argwhere(I != Io)
            print(f"FAIL case {case}: d={d} nb={nb} nq={nq} k={k} metric={metric} flags={flags} qt={qt} nch={nch} kind={kind} "
                  f"bad_ids={len(bad)} first={bad[:3].tolist()}", flush=True)
        elif case % 20 == 0:
            print(f"case {case} ok ({time.time()-t0:.0f}s): d={d} nb={nb} nq={nq} k={k} m={metric} flags={flags} kind={kind}", flush=True)
    print(f"FUZZ FAILS: {fails} of {ran}")
    return fails, ran


if __name__ == "__main__":
    _n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    _s = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    _b = float(sys.argv[3]) if len(sys.argv) > 3 else None  # stop starting new cases after that many seconds
    sys.exit(1 if run(_n, _s, _b)[0] else 0)
