#!/usr/bin/env python3
"""Developer fuzz of the STREAMING regime (few queries, >= 131 k rows: paired workgroups, shared pool of tiles,
tile-minimum seed, batches searched in pieces) against the CPU oracle, bit for bit.  Random shapes / metrics / k / tuning flags (2 = no pool, 4 = no
pairs, 2048 = sample pass instead of the tile-minimum seed, 8 = no seeding at all, bits 12-13 = publication rounds) and
data kinds (gaussian, massive ties, duplicated rows, sorted so that every tile beats the previous one, constant rows,
the best rows packed into one tile).  usage: fuzz_stream_gpu.py [ncases] [seed]"""
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
        d = int(rng.choice([8, 16, 31, 32, 48, 64]))
        nb = int(rng.choice([131_072, 140_000, 200_001, 262_144, 300_000, 400_003, 524_288, 530_000, 600_001, 777_777, 1_048_577, 1_300_000]))
        nq = int(rng.choice([1, 2, 7, 31, 32, 33, 41, 48, 64]))
        k = int(rng.choice([1, 2, 10, 64, 100, 101, 200, 256, 481, 600, 1000, 1536, 1537, 1800, 2048]))
        metric = int(rng.integers(0, 2))
        flags = int(rng.choice([0, 0, 0, 2, 4, 2048, 2048 | 4, 8, 1 << 12, 2 << 12]))
        if rng.integers(0, 5) == 0:  # a batch searched in pieces (the remainder behind the full 128-query tiles on its own)
            nq = int(rng.choice([65, 81, 96, 129, 150, 161, 170, 193, 224, 257]))
            k = min(k, 256)
            d = min(d, 32)
            flags = int(rng.choice([0, 0, 16384, 2048]))
        kind = int(rng.integers(0, 6))
        if kind == 0:
            xb = rng.standard_normal((nb, d), dtype=np.float32)
        elif kind == 1:  # few distinct values: ties everywhere
            xb = rng.integers(-1, 2, (nb, d)).astype(np.float32)
        elif kind == 2:  # duplicated rows
            base = rng.standard_normal((max(1, nb // 50), d), dtype=np.float32)
            xb = base[rng.integers(0, base.shape[0], nb)]
        elif kind == 3:  # sorted: every tile beats the previous one (for one direction of the walk at least)
            xb = np.sort(rng.standard_normal((nb, d), dtype=np.float32), axis=0)
        elif kind == 4:  # constant rows
            xb = np.full((nb, d), 0.25, np.float32)
        else:            # the best rows of every query sit in ONE tile: the published minima say little about the rest
            xb = rng.standard_normal((nb, d), dtype=np.float32)
            t = int(rng.integers(0, nb // 256)) * 256
            xb[t:t + 256] *= 8.0
        xq = rng.standard_normal((nq, d), dtype=np.float32) if kind not in (1, 4) else rng.integers(-1, 2, (nq, d)).astype(np.float32)
        if rng.integers(0, 2) == 0:
            xq[: min(nq, 4)] = xb[nb // 3: nb // 3 + min(nq, 4)]
        xb, xq = np.ascontiguousarray(xb), np.ascontiguousarray(xq)
        idx = faiss.IndexFlat(d, metric)
        idx.set_tuning(0, 0, flags)
        idx.add(xb)
        D, I = idx.search(xq, k)
        info, sd = idx.last_scan(), idx.last_seed()
        Do, Io = orc.flat_search(xb, xq, k, metric)
        ok = np.array_equal(I, Io) and np.array_equal(D.view(np.uint32), Do.view(np.uint32))
        if not ok:
            fails += 1
            bad = np.argwhere(I != Io)
            print(f"FAIL case {case}: d={d} nb={nb} nq={nq} k={k} metric={metric} flags={flags} kind={kind} grid={info['grid']} seed={sd} "
                  f"first bad {bad[:3].tolist()}", flush=True)
        elif case % 5 == 0:
            print(f"case {case} ok ({time.time() - t0:.0f}s): d={d} nb={nb} nq={nq} k={k} m={metric} flags={flags} kind={kind} grid={info['grid']} seed={sd['stride']}", flush=True)
        del idx
    print(f"STREAM FUZZ FAILS: {fails} of {ran}")
    return fails, ran


if __name__ == "__main__":
    _n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    _s = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    sys.exit(1 if run(_n, _s)[0] else 0)
