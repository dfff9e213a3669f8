#!/usr/bin/env python3
"""Developer check: HIP flat search vs the CPU oracle over a matrix of shapes/configs.
Prints one line per case; exits non-zero on the first mismatch class."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent  # repo root (this script lives in tests/: it uses the oracle)
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss, _lib  # noqa: E402
from oracle import knn_oracle as ko  # noqa: E402

orc = ko.oracle()
fails = 0


def case(name, xb, xq, k, metric, qt=0, nchunks=0, flags=0):
    global fails
    idx = faiss.IndexFlat(xb.shape[1], metric)
    idx.set_tuning(qt, nchunks, flags)
    idx.add(xb)
    t0 = time.time()
    D, I = idx.search(xq, k)
    t1 = time.time()
    Do, Io = orc.flat_search(xb, xq, k, metric)
    ok_i = np.array_equal(I, Io)
    ok_d = np.array_equal(D.view(np.uint32), Do.view(np.uint32))
    info = idx.last_scan()
    msg = f"{name:44s} nq={xq.shape[0]:6d} nb={xb.shape[0]:7d} d={xb.shape[1]:5d} k={k:5d} m={metric} {info['kernel']} chunks={info['nchunks']} flags={flags} ids={'OK' if ok_i else 'BAD'} dist={'OK' if ok_d else 'BAD'} {1e3*(t1-t0):8.1f}ms scan={info['ms']:.2f}ms"
    print(msg, flush=True)
    if not (ok_i and ok_d):
        fails += 1
        bad = np.argwhere(I != Io)
        print("   first id mismatches:", bad[:5].tolist(), "n_bad_ids", len(bad), "n_bad_d", int((D.view(np.uint32) != Do.view(np.uint32)).sum()))
        if len(bad):
            q, j = bad[0]
            print("   got ", I[q, max(0, j - 2):j + 3], D[q, max(0, j - 2):j + 3])
            print("   want", Io[q, max(0, j - 2):j + 3], Do[q, max(0, j - 2):j + 3])
        else:
            bd = np.argwhere(D.view(np.uint32) != Do.view(np.uint32))
            q, j = bd[0]
            print("   dist got", D[q, j], "want", Do[q, j], "at", q, j)
    del idx


def main():
    print("devices:", _lib.device_count(), flush=True)
    rng = np.random.default_rng(1)
    # normalize
    x = rng.standard_normal((1000, 1024), dtype=np.float32)
    a, b = x.copy(), x.copy()
    faiss.normalize_L2(a)
    orc.normalize_l2(b)
    print("normalize_L2 d=1024:", "OK" if np.array_equal(a, b) else "BAD", flush=True)
    if not np.array_equal(a, b):
        bad = a != b
        print("  bad elems", bad.sum(), "of", a.size, "bad rows", bad.any(1).sum(), "max abs", np.abs(a - b).max())
        r = np.argwhere(bad.any(1))[0, 0]
        print("  row", r, "ratio a/x", (a[r, :4] / x[r, :4]).tolist(), "b/x", (b[r, :4] / x[r, :4]).tolist())
        nr = orc.norm_l2sqr(x)[r]
        print("  nr", nr, "inv64", 1.0 / np.sqrt(np.float32(nr), dtype=np.float32).astype(np.float64), "inv32", np.float32(1.0 / np.float64(np.sqrt(np.float32(nr)))))
        j = np.argwhere(bad[r])[0, 0]
        print("  x", x[r, j].view(np.uint32) if False else x[r, j], "a", a[r, j], "b", b[r, j], "x*inv", np.float32(x[r, j]) * np.float32(1.0 / np.float64(np.sqrt(np.float32(nr)))))
    x = rng.standard_normal((333, 100), dtype=np.float32)
    x[5] = 0
    a, b = x.copy(), x.copy()
    faiss.normalize_L2(a)
    orc.normalize_l2(b)
    print("normalize_L2 d=100:", "OK" if np.array_equal(a, b) else "BAD", flush=True)
    x = rng.standard_normal((50, 37), dtype=np.float32)
    a, b = x.copy(), x.copy()
    faiss.normalize_L2(a)
    orc.normalize_l2(b)
    print("normalize_L2 d=37:", "OK" if np.array_equal(a, b) else "BAD", flush=True)

    gold = ROOT / "tests" / "golden"
    for ds in ["small-random", "pfam-20-10"]:
        tr = np.load(gold / ds / "train.npy")
        te = np.load(gold / ds / "test.npy")
        trn, ten = tr.copy(), te.copy()
        orc.normalize_l2(trn)
        orc.normalize_l2(ten)
        for flags in (0,):
            case(f"{ds} cosine", trn, ten, min(10, tr.shape[0]), 0, flags=flags)
            case(f"{ds} l2", tr, te, min(10, tr.shape[0]), 1, flags=flags)
        case(f"{ds} cosine k>nb", trn, ten, tr.shape[0] + 7, 0)
    xb = rng.standard_normal((5000, 1024), dtype=np.float32)
    xq = rng.standard_normal((300, 1024), dtype=np.float32)
    for qt in (32, 64, 128):
        for nch in (1, 3):
            for metric in (0, 1):
                case(f"rand qt={qt} nch={nch}", xb, xq[:qt + 5], 100, metric, qt=qt, nchunks=nch)
    case("rand auto k=1000", xb, xq, 1000, 0)
    case("rand auto k=301 l2", xb, xq, 301, 1)
    case("rand auto k=1", xb, xq[:7], 1, 0)
    case("rand auto k=2048", xb, xq[:40], 2048, 1)
    xb2 = rng.standard_normal((2049, 100), dtype=np.float32)
    xq2 = rng.standard_normal((77, 100), dtype=np.float32)
    case("d=100", xb2, xq2, 13, 0)
    case("d=100 l2", xb2, xq2, 13, 1)
    xb3 = rng.standard_normal((700, 37), dtype=np.float32)
    case("d=37 l2", xb3, xb3[:50].copy(), 11, 1)
    # duplicates -> ties broken by lower id
    xb4 = xb[:3000].copy()
    xb4[1500:1600] = xb4[100:200]
    xb4[2900:2950] = xb4[100:150]
    case("duplicates ip", xb4, xb4[100:164].copy(), 20, 0)
    case("duplicates l2", xb4, xb4[100:164].copy(), 20, 1, nchunks=4)
    # bigger
    xb5 = rng.standard_normal((60000, 1024), dtype=np.float32)
    case("60k x 32q k=100", xb5, xq[:32], 100, 0)
    case("60k x 200q k=100 l2", xb5, xq[:200], 100, 1)
    # seeded (recursive) searches: big enough that the strided sample kicks in
    xb6 = rng.standard_normal((40000, 128), dtype=np.float32)
    xq6 = rng.standard_normal((70, 128), dtype=np.float32)
    case("seeded 40k d=128 k=10", xb6, xq6, 10, 0, flags=16)
    case("seeded 40k d=128 k=100 l2", xb6, xq6, 100, 1, flags=16)
    case("seeded 40k d=128 k=100 q32", xb6, xq6[:20], 100, 0, flags=16)
    case("auto 40k d=128 k=10 q32", xb6, xq6[:20], 10, 0)
    case("unseeded (flags=8) same", xb6, xq6, 100, 1, flags=8)
    xb7 = np.sort(rng.standard_normal((30000, 64), dtype=np.float32), axis=0)  # adversarial order
    case("sorted columns 30k d=64", xb7, xb7[::500].copy(), 50, 1, flags=16)
    print("FAILS:", fails)
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
