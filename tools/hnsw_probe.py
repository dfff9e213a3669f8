#!/usr/bin/env python3
"""Developer probe: HNSW build/search time and recall vs the flat index."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss  # noqa: E402


def clustered(n, d, ncent, seed):
    rng = np.random.default_rng(seed)
    cent = rng.standard_normal((ncent, d), dtype=np.float32)
    lab = rng.integers(0, ncent, n)
    x = cent[lab] + 0.35 * rng.standard_normal((n, d), dtype=np.float32)
    return x


def recall(I, It):
    hit = 0
    for a, b in zip(I, It):
        hit += len(np.intersect1d(a[a >= 0], b))
    return hit / It.size


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    M = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    nq = min(n, 2000)
    x = clustered(n, d, max(10, n // 100), 21)
    faiss.normalize_L2(x)
    for metric in (faiss.METRIC_INNER_PRODUCT, faiss.METRIC_L2):
        flat = faiss.IndexFlat(d, metric)
        flat.add(x)
        t0 = time.time()
        Dt, It = flat.search(x[:nq], 100)
        tf = time.time() - t0
        idx = faiss.IndexHNSWFlat(d, M, metric)
        t0 = time.time()
        idx.add(x)
        tb = time.time() - t0
        st = idx.stats(reset=True)
        levels, offsets, nbrs, cum, _ = idx.graph()
        deg0 = [(nbrs[offsets[i]:offsets[i] + cum[1]] >= 0).sum() for i in range(0, n, max(1, n // 2000))]
        print(f"metric={metric} n={n} d={d} M={M}: build {tb:.2f}s  {st}  max_level={idx.hnsw.max_level} mean deg0={np.mean(deg0):.1f}", flush=True)
        for efs, k, ex in ((16, 10, 4), (64, 10, 4), (256, 100, 1), (256, 100, 2), (256, 100, 4), (256, 100, 8), (256, 10, 4)):
            idx.hnsw.efSearch = efs
            idx.set_walk(ex, 0)
            t0 = time.time()
            D, I = idx.search(x[:nq], k)
            ts = time.time() - t0
            st = idx.stats(reset=True)
            r = recall(I, It[:, :k])
            same = np.array_equal(D[:, 0], Dt[:, 0])
            print(f"   efSearch={efs:4d} k={k:4d} expand={ex}: {ts:.3f}s ({nq/ts:.0f} q/s; flat {nq/tf:.0f} q/s) recall@{k}={r:.4f} top1dist_equal={same} rounds={st['rounds']} pairs={st['pairs']} gpu={st['gpu_s']:.3f}s host={st['host_s']:.3f}s", flush=True)


if __name__ == "__main__":
    main()
