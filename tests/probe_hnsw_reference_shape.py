#!/usr/bin/env python3
"""Why is recall@1000 ~0.81-0.84 on the Pfam-sized synthetic set at the reference's HNSW parameters (M = 42, ef = k = 1000)?
The same data STRUCTURE (d = 1024, 100 rows per cluster, random centres) at a size the sequential CPU oracle can build:
device walk vs oracle walk vs flat, at ef = 1000 and ef = 2000.  usage: tests/probe_hnsw_reference_shape.py [n] [d]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss  # noqa: E402
from oracle import knn_oracle as ko  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
k, nq = 1000, 256
rng = np.random.default_rng(21)
cent = rng.standard_normal((n // 100, d), dtype=np.float32)
x = cent[rng.integers(0, n // 100, n)] + 0.35 * rng.standard_normal((n, d), dtype=np.float32)
faiss.normalize_L2(x)
flat = faiss.IndexFlat(d, 0)
flat.add(x)
q = x[:nq]
Dt, It = flat.search(q, k)
rec = lambda I: sum(len(np.intersect1d(a[a >= 0], b)) for a, b in zip(I, It)) / It.size
t0 = time.time(); ref = ko.OracleHNSW(d, 42, 0, 40); ref.add(x); print(f"oracle build {time.time() - t0:.1f}s", flush=True)
for ef in (1000, 2000):
    _, Ir = ref.search(q, k, ef)
    print(f"sequential oracle  ef={ef}: recall@1000 {rec(Ir):.4f}  recall@100 {sum(len(np.intersect1d(a[:100], b[:100])) for a, b in zip(Ir, It)) / (nq * 100):.4f}", flush=True)
idx = faiss.IndexHNSWFlat(d, 42, 0)
idx.add(x)
for ef in (256, 2000):  # (efSearch; the walk uses max(efSearch, k))
    idx.hnsw.efSearch = ef
    _, I = idx.search(q, k)
    print(f"device             ef={max(ef, k)}: recall@1000 {rec(I):.4f}  recall@100 {sum(len(np.intersect1d(a[:100], b[:100])) for a, b in zip(I, It)) / (nq * 100):.4f}", flush=True)
# how flat is the score profile beyond the query's own cluster?
print("score at rank 1 / 50 / 100 / 150 / 500 / 1000 (median over queries):", [round(float(np.median(Dt[:, r - 1])), 4) for r in (1, 50, 100, 150, 500, 1000)])
