#!/usr/bin/env python3
"""Developer measurement (not collected by pytest): the consumers of (hits, scores) -- SURVEY.md 8(f) N3 / N4 -- at the
sizes the survey names, the HIP / native path through the package's own entry points beside the reference's per-row
Python loops (restated in oracle/consumers_oracle.py, timed on a bounded sample of the rows and extrapolated).
Prints JSON (profiles/rNN_consumers.json).  usage: bench_consumers_gpu.py [rows=200000] [sample=2000]"""
import json
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import evaluation  # noqa: E402
from knn_for_homology_amd.seqvec_search.mmseqs import write_prefilter_db  # noqa: E402
from oracle import consumers_oracle as co  # noqa: E402

opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
N = int(opts.get("rows", 200_000))
S = int(opts.get("sample", 2000))
rng = np.random.default_rng(11)
out = {"rows": N, "cpu_sample_rows": S, "note": "GPU/native: whole arrays through the package's entry points (host numpy in and out, "
       "so PCIe both ways is inside); CPU: the reference's loops on the first cpu_sample_rows rows, seconds scaled to all rows"}


def timed(fn, reps=3):
    best = None
    res = None
    for _ in range(reps):
        t0 = time.perf_counter()
        res = fn()
        t = time.perf_counter() - t0
        best = t if best is None else min(best, t)
    return best, res


def silent(fn):
    import contextlib
    import io

    def run():
        with contextlib.redirect_stdout(io.StringIO()):
            return fn()
    return run


# ---- remove_self_hit, 200 k x 1000 (pfam/proteins.py:85-122): 2 % of the rows have their self hit misplaced, 0.1 % lack it
k = 1000
hits = rng.integers(0, N, (N, k), dtype=np.int64)
scores = np.sort(rng.random((N, k), dtype=np.float32), axis=1)[:, ::-1].copy()
self_ids = np.arange(N, dtype=np.int64)
hits[:, 0] = self_ids
mis = rng.choice(N, N // 50, replace=False)
pos = rng.integers(1, 20, mis.size)
hits[mis, pos] = self_ids[mis]
hits[mis, 0] = (self_ids[mis] + 1) % N
lack = rng.choice(N, N // 1000, replace=False)
hits[lack, :] = np.where(hits[lack, :] == self_ids[lack, None], (self_ids[lack, None] + 7) % N, hits[lack, :])
t_gpu, (ho, so) = timed(silent(lambda: evaluation.remove_self_hit(hits, scores, self_ids)))
t_cpu, (hc, sc, _) = timed(lambda: co.remove_self_hit(hits[:S], scores[:S], self_ids[:S]), reps=1)
assert np.array_equal(ho[:S], hc) and np.array_equal(so[:S].view(np.uint32), sc.view(np.uint32))
byts = N * k * 12 + N * (k - 1) * 12
out["remove_self_hit"] = {"shape": [N, k], "seconds": t_gpu, "algorithmic_bytes": byts, "GBs": byts / t_gpu / 1e9,
                          "cpu_seconds_scaled": t_cpu * N / S, "cpu_sample_seconds": t_cpu}
print(json.dumps(out["remove_self_hit"]), file=sys.stderr, flush=True)

# ---- AUC1 / TP with one family label per id (seqvec_search/main.py:53-82), 200 k x 1000
labels = rng.integers(0, 5000, N).astype(np.int32)
t_gpu, (_, lead, tp) = timed(lambda: evaluation.label_matches(hits, labels, labels, want_matrix=False))
ids = [str(i) for i in range(N)]
fam = {str(i): int(labels[i]) for i in range(N)}
t_cpu, (auc_c, tp_c) = timed(lambda: co.evaluate(fam, ids, ids[:S], hits[:S]), reps=1)
sizes = np.bincount(labels, minlength=5000)[labels[:S]]
assert np.allclose(lead[:S] / sizes, auc_c) and np.allclose(tp[:S] / sizes, tp_c)
byts = N * k * 8 + N * k * 4  # hits + one label gather per hit
out["auc1_tp_labels"] = {"shape": [N, k], "seconds": t_gpu, "algorithmic_bytes": byts, "GBs": byts / t_gpu / 1e9,
                         "cpu_seconds_scaled": t_cpu * N / S, "cpu_sample_seconds": t_cpu}
print(json.dumps(out["auc1_tp_labels"]), file=sys.stderr, flush=True)

# ---- compute_is_correct (cath/cath.py:76-84): 14433 x 300 hits, 4 levels
n_c, k_c = 14433, 300
hits_c = rng.integers(0, n_c, (n_c, k_c), dtype=np.int64)
mapping = rng.integers(0, 40, (n_c, 4)).astype(np.int64)
t_gpu, ic = timed(lambda: evaluation.compute_is_correct(hits_c, mapping))
t_cpu, ic_c = timed(lambda: co.compute_is_correct(hits_c, mapping), reps=1)
assert np.array_equal(ic, ic_c)
out["compute_is_correct"] = {"shape": [n_c, k_c, 4], "seconds": t_gpu, "cpu_seconds": t_cpu}
print(json.dumps(out["compute_is_correct"]), file=sys.stderr, flush=True)

# ---- MMseqs2 prefilter database (seqvec_search/mmseqs/_write_prefilter_db.py:52-97): 200 k queries x 300 hits
k_p = 300
hp = np.ascontiguousarray(hits[:, :k_p])
hp[rng.integers(0, N, N // 100), rng.integers(0, k_p, N // 100)] = -1
sp = np.ascontiguousarray(scores[:, :k_p])
queries = np.arange(N, dtype=np.int64)
tmap = rng.permutation(N).astype(np.int64)
rmap = rng.permutation(N).astype(np.int64)
with tempfile.TemporaryDirectory() as tmp:
    db = Path(tmp) / "pref"
    t_nat, _ = timed(lambda: write_prefilter_db(hp, db, queries, sp, tmap, rmap), reps=2)
    size = db.with_suffix(".0").stat().st_size + db.with_suffix(".index").stat().st_size
    S2 = min(S, 1000)
    t_cpu, (data_c, index_c) = timed(lambda: co.write_prefilter_db(hp[:S2], queries[:S2], sp[:S2], tmap, rmap), reps=1)
    assert db.with_suffix(".0").read_bytes()[: len(data_c)] == bytes(data_c)
out["write_prefilter_db"] = {"shape": [N, k_p], "seconds": t_nat, "file_bytes": size, "MBs_written": size / t_nat / 1e6,
                             "cpu_seconds_scaled": t_cpu * N / S2, "cpu_sample_rows": S2, "cpu_sample_seconds": t_cpu,
                             "what": "native formatter (OpenMP) + one write; the reference formats every line in a Python double loop"}
print(json.dumps(out["write_prefilter_db"]), file=sys.stderr, flush=True)
print(json.dumps(out))
