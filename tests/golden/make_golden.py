#!/usr/bin/env python3
"""Regenerates tests/golden/ -- run ONLY in the build container (needs /root/reference).

What it does
------------
1. Copies the reference's own test fixtures (data files: ``*.npy`` embeddings and the
   ``*.json`` id lists of ``test-data/{small-random,pfam-20-10,pfam-20-10-sum,pfam-20-dist}``)
   into ``tests/golden/<dataset>/``.  These are the inputs of the reference's
   known-answer tests (``tests/test_main.py:10-27``).
2. Imports the reference's *own* ``seqvec_search.main`` (``faiss_search``,
   ``evaluate_faiss``) with the CPU oracle registered under the name ``faiss``
   (FAISS itself -- faiss-cpu 1.7.2, ``poetry.lock:100-101`` -- is not installed and
   cannot be fetched) and checks that the reference's asserted known answers come
   out: small-random cosine k=5 AUC1/TP lists, pfam-20-10 cosine k=10 mean AUC1
   0.871 / mean TP 0.91.  The ids/scores/auc1s/tps of those runs are saved as
   ``reference_driven.npz``.
   (``matplotlib.rcParams['svg.hashsalt'] = 42`` in ``seqvec_search/utils.py:18`` is
   rejected by matplotlib >= 3.x; the validator is relaxed to accept it, nothing
   else is touched.)
3. Writes oracle outputs for every fixture: cosine (normalised inner product) and L2,
   k in {5, 10, 11, 13, 100, min(1000, nb)}, plus fp64 ground-truth ids, as
   ``oracle_<dataset>.npz``.

Nothing from the reference's *source* is copied; only data files.
"""
import json
import os
import shutil
import sys
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
REF = Path("/root/reference")
DATASETS = ["small-random", "pfam-20-10", "pfam-20-10-sum", "pfam-20-dist"]
KS = [5, 10, 11, 13, 100, 1000]

sys.path.insert(0, str(REPO))
from oracle import knn_oracle as ko  # noqa: E402


def copy_fixtures():
    for ds in DATASETS:
        dst = HERE / ds
        dst.mkdir(exist_ok=True)
        for name in ["train.npy", "test.npy", "train.json", "test.json", "ids_to_family.json"]:
            shutil.copyfile(REF / "test-data" / ds / name, dst / name)
            os.chmod(dst / name, 0o644)


def reference_driven():
    fake = types.ModuleType("faiss")
    for n in dir(ko.OracleFaiss):
        if not n.startswith("__"):
            setattr(fake, n, getattr(ko.OracleFaiss, n))
    sys.modules["faiss"] = fake
    import matplotlib
    matplotlib.use("Agg")
    matplotlib.rcParams.validate["svg.hashsalt"] = lambda v: None if v is None else str(v)
    sys.path.insert(0, str(REF))
    cwd = os.getcwd()
    os.chdir(REF)  # the reference's tests use repo-relative fixture paths
    try:
        import seqvec_search.main as ref_main
        from seqvec_search.data import LoadedData
        out = {}
        for ds, hits in [("small-random", 5), ("pfam-20-10", 10)]:
            data = LoadedData.from_options(path=Path("test-data") / ds, hits=hits)
            queries = np.load(str(data.test))
            haystack = np.load(str(data.train))
            results, scores, _ = ref_main.faiss_search(haystack, queries, data.hits)
            auc1s, tps = ref_main.evaluate_faiss(data, results)
            key = ds.replace("-", "_")
            out[f"{key}_ids"] = results
            out[f"{key}_scores"] = scores
            out[f"{key}_auc1s"] = np.asarray(auc1s)
            out[f"{key}_tps"] = np.asarray(tps)
            # in-place normalisation of the caller's arrays (seqvec_search/main.py:31,34)
            out[f"{key}_queries_after"] = queries
            out[f"{key}_haystack_after"] = haystack
            if ds == "small-random":  # tests/test_main.py:17-18
                assert auc1s == [1.0, 1 / 3, 2 / 3, 0.0, 0.0, 1 / 3], auc1s
                assert tps == [1.0, 2 / 3, 2 / 3, 1.0, 1.0, 1.0], tps
            else:  # tests/test_main.py:26-27
                assert np.mean(auc1s) == 0.871, np.mean(auc1s)
                assert np.mean(tps) == 0.91, np.mean(tps)
        np.savez_compressed(HERE / "reference_driven.npz", **out)
    finally:
        os.chdir(cwd)
    print("reference-driven known answers reproduced")


def oracle_vectors():
    orc = ko.oracle()
    for ds in DATASETS:
        train = np.load(HERE / ds / "train.npy")
        test = np.load(HERE / ds / "test.npy")
        out = {}
        tn, qn = train.copy(), test.copy()
        orc.normalize_l2(tn)
        orc.normalize_l2(qn)
        if ds == "small-random":  # normalised rows themselves, small enough to commit
            out["train_normalized"] = tn
            out["test_normalized"] = qn
        for k in KS:
            k = min(k, train.shape[0])
            D, I = orc.flat_search(tn, qn, k, ko.METRIC_INNER_PRODUCT)
            out[f"ip_k{k}_D"], out[f"ip_k{k}_I"] = D, I
            _, out[f"ip_k{k}_I64"] = ko.exact_knn_f64(tn, qn, k, ko.METRIC_INNER_PRODUCT)
            D, I = orc.flat_search(train, test, k, ko.METRIC_L2)
            out[f"l2_k{k}_D"], out[f"l2_k{k}_I"] = D, I
            _, out[f"l2_k{k}_I64"] = ko.exact_knn_f64(train, test, k, ko.METRIC_L2)
        # cath.search.search semantics: all-vs-all with the self hit stripped (cath/search.py:13-26)
        D, I = orc.flat_search(tn, tn, 11, ko.METRIC_INNER_PRODUCT)
        out["self_ip_k11_D"], out["self_ip_k11_I"] = D, I
        D, I = orc.flat_search(train, train, 11, ko.METRIC_L2)
        out["self_l2_k11_D"], out["self_l2_k11_I"] = D, I
        np.savez_compressed(HERE / f"oracle_{ds}.npz", **out)
        print(ds, "written")


if __name__ == "__main__":
    copy_fixtures()
    reference_driven()
    oracle_vectors()
