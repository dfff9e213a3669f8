"""oracle/cpu_scan.c -- the OpenMP + AVX-512 scan bench.py times as `cpu_baseline` -- is a faithful flat search:
same neighbours as the bit-level oracle wherever fp32 summation order cannot decide, FAISS's padding, both metrics.
(The baseline restates /root/reference/seqvec_search/main.py:45 `index.search` on FAISS's CPU path.)"""
import numpy as np
import pytest

from oracle import cpu_scan as cs
from oracle import knn_oracle as ko


@pytest.mark.parametrize("metric", [ko.METRIC_INNER_PRODUCT, ko.METRIC_L2])
@pytest.mark.parametrize("nb,nq,d,k,threads", [(1003, 7, 1024, 10, 5), (4001, 32, 1024, 100, 8), (11, 6, 1024, 20, 3),
                                                (777, 33, 100, 5, 4), (3, 1, 8, 4, 2), (130, 5, 37, 130, 7)])
def test_cpu_scan_matches_oracle(metric, nb, nq, d, k, threads):
    rng = np.random.default_rng(nb + 7 * k)
    xb = rng.standard_normal((nb, d), dtype=np.float32)
    xq = rng.standard_normal((nq, d), dtype=np.float32)
    D, I = cs.flat_search(xb, xq, k, metric, threads=threads)
    Do, Io = ko.oracle().flat_search(xb, xq, k, metric)
    kk = min(k, nb)
    # padding beyond nb rows: FAISS's -1 / -+FLT_MAX
    assert np.array_equal(I[:, kk:], Io[:, kk:]) and np.array_equal(D[:, kk:], Do[:, kk:])
    ko.compare_tie_tolerant(I[:, :kk], D[:, :kk], xb, xq, metric)  # raises on a violation
    assert ko.recall_at_k(I[:, :kk], Io[:, :kk]) >= 0.999


def test_cpu_scan_on_reference_fixture_reproduces_known_families():
    """pfam-20-10, cosine, k=10: the reference's test asserts mean AUC1 0.871 / TP 0.91 (tests/test_main.py:26-27);
    the committed golden ids are the oracle's -- the CPU baseline returns the same ids."""
    from pathlib import Path
    gold = Path(__file__).parent / "golden" / "pfam-20-10"
    train, test = np.load(gold / "train.npy").copy(), np.load(gold / "test.npy").copy()
    orc = ko.oracle()
    orc.normalize_l2(train)
    orc.normalize_l2(test)
    D, I = cs.flat_search(train, test, 10, ko.METRIC_INNER_PRODUCT, threads=4)
    Do, Io = orc.flat_search(train, test, 10, ko.METRIC_INNER_PRODUCT)
    assert (I == Io).mean() > 0.995  # one rank pair of this fixture sits inside fp32 reorder noise
    assert np.allclose(D, Do, atol=2e-6)


def test_ties_go_to_the_lower_id_and_nan_rows_never_hit():
    xb = np.ones((40, 16), np.float32)
    xb[7] = np.nan
    xq = np.ones((2, 16), np.float32)
    D, I = cs.flat_search(xb, xq, 5, ko.METRIC_INNER_PRODUCT, threads=3)
    assert I.tolist() == [[0, 1, 2, 3, 4]] * 2
    D, I = cs.flat_search(xb, xq, 45, ko.METRIC_L2, threads=3)
    assert 7 not in I and (I[:, 39:] == -1).all()


def test_dram_read_probe_runs():
    rows = cs.Rows(2048, 256, 2)
    rows.fill(np.ones((2048, 256), np.float32))
    assert rows.read_seconds() > 0
    rows.close()
