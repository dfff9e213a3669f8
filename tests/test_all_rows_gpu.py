"""All-rows parity at configuration size and a bounded batch of every fuzzer.

The config-size tests of test_flat_gpu.py / test_configs_gpu.py hold the product to the oracle's BITS on 16-24 sampled
query rows (the C oracle needs seconds per query at these sizes) and to size-independent properties on all rows.  Here
EVERY row of the result is checked against an fp64 ground truth computed on the GPU by torch -- a checker, never the
product: torch.matmul in float64 + torch.topk -- with the tie-tolerant rule of oracle/knn_oracle.py::compare_tie_tolerant
(restated below for torch tensors): ids may only be permuted inside clusters of fp32 noise, every returned distance is
within 1e-5 (relative to the magnitudes its fp32 sum is formed from: BASELINE.json's north-star tolerance) of the true
score of the returned id, no id twice.

The three developer fuzzers (tests/fuzz_*_gpu.py: random shapes / metrics / k / tuning flags / adversarial data against
the oracle, bit for bit) run a time-bounded batch each, so that the driver's `-m gpu` run executes them too."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))

DIST_RTOL = 1e-5   # BASELINE.json north_star: "within 1e-5 on distances"
TAU_REL = 8e-6     # oracle/knn_oracle.py::compare_tie_tolerant


def check_all_rows_fp64(xb, rows_of_queries, I, D, metric, block=1024):
    """xb: float32 [nb, d] torch tensor on the GPU; rows_of_queries: int64 indices into xb (the queries are database rows);
    I / D: numpy [nq, k] results of the product.  Raises AssertionError on the first violated rule; returns statistics."""
    dev = xb.device
    nb, d = xb.shape
    nq, k = I.shape
    kk = min(k, nb)
    xb64 = xb.double()
    bn2 = (xb64 * xb64).sum(1)
    permuted, max_rank_err, max_dist_rel = 0, 0.0, 0.0
    for b0 in range(0, nq, block):
        b1 = min(nq, b0 + block)
        qidx = torch.as_tensor(rows_of_queries[b0:b1], device=dev)
        q64 = xb64[qidx]
        qn2 = (q64 * q64).sum(1)
        s = q64 @ xb64.T                                    # fp64 inner products [m, nb]
        if metric == 1:
            s = qn2[:, None] + bn2[None, :] - 2.0 * s       # squared L2
            s.clamp_(min=0.0)
        got = torch.as_tensor(I[b0:b1, :kk], device=dev)
        Dg = torch.as_tensor(D[b0:b1, :kk], device=dev).double()
        assert bool(((got >= 0) & (got < nb)).all()), "id out of range"
        srt = got.sort(1).values
        assert bool((srt[:, 1:] != srt[:, :-1]).all()), "duplicate ids in a result row"
        D_true, I_true = torch.topk(s, kk, dim=1, largest=(metric == 0), sorted=True)
        if metric == 0:
            scale = torch.clamp(qn2.sqrt()[:, None] * bn2.sqrt()[got], min=1.0)
            scale_max = torch.clamp(qn2.sqrt() * bn2.max().sqrt(), min=1.0)
        else:
            scale = torch.clamp(qn2[:, None] + bn2[got], min=1.0)
            scale_max = torch.clamp(qn2 + bn2.max(), min=1.0)
        tau = TAU_REL * (d ** 0.5) * scale_max
        true_of_got = torch.gather(s, 1, got)
        rank_err = (true_of_got - D_true).abs()
        assert bool((rank_err <= tau[:, None]).all()), f"rank error {float(rank_err.max())} exceeds tau {float(tau.min())} (queries {b0}..{b1})"
        derr = (Dg - true_of_got).abs() / scale
        assert bool((derr <= DIST_RTOL).all()), f"distance error {float(derr.max())} relative to its scale (queries {b0}..{b1})"
        permuted += int((got != I_true).sum())
        max_rank_err = max(max_rank_err, float(rank_err.max()))
        max_dist_rel = max(max_dist_rel, float(derr.max()))
        del s, D_true, I_true, true_of_got, rank_err, derr
    if k > kk:
        assert (I[:, kk:] == -1).all(), "unfilled slots must be id -1"
    return {"permuted": permuted, "max_rank_err": max_rank_err, "max_dist_err_rel": max_dist_rel}


def test_checker_rejects_a_wrong_result():
    """the checker itself: a swapped-in far row, a perturbed distance and a duplicated id are each refused"""
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    x = torch.randn((3000, 64), generator=g, device=dev)
    s = x.double() @ x.double().T
    Dt, It = torch.topk(s, 10, dim=1)
    I, D = It.cpu().numpy(), Dt.float().cpu().numpy()
    rows = np.arange(3000)
    check_all_rows_fp64(x, rows, I, D, 0)
    bad = I.copy()
    bad[17, 9] = int(torch.argmin(s[17]))
    with pytest.raises(AssertionError, match="rank error"):
        check_all_rows_fp64(x, rows, bad, D, 0)
    badD = D.copy()
    badD[5, 3] += 1e-2
    with pytest.raises(AssertionError, match="distance error"):
        check_all_rows_fp64(x, rows, I, badD, 0)
    dup = I.copy()
    dup[8, 4] = dup[8, 3]
    with pytest.raises(AssertionError, match="duplicate"):
        check_all_rows_fp64(x, rows, dup, D, 0)


def test_cath20_sized_every_row_against_fp64(gpu_faiss):
    """BASELINE configs[1] (cath/search.py:13-26): 14433 x 1024 all-vs-all -- squared L2 with k = 301 through the plain
    search AND the symmetric self-search, cosine with k = 11 (the reference's default hits = 10 + self) -- all 14433 rows."""
    dev = torch.device("cuda:0")
    n, d = 14433, 1024
    xh = np.random.default_rng(20).standard_normal((n, d), dtype=np.float32)
    x = torch.from_numpy(xh).to(dev)
    rows = np.arange(n)
    idx = gpu_faiss.IndexFlat(d, gpu_faiss.METRIC_L2)
    idx.add(xh)
    D, I = idx.search(xh, 301)
    st = check_all_rows_fp64(x, rows, I, D, 1)
    assert (I[:, 0] == rows).all() and (D[:, 0] == 0).all(), "every row finds itself first at distance exactly 0"
    Ds, Is = idx.search_self(301)
    assert idx.last_scan()["kernel"].endswith("_sym")
    assert np.array_equal(Is, I) and np.array_equal(Ds.view(np.uint32), D.view(np.uint32)), "symmetric self-search = plain search, bit for bit"
    print("L2 k=301:", st)
    xn = xh.copy()
    gpu_faiss.normalize_L2(xn)
    cos = gpu_faiss.IndexFlat(d, gpu_faiss.METRIC_INNER_PRODUCT)
    cos.add(xn)
    Dc, Ic = cos.search_self(11)
    st = check_all_rows_fp64(torch.from_numpy(xn).to(dev), rows, Ic, Dc, 0)
    assert (Ic[:, 0] == rows).all()
    print("cosine k=11:", st)


def test_pfam_sized_slice_every_row_against_fp64(gpu_faiss):
    """BASELINE configs[2] at the reference's own k (pfam/proteins_search.py:49: k = 1000): 200 000 clustered rows with
    0.5 % exact duplicates, cosine; the all-vs-all self-search's rows 0, 49, 98, ... (4096 of them) are checked in full
    -- every one of their 1000 hits -- and so is the same slice searched as a 4096-query batch."""
    dev = torch.device("cuda:0")
    n, d, k, ncent = 200_000, 1024, 1000, 2000
    g = torch.Generator(device=dev)
    g.manual_seed(21)
    cent = torch.randn((ncent, d), generator=g, device=dev)
    which = torch.randint(0, ncent, (n,), generator=g, device=dev)
    x = cent[which] + 0.35 * torch.randn((n, d), generator=g, device=dev)
    x[-1000:] = x[:1000]
    del cent, which
    xh = x.cpu().numpy()
    idx = gpu_faiss.IndexFlat(d, gpu_faiss.METRIC_INNER_PRODUCT)
    idx.add(xh)
    idx.normalize_rows()
    idx.reconstruct_into(xh)
    x = torch.from_numpy(xh).to(dev)
    sample = np.arange(0, n, 48)[:4096]
    D, I = idx.search_self(k)
    st = check_all_rows_fp64(x, sample, I[sample], D[sample], 0, block=512)
    print("all-vs-all k=1000, 4096 rows:", st)
    # duplicates: row i and row n - 1000 + i tie exactly -- the lower id first
    assert (I[:1000, 0] == np.arange(1000)).all() and (I[:1000, 1] == np.arange(n - 1000, n)).all()
    assert (I[n - 1000:, 0] == np.arange(1000)).all() and (I[n - 1000:, 1] == np.arange(n - 1000, n)).all()
    Db, Ib = idx.search(xh[sample], k)
    assert np.array_equal(Ib, I[sample]) and np.array_equal(Db.view(np.uint32), D[sample].view(np.uint32)), "a batch of the same queries: the same bits"


@pytest.mark.parametrize("which,ncases,budget_s", [("fuzz_gpu", 400, 25.0), ("fuzz_sym_gpu", 60, 20.0), ("fuzz_stream_gpu", 60, 25.0)])
def test_bounded_fuzz_batch(gpu_faiss, which, ncases, budget_s):
    """tests/fuzz_*_gpu.py, a time-bounded batch (seed 4): HIP path vs oracle bit for bit on random shapes, metrics, k, tuning
    flags and adversarial data (ties everywhere, duplicated rows, sorted columns, constant rows, tiny magnitudes)."""
    mod = __import__(which)
    fails, ran = mod.run(ncases, 4, budget_s)
    assert fails == 0 and ran >= 5, (fails, ran)
