"""GPU parity tests (run with -m gpu on an MI355X): everything goes through the C ABI of
libknn355.so (via the faiss-shaped facade) and is compared BIT FOR BIT -- neighbour ids
and float32 distances -- with the CPU oracle, with the committed golden vectors, and
with the reference's own known answers."""
import numpy as np
import pytest

from conftest import GOLDEN, DATASETS

pytestmark = pytest.mark.gpu


def _load(ds):
    return np.load(GOLDEN / ds / "train.npy"), np.load(GOLDEN / ds / "test.npy")


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _assert_same(D, I, Do, Io):
    assert np.array_equal(I, Io), f"{int((I != Io).sum())} neighbour ids differ"
    assert np.array_equal(_bits(D), _bits(Do)), f"{int((_bits(D) != _bits(Do)).sum())} distances differ"


# ---- the reference's own tests, read through the drop-in entry points -------------
def test_search_ann(gpu_faiss):
    """tests/test_main.py:10-18 of the reference."""
    from knn_for_homology_amd.seqvec_search.data import LoadedData
    from knn_for_homology_amd.seqvec_search.main import faiss_search, evaluate_faiss
    data = LoadedData.from_options(path=GOLDEN / "small-random", hits=5)
    queries = np.load(str(data.test))
    results, scores, search_time = faiss_search(np.load(str(data.train)), queries, data.hits)
    auc1s, tps = evaluate_faiss(data, results)
    assert auc1s == [1.0, 1 / 3, 2 / 3, 0.0, 0.0, 1 / 3]
    assert tps == [1.0, 2 / 3, 2 / 3, 1.0, 1.0, 1.0]
    assert results.dtype == np.int64 and scores.dtype == np.float32 and search_time >= 0


def test_ann_alignment_knn_half(gpu_faiss):
    """tests/test_main.py:21-27 of the reference (the MMseqs2 half needs the external binary)."""
    from knn_for_homology_amd.seqvec_search.data import LoadedData
    from knn_for_homology_amd.seqvec_search.main import faiss_search, evaluate_faiss
    data = LoadedData.from_options(path=GOLDEN / "pfam-20-10", hits=10)
    queries = np.load(str(data.test))
    results, scores, _ = faiss_search(np.load(str(data.train)), queries, data.hits)
    auc1s_ann, tps_ann = evaluate_faiss(data, results)
    assert np.mean(auc1s_ann) == 0.871
    assert np.mean(tps_ann) == 0.91


def test_faiss_search_matches_reference_driven_vectors(gpu_faiss):
    """ids / scores / in-place normalisation equal what the REFERENCE's faiss_search
    produced when driving the oracle (tests/golden/reference_driven.npz)."""
    from knn_for_homology_amd.seqvec_search.main import faiss_search
    g = np.load(GOLDEN / "reference_driven.npz")
    for ds, k in (("small-random", 5), ("pfam-20-10", 10)):
        key = ds.replace("-", "_")
        hay, qs = _load(ds)
        ids, scores, _ = faiss_search(hay, qs, k)
        assert np.array_equal(hay, g[f"{key}_haystack_after"]), "haystack must be normalised in place"
        assert np.array_equal(qs, g[f"{key}_queries_after"]), "queries must be normalised in place"
        _assert_same(scores, ids, g[f"{key}_scores"], g[f"{key}_ids"])


# ---- golden vectors ----------------------------------------------------------------
@pytest.mark.parametrize("ds", DATASETS)
def test_golden_vectors(gpu_faiss, ds):
    g = np.load(GOLDEN / f"oracle_{ds}.npz")
    train, test = _load(ds)
    tn, qn = train.copy(), test.copy()
    gpu_faiss.normalize_L2(tn)
    gpu_faiss.normalize_L2(qn)
    if "train_normalized" in g:
        assert np.array_equal(tn, g["train_normalized"]) and np.array_equal(qn, g["test_normalized"])
    ip = gpu_faiss.IndexFlat(train.shape[1], gpu_faiss.METRIC_INNER_PRODUCT)
    ip.add(tn)
    l2 = gpu_faiss.IndexFlat(train.shape[1], gpu_faiss.METRIC_L2)
    l2.add(train)
    assert ip.ntotal == train.shape[0] and ip.d == 1024
    for k in (5, 10, 11, 13, 100, 1000):
        k = min(k, train.shape[0])
        D, I = ip.search(qn, k)
        _assert_same(D, I, g[f"ip_k{k}_D"], g[f"ip_k{k}_I"])
        D, I = l2.search(test, k)
        _assert_same(D, I, g[f"l2_k{k}_D"], g[f"l2_k{k}_I"])


@pytest.mark.parametrize("ds", DATASETS)
def test_cath_search_semantics(gpu_faiss, ds):
    """cath/search.py:13-26: self hit stripped, (hits, scores) order, input untouched."""
    from knn_for_homology_amd.cath.search import search
    g = np.load(GOLDEN / f"oracle_{ds}.npz")
    train, _ = _load(ds)
    before = train.copy()
    hits, scores = search(train, hits=10, metric=gpu_faiss.METRIC_INNER_PRODUCT)
    assert np.array_equal(train, before), "cosine search must normalise a copy"
    assert hits.shape == (train.shape[0], 10) and hits.dtype == np.int64 and scores.dtype == np.float32
    _assert_same(scores, hits, g["self_ip_k11_D"][:, 1:], g["self_ip_k11_I"][:, 1:])
    hits, scores = search(train, hits=10, metric=gpu_faiss.METRIC_L2)
    assert np.array_equal(train, before)
    _assert_same(scores, hits, g["self_l2_k11_D"][:, 1:], g["self_l2_k11_I"][:, 1:])
    # the stripped column is the self hit, at distance exactly 0 (nrm(x) == dot(x, x))
    idx = gpu_faiss.IndexFlat(1024, gpu_faiss.METRIC_L2)
    idx.add(train)
    D, I = idx.search(train, 1)
    assert (I[:, 0] == np.arange(train.shape[0])).all() and (D == 0).all()


# ---- synthetic shapes against the live oracle ---------------------------------------
CASES = [
    # nq, nb, d, k, query_tile, nchunks
    (1, 1000, 1024, 10, 0, 0),
    (7, 5000, 1024, 1, 0, 0),
    (37, 5000, 1024, 100, 32, 1),
    (37, 5000, 1024, 100, 32, 3),
    (69, 5000, 1024, 100, 64, 1),
    (69, 5000, 1024, 100, 64, 5),
    (133, 5000, 1024, 100, 128, 1),
    (133, 5000, 1024, 100, 128, 3),
    (300, 5000, 1024, 301, 0, 0),
    (300, 5000, 1024, 1000, 0, 0),
    (40, 5000, 1024, 2048, 0, 0),
    (33, 6000, 256, 1536, 0, 0),
    (33, 6000, 256, 1537, 0, 0),
    (130, 6000, 128, 1025, 0, 0),
    (77, 2049, 100, 13, 0, 0),
    (50, 700, 37, 11, 0, 0),
    (260, 4100, 128, 64, 0, 0),
    (32, 60000, 1024, 100, 0, 0),
    (50, 3000, 1280, 11, 0, 0),   # ESM-1b width (cath/search.py searches every embedder's file)
    (20, 1500, 2560, 11, 0, 0),
    (9, 700, 4096, 5, 0, 0),
]


@pytest.mark.parametrize("nq,nb,d,k,qt,nch", CASES)
@pytest.mark.parametrize("metric", [0, 1])
def test_random_vs_oracle(gpu_faiss, oracle, nq, nb, d, k, qt, nch, metric):
    rng = np.random.default_rng(nq * 7919 + nb + d + k)
    xb = rng.standard_normal((nb, d), dtype=np.float32)
    xq = rng.standard_normal((nq, d), dtype=np.float32)
    idx = gpu_faiss.IndexFlat(d, metric)
    idx.set_tuning(qt, nch, 0)
    idx.add(xb)
    D, I = idx.search(xq, k)
    Do, Io = oracle.flat_search(xb, xq, k, metric)
    _assert_same(D, I, Do, Io)


# ---- the builds on 16-query blocks (v_mfma_f32_16x16x4_f32): 33..48 queries -> 48-query tile, 65..96 -> 96-query tile --------
@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("nq,nb,d,k,nch", [
    (33, 5000, 1024, 100, 0), (48, 5000, 1024, 100, 3), (41, 700, 37, 11, 0), (40, 6000, 256, 1536, 0), (45, 9001, 100, 2048, 0),
    (65, 5000, 1024, 100, 0), (96, 5000, 1024, 100, 5), (80, 2049, 100, 13, 0), (90, 6000, 128, 1025, 0), (70, 300, 64, 301, 1),
    (36, 70001, 64, 10, 0),     # >= 64 tiles of 256 rows: paired walk + tile-minimum seed in the 48-query build
    (47, 140000, 32, 600, 0),   # ... with one published key per wave (k beyond the workgroup count)
    (77, 70001, 64, 10, 0),     # 32 k .. 262 k rows, k <= 200: two self-seeding pieces (64 + 13) stay ahead of one unseeded 96-query launch
    (77, 70001, 64, 300, 0),    # ... a larger k: the one 96-query launch under the statistical seed
])
def test_sixteen_query_block_builds_vs_oracle(gpu_faiss, oracle, nq, nb, d, k, nch, metric):
    """a query tile costs its whole width -- 33 queries paid for 64, 65 for 128.  The 48- and 96-query
    tiles are built from 16 x 16 x 4 MFMAs (the same k-ordered fma chain: every bit as before).  The plan picks them by itself;
    flags 131072 switches them off: the same bits either way."""
    rng = np.random.default_rng(nq * 7919 + nb + d + k)
    xb = rng.standard_normal((nb, d), dtype=np.float32)
    xb[nb // 2: nb // 2 + 5] = xb[:5]  # exact duplicates: ties -> lower id
    xq = rng.standard_normal((nq, d), dtype=np.float32)
    xq[:3] = xb[:3]
    idx = gpu_faiss.IndexFlat(d, metric)
    idx.set_tuning(0, nch, 0)
    idx.add(xb)
    D, I = idx.search(xq, k)
    in_pieces = nq > 64 and 32768 <= nb < 262144 and k <= 200 and nch == 0
    assert idx.last_scan()["query_tile"] == (32 if in_pieces else (48 if nq <= 48 else 96)), idx.last_scan()
    Do, Io = oracle.flat_search(xb, xq, k, metric)
    _assert_same(D, I, Do, Io)
    idx.set_tuning(0, nch, 131072)
    D2, I2 = idx.search(xq, k)
    assert idx.last_scan()["query_tile"] in (32, 64, 128)
    _assert_same(D2, I2, Do, Io)
    # forced on a smaller batch (a tile holds any batch that fits it)
    idx.set_tuning(48 if nq <= 48 else 96, nch, 0)
    D3, I3 = idx.search(xq[:nq - 7], k)
    _assert_same(D3, I3, Do[:nq - 7], Io[:nq - 7])


def test_remainders_behind_full_query_tiles_use_their_own_width(gpu_faiss, oracle):
    """128 k + r queries on a database that is streamed from HBM: the remainder r is searched with the narrowest build that
    holds it (32 / 48 / 64 / 96 queries); the pieces return the bits of one launch (flags 16384)."""
    rng = np.random.default_rng(5)
    nb, d, k = 300_000, 32, 20
    xb = rng.standard_normal((nb, d), dtype=np.float32)
    idx = gpu_faiss.IndexFlat(d, 0)
    idx.add(xb)
    for nq, tile in ((128 + 40, 48), (256 + 90, 96), (70, 96), (44, 48)):
        xq = rng.standard_normal((nq, d), dtype=np.float32)
        idx.set_tuning(0, 0, 0)
        D, I = idx.search(xq, k)
        assert idx.last_scan()["query_tile"] == tile, (nq, idx.last_scan())
        idx.set_tuning(0, 0, 16384)
        D1, I1 = idx.search(xq, k)
        _assert_same(D, I, D1, I1)
        Do, Io = oracle.flat_search(xb, xq[-8:], k, 0)
        _assert_same(D[-8:], I[-8:], Do, Io)


def test_query_batching_over_16384(gpu_faiss, oracle):
    """Host searches run in batches of 16384 queries; results must not depend on it."""
    rng = np.random.default_rng(41)
    xb = rng.standard_normal((3000, 32), dtype=np.float32)
    xq = rng.standard_normal((40000, 32), dtype=np.float32)
    for metric in (0, 1):
        idx = gpu_faiss.IndexFlat(32, metric)
        idx.add(xb)
        D, I = idx.search(xq, 7)
        Do, Io = oracle.flat_search(xb, xq, 7, metric)
        _assert_same(D, I, Do, Io)


@pytest.mark.parametrize("k,nq", [(1500, 20), (1434, 32), (1536, 7), (1000, 20), (300, 20)])
@pytest.mark.parametrize("metric", [0, 1])
def test_sorted_database_never_overflows_lists(gpu_faiss, oracle, k, nq, metric):
    """Every new tile beats everything seen so far (rows sorted by increasing similarity to all queries):
    each tile appends all of its rows to every list.  With the 32-query / 256-row tile and k in
    (1433, 1536] a cut to 1.25 k keys would leave less than one tile of room in a 2048-key list --
    the cut must stop at cap - tile rows."""
    rng = np.random.default_rng(k + nq)
    d, nb = 64, 9000
    base = rng.standard_normal(d).astype(np.float32)
    base /= np.linalg.norm(base)
    noise = 0.01 * rng.standard_normal((nb, d)).astype(np.float32)
    if metric == 0:   # inner product with the queries grows with the row number
        xb = (np.linspace(0.1, 3.0, nb, dtype=np.float32)[:, None] * base[None, :] + noise).astype(np.float32)
        xq = (base[None, :] + 0.01 * rng.standard_normal((nq, d)).astype(np.float32)).astype(np.float32)
    else:             # distance to the queries shrinks with the row number
        xb = (np.linspace(3.0, 0.1, nb, dtype=np.float32)[:, None] * base[None, :] + noise).astype(np.float32)
        xq = (0.01 * rng.standard_normal((nq, d))).astype(np.float32)
    idx = gpu_faiss.IndexFlat(d, metric)
    idx.add(xb)
    for nch in (1, 0):
        idx.set_tuning(0, nch, 8)  # no seeding: the lists warm up on their own
        D, I = idx.search(xq, k)
        _assert_same(D, I, *oracle.flat_search(xb, xq, k, metric))


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("k", [10, 100, 301])
def test_statistical_seed_matches_oracle(gpu_faiss, oracle, k, metric):
    """flags=128 forces the statistical seed (batch regime: threshold = j-th score of a strided sample, j << k,
    result verified against it): same bits as the oracle, through the host entry and through search_self."""
    rng = np.random.default_rng(100 + k)
    nb, d, nq = 20000, 64, 300
    xb = rng.standard_normal((nb, d), dtype=np.float32)
    xq = np.concatenate([rng.standard_normal((nq - 40, d), dtype=np.float32), xb[:40]])
    idx = gpu_faiss.IndexFlat(d, metric)
    idx.set_tuning(0, 0, 128)
    idx.add(xb)
    D, I = idx.search(xq, k)
    seed = idx.last_seed()
    assert seed["stride"] > 0 and 0 < seed["stat_rank"] < k, seed
    _assert_same(D, I, *oracle.flat_search(xb, xq, k, metric))
    Ds, Is = idx.search_self(k, 0, 500)
    _assert_same(Ds, Is, *oracle.flat_search(xb, xb[:500], k, metric))
    assert idx.last_seed()["stat_redo"] == 0


def test_statistical_seed_is_automatic_in_the_batch_regime(gpu_faiss, oracle):
    """>= 1024 queries against >= 8192 rows with a k that makes every chunk warm up on its own: the statistical
    seed is chosen without being asked for; a caller stream (asynchronous entry) never uses it."""
    import torch
    from knn_for_homology_amd import _lib
    rng = np.random.default_rng(7)
    nb, d, nq, k = 16000, 48, 1100, 200
    xb = rng.standard_normal((nb, d), dtype=np.float32)
    xq = rng.standard_normal((nq, d), dtype=np.float32)
    idx = gpu_faiss.IndexFlat(d, 1)
    idx.add(xb)
    D, I = idx.search(xq, k)
    assert idx.last_seed()["stat_rank"] > 0
    Do, Io = oracle.flat_search(xb, xq, k, 1)
    _assert_same(D, I, Do, Io)
    dev = torch.device("cuda:0")
    q = torch.from_numpy(xq).to(dev)
    Dd = torch.empty((nq, k), device=dev, dtype=torch.float32)
    Id = torch.empty((nq, k), device=dev, dtype=torch.int64)
    L = _lib.lib()
    _lib.check(L.knn_flat_search_dev(idx._h, q.data_ptr(), nq, k, Dd.data_ptr(), Id.data_ptr(), None))  # synchronous: may use it
    assert idx.last_seed()["stat_rank"] > 0
    _assert_same(Dd.cpu().numpy(), Id.cpu().numpy(), Do, Io)
    side = torch.cuda.Stream(dev)
    with torch.cuda.stream(side):
        import ctypes
        _lib.check(L.knn_flat_search_dev(idx._h, q.data_ptr(), nq, k, Dd.data_ptr(), Id.data_ptr(), ctypes.c_void_p(side.cuda_stream)))
    side.synchronize()
    assert idx.last_seed()["stat_rank"] == 0 and idx.last_seed()["stride"] == 0
    _assert_same(Dd.cpu().numpy(), Id.cpu().numpy(), Do, Io)


@pytest.mark.parametrize("metric", [0, 1])
def test_statistical_seed_failure_is_caught_and_repaired(gpu_faiss, oracle, metric):
    """A database built against the sampler: exactly the sampled rows (every 32nd) are close to the queries, nothing
    else is.  The j-th sample score then promises ~32 j rows that do not exist; the verification (k-th score found
    <= threshold) must fail and the search must be repeated without the estimate -- exact result, stat_redo counts."""
    rng = np.random.default_rng(55 + metric)
    nb, d, nq, k = 16384, 64, 260, 100
    base = rng.standard_normal(d).astype(np.float32)
    base /= np.linalg.norm(base)
    xb = rng.standard_normal((nb, d), dtype=np.float32)
    xb /= np.linalg.norm(xb, axis=1, keepdims=True)
    near = np.arange(0, nb, 32)
    xb[near] = base[None, :] + 0.05 * rng.standard_normal((near.size, d)).astype(np.float32)
    xq = (base[None, :] + 0.05 * rng.standard_normal((nq, d))).astype(np.float32)
    idx = gpu_faiss.IndexFlat(d, metric)
    idx.set_tuning(0, 0, 128)
    idx.add(xb)
    before = idx.last_seed()["stat_redo"]
    D, I = idx.search(xq, k)
    assert idx.last_seed()["stat_redo"] > before, "the adversarial sample must trip the verification"
    _assert_same(D, I, *oracle.flat_search(xb, xq, k, metric))
    assert np.isin(I[:, :50], near).all()
    # pipelined multi-batch host path (> 16384 queries): failed batches are repeated after the pipeline drains
    xq2 = np.concatenate([xq] * 70)[:17000]
    D2, I2 = idx.search(xq2, k)
    assert idx.last_seed()["stat_redo"] >= before + 3
    _assert_same(D2[:nq], I2[:nq], D, I)
    _assert_same(D2[-100:], I2[-100:], *oracle.flat_search(xb, xq2[-100:], k, metric))


@pytest.mark.parametrize("flags", [8, 16, 0])
def test_long_candidate_arrays_take_the_segment_pass(gpu_faiss, oracle, flags):
    """Few queries against many rows: hundreds of chunks hand on their survivors, the per-query candidate arrays are
    sized for more than 32768 keys and the final selection runs its segment pass first (flags=8: unseeded, arrays
    really that long; 16: exact seed; 0: whatever the plan picks)."""
    rng = np.random.default_rng(900 + flags)
    nb, d, nq, k = 300_000, 32, 5, 100
    xb = rng.standard_normal((nb, d), dtype=np.float32)
    xq = rng.standard_normal((nq, d), dtype=np.float32)
    for metric in (0, 1):
        idx = gpu_faiss.IndexFlat(d, metric)
        idx.set_tuning(0, 0, flags)
        idx.add(xb)
        D, I = idx.search(xq, k)
        assert idx.last_scan()["nchunks"] * 125 > 32768
        _assert_same(D, I, *oracle.flat_search(xb, xq, k, metric))


def test_large_results_live_in_page_locked_memory(gpu_faiss, oracle):
    """(D, I) of >= 4 MB are numpy arrays over pooled hipHostMalloc blocks (one DMA at PCIe line rate instead of a
    staged copy): ordinary writable arrays that outlive their index, whose block goes back to the pool with the last view."""
    import gc
    from knn_for_homology_amd import _lib
    rng = np.random.default_rng(3)
    xb = rng.standard_normal((3000, 32), dtype=np.float32)
    xq = rng.standard_normal((6000, 32), dtype=np.float32)
    idx = gpu_faiss.IndexFlat(32, 1)
    idx.add(xb)
    D, I = idx.search(xq, 200)          # 4.8 MB + 9.6 MB
    assert not D.flags.owndata and not I.flags.owndata and D.flags.writeable and D.flags.c_contiguous
    del idx
    gc.collect()
    Do, Io = oracle.flat_search(xb, xq, 200, 1)
    _assert_same(D, I, Do, Io)
    view = I[:, 1:]                      # what cath.search.search hands out
    del I
    gc.collect()
    assert np.array_equal(view, Io[:, 1:])
    D[0, 0] = 7.0                        # writable (pfam/proteins.py:85-122 swaps entries in place)
    cls = 1 << (D.nbytes - 1).bit_length()
    before = len(_lib._pinned_free.get(cls, []))
    del D
    gc.collect()
    assert len(_lib._pinned_free.get(cls, [])) == before + 1
    small = gpu_faiss.IndexFlat(32, 1)
    small.add(xb)
    assert small.search(xq[:10], 5)[0].flags.owndata  # small results stay plain numpy
    # a block above PINNED_FIRST_MAX is page-locked from the second request of its size class on
    big = 1 << 27
    _lib._pinned_seen.pop(big, None)
    cached = _lib._pinned_free.pop(big, [])   # (blocks of this class an earlier test may have left in the pool)
    first = _lib.result_array((big // 4 - 5,), np.float32)
    second = _lib.result_array((big // 4 - 5,), np.float32)
    assert first.flags.owndata and not second.flags.owndata
    del first, second
    gc.collect()
    _lib._pinned_free.setdefault(big, []).extend(cached)


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("n,d,k", [(9000, 64, 10), (14433, 128, 301), (20011, 32, 1000)])
def test_symmetric_self_search_matches_plain_and_oracle(gpu_faiss, oracle, n, d, k, metric):
    """search_self over the whole index multiplies only the score tiles on and above the diagonal (every tile serves the
    queries of its row tile and of its column tile); flags=1024 forces the plain launch.  Same bits, with duplicates,
    a ragged last tile and k up to 1000; oracle bits on sampled rows."""
    rng = np.random.default_rng(n + k)
    x = rng.standard_normal((n, d), dtype=np.float32)
    x[n - 300:n - 200] = x[100:200]          # exact duplicates across distant tiles: ties broken by the lower id
    x[5000:5050] = x[4950:5000]
    idx = gpu_faiss.IndexFlat(d, metric)
    idx.add(x)
    D, I = idx.search_self(k)
    assert idx.last_scan()["kernel"] == "flat_scan_q128_d128_sym" and idx.last_seed()["stat_rank"] > 0
    idx.set_tuning(0, 0, 1024)
    Dp, Ip = idx.search_self(k)
    assert idx.last_scan()["kernel"] == "flat_scan_q128_d128"
    _assert_same(D, I, Dp, Ip)
    sample = np.concatenate([rng.choice(n, 24, replace=False), [0, 127, 128, n - 1, 100, n - 300, 5000, 4950]])
    Do, Io = oracle.flat_search(x, x[sample], k, metric)
    _assert_same(D[sample], I[sample], Do, Io)


def test_symmetric_self_search_repairs_a_failed_estimate(gpu_faiss, oracle):
    """The adversarial database of the statistical-seed test, searched against itself: the sampled rows form a tight
    cluster nobody else belongs to, their thresholds promise neighbours that do not exist, the symmetric launch's
    verification fails and the plain path repeats the search -- exact result."""
    rng = np.random.default_rng(77)
    n, d, k = 16384, 64, 100
    base = rng.standard_normal(d).astype(np.float32)
    base /= np.linalg.norm(base)
    x = rng.standard_normal((n, d), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    near = np.arange(0, n, 32)
    x[near] = base[None, :] + 0.05 * rng.standard_normal((near.size, d)).astype(np.float32)
    idx = gpu_faiss.IndexFlat(d, 0)
    idx.add(x)
    before = idx.last_seed()["stat_redo"]
    D, I = idx.search_self(k)
    assert idx.last_seed()["stat_redo"] > before
    sample = np.concatenate([near[:20], rng.choice(n, 20, replace=False)])
    _assert_same(D[sample], I[sample], *oracle.flat_search(x, x[sample], k, 0))


def test_normalize_matches_oracle(gpu_faiss, oracle):
    rng = np.random.default_rng(11)
    for n, d in ((1000, 1024), (333, 100), (50, 37), (1, 1024), (129, 8)):
        x = rng.standard_normal((n, d), dtype=np.float32)
        if n > 5:
            x[5] = 0
        a, b = x.copy(), x.copy()
        gpu_faiss.normalize_L2(a)
        oracle.normalize_l2(b)
        assert np.array_equal(_bits(a), _bits(b))
        if n > 5:
            assert (a[5] == 0).all()


# ---- edge cases ------------------------------------------------------------------------
def test_edge_cases(gpu_faiss, oracle):
    from knn_for_homology_amd._lib import Knn355Error
    rng = np.random.default_rng(2)
    xb = rng.standard_normal((50, 24), dtype=np.float32)
    xq = rng.standard_normal((3, 24), dtype=np.float32)
    fmax = np.finfo(np.float32).max
    for metric, pad in ((0, -fmax), (1, fmax)):
        idx = gpu_faiss.IndexFlat(24, metric)
        # empty index: all slots unfilled
        D, I = idx.search(xq, 4)
        assert (I == -1).all() and (D == pad).all()
        idx.add(xb)
        # k > ntotal
        D, I = idx.search(xq, 60)
        _assert_same(D, I, *oracle.flat_search(xb, xq, 60, metric))
        assert (I[:, 50:] == -1).all() and (D[:, 50:] == pad).all()
        # no queries
        D, I = idx.search(xq[:0], 4)
        assert D.shape == (0, 4) and I.shape == (0, 4)
        # several add calls == one add call (ids are insertion order)
        idx2 = gpu_faiss.IndexFlat(24, metric)
        idx2.add(xb[:7])
        idx2.add(xb[7:8])
        idx2.add(xb[8:])
        assert idx2.ntotal == 50
        _assert_same(*idx2.search(xq, 9), *oracle.flat_search(xb, xq, 9, metric))
        assert np.array_equal(idx2.reconstruct_n(0, 50), xb)
        idx2.reset()
        assert idx2.ntotal == 0
        # argument errors
        with pytest.raises(Knn355Error):
            idx.search(xq, 4096)
        with pytest.raises(AssertionError):
            idx.search(xq[:, :20].copy(), 3)
        with pytest.raises(TypeError):
            idx.add(xb.astype(np.float64))
    with pytest.raises(Knn355Error):
        gpu_faiss.IndexFlat(24, 7)


def test_exact_duplicates_break_ties_by_lower_id(gpu_faiss, oracle):
    rng = np.random.default_rng(4)
    xb = rng.standard_normal((3000, 1024), dtype=np.float32)
    xb[1500:1600] = xb[100:200]
    xb[2900:2950] = xb[100:150]
    q = xb[100:164].copy()
    for metric, nch in ((0, 0), (1, 4), (1, 1)):
        idx = gpu_faiss.IndexFlat(1024, metric)
        idx.set_tuning(0, nch, 0)
        idx.add(xb)
        D, I = idx.search(q, 20)
        _assert_same(D, I, *oracle.flat_search(xb, q, 20, metric))
        assert (I[:50, 0] == np.arange(100, 150)).all()
        assert (I[:50, 1] == np.arange(1500, 1550)).all() and (I[:50, 2] == np.arange(2900, 2950)).all()
        if metric == 1:
            assert (D[:50, :3] == 0).all()


@pytest.mark.parametrize("metric", [0, 1])
def test_seeded_search_matches_oracle(gpu_faiss, oracle, metric):
    """flags=16 forces the recursive strided-sample seeding (normally reserved for the
    streaming regime); flags=8 disables it.  Both must return the oracle's bits, also on a
    database sorted so that every new tile beats the previous ones."""
    rng = np.random.default_rng(31)
    xb = rng.standard_normal((40000, 128), dtype=np.float32)
    xq = rng.standard_normal((70, 128), dtype=np.float32)
    Do, Io = oracle.flat_search(xb, xq, 100, metric)
    for flags, nq in ((16, 70), (16, 20), (8, 70), (0, 20)):
        idx = gpu_faiss.IndexFlat(128, metric)
        idx.set_tuning(0, 0, flags)
        idx.add(xb)
        D, I = idx.search(xq[:nq], 100)
        _assert_same(D, I, Do[:nq], Io[:nq])
    xs = np.sort(rng.standard_normal((30000, 64), dtype=np.float32), axis=0)
    q = xs[::500].copy()
    idx = gpu_faiss.IndexFlat(64, metric)
    idx.set_tuning(0, 0, 16)
    idx.add(xs)
    _assert_same(*idx.search(q, 50), *oracle.flat_search(xs, q, 50, metric))


@pytest.mark.parametrize("nb", [8193, 8199, 40003, 65535, 65543])
def test_seed_sample_blocks_with_ragged_tail(gpu_faiss, oracle, nb):
    """The seed sample is every s-th block of 8 rows; a database that does not end on a block
    (or whose last sample block is cut short) must still be searched exactly, every row counted
    once: queries are planted rows, including the very last ones."""
    rng = np.random.default_rng(nb)
    d, k = 96, 37
    xb = rng.standard_normal((nb, d), dtype=np.float32)
    planted = np.array([0, 7, 8, nb - 9, nb - 8, nb - 2, nb - 1])
    xq = np.concatenate([xb[planted], rng.standard_normal((9, d), dtype=np.float32)])
    for metric in (0, 1):
        Do, Io = oracle.flat_search(xb, xq, k, metric)
        for flags in (16, 0):
            idx = gpu_faiss.IndexFlat(d, metric)
            idx.set_tuning(0, 0, flags)
            idx.add(xb)
            D, I = idx.search(xq, k)
            _assert_same(D, I, Do, Io)
            if metric == 1:
                assert (I[:len(planted), 0] == planted).all() and (D[:len(planted), 0] == 0).all()


def test_result_independent_of_tiling(gpu_faiss):
    """The same search under every query-tile / chunk split returns identical bits."""
    rng = np.random.default_rng(8)
    xb = rng.standard_normal((20000, 1024), dtype=np.float32)
    xq = rng.standard_normal((96, 1024), dtype=np.float32)
    idx = gpu_faiss.IndexFlat(1024, 0)
    idx.add(xb)
    ref = None
    for qt, nch in ((0, 0), (32, 1), (32, 40), (64, 7), (128, 1), (128, 19)):
        idx.set_tuning(qt, nch, 0)
        D, I = idx.search(xq, 100)
        if ref is None:
            ref = (D, I)
        else:
            _assert_same(D, I, *ref)


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("n,d,k", [(3000, 200, 11), (20000, 64, 5), (700, 1024, 301)])
def test_search_self_equals_host_path(gpu_faiss, oracle, n, d, k, metric):
    """The one-upload all-vs-all (add, normalize_rows, search_self) returns the bits of the
    reference's sequence normalize_L2(copy) / add / search(same array); 20000 rows cross the
    16384-query batch boundary of the pipelined host search."""
    rng = np.random.default_rng(n + d)
    x = rng.standard_normal((n, d), dtype=np.float32)
    host = x.copy()
    if metric == 0:
        gpu_faiss.normalize_L2(host)
    a = gpu_faiss.IndexFlat(d, metric)
    a.add(host)
    Dh, Ih = a.search(host, k)
    b = gpu_faiss.IndexFlat(d, metric)
    b.add(x)
    if metric == 0:
        b.normalize_rows()
        assert np.array_equal(b.reconstruct_n(0, n).view(np.uint32), host.view(np.uint32))
    Ds, Is = b.search_self(k)
    _assert_same(Ds, Is, Dh, Ih)
    D2, I2 = b.search_self(k, row0=n // 3, nrows=257)
    _assert_same(D2, I2, Dh[n // 3:n // 3 + 257], Ih[n // 3:n // 3 + 257])
    sample = rng.choice(n, 16, replace=False)
    Do, Io = oracle.flat_search(host, host[sample], k, metric, l2_mode=1)  # (sampled rows of a big batch: the norm formula)
    _assert_same(Ds[sample], Is[sample], Do, Io)
    with pytest.raises(RuntimeError):
        b.search_self(k, row0=n - 5, nrows=10)


# ---- full-size configs: size-independent properties + sampled oracle rows ---------------
def test_cath20_sized_all_vs_all(gpu_faiss, oracle):
    """BASELINE config 2: 14433 x 1024 all-vs-all, L2, k=300 (+ self)."""
    from knn_for_homology_amd.cath.search import search
    rng = np.random.default_rng(20)
    x = rng.standard_normal((14433, 1024), dtype=np.float32)
    hits, scores = search(x, hits=300, metric=gpu_faiss.METRIC_L2)
    assert hits.shape == (14433, 300)
    assert (np.diff(scores, axis=1) >= 0).all(), "distances must be ascending"
    assert (hits != np.arange(14433)[:, None]).all(), "self hit must be stripped"
    srt = np.sort(hits, axis=1)
    assert (srt[:, 1:] != srt[:, :-1]).all() and hits.min() >= 0 and hits.max() < 14433
    sample = rng.choice(14433, 24, replace=False)
    Do, Io = oracle.flat_search(x, x[sample], 301, 1, l2_mode=1)  # (sampled rows of a 14433-query batch: the norm formula)
    _assert_same(scores[sample], hits[sample], Do[:, 1:], Io[:, 1:])
    # and against fp64 truth: ids only permuted inside fp32-noise clusters, distances within
    # 1e-5 of |x|^2+|y|^2 (the north star's tolerance; oracle/knn_oracle.py compare_tie_tolerant)
    from oracle import knn_oracle as ko
    full_I = np.concatenate([sample[:, None], hits[sample]], 1)
    full_D = np.concatenate([np.zeros((24, 1), np.float32), scores[sample]], 1)
    stats = ko.compare_tie_tolerant(full_I, full_D, x, x[sample], ko.METRIC_L2, dist_rtol=1e-5)
    assert stats["max_dist_err_rel"] < 1e-5
    # cosine, reference default hits=10
    hits, scores = search(x, hits=10)
    xn = x.copy()
    oracle.normalize_l2(xn)
    Do, Io = oracle.flat_search(xn, xn[sample], 11, 0)
    _assert_same(scores[sample], hits[sample], Do[:, 1:], Io[:, 1:])
    assert (np.diff(scores, axis=1) <= 0).all()


def test_large_shard_streaming(gpu_faiss, oracle):
    """BASELINE config 4 shard shape: 1.25M x 1024 rows resident, inner product, k=100.
    Properties: planted queries find themselves first; result identical under another
    tiling; sampled queries equal the oracle bit for bit."""
    import torch
    from knn_for_homology_amd import _lib
    L = _lib.lib()
    nb, d, k = 1_250_000, 1024, 100
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(23)
    idx = gpu_faiss.IndexFlat(d, 0)
    _lib.check(L.knn_flat_reserve(idx._h, nb))
    for i0 in range(0, nb, 250_000):
        x = torch.randn((250_000, d), generator=g, device=dev)
        _lib.check(L.knn_normalize_l2_dev(x.data_ptr(), 250_000, d, None))
        _lib.check(L.knn_flat_add_dev(idx._h, x.data_ptr(), 250_000, None))
        del x
    assert idx.ntotal == nb
    planted = np.array([0, 1, 777_777, nb - 1])
    q = np.concatenate([idx.reconstruct(int(i))[None] for i in planted]
                       + [np.random.default_rng(24).standard_normal((28, d), dtype=np.float32)])
    gpu_faiss.normalize_L2(q)
    D, I = idx.search(q, k)
    assert (I[:4, 0] == planted).all() and np.allclose(D[:4, 0], 1.0, atol=1e-5)
    assert (np.diff(D, axis=1) <= 0).all() and I.min() >= 0 and I.max() < nb
    idx.set_tuning(128, 37, 0)
    D2, I2 = idx.search(q, k)
    _assert_same(D2, I2, D, I)
    # oracle on a 200k-row window that contains two planted rows' neighbourhoods
    w0, w1 = 700_000, 900_000
    win = idx.reconstruct_n(w0, w1 - w0)
    sub = gpu_faiss.IndexFlat(d, 0)
    sub.add(win)
    Ds, Is = sub.search(q[:8], k)
    _assert_same(Ds, Is, *oracle.flat_search(win, q[:8], k, 0))


def test_full_size_10m_streaming(gpu_faiss, oracle):
    """BASELINE config 4 at its full single-GPU size: 10 M x 1024 rows resident (41 GB), inner
    product, k = 100, 32 queries.  Size-independent properties: planted queries find themselves
    first with score 1; scores descend, ids are valid and distinct; the seeded plan, an unseeded
    plan with another chunk count and a 128-query-tile plan return the same bits; a 200 k-row
    window around planted rows equals the oracle bit for bit."""
    import torch
    from knn_for_homology_amd import _lib
    L = _lib.lib()
    nb, d, k = 10_000_000, 1024, 100
    dev = torch.device("cuda:0")
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 60 * (1 << 30):
        pytest.skip("needs 60 GB of free HBM")
    g = torch.Generator(device=dev)
    g.manual_seed(23)
    idx = gpu_faiss.IndexFlat(d, 0)
    _lib.check(L.knn_flat_reserve(idx._h, nb))
    for i0 in range(0, nb, 1_000_000):
        x = torch.randn((1_000_000, d), generator=g, device=dev)
        _lib.check(L.knn_normalize_l2_dev(x.data_ptr(), 1_000_000, d, None))
        _lib.check(L.knn_flat_add_dev(idx._h, x.data_ptr(), 1_000_000, None))
        del x
    assert idx.ntotal == nb
    planted = np.array([0, 7, 4_999_999, 7_654_321, nb - 1])
    q = np.concatenate([idx.reconstruct(int(i))[None] for i in planted]
                       + [np.random.default_rng(24).standard_normal((27, d), dtype=np.float32)])
    gpu_faiss.normalize_L2(q)
    D, I = idx.search(q, k)
    assert idx.last_scan()["kernel"] == "flat_scan_q32_d256"
    assert (I[:5, 0] == planted).all() and np.allclose(D[:5, 0], 1.0, atol=1e-5)
    assert (np.diff(D, axis=1) <= 0).all() and I.min() >= 0 and I.max() < nb
    srt = np.sort(I, axis=1)
    assert (srt[:, 1:] != srt[:, :-1]).all()
    for qt, nch, flags in ((0, 301, 8), (128, 0, 0)):
        idx.set_tuning(qt, nch, flags)
        D2, I2 = idx.search(q, k)
        _assert_same(D2, I2, D, I)
    w0 = 7_600_000
    win = idx.reconstruct_n(w0, 200_000)
    sub = gpu_faiss.IndexFlat(d, 0)
    sub.add(win)
    Ds, Is = sub.search(q[:8], k)
    _assert_same(Ds, Is, *oracle.flat_search(win, q[:8], k, 0))
    # every window row that made the global top-k must also be in the window's own top-k
    for r in range(8):
        inside = I[r][(I[r] >= w0) & (I[r] < w0 + 200_000)] - w0
        assert np.isin(inside, Is[r]).all()


def test_pfam_sized_all_vs_all(gpu_faiss, oracle):
    """BASELINE config 3 at full size: 200 000 x 1024 clustered rows with 0.5 % exact duplicates,
    cosine all-vs-all, k = 1000 (the reference's pfam/proteins_search.py run), through the pipelined
    host path.  Properties over all rows + the oracle's bits on sampled rows."""
    n, d, k = 200_000, 1024, 1000
    cent = np.random.default_rng(21).standard_normal((2000, d)).astype(np.float32)
    rng = np.random.default_rng(22)
    x = cent[rng.integers(0, 2000, n)] + 0.35 * rng.standard_normal((n, d), dtype=np.float32)
    dst = rng.choice(n, n // 200, replace=False)
    src = (dst + 1 + rng.integers(0, n - 1, dst.size)) % n
    keep = ~np.isin(src, dst)
    dst, src = dst[keep], src[keep]
    x[dst] = x[src]
    gpu_faiss.normalize_L2(x)
    idx = gpu_faiss.IndexFlat(d, gpu_faiss.METRIC_INNER_PRODUCT)
    idx.add(x)
    D, I = idx.search_self(k)
    assert D.shape == (n, k) and I.dtype == np.int64
    assert (np.diff(D, axis=1) <= 0).all() and I.min() >= 0 and I.max() < n
    # a row is its own best hit unless an exact duplicate with a lower id ties with it
    lower_twin = np.full(n, -1)
    a, b = np.minimum(dst, src), np.maximum(dst, src)
    lower_twin[b] = a
    expect_first = np.where(lower_twin >= 0, lower_twin, np.arange(n))
    has_twin = np.zeros(n, bool)
    has_twin[a] = has_twin[b] = True
    assert (I[~has_twin, 0] == np.arange(n)[~has_twin]).all()
    assert (np.isin(I[has_twin, 0], np.concatenate([a, b]))).all() and (I[b, 0] <= b).all() and (I[expect_first == np.arange(n), 0] <= np.arange(n)[expect_first == np.arange(n)]).all()
    sample = np.concatenate([rng.choice(n, 12, replace=False), a[:2], b[:2]])
    Do, Io = oracle.flat_search(x, x[sample], k, 0)
    _assert_same(D[sample], I[sample], Do, Io)
    for r in rng.choice(n, 64, replace=False):  # no id twice in a row of results
        assert len(np.unique(I[r])) == k


@pytest.mark.parametrize("metric", [0, 1])
def test_device_search_of_many_queries_runs_in_blocks(gpu_faiss, oracle, metric):
    """knn_flat_search_dev with more than 24576 queries launches blocks of 16384 (the last one ragged); synchronous
    (statistical seed allowed, its verification flag accumulates over the blocks) and on a caller's stream (exact seed
    only) -- same bits as the host path, oracle bits on sampled queries of every block."""
    import ctypes
    import torch
    from knn_for_homology_amd import _lib
    rng = np.random.default_rng(40 + metric)
    nb, nq, d, k = 30000, 40001, 64, 20
    xb = rng.standard_normal((nb, d), dtype=np.float32)
    xq = rng.standard_normal((nq, d), dtype=np.float32)
    xq[:500] = xb[:500]
    idx = gpu_faiss.IndexFlat(d, metric)
    idx.add(xb)
    dev = torch.device("cuda", 0)
    q = torch.from_numpy(xq).to(dev)
    D = torch.empty((nq, k), dtype=torch.float32, device=dev)
    I = torch.empty((nq, k), dtype=torch.int64, device=dev)
    L = _lib.lib()
    _lib.check(L.knn_flat_search_dev(idx._h, q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), None))
    Dh, Ih = idx.search(xq, k)
    _assert_same(D.cpu().numpy(), I.cpu().numpy(), Dh, Ih)
    side = torch.cuda.Stream(dev)
    D2, I2 = torch.empty_like(D), torch.empty_like(I)
    with torch.cuda.stream(side):
        _lib.check(L.knn_flat_search_dev(idx._h, q.data_ptr(), nq, k, D2.data_ptr(), I2.data_ptr(), ctypes.c_void_p(side.cuda_stream)))
    side.synchronize()
    _assert_same(D2.cpu().numpy(), I2.cpu().numpy(), Dh, Ih)
    sample = np.concatenate([rng.choice(nq, 40, replace=False), [0, 16383, 16384, 32767, 32768, nq - 1]])
    Do, Io = oracle.flat_search(xb, xq[sample], k, metric)
    _assert_same(Dh[sample], Ih[sample], Do, Io)


# ---- tile-minimum seed (streaming regime, round 3) ---------------------------------------------------------
@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("nb,d,nq,k,rounds", [(600_000, 64, 32, 10, 0), (600_000, 32, 5, 10, 1), (700_001, 32, 40, 20, 2),
                                              (900_000, 32, 1, 100, 0), (650_003, 40, 33, 1, 0)])
def test_tile_minimum_seed_matches_oracle(gpu_faiss, oracle, metric, nb, d, nq, k, rounds):
    """Few queries, many rows: the scan seeds itself (every chunk publishes the best key of its first tiles, the k-th
    smallest published key bounds the global k-th) instead of running a sample pass first.  Same bits as the oracle and
    as the sample-pass build (flags 2048), with duplicates and ties in the data."""
    rng = np.random.default_rng(nb + k)
    xb = rng.standard_normal((nb, d), dtype=np.float32)
    xb[nb // 2: nb // 2 + 300] = xb[:300]          # exact duplicates: ties on the score word
    xb[-50:] = 0.0                                  # and a block of identical rows at the ragged end
    xq = np.concatenate([rng.standard_normal((max(nq - 2, 0), d), dtype=np.float32), xb[:min(2, nq)]])[:nq]
    idx = gpu_faiss.IndexFlat(d, metric)
    idx.add(xb)
    idx.set_tuning(0, 0, rounds << 12)
    D, I = idx.search(xq, k)
    seed = idx.last_seed()
    assert seed["stride"] < 0, f"expected the tile-minimum seed, got {seed} {idx.last_scan()}"
    if rounds:
        assert seed["stride"] == -rounds
    Do, Io = oracle.flat_search(xb, xq, k, metric)
    _assert_same(D, I, Do, Io)
    idx.set_tuning(0, 0, 2048)
    D2, I2 = idx.search(xq, k)
    assert idx.last_seed()["stride"] >= 0
    _assert_same(D2, I2, Do, Io)


def test_tile_minimum_seed_with_adversarial_order(gpu_faiss, oracle):
    """Rows sorted so that every chunk's first tile holds its WORST rows and the best rows of the whole database sit in
    the last tiles of the last chunks: the published bound is loose, never wrong."""
    rng = np.random.default_rng(5)
    nb, d, nq, k = 640_000, 48, 32, 50
    base = rng.standard_normal(d).astype(np.float32)
    base /= np.linalg.norm(base)
    scale = np.linspace(0.05, 2.0, nb, dtype=np.float32)
    xb = (scale[:, None] * base[None, :] + 0.01 * rng.standard_normal((nb, d)).astype(np.float32)).astype(np.float32)
    xq = (base[None, :] + 0.01 * rng.standard_normal((nq, d)).astype(np.float32)).astype(np.float32)
    idx = gpu_faiss.IndexFlat(d, 0)
    idx.add(xb)
    D, I = idx.search(xq, k)
    assert idx.last_seed()["stride"] < 0
    _assert_same(D, I, *oracle.flat_search(xb, xq, k, 0))


@pytest.mark.parametrize("metric,nb,d,nq,k,keys", [(0, 620_000, 32, 32, 1000, 4), (1, 620_000, 32, 7, 600, 4), (0, 560_000, 24, 32, 1536, 4),
                                                   (1, 620_000, 32, 48, 600, 4), (0, 620_000, 32, 64, 700, 2), (1, 600_000, 40, 21, 481, 4)])
def test_tile_minimum_seed_with_one_key_per_wave_for_large_k(gpu_faiss, oracle, metric, nb, d, nq, k, keys):
    """k beyond the number of workgroups (two rounds of 512 publications carry k <= 480): every wave publishes the best
    key of its own rows of the first tile (4 per workgroup and query with the 32-query tile, 2 with the 64-query one).
    Same bits as the oracle and as the build without the tile-minimum seed, duplicates and ties included."""
    rng = np.random.default_rng(nb + k + nq)
    xb = rng.standard_normal((nb, d), dtype=np.float32)
    xb[nb // 2: nb // 2 + 700] = xb[:700]
    xb[-90:] = 0.0
    xq = np.concatenate([rng.standard_normal((max(nq - 2, 0), d), dtype=np.float32), xb[:min(2, nq)]])[:nq]
    idx = gpu_faiss.IndexFlat(d, metric)
    idx.add(xb)
    D, I = idx.search(xq, k)
    assert idx.last_seed()["stride"] == -keys, (idx.last_seed(), idx.last_scan())
    Do, Io = oracle.flat_search(xb, xq, k, metric)
    _assert_same(D, I, Do, Io)
    D1, I1 = idx.search(xq, k)  # (back to back: the state the first search's selection left)
    _assert_same(D1, I1, Do, Io)
    idx.set_tuning(0, 0, 2048)
    D2, I2 = idx.search(xq, k)
    assert idx.last_seed()["stride"] >= 0
    _assert_same(D2, I2, Do, Io)


@pytest.mark.parametrize("metric,nb,d,nq,k,seed", [(0, 620_000, 32, 32, 1537, -4), (1, 620_000, 24, 9, 2048, -8), (0, 560_000, 32, 32, 1800, -8),
                                                   (1, 620_000, 32, 50, 2048, None), (0, 30_000, 48, 300, 2048, None), (1, 9_000, 64, 130, 1700, None),
                                                   (0, 4_100, 32, 5, 2048, None), (1, 2_049, 40, 33, 2048, None)])
def test_k_between_1537_and_2048_selects_by_probing_the_list(gpu_faiss, oracle, metric, nb, d, nq, k, seed):
    """k in (1536, 2048]: the 4096-key candidate lists are cut by one wave that probes them in memory (wave_select_mem)
    instead of a workgroup-wide sort, and a streaming launch seeds itself from 2048 or 4096 published keys.  Streaming,
    batch and small-database shapes, duplicates and ties: the oracle's bits, with and without seeding."""
    rng = np.random.default_rng(nb + k + nq)
    xb = rng.standard_normal((nb, d), dtype=np.float32)
    dup = min(900, nb // 4)
    xb[nb // 2: nb // 2 + dup] = xb[:dup]
    xb[-40:] = 0.0
    xq = np.concatenate([rng.standard_normal((max(nq - 2, 0), d), dtype=np.float32), xb[:min(2, nq)]])[:nq]
    idx = gpu_faiss.IndexFlat(d, metric)
    idx.add(xb)
    D, I = idx.search(xq, k)
    if seed is not None:
        assert idx.last_seed()["stride"] == seed, (idx.last_seed(), idx.last_scan())
    Do, Io = oracle.flat_search(xb, xq, k, metric)
    _assert_same(D, I, Do, Io)
    for flags in (8, 2048, 4):
        idx.set_tuning(0, 0, flags)
        D2, I2 = idx.search(xq, k)
        _assert_same(D2, I2, Do, Io)


@pytest.mark.parametrize("metric,nb,d,nq,k", [(0, 131_072, 32, 32, 100), (1, 131_073, 32, 7, 10), (0, 140_000, 24, 32, 300), (1, 200_000, 32, 64, 100),
                                              (0, 262_144, 32, 32, 100), (1, 300_001, 40, 19, 50), (0, 400_000, 32, 48, 600), (1, 66_000, 32, 40, 20)])
def test_short_streaming_launches_pair_and_seed_themselves(gpu_faiss, oracle, metric, nb, d, nq, k):
    """From two tiles per CU on (131 k rows with the 32-query tile, 66 k with the 64-query tile) a one-query-tile launch
    pairs its workgroups and seeds itself: every pair shares 2-8 tiles, a workgroup may walk a single one (whose parked
    scores it filters on the spot).  Shards of a 1-4 M-row database on eight GPUs are this size.  Same bits as the
    oracle, as static chunks (flags 4), as the sample pass (2048) and unseeded (8)."""
    rng = np.random.default_rng(nb + k)
    xb = rng.standard_normal((nb, d), dtype=np.float32)
    xb[nb // 2: nb // 2 + 200] = xb[:200]
    xb[-30:] = 0.0
    xq = np.concatenate([rng.standard_normal((max(nq - 2, 0), d), dtype=np.float32), xb[:min(2, nq)]])[:nq]
    idx = gpu_faiss.IndexFlat(d, metric)
    idx.add(xb)
    D, I = idx.search(xq, k)
    tile = 256 if nq <= 48 else 128  # (the 32- and 48-query builds walk 256-row tiles)
    pairs = min(256, -(-nb // tile) // 2)
    assert idx.last_seed()["stride"] < 0 and idx.last_scan()["grid"] == 2 * pairs, (idx.last_seed(), idx.last_scan())
    Do, Io = oracle.flat_search(xb, xq, k, metric)
    _assert_same(D, I, Do, Io)
    D1, I1 = idx.search(xq, k)
    _assert_same(D1, I1, Do, Io)
    for flags in (4, 2048, 8, 2):
        idx.set_tuning(0, 0, flags)
        D2, I2 = idx.search(xq, k)
        _assert_same(D2, I2, Do, Io)


@pytest.mark.parametrize("metric", [0, 1])
def test_batches_of_65_to_128_queries_on_a_mid_sized_database_go_as_two_pieces(gpu_faiss, oracle, metric):
    """32 k .. 262 k rows: no sample pass exists to seed a 128-query launch with, the 64-query build seeds itself -- the
    batch is searched as two 64-query pieces.  Same bits as the one 128-query launch (flags 16384) and as the oracle."""
    rng = np.random.default_rng(65 + metric)
    nb, d, k = 100_000, 32, 40
    xb = rng.standard_normal((nb, d), dtype=np.float32)
    xb[700:760] = xb[:60]
    idx = gpu_faiss.IndexFlat(d, metric)
    idx.add(xb)
    for nq in (65, 100, 128, 64, 129):
        xq = rng.standard_normal((nq, d), dtype=np.float32)
        xq[0] = xb[3]
        idx.set_tuning(0, 0, 0)
        D, I = idx.search(xq, k)
        rest = nq - 64  # (the second piece: the narrowest build that holds it)
        second = "flat_scan_q32_d256" if rest <= 32 else ("flat_scan_q48_d256" if rest <= 48 else "flat_scan_q64_d128")
        assert idx.last_scan()["kernel"] == ("flat_scan_q128_d128" if nq > 128 else ("flat_scan_q64_d128" if nq <= 64 else second)), (nq, idx.last_scan())
        idx.set_tuning(0, 0, 16384)
        D1, I1 = idx.search(xq, k)
        assert idx.last_scan()["kernel"] == ("flat_scan_q128_d128" if nq > 96 else ("flat_scan_q96_d128" if nq > 64 else "flat_scan_q64_d128"))
        _assert_same(D, I, D1, I1)
        _assert_same(D, I, *oracle.flat_search(xb, xq, k, metric))


@pytest.mark.parametrize("metric,nb,d,nq,k", [(0, 50_000, 32, 256, 1000), (1, 50_000, 24, 129, 100), (0, 200_000, 16, 700, 1000), (1, 9_000, 48, 512, 300),
                                              (0, 300_000, 16, 128, 1000), (1, 400_000, 16, 100, 700)])
def test_statistical_seed_serves_batches_from_65_queries_on(gpu_faiss, oracle, metric, nb, d, nq, k):
    """129 .. 1023 queries used to run unseeded, with few long chunks so that every chunk's warm-up was amortised (50 k rows
    x 256 queries, k = 1000: 38 workgroups, 2.4 ms); now the statistical seed (verified, repaired on failure) serves them
    too: 0.52 ms.  Same bits as the oracle and as the unseeded search (flags 8)."""
    rng = np.random.default_rng(nb + nq + k)
    xb = rng.standard_normal((nb, d), dtype=np.float32)
    xb[nb // 3: nb // 3 + 400] = xb[:400]
    xq = rng.standard_normal((nq, d), dtype=np.float32)
    xq[:3] = xb[5:8]
    idx = gpu_faiss.IndexFlat(d, metric)
    idx.add(xb)
    D, I = idx.search(xq, k)
    assert idx.last_seed()["stat_rank"] > 0, (idx.last_seed(), idx.last_scan())
    Do, Io = oracle.flat_search(xb, xq, k, metric)
    _assert_same(D, I, Do, Io)
    idx.set_tuning(0, 0, 8)
    D2, I2 = idx.search(xq, k)
    assert idx.last_seed()["stat_rank"] == 0
    _assert_same(D2, I2, Do, Io)


@pytest.mark.parametrize("metric,n,d,k", [(0, 3000, 64, 11), (1, 3001, 32, 301), (0, 4096, 48, 1001), (1, 5000, 64, 100), (0, 7000, 32, 301), (1, 2999, 32, 50)])
def test_symmetric_self_search_from_3000_rows_on(gpu_faiss, oracle, metric, n, d, k):
    """The symmetric launch (runs of one tile on, so that a small index still fills the chip) serves whole-index
    self-searches from 3000 rows on (before: 8192; 5000 rows: 0.81 -> 0.54 ms).  Same bits as the plain launch and as
    the oracle on every row."""
    rng = np.random.default_rng(n + k)
    cent = rng.standard_normal((20, d), dtype=np.float32)
    x = (cent[np.sort(rng.integers(0, 20, n))] + 0.4 * rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
    x[n // 2: n // 2 + 50] = x[:50]
    idx = gpu_faiss.IndexFlat(d, metric)
    idx.add(x)
    D, I = idx.search_self(k)
    assert idx.last_scan()["kernel"].endswith("_sym") == (n >= 3000), idx.last_scan()
    Do, Io = oracle.flat_search(x, x, k, metric, l2_mode=1)
    _assert_same(D, I, Do, Io)
    idx.set_tuning(0, 0, 1024)
    Dp, Ip = idx.search_self(k)
    assert not idx.last_scan()["kernel"].endswith("_sym")
    _assert_same(Dp, Ip, Do, Io)


def test_per_wave_publications_with_adversarial_order(gpu_faiss, oracle):
    """every chunk's first tile holds its worst rows, the best rows of the database sit at its end: a loose bound, never a
    wrong one (k = 1000, four keys per workgroup)"""
    rng = np.random.default_rng(6)
    nb, d, nq, k = 640_000, 48, 32, 1000
    base = rng.standard_normal(d).astype(np.float32)
    base /= np.linalg.norm(base)
    scale = np.linspace(0.05, 2.0, nb, dtype=np.float32)
    xb = (scale[:, None] * base[None, :] + 0.01 * rng.standard_normal((nb, d)).astype(np.float32)).astype(np.float32)
    xq = (base[None, :] + 0.01 * rng.standard_normal((nq, d)).astype(np.float32)).astype(np.float32)
    for order in (1, -1):
        idx = gpu_faiss.IndexFlat(d, 0)
        idx.add(np.ascontiguousarray(xb[::order]))
        D, I = idx.search(xq, k)
        assert idx.last_seed()["stride"] == -4
        _assert_same(D, I, *oracle.flat_search(np.ascontiguousarray(xb[::order]), xq, k, 0))


# ---- FAISS's small-batch squared L2: the sum of squared differences (round 3) --------------------------------------
@pytest.mark.parametrize("nb,d,nq,k", [(1000, 5, 1, 3), (5000, 64, 7, 10), (4097, 100, 19, 50), (300, 1024, 6, 11),
                                       (20000, 33, 19, 100), (600_000, 32, 5, 10), (700_001, 16, 19, 100)])
def test_l2_small_batch_is_the_sum_of_squared_differences(gpu_faiss, oracle, nb, d, nq, k):
    """IndexFlat(d, METRIC_L2).search with fewer than 20 queries returns sum (x - y)^2 (FAISS's knn_L2sqr takes its SIMD
    path below distance_compute_blas_threshold = 20 [ext]; reachable through seqvec_search/main.py:22-45 with
    metric=METRIC_L2): bit for bit the oracle's difference chain, in every seeding / pairing regime; flags=32 keeps the
    norm formula."""
    rng = np.random.default_rng(nb + nq)
    xb = rng.standard_normal((nb, d), dtype=np.float32)
    xb[nb // 2: nb // 2 + 20] = xb[:20]
    xq = np.concatenate([rng.standard_normal((nq, d), dtype=np.float32)[: max(nq - 1, 0)], xb[:1]])[:nq]
    idx = gpu_faiss.IndexFlat(d, 1)
    idx.add(xb)
    D, I = idx.search(xq, k)
    assert idx.last_scan()["kernel"] == "flat_scan_q32_d256_l2diff"
    _assert_same(D, I, *oracle.flat_search(xb, xq, k, 1))
    assert (D[-1, 0] == 0.0) and I[-1, 0] == 0, "a row's distance to itself is exactly 0, the lower id of a duplicate first"
    idx.set_tuning(0, 0, 32)
    D2, I2 = idx.search(xq, k)
    assert idx.last_scan()["kernel"] == "flat_scan_q32_d256"
    _assert_same(D2, I2, *oracle.flat_search(xb, xq, k, 1, l2_mode=1))
    # both within the north star's 1e-5 (relative to the magnitudes the sums are formed from) of the fp64 truth
    if nb <= 5000:
        ref = np.sort(((xq[:, None, :].astype(np.float64) - xb[None].astype(np.float64)) ** 2).sum(2), 1)[:, :k]
        scale = float((xb.astype(np.float64) ** 2).sum(1).max() + (xq.astype(np.float64) ** 2).sum(1).max())
        assert np.abs(D - ref).max() <= 1e-5 * scale and np.abs(D2 - ref).max() <= 1e-5 * scale


def test_l2_formula_switches_at_twenty_queries(gpu_faiss, oracle):
    rng = np.random.default_rng(20)
    xb = rng.standard_normal((3000, 48), dtype=np.float32)
    xq = rng.standard_normal((20, 48), dtype=np.float32)
    idx = gpu_faiss.IndexFlat(48, 1)
    idx.add(xb)
    D19, I19 = idx.search(xq[:19], 5)
    assert idx.last_scan()["kernel"].endswith("l2diff")
    D20, I20 = idx.search(xq, 5)
    assert not idx.last_scan()["kernel"].endswith("l2diff")
    _assert_same(D19, I19, *oracle.flat_search(xb, xq[:19], 5, 1))
    _assert_same(D20, I20, *oracle.flat_search(xb, xq, 5, 1))
    assert (D19.view(np.uint32) != D20[:19].view(np.uint32)).any(), "two formulas, two roundings"
    # inner product has one formula
    ip = gpu_faiss.IndexFlat(48, 0)
    ip.add(xb)
    Da, Ia = ip.search(xq[:19], 5)
    Db, Ib = ip.search(xq, 5)
    assert np.array_equal(Da.view(np.uint32), Db[:19].view(np.uint32)) and np.array_equal(Ia, Ib[:19])


def test_streaming_searches_back_to_back_reuse_the_reset_state(gpu_faiss, oracle):
    """A streaming search whose predecessor had the same shape skips its state-reset launch (the predecessor's final
    selection left the thresholds, counts, publications and ticket counters reset).  Sequences of searches with changing
    queries, batch sizes, k and metrics -- and the same sequence with the reset launch forced (flags 64) -- all return the
    oracle's bits."""
    rng = np.random.default_rng(64)
    nb, d = 620_000, 32
    xb = rng.standard_normal((nb, d), dtype=np.float32)
    xb[1000:1100] = xb[:100]
    qs = [rng.standard_normal((nq, d), dtype=np.float32) for nq in (32, 32, 32, 7, 7, 32, 1, 1, 64, 64, 32)]
    ks = [100, 100, 10, 10, 10, 100, 5, 5, 20, 20, 100]
    for metric in (0, 1):
        for flags in (0, 64):
            idx = gpu_faiss.IndexFlat(d, metric)
            idx.add(xb)
            idx.set_tuning(0, 0, flags)
            for q, k in zip(qs, ks):
                D, I = idx.search(q, k)
                _assert_same(D, I, *oracle.flat_search(xb, q, k, metric))
            # the index's own rows as queries, twice in a row (squared L2: a row finds its lower-id duplicate first)
            for _ in range(2):
                D, I = idx.search(xb[1000:1032], 3)
                _assert_same(D, I, *oracle.flat_search(xb, xb[1000:1032], 3, metric))
                if metric == 1:
                    assert (I[:, 0] == np.arange(32)).all() and (D[:, 0] == 0).all()


def test_normalize_l2_in_chunks_matches_oracle(gpu_faiss, oracle):
    """faiss.normalize_L2 on a host array larger than one staging chunk (64 MB): two pooled device buffers and two streams
    take turns; every row carries the oracle's bits, zero rows stay untouched, the array is normalised in place."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((70_001, 512), dtype=np.float32) * rng.uniform(0.01, 30.0, (70_001, 1)).astype(np.float32)
    x[123] = 0.0
    x[-1] = 0.0
    want = x.copy()
    oracle.normalize_l2(want)
    held = x
    gpu_faiss.normalize_L2(held)
    assert held is x and np.array_equal(x.view(np.uint32), want.view(np.uint32))
    assert not x[123].any() and not x[-1].any()


@pytest.mark.gpu
def test_l2_formula_follows_the_whole_batch_not_its_pieces(gpu_faiss, oracle):
    """FAISS picks the squared-L2 formula by the batch handed to index.search (fewer than 20 queries: differences).  The
    library works through a large batch in blocks of 16384 queries: a last block of fewer than 20 must keep the norm
    formula of the whole; knn_flat_set_batch lets a caller that splits its batch itself say the same."""
    rng = np.random.default_rng(16389)
    nb, d, k = 400, 40, 6
    xb = rng.standard_normal((nb, d), dtype=np.float32)
    xq = rng.standard_normal((3 * 16384 + 5, d), dtype=np.float32)  # (host blocks of 16384: the last one has 5 queries)
    idx = gpu_faiss.IndexFlat(d, 1)
    idx.add(xb)
    D, I = idx.search(xq, k)
    _assert_same(D, I, *oracle.flat_search(xb, xq, k, 1))
    # the same five queries as a batch of their own: the other formula
    D5, I5 = idx.search(xq[-5:], k)
    assert idx.last_scan()["kernel"].endswith("l2diff")
    _assert_same(D5, I5, *oracle.flat_search(xb, xq[-5:], k, 1))
    assert (D5.view(np.uint32) != D[-5:].view(np.uint32)).any()
    # ... and as a declared piece of a batch of 40: the norm formula again
    idx.set_batch(40)
    Dp, Ip = idx.search(xq[-5:], k)
    assert not idx.last_scan()["kernel"].endswith("l2diff")
    _assert_same(Dp, Ip, *oracle.flat_search(xb, xq[-5:], k, 1, l2_mode=1))
    idx.set_batch(0)
    Dq, Iq = idx.search(xq[-5:], k)
    _assert_same(Dq, Iq, D5, I5)
    # whole-index self-search in blocks (symmetric launch off): 16384 + 7 rows
    xs = rng.standard_normal((16384 + 7, 24), dtype=np.float32)
    s = gpu_faiss.IndexFlat(24, 1)
    s.add(xs)
    s.set_tuning(0, 0, 1024)
    Ds, Is = s.search_self(4)
    Do, Io = oracle.flat_search(xs, xs[-40:], 4, 1)
    _assert_same(Ds[-40:], Is[-40:], Do, Io)


@pytest.mark.gpu
@pytest.mark.parametrize("metric", [0, 1])
def test_remainder_behind_the_full_query_tiles_is_searched_on_its_own(gpu_faiss, oracle, metric):
    """On a database that is streamed from HBM the queries behind the last full 128-query tile are searched with the
    narrowest build that holds them (<= 32: streaming build, <= 48 / 64 / 96: the 48- / 64- / 96-query builds) instead of a padded
    128-query tile.  Same bits as one launch (flags 16384) and as the oracle, whatever the split."""
    rng = np.random.default_rng(129 + metric)
    nb, d, k = 300_000, 32, 20
    xb = rng.standard_normal((nb, d), dtype=np.float32)
    xb[5000:5050] = xb[:50]
    idx = gpu_faiss.IndexFlat(d, metric)
    idx.add(xb)
    for nq in (129, 150, 161, 200, 96, 70, 64, 128, 257, 353):
        xq = rng.standard_normal((nq, d), dtype=np.float32)
        xq[-1] = xb[17]
        idx.set_tuning(0, 0, 0)
        D, I = idx.search(xq, k)
        last = idx.last_scan()["kernel"]
        r = nq % 128

        def build(m):  # the narrowest build that holds m queries
            return ("flat_scan_q32_d256" if m <= 32 else ("flat_scan_q48_d256" if m <= 48 else ("flat_scan_q64_d128" if m <= 64 else
                    ("flat_scan_q96_d128" if m <= 96 else "flat_scan_q128_d128"))))
        assert last == (build(r) if 0 < r <= 96 else "flat_scan_q128_d128"), (nq, last)
        idx.set_tuning(0, 0, 16384)
        D1, I1 = idx.search(xq, k)
        assert idx.last_scan()["kernel"] == (build(nq) if nq <= 128 else "flat_scan_q128_d128")
        _assert_same(D, I, D1, I1)
        _assert_same(D, I, *oracle.flat_search(xb, xq, k, metric))


def test_read_rate_aid_reads_the_index_rows(gpu_faiss):
    """knn_flat_read_rate (bench.py's "box_read_rate"): a plain read pass over the index's own (padded) rows"""
    import ctypes
    from knn_for_homology_amd import _lib
    rng = np.random.default_rng(1)
    idx = gpu_faiss.IndexFlat(100, 0)  # (rows are padded to 128 floats)
    idx.add(rng.standard_normal((50_000, 100), dtype=np.float32))
    ms, nbytes = ctypes.c_float(), ctypes.c_int64()
    _lib.check(_lib.lib().knn_flat_read_rate(idx._h, 2, ctypes.byref(ms), ctypes.byref(nbytes)))
    assert nbytes.value == 50_000 * 128 * 4 and 0.0 < ms.value < 100.0
    empty = gpu_faiss.IndexFlat(100, 0)
    _lib.check(_lib.lib().knn_flat_read_rate(empty._h, 2, ctypes.byref(ms), ctypes.byref(nbytes)))
    assert nbytes.value == 0 and ms.value == 0.0


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("k", [1, 40, 64, 100, 129, 301, 420, 600, 819, 820])
def test_one_wave_final_selection_sorts_in_registers(gpu_faiss, oracle, k, metric):
    """The batch regime's final selection (one wave per query, select_topk_kernel<R, 64>) sorts its k..1.25 k survivors in
    registers (wave_bitonic_sort_regs<E>: 64 E keys, E = 1 .. 16) and falls back to the LDS sort beyond 1024: every E, the
    boundary (k = 819 -> 1023 survivors at most, 820 -> 1025), ties (duplicated rows) -- oracle bits."""
    rng = np.random.default_rng(1000 + k)
    nb, nq, d = 16384, 640, 32  # (>= 512 queries, statistically seeded: ~1.3 j N / S + 1.25 k <= 2048 candidates per query up to k ~ 650)
    xb = rng.standard_normal((nb, d), dtype=np.float32)
    xb[3000:3100] = xb[:100]
    xq = rng.standard_normal((nq, d), dtype=np.float32)
    xq[:5] = xb[:5]
    idx = gpu_faiss.IndexFlat(d, metric)
    idx.add(xb)
    D, I = idx.search(xq, k)
    assert k < 40 or idx.last_seed()["stat_rank"] > 0, idx.last_seed()  # (no estimate for a single neighbour)
    Do, Io = oracle.flat_search(xb, xq, k, metric)
    _assert_same(D, I, Do, Io)


def test_results_that_are_kept_stop_taking_new_page_locked_blocks(gpu_faiss):
    """cath/search.py:37-50 keeps the hits of every file of a metric until it saves them: from the third live result of a size
    class on, a search returns plain arrays (a staged download) instead of page-locking one more block per search; dropping
    the results brings the pooled blocks back."""
    import gc
    from knn_for_homology_amd import _lib
    rng = np.random.default_rng(5)
    xb = rng.standard_normal((2000, 16), dtype=np.float32)
    xq = rng.standard_normal((5000, 16), dtype=np.float32)
    idx = gpu_faiss.IndexFlat(16, 0)
    idx.add(xb)
    gc.collect()
    cls = 1 << (5000 * 300 * 4 - 1).bit_length()
    live0 = _lib._pinned_live.get(cls, 0)
    kept = [idx.search(xq, 300) for _ in range(_lib.PINNED_LIVE_PER_CLASS - live0 + 2)]   # D: 6 MB (class 8 MB), I: 12 MB (class 16 MB)
    pinned = [not D.flags.owndata for D, _ in kept]
    assert pinned[0] and not pinned[-1] and not pinned[-2], pinned
    assert all(np.array_equal(kept[0][1], I) for _, I in kept)   # the same result either way
    del kept
    gc.collect()
    D, I = idx.search(xq, 300)
    assert not D.flags.owndata and not I.flags.owndata


@pytest.mark.parametrize("k", [20, 1000])
def test_symmetric_self_search_beyond_131072_rows_keeps_its_estimate(gpu_faiss, oracle, k):
    """From 131 072 rows on the symmetric self-search samples every 64th row (every 32nd below): the rank of its bound and the
    sample it is read from belong together -- the search stays on the symmetric launch, no estimate fails (a mismatch would
    still return the exact result, through the plain path's repeat, at three times the cost)."""
    rng = np.random.default_rng(140 + k)
    n, d = 140_000, 16
    cent = rng.standard_normal((700, d), dtype=np.float32)
    x = (cent[rng.integers(0, 700, n)] + 0.4 * rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
    idx = gpu_faiss.IndexFlat(d, 0)
    idx.add(x)
    before = idx.last_seed()["stat_redo"]
    D, I = idx.search_self(k)
    assert idx.last_scan()["kernel"] == "flat_scan_q128_d128_sym" and idx.last_seed()["stride"] == 64, (idx.last_scan(), idx.last_seed())
    assert idx.last_seed()["stat_redo"] == before
    sample = rng.choice(n, 16, replace=False)
    Do, Io = oracle.flat_search(x, x[sample], k, 0, l2_mode=1)
    _assert_same(D[sample], I[sample], Do, Io)
