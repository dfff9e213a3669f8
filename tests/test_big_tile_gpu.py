"""GPU parity tests of the 256 x 256 tile (flat_scan_kernel<2, 2, 4, 4>: one workgroup per CU, the accumulators of a wave's
128 x 128 scores in AGPRs, the late-barrier K loop): the batch regime's build for Pfam-sized searches.  The plan picks it by
itself for long chunks only (make_plan / big_tile_pays); here it is FORCED (set_tuning query_tile = 256, or flags 524288 --
the symmetric self-search ignores a forced tile) onto small shapes and compared bit for bit with the CPU oracle and with the
128 x 128 build (flags 262144: never the 256 tile).  Every distance is one fma chain whatever the tile: same bits."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BIG, NOBIG = 524288, 262144


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _assert_same(D, I, Do, Io):
    assert np.array_equal(I, Io), f"{int((I != Io).sum())} neighbour ids differ"
    assert np.array_equal(_bits(D), _bits(Do)), f"{int((_bits(D) != _bits(Do)).sum())} distances differ"


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("nq,nb,d,k,nch", [
    (129, 5000, 1024, 100, 0),     # two query tiles, the second almost empty; ragged last database tile
    (256, 4096, 1024, 1, 1),       # exactly one tile each way, one chunk
    (300, 5000, 1024, 301, 3),
    (300, 5000, 1024, 1000, 0),
    (257, 6000, 256, 1536, 0),     # 4096-key lists (k beyond the register select)
    (260, 6000, 128, 2048, 2),
    (513, 2049, 100, 13, 0),       # d not a multiple of 32 (zero padded), nb = 8 tiles + 1 row
    (700, 700, 37, 11, 0),
    (1000, 20000, 64, 10, 7),
    (2100, 9000, 32, 50, 0),       # one K step per tile: the K loop's last-step form alone
    (400, 3000, 64, 5, 0),         # two K steps: first and last form
    (400, 3000, 96, 5, 0),         # three K steps: all three forms once
])
def test_big_tile_vs_oracle(gpu_faiss, oracle, nq, nb, d, k, nch, metric):
    rng = np.random.default_rng(nq * 7919 + nb + d + k)
    xb = rng.standard_normal((nb, d), dtype=np.float32)
    xb[nb // 2: nb // 2 + 5] = xb[:5]  # exact duplicates: ties -> lower id
    xq = rng.standard_normal((nq, d), dtype=np.float32)
    xq[:3] = xb[:3]
    idx = gpu_faiss.IndexFlat(d, metric)
    idx.add(xb)
    idx.set_tuning(256, nch, 0)
    D, I = idx.search(xq, k)
    assert idx.last_scan()["query_tile"] == 256 and idx.last_scan()["kernel"] == "flat_scan_q256_d256", idx.last_scan()
    Do, Io = oracle.flat_search(xb, xq, k, metric)
    _assert_same(D, I, Do, Io)
    # the flag form (what a whole-index self-search honours), unseeded and under a forced statistical seed
    for flags in (BIG, BIG | 8, BIG | 128):
        idx.set_tuning(0, nch, flags)
        D2, I2 = idx.search(xq, k)
        assert idx.last_scan()["query_tile"] == (256 if nb >= 1024 else 128)  # (the flag spares tiny databases: seed samples)
        _assert_same(D2, I2, Do, Io)
    idx.set_tuning(0, nch, NOBIG)
    D3, I3 = idx.search(xq, k)
    assert idx.last_scan()["query_tile"] != 256
    _assert_same(D3, I3, Do, Io)


def test_big_tile_result_is_independent_of_the_chunk_split(gpu_faiss):
    rng = np.random.default_rng(81)
    xb = rng.standard_normal((30000, 128), dtype=np.float32)
    xq = rng.standard_normal((777, 128), dtype=np.float32)
    idx = gpu_faiss.IndexFlat(128, 0)
    idx.add(xb)
    ref = None
    for qt, nch in ((128, 0), (256, 1), (256, 5), (256, 37), (256, 118), (0, 0)):
        idx.set_tuning(qt, nch, 0)
        D, I = idx.search(xq, 100)
        if ref is None:
            ref = (D, I)
        else:
            _assert_same(D, I, *ref)


def test_small_batches_never_take_the_big_tile(gpu_faiss):
    rng = np.random.default_rng(82)
    xb = rng.standard_normal((5000, 64), dtype=np.float32)
    idx = gpu_faiss.IndexFlat(64, 0)
    idx.add(xb)
    for nq, want in ((20, 32), (100, 128), (128, 128)):
        idx.set_tuning(256, 0, 0)
        idx.search(xb[:nq], 10)
        assert idx.last_scan()["query_tile"] == want, (nq, idx.last_scan())
    # the plan's own choice: a CATH-sized search stays on the 128 x 128 tile (short chunks), flags or not
    idx.set_tuning(0, 0, 0)
    idx.search(xb[:3000], 10)
    assert idx.last_scan()["query_tile"] == 128


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("n,d,k", [(3000, 200, 11), (5000, 1024, 301), (8195, 64, 1000), (20000, 96, 100)])
def test_symmetric_self_search_on_big_tiles(gpu_faiss, oracle, n, d, k, metric):
    """The symmetric launch on 256-row tiles (flat_scan_kernel<2, 2, 4, 4, ..., SYM>): the bits of the 128-row symmetric
    launch, of the plain search, and of the oracle -- duplicates across distant tiles, a ragged last tile."""
    rng = np.random.default_rng(n + k)
    x = rng.standard_normal((n, d), dtype=np.float32)
    x[n - 300:n - 200] = x[100:200]
    x[2000:2050] = x[1950:2000]
    idx = gpu_faiss.IndexFlat(d, metric)
    idx.add(x)
    idx.set_tuning(0, 0, BIG)
    D, I = idx.search_self(k)
    assert idx.last_scan()["kernel"] == "flat_scan_q256_d256_sym" and idx.last_seed()["stat_rank"] > 0, idx.last_scan()
    idx.set_tuning(0, 0, NOBIG)
    D1, I1 = idx.search_self(k)
    assert idx.last_scan()["kernel"] == "flat_scan_q128_d128_sym"
    _assert_same(D, I, D1, I1)
    idx.set_tuning(0, 0, BIG | 1024)
    Dp, Ip = idx.search_self(k)
    assert idx.last_scan()["kernel"] == "flat_scan_q256_d256"
    _assert_same(D, I, Dp, Ip)
    sample = np.concatenate([rng.choice(n, 24, replace=False), [0, 255, 256, n - 1, 100, n - 300, 2000, 1950]])
    Do, Io = oracle.flat_search(x, x[sample], k, metric)
    _assert_same(D[sample], I[sample], Do, Io)


def test_symmetric_big_tiles_repair_a_failed_estimate(gpu_faiss, oracle):
    """The adversarial database of the statistical-seed test against itself on 256-row tiles: the verification fails, the plain
    path repeats the search -- exact result."""
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
    idx.set_tuning(0, 0, BIG)
    before = idx.last_seed()["stat_redo"]
    D, I = idx.search_self(k)
    assert idx.last_seed()["stat_redo"] > before
    sample = np.concatenate([near[:20], rng.choice(n, 20, replace=False)])
    _assert_same(D[sample], I[sample], *oracle.flat_search(x, x[sample], k, 0))


def test_half_a_wide_query_tile_goes_with_the_remainder(gpu_faiss, oracle):
    """A database that is streamed from HBM (>= 2^18 rows: batches are cut into pieces) and a batch the 256 x 256 tile serves whose
    full 128-query tiles end in half a wide tile: 640 queries = 512 on the wide tile + 128 on the 128 x 128 tile, 936 = 768 +
    128 + a 40-query remainder -- not a third / fourth wide query tile that is half empty.  Same bits as the oracle."""
    rng = np.random.default_rng(91)
    nb, d, k = 300_000, 32, 10
    xb = rng.standard_normal((nb, d), dtype=np.float32)
    xb[200_000:200_050] = xb[:50]
    idx = gpu_faiss.IndexFlat(d, 0)
    idx.add(xb)
    idx.set_tuning(0, 0, BIG)
    for nq, last_tile in ((512, 256), (640, 128), (936, 48), (1024, 256)):
        xq = rng.standard_normal((nq, d), dtype=np.float32)
        xq[:3] = xb[:3]
        D, I = idx.search(xq, k)
        assert idx.last_scan()["query_tile"] == last_tile, (nq, idx.last_scan())
        _assert_same(D, I, *oracle.flat_search(xb, xq, k, 0))
