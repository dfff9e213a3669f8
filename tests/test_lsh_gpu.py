"""GPU: IndexLSH (sign-bit codes + Hamming top-k) and the entry points that use it
(seqvec_search/create_index.py, pfam/search.py).  FAISS's rotation matrix comes from its own
RNG, so the check is (a) exactness of OUR pipeline against a numpy restatement using the same
matrix, (b) recall against the flat search, (c) the reference's file protocol."""
import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _numpy_lsh(rot, xb, xq, k):
    def codes(x):
        return (x.astype(np.float64) @ rot.astype(np.float64).T) >= 0
    cb, cq = codes(xb), codes(xq)
    dist = (cq[:, None, :] != cb[None, :, :]).sum(-1)
    order = np.argsort(dist, axis=1, kind="stable")[:, :k]
    return np.take_along_axis(dist, order, 1).astype(np.float32), order.astype(np.int64), cb


def test_lsh_matches_numpy_restatement(gpu_faiss):
    from knn_for_homology_amd.lsh import IndexLSH
    rng = np.random.default_rng(3)
    for d, nbits, nb, nq, k in ((64, 128, 3000, 40, 17), (100, 256, 2000, 33, 50), (1024, 1024, 1500, 20, 100), (96, 200, 700, 9, 5)):
        xb = rng.standard_normal((nb, d), dtype=np.float32)
        xq = rng.standard_normal((nq, d), dtype=np.float32)
        idx = IndexLSH(d, nbits)
        assert idx.is_trained
        idx.train(xb)
        idx.add(xb)
        assert idx.ntotal == nb and idx.code_size == (nbits + 7) // 8
        D, I = idx.search(xq, k)
        Dn, In, cb = _numpy_lsh(idx._rot, xb, xq, k)
        # projections that land within fp32 noise of 0 may flip a bit: compare codes first
        got = np.unpackbits(idx.codes(), axis=1, bitorder="little")[:, :nbits].astype(bool)
        flips = (got != cb).sum()
        assert flips <= 1e-5 * cb.size + 2
        if flips == 0:
            assert np.array_equal(D, Dn) and np.array_equal(I, In)
        else:
            assert np.abs(D - Dn).max() <= flips
        assert (np.diff(D, axis=1) >= 0).all() and D.dtype == np.float32 and I.dtype == np.int64


def test_lsh_recall_and_roundtrip(gpu_faiss, tmp_path):
    x = np.load(GOLDEN / "pfam-20-10" / "train.npy")
    q = np.load(GOLDEN / "pfam-20-10" / "test.npy")
    gpu_faiss.normalize_L2(x)
    gpu_faiss.normalize_L2(q)
    flat = gpu_faiss.IndexFlat(1024, gpu_faiss.METRIC_INNER_PRODUCT)
    flat.add(x)
    _, It = flat.search(q, 10)
    lsh = gpu_faiss.IndexLSH(1024, 1024)
    lsh.train(x)
    lsh.add(x)
    D, I = lsh.search(q, 10)
    rec = sum(len(np.intersect1d(a, b)) for a, b in zip(I, It)) / It.size
    assert rec > 0.6, rec  # 1024 sign bits of 1024-d cosine data
    f = tmp_path / "lsh.index"
    gpu_faiss.write_index(lsh, str(f))
    assert f.read_bytes()[:4] == b"IxHe"
    back = gpu_faiss.read_index(str(f))
    D2, I2 = back.search(q, 10)
    assert np.array_equal(D, D2) and np.array_equal(I, I2) and back.ntotal == 200 and back.nbits == 1024


def test_a_search_of_several_batches_equals_its_pieces(gpu_faiss):
    """IndexLSH.search scans 16384 queries at a time; with several batches the results of batch b - 1 are downloaded on the
    copy stream while batch b is scanned (two sets of result buffers): 40 000 queries in one call = the same queries in calls
    of one batch each, bit for bit (pfam/search.py searches every row of its index, k = 1000)."""
    n, d, nbits, k = 40_000, 64, 128, 25
    x = np.random.default_rng(3).standard_normal((n, d)).astype(np.float32)
    idx = gpu_faiss.IndexLSH(d, nbits)
    idx.train(x)
    idx.add(x)
    D, I = idx.search(x, k)
    assert (D[:, 0] == 0).all(), "every row's own code is at Hamming distance 0"
    for a in range(0, n, 16384):
        b = min(n, a + 16384)
        Dp, Ip = idx.search(x[a:b], k)
        assert np.array_equal(Ip, I[a:b]) and np.array_equal(Dp, D[a:b]), (a, b)


def test_create_index_entry_point(gpu_faiss, tmp_path):
    """tests/test_utils.py:17-21 of the reference: the index file gets written."""
    from knn_for_homology_amd.seqvec_search import create_index
    out = tmp_path / "index.bin"
    create_index.main(["--dir", str(GOLDEN / "pfam-20-10"), "--index", str(out)])
    assert out.exists()
    idx = gpu_faiss.read_index(str(out))
    assert idx.ntotal == 200 and idx.nbits == 1024
    # and the prebuilt-index branch of faiss_search (seqvec_search/main.py:40-41,132)
    from knn_for_homology_amd.seqvec_search.main import faiss_search
    q = np.load(GOLDEN / "pfam-20-10" / "test.npy")
    result, scores, _ = faiss_search(idx, q, 13)
    assert result.shape == (200, 13) and (np.diff(scores, axis=1) >= 0).all()


def test_pfam_search_entry_points(gpu_faiss, tmp_path):
    """pfam/search.py:14-53: flat and LSH searches of a train/test embedding set."""
    import shutil
    from knn_for_homology_amd.pfam import search as pfam_search
    for name in ("train.npy", "test.npy"):
        shutil.copyfile(GOLDEN / "pfam-20-10" / name, tmp_path / name)
    pfam_search.search_flat(tmp_path, k=100)
    pfam_search.search_index(tmp_path, k=100)
    assert (tmp_path / "index_lsh_1024.bin").is_file()
    fs, fh = np.load(tmp_path / "flat_scores.npy"), np.load(tmp_path / "flat_hits.npy")
    is_, ih = np.load(tmp_path / "index_scores.npy"), np.load(tmp_path / "index_hits.npy")
    assert fs.shape == fh.shape == is_.shape == ih.shape == (200, 100)
    assert (np.diff(fs, axis=1) <= 0).all() and (np.diff(is_, axis=1) >= 0).all()
    pfam_search.search_index(tmp_path, k=100)  # second call loads the cached index file
    assert np.array_equal(ih, np.load(tmp_path / "index_hits.npy"))
