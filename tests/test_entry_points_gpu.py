"""GPU: the file protocols of the remaining reference entry points (pfam/search.py,
pfam/slices/slices_search.py, seqvec_search/create_index.py) on fixture-sized data."""
import shutil

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_pfam_search_flat_and_lsh_files(gpu_faiss, oracle, tmp_path):
    """pfam/search.py:14-53: normalised train/test, flat IP and LSH-1024 searches, four .npy outputs,
    the LSH index cached as index_lsh_1024.bin and re-used."""
    from knn_for_homology_amd.pfam import search as pfam_search
    for name in ("train.npy", "test.npy"):
        shutil.copy(GOLDEN / "pfam-20-10" / name, tmp_path / name)
    train, test = pfam_search.load_embeddings(tmp_path)
    assert train.dtype == np.float32 and np.allclose((train * train).sum(1), 1.0, atol=1e-5)
    k = 17
    pfam_search.search_flat(tmp_path, k=k)
    scores, hits = np.load(tmp_path / "flat_scores.npy"), np.load(tmp_path / "flat_hits.npy")
    Do, Io = oracle.flat_search(train, test, k, 0)
    assert hits.dtype == np.int64 and np.array_equal(hits, Io) and np.array_equal(_bits(scores), _bits(Do))
    pfam_search.search_index(tmp_path, k=k)
    cache = tmp_path / "index_lsh_1024.bin"
    assert cache.is_file()
    first = np.load(tmp_path / "index_hits.npy").copy()
    stamp = cache.stat().st_mtime_ns
    pfam_search.search_index(tmp_path, k=k)  # second run reads the cached index
    assert cache.stat().st_mtime_ns == stamp and np.array_equal(np.load(tmp_path / "index_hits.npy"), first)
    assert first.shape == (test.shape[0], k) and np.load(tmp_path / "index_scores.npy").dtype == np.float32
    # 1024 sign bits of 1024-d vectors: the Hamming ranking agrees with the cosine ranking on the first hit mostly
    assert (first[:, 0] == Io[:, 0]).mean() > 0.5


def test_slices_search_protocol(gpu_faiss, oracle, tmp_path, capsys):
    """pfam/slices/slices_search.py:9-31: both sets searched against themselves, existing results skipped."""
    from knn_for_homology_amd.pfam.slices import slices_search
    x = np.load(GOLDEN / "pfam-20-10" / "train.npy")
    np.save(tmp_path / "slices.npy", x.astype(np.float16))  # the reference casts to float32 itself
    np.save(tmp_path / "full_sequences.npy", x[:77])
    np.save(tmp_path / "full_sequences_scores.npy", np.zeros(3, np.float32))  # pretend this set is done
    slices_search.main(data_dir=tmp_path, k=9)
    out = capsys.readouterr().out.split("\n")
    assert out[0].startswith("slices (") and float(out[1]) >= 0 and "full_sequences" not in "".join(out)
    assert np.load(tmp_path / "full_sequences_scores.npy").shape == (3,) and not (tmp_path / "full_sequences_hits.npy").exists()
    xs = x.astype(np.float16).astype(np.float32)
    oracle.normalize_l2(xs)
    Do, Io = oracle.flat_search(xs, xs, 9, 0)
    assert np.array_equal(np.load(tmp_path / "slices_hits.npy"), Io)
    assert np.array_equal(_bits(np.load(tmp_path / "slices_scores.npy")), _bits(Do))


def test_create_index_cli(gpu_faiss, tmp_path):
    """seqvec_search/create_index.py:16-47 and tests/test_utils.py:17-21 of the reference: the index file exists
    (and, here, loads back and searches)."""
    from knn_for_homology_amd.seqvec_search import create_index
    shutil.copy(GOLDEN / "pfam-20-10" / "train.npy", tmp_path / "train.npy")
    target = tmp_path / "out" / "lsh.index"
    target.parent.mkdir()
    create_index.main(["--dir", str(tmp_path), "--index", str(target), "--param", "256"])
    assert target.is_file() and target.stat().st_size > 0
    index = gpu_faiss.read_index(str(target))
    x = np.load(tmp_path / "train.npy").astype(np.float32)
    assert index.ntotal == x.shape[0] and index.d == x.shape[1]
    D, I = index.search(x[:20], 3)
    assert (I[:, 0] == np.arange(20)).all() and (D[:, 0] == 0).all()
    with pytest.raises(SystemExit):
        create_index.main(["--dir", str(tmp_path)])  # --index is required
