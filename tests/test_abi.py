"""CPU: the C-ABI library loads and exports every symbol include/knn355.h declares; the
Python boundary mirrors the faiss wrapper's argument checks; no compute happens here."""
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / "include" / "knn355.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(knn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from knn_for_homology_amd import _lib
    L = _lib.lib()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"libknn355.so does not export {n}"
    # and the ctypes table covers the header
    assert set(names) <= set(_lib.EXPORTS), set(names) - set(_lib.EXPORTS)


def test_version_and_device_count_need_no_gpu():
    from knn_for_homology_amd import _lib
    L = _lib.lib()
    assert b"knn355" in L.knn_version()
    assert _lib.device_count() >= 0


def test_argument_checks_mirror_faiss():
    from knn_for_homology_amd import faiss
    with pytest.raises(TypeError):
        faiss.normalize_L2(np.zeros((2, 4), np.float64))
    with pytest.raises(ValueError):
        faiss.normalize_L2(np.zeros(8, np.float32))
    with pytest.raises(ValueError):
        faiss.normalize_L2(np.zeros((4, 8), np.float32)[:, ::2])
    with pytest.raises(TypeError):
        faiss.normalize_L2([[1.0, 2.0]])
    assert faiss.METRIC_INNER_PRODUCT == 0 and faiss.METRIC_L2 == 1


def test_no_silent_cpu_fallback():
    """Without a device the compute entry points raise; with one they must work -- either
    way nothing is served from the CPU."""
    from knn_for_homology_amd import _lib, faiss
    if _lib.device_count() == 0:
        with pytest.raises(_lib.Knn355Error, match="no HIP device"):
            faiss.IndexFlat(8, faiss.METRIC_L2)
        with pytest.raises(_lib.Knn355Error, match="no HIP device"):
            faiss.normalize_L2(np.ones((2, 8), np.float32))


def test_product_never_imports_oracle():
    pkg = ROOT / "knn-for-homology_amd"
    for f in pkg.rglob("*.py"):
        assert "oracle" not in f.read_text(), f"{f} mentions the oracle"
    for f in (pkg / "csrc").glob("*"):
        if f.suffix in (".hip", ".cpp", ".h"):
            assert "knn_oracle" not in f.read_text().replace("oracle/knn_oracle.c", "")
    # developer tools are not allowed to use it either: only tests/, smoke() and bench.py's cpu_baseline
    for f in (ROOT / "tools").rglob("*.py"):
        assert "oracle" not in f.read_text(), f"{f} mentions the oracle"


def test_entry_points_importable():
    from knn_for_homology_amd.cath.search import search, search_and_save  # noqa: F401
    from knn_for_homology_amd.pfam.proteins_search import main, naturalsize  # noqa: F401
    from knn_for_homology_amd.pfam.search import load_embeddings, search_flat, search_index  # noqa: F401
    from knn_for_homology_amd.pfam.slices.slices_search import main as m2  # noqa: F401
    from knn_for_homology_amd.seqvec_search.main import faiss_search, evaluate_faiss, evaluate  # noqa: F401
    from knn_for_homology_amd.seqvec_search.create_index import main as m3  # noqa: F401
    assert naturalsize(819200128) == "819.2 MB"
    assert naturalsize(512) == "512 Bytes"
