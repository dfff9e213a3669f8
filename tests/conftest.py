import os
import sys
from pathlib import Path

import pytest

try:  # torch first: it and libknn355 must share one HIP runtime in this process
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"
DATASETS = ["small-random", "pfam-20-10", "pfam-20-10-sum", "pfam-20-dist"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import knn_oracle as ko
    ko.build()
    return ko.oracle()


@pytest.fixture(scope="session")
def ko():
    from oracle import knn_oracle
    return knn_oracle


@pytest.fixture(scope="session")
def gpu_faiss():
    """The product facade; fails loudly (no skip, no fallback) when the HIP library or a
    device is missing -- GPU tests must never pass on a silent CPU path."""
    from knn_for_homology_amd import _lib, faiss
    _lib.lib()
    if _lib.device_count() < 1:
        raise RuntimeError("gpu test collected but no HIP device is visible")
    return faiss
