"""Data locations used by the reference's scripts.

The reference resolves everything from ``git rev-parse --show-toplevel``
(cath/cath_shared.py:14-25, pfam/pfam_shared.py:10-22, pfam/proteins_shared.py:19-21,
pfam/slices/slices_shared.py:3-6).  Here the root is ``$KNN355_PROJECT_ROOT`` when set,
otherwise the git toplevel of the current directory, otherwise the current directory --
so the drop-in scripts read and write the very same files when run from a checkout of
the reference.
"""
import os
import subprocess
from pathlib import Path


def project_root() -> Path:
    env = os.environ.get("KNN355_PROJECT_ROOT")
    if env:
        return Path(env)
    try:
        top = subprocess.check_output(["git", "rev-parse", "--show-toplevel"], text=True,
                                      stderr=subprocess.DEVNULL).strip()
        return Path(top)
    except (subprocess.CalledProcessError, FileNotFoundError):
        return Path()


def cath_data() -> Path:
    return project_root() / "cath" / "data"


def pfam_dir() -> Path:
    return project_root() / "pfam"


def full_sequences_data() -> Path:
    return pfam_dir() / "full_sequences_data"


def slices_data() -> Path:
    return pfam_dir() / "slices_data"


def subset10() -> Path:
    return pfam_dir() / "subset10"


def subset10_t5() -> Path:
    return pfam_dir() / "subset10_t5"
