"""Dataset descriptor with the fields, properties and constructor of the reference's
``LoadedData`` (seqvec_search/data.py:9-50): a directory holding train/test ``.npy``
embeddings, ``.json`` id lists, ``ids_to_family.json`` and the FASTA files."""
import json
from pathlib import Path

from .constants import default_hits

# (attribute, file inside the dataset directory or None, loader)
_LAYOUT = (
    ("train", "train.npy", None),
    ("train_ids", "train.json", json.loads),
    ("test", "test.npy", None),
    ("test_ids", "test.json", json.loads),
    ("ids_to_family", "ids_to_family.json", json.loads),
    ("train_sequences", "train.fasta", None),
    ("test_sequences", "test.fasta", None),
)
_ORDER = ("path", "train", "train_ids", "knn_index", "test", "test_ids", "ids_to_family", "train_sequences",
          "test_sequences", "hits")


class LoadedData:
    """Positional / keyword construction in the reference's field order; ``hits`` defaults to
    ``default_hits``.  Compares and prints like the reference's dataclass."""

    def __init__(self, *args, **kwargs):
        values = dict(zip(_ORDER, args))
        if len(args) > len(_ORDER):
            raise TypeError(f"LoadedData takes at most {len(_ORDER)} arguments")
        for key, value in kwargs.items():
            if key not in _ORDER or key in values:
                raise TypeError(f"LoadedData: unexpected or repeated argument {key!r}")
            values[key] = value
        values.setdefault("hits", default_hits)
        missing = [name for name in _ORDER if name not in values]
        if missing:
            raise TypeError(f"LoadedData: missing {', '.join(missing)}")
        for name in _ORDER:
            setattr(self, name, values[name])

    def _astuple(self):
        return tuple(getattr(self, name) for name in _ORDER)

    def __eq__(self, other):
        return isinstance(other, LoadedData) and self._astuple() == other._astuple()

    def __repr__(self):
        return "LoadedData(" + ", ".join(f"{name}={getattr(self, name)!r}" for name in _ORDER) + ")"

    def _mmseqs(self, leaf=None):
        base = Path(self.path) / "mmseqs_dbs"
        return base if leaf is None else base / leaf

    mmseqs_dir = property(lambda self: self._mmseqs())
    mmseqs_test = property(lambda self: self._mmseqs("test"))
    mmseqs_train = property(lambda self: self._mmseqs("train"))

    @classmethod
    def from_options(cls, path, hits=default_hits, knn_index=None):
        root = Path(path)
        found = {}
        for attribute, filename, loader in _LAYOUT:
            location = root / filename
            found[attribute] = loader(location.read_text()) if loader else location
        return cls(path=root, knn_index=knn_index, hits=hits, **found)
