"""Dataset descriptor with the same fields as the reference's ``LoadedData``
(seqvec_search/data.py:9-50): a directory holding train/test ``.npy`` embeddings,
``.json`` id lists, ``ids_to_family.json`` and the FASTA files."""
import json
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, List, Optional

from .constants import default_hits


@dataclass
class LoadedData:
    path: Path
    train: Path
    train_ids: List[str]
    knn_index: Optional[Path]
    test: Path
    test_ids: List[str]
    ids_to_family: Dict[str, str]
    train_sequences: Path
    test_sequences: Path
    hits: int = default_hits

    @property
    def mmseqs_dir(self) -> Path:
        return self.path / "mmseqs_dbs"

    @property
    def mmseqs_test(self) -> Path:
        return self.mmseqs_dir / "test"

    @property
    def mmseqs_train(self) -> Path:
        return self.mmseqs_dir / "train"

    @classmethod
    def from_options(cls, path: Path, hits: int = default_hits, knn_index: Optional[Path] = None) -> "LoadedData":
        path = Path(path)

        def ids(name):
            return json.loads((path / name).read_text())

        return cls(path=path, train=path / "train.npy", train_ids=ids("train.json"), knn_index=knn_index,
                   test=path / "test.npy", test_ids=ids("test.json"), ids_to_family=ids("ids_to_family.json"),
                   train_sequences=path / "train.fasta", test_sequences=path / "test.fasta", hits=hits)
