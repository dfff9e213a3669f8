"""GPU: the all-vs-all entry points under a multi-rank launch (knn_for_homology_amd/ranks.py), rehearsed with two and
three ranks on ONE GPU (KNN355_REHEARSE_ONE_GPU=1: gloo, every rank on GPU 0 -- RCCL refuses two ranks per device).
The ranks find RANK / WORLD_SIZE / LOCAL_RANK in their environment, as under ``python -m torch.distributed.run``, and join
the group themselves.  What they return and write must be what ONE process returns and writes, bit for bit."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world),
                       "LOCAL_RANK": str(rank), "KNN355_REHEARSE_ONE_GPU": "1"})
    from knn_for_homology_amd import ranks
    from knn_for_homology_amd.cath import search as cath_search
    from knn_for_homology_amd.pfam import proteins_search, search as pfam_search
    from knn_for_homology_amd.pfam.slices import slices_search
    out = Path(out_dir)
    assert ranks.launched_group() == (rank, world)
    import torch.distributed as dist
    assert dist.is_initialized() and dist.get_backend() == "gloo"
    x = np.load(out / "cath" / "a.npy").astype(np.float32)
    res = {}
    for metric in (0, 1):
        res[f"h{metric}"], res[f"s{metric}"] = cath_search.search(x, hits=300, metric=metric)
    np.savez(out / f"e{rank}.npz", **res)
    cath_search.search_and_save(out / "cath")
    for mode in ("flat", "hnsw"):
        proteins_search.main(["proteins_search", mode], data_dir=out / "proteins", k=50)
    pfam_search.search_flat(out / "pfam", k=40)
    pfam_search.search_index(out / "pfam", k=40)
    slices_search.main(out / "slices", k=30)
    dist.barrier()
    dist.destroy_process_group()


def _make_inputs(root: Path):
    rng = np.random.default_rng(81)
    for sub in ("cath", "proteins", "pfam", "slices"):
        (root / sub).mkdir(parents=True)
    cent = rng.standard_normal((40, 96), dtype=np.float32)
    def clustered(n):
        return (cent[rng.integers(0, 40, n)] + 0.4 * rng.standard_normal((n, 96), dtype=np.float32)).astype(np.float32)
    np.save(root / "cath" / "a.npy", clustered(9001).astype(np.float16))  # (>= 8192 rows: one process takes the symmetric launch)
    np.save(root / "cath" / "b.npy", clustered(700))
    np.save(root / "proteins" / "full_sequences.npy", clustered(3000))
    np.save(root / "pfam" / "train.npy", clustered(2500))
    np.save(root / "pfam" / "test.npy", clustered(333))
    np.save(root / "slices" / "slices.npy", clustered(1200))
    np.save(root / "slices" / "full_sequences.npy", clustered(10))


@pytest.mark.parametrize("world", [2, 3])
def test_entry_points_under_a_multi_rank_launch(tmp_path, oracle, world):
    import torch.multiprocessing as mp
    many, one = tmp_path / "many", tmp_path / "one"
    _make_inputs(many)
    _make_inputs(one)
    mp.spawn(_worker, args=(world, _free_port(), str(many)), nprocs=world, join=True)
    # the same calls in this process: one GPU, no group
    sys.path.insert(0, str(ROOT))
    from knn_for_homology_amd import ranks
    from knn_for_homology_amd.cath import search as cath_search
    from knn_for_homology_amd.pfam import proteins_search, search as pfam_search
    from knn_for_homology_amd.pfam.slices import slices_search
    assert ranks.launched_group() == (0, 1)
    x = np.load(one / "cath" / "a.npy").astype(np.float32)
    for metric in (0, 1):
        h, s = cath_search.search(x, hits=300, metric=metric)
        for r in range(world):
            got = np.load(many / f"e{r}.npz")
            assert np.array_equal(got[f"h{metric}"], h), (metric, r)
            assert np.array_equal(got[f"s{metric}"].view(np.uint32), s.view(np.uint32)), (metric, r)
        # ... and the oracle on a sample of the rows
        xs = x.copy()
        if metric == 0:
            oracle.normalize_l2(xs)
        Do, Io = oracle.flat_search(xs, xs[4000:4040], 301, metric, l2_mode=1)
        assert np.array_equal(h[4000:4040], Io[:, 1:]) and np.array_equal(s[4000:4040].view(np.uint32), Do[:, 1:].view(np.uint32))
    cath_search.search_and_save(one / "cath")
    for mode in ("flat", "hnsw"):
        proteins_search.main(["proteins_search", mode], data_dir=one / "proteins", k=50)
    pfam_search.search_flat(one / "pfam", k=40)
    pfam_search.search_index(one / "pfam", k=40)
    slices_search.main(one / "slices", k=30)

    def same_npy(rel):
        a, b = np.load(many / rel), np.load(one / rel)
        assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8)), rel

    for label in ("cosine", "euclidean"):
        for kind in ("hits", "scores"):
            a, b = np.load(many / "cath" / f"{kind}_{label}.npz"), np.load(one / "cath" / f"{kind}_{label}.npz")
            assert sorted(a.files) == sorted(b.files) == ["a", "b"]
            for stem in a.files:
                assert np.array_equal(a[stem].view(np.uint8), b[stem].view(np.uint8)), (label, kind, stem)
        assert (many / "cath" / f"a.{label}-search-time.txt").is_file()
    for rel in ("proteins/full_sequences_flat_scores.npy", "proteins/full_sequences_flat_hits.npy",
                "proteins/full_sequences_hnsw_scores.npy", "proteins/full_sequences_hnsw_hits.npy",
                "pfam/flat_scores.npy", "pfam/flat_hits.npy", "pfam/index_scores.npy", "pfam/index_hits.npy",
                "slices/slices_scores.npy", "slices/slices_hits.npy", "slices/full_sequences_scores.npy", "slices/full_sequences_hits.npy"):
        same_npy(rel)
    assert (many / "proteins" / "full_sequences_flat.index").read_bytes() == (one / "proteins" / "full_sequences_flat.index").read_bytes()


def test_module_launch_under_torch_distributed_run(tmp_path):
    """the command INTEGRATION.md gives: ``python -m torch.distributed.run ... -m knn_for_homology_amd.<script>`` on an
    unchanged script -- here two ranks on one GPU (KNN355_REHEARSE_ONE_GPU=1), files compared with a plain run"""
    import subprocess
    rng = np.random.default_rng(82)
    x = rng.standard_normal((2100, 64), dtype=np.float32)
    outs = []
    for name, launcher in (("many", [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                     "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "-m"]),
                           ("one", [sys.executable, "-m"])):
        root = tmp_path / name
        (root / "pfam" / "full_sequences_data").mkdir(parents=True)
        (root / "cath" / "data").mkdir(parents=True)
        np.save(root / "pfam" / "full_sequences_data" / "full_sequences.npy", x)
        np.save(root / "cath" / "data" / "emb.npy", x[:900].astype(np.float16))
        env = dict(os.environ, KNN355_PROJECT_ROOT=str(root), KNN355_REHEARSE_ONE_GPU="1",
                   PYTHONPATH=str(ROOT) + os.pathsep + os.environ.get("PYTHONPATH", ""))
        for var in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(var, None)
        for module, extra in (("knn_for_homology_amd.pfam.proteins_search", ["flat"]), ("knn_for_homology_amd.cath.search", [])):
            r = subprocess.run(launcher + [module] + extra, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
            outs.append((name, module, r.stdout))
    # rank 0 alone printed: the protocol lines appear once
    for name, module, stdout in outs:
        if module.endswith("proteins_search"):
            assert stdout.count("Index creation took") == 1 and stdout.count("Search took") == 1, (name, stdout)
        else:
            assert stdout.count("Searching with Cosine") == 1, (name, stdout)
    for rel in ("pfam/full_sequences_data/full_sequences_flat_scores.npy", "pfam/full_sequences_data/full_sequences_flat_hits.npy",
                "pfam/full_sequences_data/full_sequences_flat.index"):
        assert (tmp_path / "many" / rel).read_bytes() == (tmp_path / "one" / rel).read_bytes(), rel
    for label in ("cosine", "euclidean"):
        a = np.load(tmp_path / "many" / "cath" / "data" / f"hits_{label}.npz")
        b = np.load(tmp_path / "one" / "cath" / "data" / f"hits_{label}.npz")
        assert np.array_equal(a["emb"], b["emb"])
