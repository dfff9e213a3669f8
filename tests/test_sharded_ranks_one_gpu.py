"""GPU: two and four ranks on ONE GPU over the gloo backend (NCCL refuses two ranks per device): the real
multi-rank flow of ShardedFlatIndex -- per-rank HIP shard scan, all-gather of the keys, merge on
every rank, two searches in flight on the two lanes -- must reproduce the unsharded result."""
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
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from knn_for_homology_amd import faiss
    from knn_for_homology_amd.sharded import ShardedFlatIndex, shard_bounds
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)
        nb, d, k = 60001, 128, 40
        xb = rng.standard_normal((nb, d), dtype=np.float32)
        xb[50000:50010] = xb[3:13]  # ties across the shard boundary -> lower global id
        qs = [rng.standard_normal((nq, d), dtype=np.float32) for nq in (32, 5, 32, 17, 32, 1)]
        qs[1][:3] = xb[3:6]
        lo, hi = shard_bounds(nb, world, rank)
        dev = torch.device("cuda", 0)
        for metric in (faiss.METRIC_INNER_PRODUCT, faiss.METRIC_L2):
            idx = ShardedFlatIndex(d, metric, row_offset=lo)
            idx.add(xb[lo:hi])
            pend = [idx.submit(torch.from_numpy(q).to(dev), k) for q in qs]  # two in flight, alternating lanes
            res = [p.result() for p in pend]
            torch.cuda.synchronize()
            np.savez(os.path.join(out_dir, f"r{rank}_m{metric}.npz"),
                     **{f"D{i}": D.cpu().numpy() for i, (D, I) in enumerate(res)},
                     **{f"I{i}": I.cpu().numpy() for i, (D, I) in enumerate(res)})
            dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_share_one_gpu(tmp_path, oracle, world):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(5)
    nb, d, k = 60001, 128, 40
    xb = rng.standard_normal((nb, d), dtype=np.float32)
    xb[50000:50010] = xb[3:13]
    qs = [rng.standard_normal((nq, d), dtype=np.float32) for nq in (32, 5, 32, 17, 32, 1)]
    qs[1][:3] = xb[3:6]
    for metric in (0, 1):
        ranks = [np.load(tmp_path / f"r{r}_m{metric}.npz") for r in range(world)]
        for i, q in enumerate(qs):
            Do, Io = oracle.flat_search(xb, q, k, metric)
            for r in ranks:
                assert np.array_equal(r[f"I{i}"], Io), (metric, i)
                assert np.array_equal(r[f"D{i}"].view(np.uint32), Do.view(np.uint32)), (metric, i)
