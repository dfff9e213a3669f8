"""GPU: the sharded index over RCCL.  Only one GPU is available to the test box, so the
process group has a single rank, but the full exchange path runs (per-shard keys ->
all_gather_into_tensor on the nccl backend -> merge kernel on a [world, nq, k] buffer), and
a 4-shard exchange is emulated by concatenating per-shard key lists on the device."""
import ctypes
import os
import socket
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_single_rank_nccl_collective_path(gpu_faiss, oracle):
    import torch
    import torch.distributed as dist
    from knn_for_homology_amd.sharded import ShardedFlatIndex
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        os.environ["KNN355_FORCE_COLLECTIVE"] = "1"
        rng = np.random.default_rng(12)
        xb = rng.standard_normal((5000, 1024), dtype=np.float32)
        xq = rng.standard_normal((40, 1024), dtype=np.float32)
        for metric in (0, 1):
            idx = ShardedFlatIndex(1024, metric, row_offset=1000)
            assert idx.force_collective
            idx.add(xb)
            D, I = idx.search(xq, 100)
            Do, Io = oracle.flat_search(xb, xq, 100, metric)
            assert np.array_equal(I, Io + 1000) and np.array_equal(D.view(np.uint32), Do.view(np.uint32))
    finally:
        os.environ.pop("KNN355_FORCE_COLLECTIVE", None)
        dist.destroy_process_group()


def test_emulated_four_shard_merge(gpu_faiss, oracle):
    """Four shard indexes on one GPU; their key lists are stacked [4, nq, k] exactly as an
    all-gather would deliver them and merged by knn_merge_keys_dev."""
    import torch
    from knn_for_homology_amd import _lib
    from knn_for_homology_amd.sharded import HipShardBackend, shard_bounds
    rng = np.random.default_rng(13)
    nb, d, nq, k = 9001, 256, 70, 64
    xb = rng.standard_normal((nb, d), dtype=np.float32)
    xb[8000:8010] = xb[5:15]  # ties across shards -> lower global id
    xq = np.concatenate([rng.standard_normal((nq - 5, d), dtype=np.float32), xb[5:10]])
    dev = torch.device("cuda", 0)
    q = torch.from_numpy(xq).to(dev)
    for metric in (0, 1):
        parts = []
        backs = []
        for r in range(4):
            lo, hi = shard_bounds(nb, 4, r)
            b = HipShardBackend(d, metric)
            b.add(xb[lo:hi])
            backs.append(b)
            parts.append(b.search_keys(q, k, lo))
        gathered = torch.stack(parts).contiguous()
        D, I = backs[0].merge(gathered, 4, nq, k)
        torch.cuda.synchronize()
        Do, Io = oracle.flat_search(xb, xq, k, metric)
        assert np.array_equal(I.cpu().numpy(), Io)
        assert np.array_equal(D.cpu().numpy().view(np.uint32), Do.view(np.uint32))


def test_submit_overlaps_two_lanes_and_matches_sync(gpu_faiss, oracle, monkeypatch):
    """Searches submitted back to back alternate between the index and its read-only view (own
    streams and scratch memory) and must return what a blocking search returns, in any
    interleaving, with and without the collective path; a view refuses writes."""
    import torch
    from knn_for_homology_amd.sharded import ShardedFlatIndex
    from knn_for_homology_amd._lib import Knn355Error
    rng = np.random.default_rng(77)
    xb = rng.standard_normal((70000, 256), dtype=np.float32)
    qs = [rng.standard_normal((nq, 256), dtype=np.float32) for nq in (32, 7, 32, 1, 19, 32)]
    for force in ("0", "1"):
        monkeypatch.setenv("KNN355_FORCE_COLLECTIVE", force)
        if force == "1" and not torch.distributed.is_initialized():
            monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
            monkeypatch.setenv("MASTER_PORT", str(_free_port()))
            torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        idx = ShardedFlatIndex(256, gpu_faiss.METRIC_INNER_PRODUCT)
        idx.add(xb)
        dev = idx.backend.device
        pend = [idx.submit(torch.from_numpy(q).to(dev), 50) for q in qs]
        torch.cuda.synchronize()
        for q, p in zip(qs, pend):
            D, I = p.result()
            Do, Io = oracle.flat_search(xb, q, 50, 0)
            assert np.array_equal(I.cpu().numpy(), Io) and np.array_equal(D.cpu().numpy().view(np.uint32), Do.view(np.uint32))
        # rows added later are seen by both lanes (the view is rebuilt)
        idx.add(xb[:1000] * 2.0)
        D, I = idx.search(qs[0], 50)
        x2 = np.concatenate([xb, xb[:1000] * 2.0])
        Do, Io = oracle.flat_search(x2, qs[0], 50, 0)
        assert np.array_equal(I, Io)
        D2, I2 = idx.search(qs[0], 50)  # the other lane
        assert np.array_equal(I2, Io)
        v = idx.local.view()
        with pytest.raises(Knn355Error):
            v.add(xb[:8])
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def test_in_library_rccl_single_rank(gpu_faiss, oracle):
    """The torch-free multi-GPU path of the C ABI (knn_comm_* + knn_sharded_search_dev: local scan -> ncclAllGather of
    the packed keys -> selection, RCCL opened by the library itself).  One GPU here, so one rank: the collective is real
    (a 1-rank communicator), the result must be the flat search's with the shard's id offset."""
    import ctypes
    import torch
    from knn_for_homology_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(99)
    nb, d, nq, k = 30000, 128, 40, 100
    xb = rng.standard_normal((nb, d), dtype=np.float32)
    xq = rng.standard_normal((nq, d), dtype=np.float32)
    ident = (ctypes.c_uint8 * 128)()
    _lib.check(L.knn_comm_unique_id(ident))
    comm = ctypes.c_void_p()
    _lib.check(L.knn_comm_create(ident, 1, 0, 0, ctypes.byref(comm)))
    try:
        dev = torch.device("cuda:0")
        q = torch.from_numpy(xq).to(dev)
        D = torch.empty((nq, k), device=dev, dtype=torch.float32)
        I = torch.empty((nq, k), device=dev, dtype=torch.int64)
        for metric in (0, 1):
            idx = gpu_faiss.IndexFlat(d, metric)
            idx.add(xb)
            _lib.check(L.knn_sharded_search_dev(idx._h, comm, q.data_ptr(), nq, k, 5000, D.data_ptr(), I.data_ptr(), None))
            Do, Io = oracle.flat_search(xb, xq, k, metric)
            assert np.array_equal(I.cpu().numpy(), Io + 5000)
            assert np.array_equal(D.cpu().numpy().view(np.uint32), Do.view(np.uint32))
        assert L.knn_comm_create(ident, 2, 2, 0, ctypes.byref(ctypes.c_void_p())) != 0  # rank out of range
    finally:
        L.knn_comm_free(comm)


_ABSENT_PEER = r"""
import ctypes, os, sys, time
sys.path.insert(0, sys.argv[1])
import torch
from knn_for_homology_amd import _lib
L = _lib.lib()
_lib.check(L.knn_init(0))
ident = (ctypes.c_uint8 * 128)()
_lib.check(L.knn_comm_unique_id(ident))
comm = ctypes.c_void_p()
t0 = time.time()
rc = L.knn_comm_create(ident, 2, 0, 0, ctypes.byref(comm))   # rank 0 of 2; rank 1 never calls
took = time.time() - t0
msg = L.knn_last_error().decode()
print(f"rc={rc} took={took:.1f} msg={msg}", flush=True)
os._exit(0 if (rc == -6 and took < 30 and "rank 0 of 2" in msg and "did not return within" in msg) else 1)
"""


def test_in_library_comm_create_gives_up_on_an_absent_peer(tmp_path):
    """VERDICT r4 item 3 (the in-library path): knn_comm_create(world = 2) with a peer that never calls returns
    KNN_ERR_TIMEOUT after KNN355_COMM_TIMEOUT_S seconds, naming the rank -- it does not hang.  In a process of its own
    (left with os._exit: the abandoned helper thread is still inside ncclCommInitRank)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, KNN355_COMM_TIMEOUT_S="6")
    p = subprocess.run([sys.executable, "-c", _ABSENT_PEER, str(ROOT)], env=env, text=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=180)
    assert p.returncode == 0, p.stdout[-2000:]


def test_query_sharded_slices_equal_the_unsharded_search(gpu_faiss, oracle):
    """QueryShardedFlatIndex (rows replicated, queries split): rank r of 3 returns rows [lo_r, hi_r) of the
    single-GPU result, bit for bit -- constructed with explicit rank/world, no process group needed without gather."""
    import torch
    from knn_for_homology_amd.sharded import QueryShardedFlatIndex, shard_bounds
    rng = np.random.default_rng(31)
    n, d, k = 4001, 256, 33
    xb = rng.standard_normal((n, d), dtype=np.float32)
    xb[3000:3010] = xb[:10]
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(xb).to(dev)
    for metric in (0, 1):
        Do, Io = oracle.flat_search(xb, xb, k, metric)
        covered = 0
        for r in range(3):
            idx = QueryShardedFlatIndex(d, metric, rank=r, world=3)
            idx.add_dev(x)
            lo, hi = idx.query_bounds(n)
            assert (lo, hi) == shard_bounds(n, 3, r)
            D, I = idx.search_dev(x, k)
            torch.cuda.synchronize()
            assert np.array_equal(I.cpu().numpy(), Io[lo:hi])
            assert np.array_equal(D.cpu().numpy().view(np.uint32), Do[lo:hi].view(np.uint32))
            covered += hi - lo
        assert covered == n


def test_query_sharded_search_waits_for_the_kernels_that_produce_q(gpu_faiss, oracle):
    """on torch's default stream the C ABI gets a NULL stream (= the library's own non-blocking stream),
    which nothing orders behind torch kernels still producing q.  The queries here are the output of a long queue of
    torch work (the buffer holds zeros until the very last kernel): the search must see the finished queries."""
    import torch
    from knn_for_homology_amd.sharded import QueryShardedFlatIndex
    rng = np.random.default_rng(77)
    n, d, k, nq = 20000, 256, 10, 512
    xb = rng.standard_normal((n, d), dtype=np.float32)
    qh = rng.standard_normal((nq, d), dtype=np.float32)
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(xb).to(dev)
    idx = QueryShardedFlatIndex(d, 1, rank=0, world=1)
    idx.add_dev(x)
    Do, Io = oracle.flat_search(xb, qh, k, 1)
    src = torch.from_numpy(qh).to(dev)
    big = torch.randn((6144, 6144), device=dev)
    torch.cuda.synchronize()
    for _ in range(3):
        q = torch.zeros((nq, d), device=dev)
        junk = big
        for _ in range(12):               # ~50 ms of queued matmuls in front of the copy that fills q
            junk = (junk @ big) * 1e-4
        q += src + 0.0 * junk[:nq, :d].nan_to_num(0.0, 0.0, 0.0)
        D, I = idx.search_dev(q, k)       # no synchronize in between
        torch.cuda.synchronize()
        assert np.array_equal(I.cpu().numpy(), Io)
        assert np.array_equal(D.cpu().numpy().view(np.uint32), Do.view(np.uint32))


def test_one_rank_large_batch_takes_the_synchronous_entry(gpu_faiss, oracle):
    """search_dev on ONE rank with more than 128 queries and a caller that waits (check=True): the synchronous entry -- it may
    take the statistical seed (the lanes never do) -- and the bits of the lanes' result and of the oracle."""
    import torch
    from knn_for_homology_amd.sharded import ShardedFlatIndex
    rng = np.random.default_rng(78)
    xb = rng.standard_normal((20000, 64), dtype=np.float32)
    xq = rng.standard_normal((300, 64), dtype=np.float32)
    idx = ShardedFlatIndex(64, gpu_faiss.METRIC_INNER_PRODUCT)
    idx.add(xb)
    q = torch.from_numpy(xq).to(idx.backend.device)
    D, I = idx.search_dev(q, 100)
    torch.cuda.synchronize()
    assert idx.local.last_seed()["stat_rank"] > 0, idx.local.last_seed()          # the statistical seed: the synchronous entry ran
    Dl, Il = idx.submit(q, 100).result()                                           # the lanes
    torch.cuda.synchronize()
    Do, Io = oracle.flat_search(xb, xq, 100, 0)
    for d_, i_ in ((D, I), (Dl, Il)):
        assert np.array_equal(i_.cpu().numpy(), Io) and np.array_equal(d_.cpu().numpy().view(np.uint32), Do.view(np.uint32))
