"""CPU, world_size 2 over gloo: the sharded index's exchange (shard bounds, global ids in
the packed keys, all-gather layout, merge order) with the local scan and the merge served
by the oracle instead of the GPU.  The HIP-backed equivalent is tests/test_sharded_gpu.py."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _f2ord(v):
    u = np.ascontiguousarray(v, np.float32).view(np.uint32).astype(np.uint64)
    neg = (u & 0x80000000) != 0
    return np.where(neg, (~u) & 0xFFFFFFFF, u | 0x80000000)


def _ord2f(o):
    o = o.astype(np.uint64)
    u = np.where((o & 0x80000000) != 0, o & 0x7FFFFFFF, (~o) & 0xFFFFFFFF).astype(np.uint32)
    return u.view(np.float32)


class OracleShardBackend:
    """Test double: same contract as HipShardBackend, arithmetic from the CPU oracle."""

    def __init__(self, d, metric):
        from oracle import knn_oracle as ko
        self.ko, self.metric, self.d = ko, metric, d
        self.rows = np.empty((0, d), np.float32)
        self.index = None
        self.device = torch.device("cpu")

    def reserve(self, n):
        pass

    def add(self, x):
        self.rows = np.concatenate([self.rows, x], 0)

    @property
    def ntotal(self):
        return self.rows.shape[0]

    fail_search_keys = None  # (tests: an exception instance the next local scan raises)

    def search_keys(self, q, k, id_base, out=None):
        if self.fail_search_keys is not None:
            raise self.fail_search_keys
        D, I = self.ko.oracle().flat_search(self.rows, q.numpy(), k, self.metric)
        v = -D if self.metric == 0 else D
        keys = (_f2ord(v + np.float32(0)) << np.uint64(32)) | (I.astype(np.uint64) + np.uint64(id_base))
        keys[I < 0] = np.uint64(0xFFFFFFFFFFFFFFFF)
        keys = torch.from_numpy(keys.view(np.int64))
        if out is not None:
            out.copy_(keys)
            return out
        return keys

    def search(self, q, k):
        D, I = self.ko.oracle().flat_search(self.rows, q.numpy(), k, self.metric)
        return torch.from_numpy(D), torch.from_numpy(I)

    def search_self(self, k, row0, nrows):
        # (the slice is a piece of a batch of ntotal queries: the L2 formula of the whole, as the product's set_batch)
        mode = 0 if self.metric == 0 else (2 if self.ntotal < 20 else 1)
        return self.ko.oracle().flat_search(self.rows, self.rows[row0:row0 + nrows], k, self.metric, l2_mode=mode)

    def normalize_rows(self):
        self.rows = np.ascontiguousarray(self.rows)
        self.ko.oracle().normalize_l2(self.rows)

    def merge(self, gathered, nlists, nq, k):
        g = gathered.numpy().view(np.uint64)                   # [world, nq, k]
        allk = np.sort(np.transpose(g, (1, 0, 2)).reshape(nq, nlists * k), axis=1)[:, :k]
        pad = allk == np.uint64(0xFFFFFFFFFFFFFFFF)
        v = _ord2f(allk >> np.uint64(32))
        D = np.where(pad, -np.finfo(np.float32).max if self.metric == 0 else np.finfo(np.float32).max,
                     -v if self.metric == 0 else v).astype(np.float32)
        I = np.where(pad, -1, (allk & np.uint64(0xFFFFFFFF)).astype(np.int64))
        return torch.from_numpy(D), torch.from_numpy(I)


def _worker(rank, world, port, metric, out_dir):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from knn_for_homology_amd.sharded import ShardedFlatIndex, shard_bounds
    from test_sharded_cpu import OracleShardBackend
    rng = np.random.default_rng(77)
    xb = rng.standard_normal((1003, 64), dtype=np.float32)
    xb[900:910] = xb[10:20]  # duplicates straddling the shard boundary: ties -> lower global id
    xq = np.concatenate([rng.standard_normal((9, 64), dtype=np.float32), xb[10:14]])
    lo, hi = shard_bounds(1003, world, rank)
    idx = ShardedFlatIndex(64, metric, row_offset=lo, backend=OracleShardBackend(64, metric))
    idx.add(xb[lo:hi])
    D, I = idx.search(xq, 25)
    np.savez(Path(out_dir) / f"r{rank}.npz", D=D, I=I)
    dist.barrier()
    dist.destroy_process_group()


def _worker_failing(rank, world, port, out_dir):
    """rank 1's local scan raises: every rank must come out of the search with ShardSearchError naming rank 1 -- nobody
    hangs in the all-gather, nobody returns a result that lacks a shard -- and the NEXT search works again"""
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    from knn_for_homology_amd.sharded import ShardedFlatIndex, QueryShardedFlatIndex, ShardSearchError, shard_bounds
    from test_sharded_cpu import OracleShardBackend
    rng = np.random.default_rng(80)
    xb = rng.standard_normal((400, 32), dtype=np.float32)
    xq = rng.standard_normal((7, 32), dtype=np.float32)
    lo, hi = shard_bounds(400, world, rank)
    idx = ShardedFlatIndex(32, 0, row_offset=lo, backend=OracleShardBackend(32, 0))
    idx.add(xb[lo:hi])
    if rank == 1:
        idx.backend.fail_search_keys = MemoryError("search: out of device memory (test)")
    res = {}
    try:
        idx.search(xq, 5)
        res["raised"] = 0
    except ShardSearchError as e:
        res["raised"] = 1
        res["failed_ranks"] = np.array(e.failed_ranks)
        res["has_local"] = int("out of device memory" in str(e))
    # asynchronous form: nothing raises until the status row is looked at
    pending = idx.submit(torch.from_numpy(xq), 5)
    D, I = pending.result(check=False)
    try:
        pending.check()
        res["raised_async"] = 0
    except ShardSearchError:
        res["raised_async"] = 1
    # search_dev(check=False): the status row stays unread (the host is free to enqueue the next search), check_pending()
    # reads it later -- and only once
    idx.search_dev(torch.from_numpy(xq), 5, check=False)
    try:
        idx.check_pending()
        res["raised_deferred"] = 0
    except ShardSearchError as e:
        res["raised_deferred"] = 1 if e.failed_ranks == [1] else -1
    idx.check_pending()  # (nothing pending any more)
    idx.backend.fail_search_keys = None
    D, I = idx.search(xq, 5)
    res["D"], res["I"] = D, I
    # the query-sharded index with gather: the same agreement in front of its gathers
    qs = QueryShardedFlatIndex(32, 0, backend=OracleShardBackend(32, 0))
    qs.add(xb)
    if rank == 1:
        qs.backend.search = lambda *a, **kw: (_ for _ in ()).throw(RuntimeError("boom"))
    try:
        qs.search(xq, 5)
        res["qs_raised"] = 0
    except ShardSearchError as e:
        res["qs_raised"] = 1
        res["qs_failed"] = np.array(e.failed_ranks)
    # the outcome rides in the id gather (no collective of its own): check=False defers reading it
    qs.search_dev(torch.from_numpy(xq), 5, gather=True, check=False)
    try:
        qs.check_pending()
        res["qs_deferred"] = 0
    except ShardSearchError as e:
        res["qs_deferred"] = 1 if e.failed_ranks == [1] else -1
    np.savez(Path(out_dir) / f"f{rank}.npz", **res)
    dist.barrier()
    dist.destroy_process_group()


def _worker_qs(rank, world, port, metric, out_dir):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from knn_for_homology_amd.sharded import QueryShardedFlatIndex
    from test_sharded_cpu import OracleShardBackend
    rng = np.random.default_rng(78)
    xb = rng.standard_normal((301, 32), dtype=np.float32)
    idx = QueryShardedFlatIndex(32, metric, backend=OracleShardBackend(32, metric))
    idx.add(xb)
    D, I = idx.search(xb, 20)                      # all-vs-all, gathered: 301 queries over 3 ranks (101 + 101 + 99)
    Dl, Il = idx.search(xb, 20, gather=False)      # this rank's slice only
    D1, I1 = idx.search(xb[:2], 5)                 # fewer queries than ranks: the last rank's slice is empty
    np.savez(Path(out_dir) / f"q{rank}.npz", D=D, I=I, Dl=Dl, Il=Il, D1=D1, I1=I1, bounds=np.array(idx.query_bounds(301)))
    dist.barrier()
    dist.destroy_process_group()


def _worker_entry(rank, world, port, out_dir):
    """the all-vs-all entry points under an initialised process group (what ranks.launched_group sees under
    torch.distributed.run): cath.search.search / search_and_save, pfam.search.search_flat, slices_search.main"""
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from knn_for_homology_amd import sharded, ranks, faiss
    from knn_for_homology_amd.cath import search as cath_search
    from knn_for_homology_amd.pfam import search as pfam_search
    from knn_for_homology_amd.pfam.slices import slices_search
    from test_sharded_cpu import OracleShardBackend
    sharded.DEFAULT_BACKEND = OracleShardBackend
    faiss.normalize_L2 = lambda x: __import__("oracle.knn_oracle", fromlist=["oracle"]).oracle().normalize_l2(x)  # (no GPU here)
    assert ranks.launched_group() == (rank, world) and ranks.writer() == (rank == 0)
    out = Path(out_dir)
    x = np.load(out / "cath" / "a.npy").astype(np.float32)
    keep = x.copy()
    res = {}
    for metric in (0, 1):
        hits, scores = cath_search.search(x, hits=7, metric=metric)
        res[f"h{metric}"], res[f"s{metric}"] = hits, scores
    assert np.array_equal(x, keep), "cath.search.search must not touch its input"
    from knn_for_homology_amd.seqvec_search.main import faiss_search
    tr, te = np.load(out / "pfam" / "train.npy"), np.load(out / "pfam" / "test.npy")
    res["fs_ids"], res["fs_scores"], _ = faiss_search(tr, te, 13)   # (normalises both in place, like the reference)
    res["fs_train"], res["fs_test"] = tr, te
    np.savez(out / f"e{rank}.npz", **res)
    cath_search.search_and_save(out / "cath")
    pfam_search.search_flat(out / "pfam", k=9)
    slices_search.main(out / "slices", k=6)
    dist.barrier()
    dist.destroy_process_group()


def test_entry_points_spread_over_three_ranks(tmp_path):
    """ranks.py: under a process group the all-vs-all entry points answer a slice of the queries per rank, every rank
    returns the whole result, rank 0 alone writes the files -- and the files hold what one process writes."""
    from oracle import knn_oracle as ko
    orc = ko.oracle()
    rng = np.random.default_rng(79)
    for sub in ("cath", "pfam", "slices"):
        (tmp_path / sub).mkdir()
    a = rng.standard_normal((203, 24)).astype(np.float16)
    b = rng.standard_normal((61, 40), dtype=np.float32)
    np.save(tmp_path / "cath" / "a.npy", a)
    np.save(tmp_path / "cath" / "b.npy", b)
    train, test = rng.standard_normal((150, 16), dtype=np.float32), rng.standard_normal((37, 16), dtype=np.float32)
    np.save(tmp_path / "pfam" / "train.npy", train)
    np.save(tmp_path / "pfam" / "test.npy", test)
    sl = rng.standard_normal((88, 12), dtype=np.float32)
    np.save(tmp_path / "slices" / "slices.npy", sl)
    np.save(tmp_path / "slices" / "full_sequences.npy", sl[:50])
    mp.spawn(_worker_entry, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)

    def self_search(x, k, metric):
        x = np.ascontiguousarray(x, dtype=np.float32).copy()
        if metric == 0:
            orc.normalize_l2(x)
        return orc.flat_search(x, x, k, metric)

    for metric in (0, 1):
        D, I = self_search(a, 8, metric)
        for r in range(3):
            got = np.load(tmp_path / f"e{r}.npz")
            assert np.array_equal(got[f"h{metric}"], I[:, 1:]) and np.array_equal(got[f"s{metric}"].view(np.uint32), D[:, 1:].view(np.uint32))
    for label, metric in (("cosine", 0), ("euclidean", 1)):
        hits, scores = np.load(tmp_path / "cath" / f"hits_{label}.npz"), np.load(tmp_path / "cath" / f"scores_{label}.npz")
        assert sorted(hits.files) == ["a", "b"]
        for stem, x in (("a", a), ("b", b)):
            D, I = self_search(x, 11, metric)
            assert np.array_equal(hits[stem], I[:, 1:]) and np.array_equal(scores[stem].view(np.uint32), D[:, 1:].view(np.uint32))
            assert float((tmp_path / "cath" / f"{stem}.{label}-search-time.txt").read_text()) >= 0.0
    tr, te = train.copy(), test.copy()
    orc.normalize_l2(tr)
    orc.normalize_l2(te)
    D, I = orc.flat_search(tr, te, 13, 0)
    for r in range(3):
        got = np.load(tmp_path / f"e{r}.npz")
        assert np.array_equal(got["fs_ids"], I) and np.array_equal(got["fs_scores"].view(np.uint32), D.view(np.uint32))
        assert np.array_equal(got["fs_train"], tr) and np.array_equal(got["fs_test"], te), "in-place normalisation of both arrays"
    D, I = orc.flat_search(tr, te, 9, 0)
    assert np.array_equal(np.load(tmp_path / "pfam" / "flat_hits.npy"), I)
    assert np.array_equal(np.load(tmp_path / "pfam" / "flat_scores.npy").view(np.uint32), D.view(np.uint32))
    for name, x in (("slices", sl), ("full_sequences", sl[:50])):
        D, I = self_search(x, 6, 0)
        assert np.array_equal(np.load(tmp_path / "slices" / f"{name}_hits.npy"), I)
        assert np.array_equal(np.load(tmp_path / "slices" / f"{name}_scores.npy").view(np.uint32), D.view(np.uint32))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("metric", [0, 1])
def test_two_rank_gloo_equals_unsharded(tmp_path, metric):
    from oracle import knn_oracle as ko
    mp.spawn(_worker, args=(2, _free_port(), metric, str(tmp_path)), nprocs=2, join=True)
    rng = np.random.default_rng(77)
    xb = rng.standard_normal((1003, 64), dtype=np.float32)
    xb[900:910] = xb[10:20]
    xq = np.concatenate([rng.standard_normal((9, 64), dtype=np.float32), xb[10:14]])
    Do, Io = ko.oracle().flat_search(xb, xq, 25, metric)
    for r in range(2):
        got = np.load(tmp_path / f"r{r}.npz")
        assert np.array_equal(got["I"], Io), f"rank {r}: ids differ from the unsharded search"
        assert np.array_equal(got["D"].view(np.uint32), Do.view(np.uint32))
    assert (Io[9:, 0] == np.arange(10, 14)).all() and (Io[9:, 1] == np.arange(900, 904)).all()


@pytest.mark.parametrize("metric", [0, 1])
def test_three_rank_query_sharded_all_vs_all(tmp_path, metric):
    """SURVEY 8(e) alternative: replicated rows, split queries, no data-path collective; the optional gather
    returns the unsharded result on every rank."""
    from oracle import knn_oracle as ko
    mp.spawn(_worker_qs, args=(3, _free_port(), metric, str(tmp_path)), nprocs=3, join=True)
    xb = np.random.default_rng(78).standard_normal((301, 32), dtype=np.float32)
    Do, Io = ko.oracle().flat_search(xb, xb, 20, metric)
    D1o, I1o = ko.oracle().flat_search(xb, xb[:2], 5, metric)
    seen = []
    for r in range(3):
        got = np.load(tmp_path / f"q{r}.npz")
        assert np.array_equal(got["I"], Io) and np.array_equal(got["D"].view(np.uint32), Do.view(np.uint32))
        lo, hi = got["bounds"]
        seen.append((lo, hi))
        assert np.array_equal(got["Il"], Io[lo:hi]) and np.array_equal(got["Dl"].view(np.uint32), Do[lo:hi].view(np.uint32))
        assert np.array_equal(got["I1"], I1o) and np.array_equal(got["D1"].view(np.uint32), D1o.view(np.uint32))
    assert seen == [(0, 101), (101, 202), (202, 301)]


def test_a_failing_rank_raises_on_every_rank(tmp_path):
    """a local failure on one rank enters the collective with padding + a status row; every rank
    raises, none hangs, the index stays usable."""
    from oracle import knn_oracle as ko
    mp.spawn(_worker_failing, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    rng = np.random.default_rng(80)
    xb = rng.standard_normal((400, 32), dtype=np.float32)
    xq = rng.standard_normal((7, 32), dtype=np.float32)
    Do, Io = ko.oracle().flat_search(xb, xq, 5, 0)
    for r in range(2):
        got = np.load(tmp_path / f"f{r}.npz")
        assert got["raised"] == 1 and got["failed_ranks"].tolist() == [1], f"rank {r}"
        assert got["has_local"] == (1 if r == 1 else 0)
        assert got["raised_async"] == 1
        assert np.array_equal(got["I"], Io) and np.array_equal(got["D"].view(np.uint32), Do.view(np.uint32)), "the search after the failure"
        assert got["qs_raised"] == 1 and got["qs_failed"].tolist() == [1]
        assert got["raised_deferred"] == 1 and got["qs_deferred"] == 1, "check=False defers the status, check_pending() raises it"


def _worker_rank0_only(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from knn_for_homology_amd import ranks
    out = {"ok": ranks.rank0_only(lambda: 41 + 1)}
    try:
        ranks.rank0_only(lambda: (_ for _ in ()).throw(ValueError("bad index")))
        out["raised"] = ""
    except Exception as e:  # noqa: BLE001
        out["raised"] = f"{type(e).__name__}: {e}"
    import json
    (Path(out_dir) / f"z{rank}.json").write_text(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()


def test_rank0_only_reports_rank0s_outcome_everywhere(tmp_path):
    """HNSW / LSH modes run on rank 0 alone -- the waiting ranks must hear about a failure instead of
    sitting in a barrier until the collective timeout."""
    import json
    mp.spawn(_worker_rank0_only, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    for r in range(3):
        got = json.loads((tmp_path / f"z{r}.json").read_text())
        assert got["ok"] == (42 if r == 0 else None)
        assert got["raised"] == ("ValueError: bad index" if r == 0 else "RuntimeError: rank 0 failed: ValueError: bad index")


def test_shard_bounds_cover_everything():
    from knn_for_homology_amd.sharded import shard_bounds
    for n in (0, 1, 7, 8, 9, 1003, 10_000_000):
        for w in (1, 2, 3, 4, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def _run_launch_check(n, extra_env, timeout):
    import subprocess
    import time
    env = dict(os.environ)
    env.update(extra_env)
    t0 = time.time()
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(n), "--launch-check"], env=env, text=True,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    return p, time.time() - t0


def test_bench_bring_up_of_four_ranks():
    """`bench.py --gpus 4 --launch-check`: the parent's preflight / child launch / result collection and the ranks' bounded
    bring-up (init_process_group + the first all-gather of the rank numbers), on gloo."""
    import json
    p, _ = _run_launch_check(4, {}, 170)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["launch_check"] is True and line["ranks_seen"] == 4


def test_bench_with_a_stalled_rank_fails_fast_and_names_it():
    """one rank of four sleeps in front of init_process_group.  The other three give up after
    KNN355_PG_TIMEOUT_S (120 s by default; 8 s here), the launcher exits non-zero well inside the driver's budget -- never a
    bare timeout -- and its report names the stalled rank and shows every rank's last progress marker."""
    p, took = _run_launch_check(4, {"KNN355_PG_TIMEOUT_S": "8", "KNN355_TEST_STALL_RANK": "2", "KNN355_TEST_STALL_S": "120"}, 170)
    assert p.returncode != 0
    assert took < 150, took
    assert "rank(s) 2 stalled before init_process_group" in p.stderr, p.stderr[-3000:]
    assert "rank 2: last progress marker: stalling (test hook)" in p.stderr
    for r in (0, 1, 3):
        assert f"rank {r}: last progress marker: init_process_group(gloo) entered" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")], "no result line from a failed run"


def test_bench_parent_timeout_kills_the_run_it_started():
    """The launcher's own bound (KNN355_BENCH_TIMEOUT_S, 480 s by default): every rank stalls -> the child is killed as a
    process group, exit code 124, the report says so."""
    p, took = _run_launch_check(2, {"KNN355_BENCH_TIMEOUT_S": "12", "KNN355_PG_TIMEOUT_S": "100", "KNN355_TEST_STALL_RANK": "1",
                                   "KNN355_TEST_STALL_S": "100"}, 120)
    assert p.returncode == 124 and took < 60, (p.returncode, took)
    assert "no result within KNN355_BENCH_TIMEOUT_S = 12 s (killed)" in p.stderr
    assert "rank(s) 1 stalled before init_process_group" in p.stderr


def test_bench_one_rank_deadline_keeps_the_headline():
    """A one-rank bench run has no parent to bound it: at KNN355_BENCH_DEADLINE_S the watchdog prints the JSON line with what
    has been measured (the headline dict registered behind the timed steps) and an `incomplete` key naming the stage, exit 0;
    with nothing registered yet it exits 124 and prints no line."""
    import json
    import subprocess
    import textwrap
    body = textwrap.dedent(f'''
        import os, sys, time, importlib.util
        sys.argv = ["bench.py"]
        spec = importlib.util.spec_from_file_location("benchmod", r"{ROOT / 'bench.py'}")
        m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
        real = os.dup(1); os.dup2(2, 1)
        m._PARTIAL["fd"] = real
        m.deadline_watch(1.0)
        if os.environ.get("WITH_HEADLINE") == "1":
            m._PARTIAL["out"] = {{"metric": "x", "value": 1.0, "roofline": {{"frac": 0.5}}}}
        m.progress("sweep")
        time.sleep(30)
    ''')
    p = subprocess.run([sys.executable, "-c", body], capture_output=True, text=True, timeout=60, env={**os.environ, "WITH_HEADLINE": "1"})
    assert p.returncode == 0, p.stderr[-500:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["value"] == 1.0 and line["incomplete"]["stage"] == "sweep"
    assert "[bench" in p.stderr and "sweep" in p.stderr
    p = subprocess.run([sys.executable, "-c", body], capture_output=True, text=True, timeout=60, env={**os.environ, "WITH_HEADLINE": "0"})
    assert p.returncode == 124 and not p.stdout.strip()
