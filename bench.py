#!/usr/bin/env python3
"""bench.py -- queries/sec of the flat kNN hot path on MI355X (one process per GPU).

Workload (BASELINE.json configs[3], the configuration the metric "queries/sec ... at
1/2/4/8 MI355X" is quoted on; it fits one GPU): synthetic L2-normalised fp32[10M, 1024]
database, inner product, k = 100, a batch of 32 queries per step.  The database is
row-sharded over the N ranks (10M/N rows each, "scaling": "strong"); every rank scans
its shard for all queries, the per-shard top-k keys are exchanged with ONE RCCL
all-gather and merged on every rank.  A "step" is one such search of the query batch
over the whole database.  Inputs are resident in HBM before the timed region.  The K
timed steps are submitted back to back (ShardedFlatIndex.submit) and all K have completed
when the clock stops.  On several GPUs two searches are in flight, on the index's two lanes
(--in-flight): the all-gather and the small launches at the ends of one search hide behind
the other's scan; on one GPU the steps run one after the other.

Launching: `python bench.py --gpus N` with no WORLD_SIZE in the environment starts the N
ranks itself (a child `python -m torch.distributed.run`, before this process has touched
the GPU) and prints rank 0's line; under torch.distributed.run it is a rank.
KNN355_REHEARSE_ONE_GPU=1 puts every rank on GPU 0 with gloo as the collective backend:
a functional rehearsal of the N-rank path on a one-GPU box, not a measurement.

Also on the same JSON line (N=1 only, outside the timed region):
  roofline      the scan kernel against the HBM roofline (each step reads the shard once); box_read_rate: a plain read
                kernel over the same rows on this box, for scale
  cpu_baseline  the reference's CPU path (faiss IndexFlat.search) on this box's host cores, on a
                2 M-row sample: `value` = the fastest faithful variant -- oracle/cpu_scan.c
                (OpenMP + AVX-512, NUMA first-touch), FAISS's blocked-sgemm algorithm on numpy's
                BLAS, the real faiss when importable -- next to the host's own DRAM read rate
  shard_unit    a 1.25 M-row shard (the per-GPU unit of the 8-GPU run, strong or weak) on this GPU: step time, one and two
                searches in flight
  sweep         the same index at nq = 1, 8, 32, 1024, 10 000 queries per search
  host_buffers  one step through the host-pointer entry (H2D of the queries + D2H of D/I)
  batch         BASELINE configs[1]: CATH20-sized all-vs-all (14433 x 1024, L2, k = 300 + self)
                device-resident and end to end through cath.search.search (incl. PCIe)
  hnsw          BASELINE configs[4]: 200 k x 1024 clustered rows, M = 32, efSearch = 256, k = 100
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import torch  # before libknn355: one shared HIP runtime (see knn_for_homology_amd/_lib.py)
import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured float4 copy)
FP32_MFMA_PEAK_TF = 157.3  # MI355X_MICROARCH.md: dense fp32 MFMA peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--nb-total", type=int, default=10_000_000)
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="strong (default, BASELINE configs[3]): --nb-total rows split over the N GPUs; weak (SURVEY 8(d)): "
                         "--nb-per-gpu rows on EVERY GPU, the database grows with N")
    ap.add_argument("--nb-per-gpu", type=int, default=1_250_000, help="rows per GPU under --scaling weak")
    ap.add_argument("--nq", type=int, default=32)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--d", type=int, default=1024)
    ap.add_argument("--in-flight", type=int, default=0, choices=(0, 1, 2),
                    help="searches in flight: 2 = steps alternate between the index's two lanes (the all-gather and the small "
                         "launches of one search hide behind the other's scan), 1 = one after the other, 0 = 2 on several GPUs, 1 on one")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-batch", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the nq sweep, the host-buffer step and the HNSW run")
    ap.add_argument("--cpu-sample-rows", type=int, default=2_000_000)
    ap.add_argument("--launch-check", action="store_true",
                    help="only bring the N ranks up (gloo, CPU) and tear them down: exercises the bounded waits and the failure report")
    return ap.parse_args()


# ---- bring-up of the N-rank run: bounded, and loud when it fails -------------------------------------------------
# The first contact with an 8-GPU node must not end as a bare driver timeout.  Every wait of the bring-up has a bound:
#   parent  preflight (enough GPUs?), the child torch.distributed.run under KNN355_BENCH_TIMEOUT_S (480 s), killed as the
#           process group this process started; on a timeout or a non-zero exit the child's stderr tail and every rank's last
#           progress marker go to stderr, the ranks that never brought their process group up are named, exit code != 0
#   rank    init_process_group and every collective under KNN355_PG_TIMEOUT_S (120 s); a first tiny all-gather of the rank
#           numbers proves the group before anything is timed (`ranks_seen` is what THAT gather returned)
# Progress markers: one line per stage and rank in $KNN355_PROGRESS_DIR/rank<r>.log (the parent makes the directory).
PG_TIMEOUT_S = float(os.environ.get("KNN355_PG_TIMEOUT_S", "120"))
BENCH_TIMEOUT_S = float(os.environ.get("KNN355_BENCH_TIMEOUT_S", "480"))
_T0 = time.time()


_PARTIAL = {"out": None, "stage": "start", "fd": None}


def deadline_watch(deadline_s):
    """One-process runs (N = 1): nothing bounds an in-process stall -- a sick box, a host leg that never returns.  After
    deadline_s/2 the Python stacks go to stderr once (which leg, which library call); at deadline_s the JSON line goes out
    with what has been measured so far and an "incomplete" key naming the stage, provided the headline is in it (exit 0);
    otherwise exit 124.  The N-rank launch has its own bounds (self_launch)."""
    import faulthandler
    import threading
    faulthandler.dump_traceback_later(max(60.0, deadline_s / 2), repeat=False, file=sys.stderr)

    def fire():
        time.sleep(deadline_s)
        print(f"bench.py: not finished after {deadline_s:.0f} s (KNN355_BENCH_DEADLINE_S), stage: {_PARTIAL['stage']}", file=sys.stderr, flush=True)
        try:
            faulthandler.dump_traceback(file=sys.stderr)
        except Exception:
            pass
        out, fd = _PARTIAL["out"], _PARTIAL["fd"]
        if out is not None and fd is not None:
            for _ in range(5):
                try:
                    rec = dict(out)
                    rec["incomplete"] = {"stage": _PARTIAL["stage"], "deadline_s": deadline_s,
                                         "note": "the legs behind this stage are missing: the run was cut at its deadline"}
                    os.write(fd, (json.dumps(rec) + "\n").encode())
                    os._exit(0)
                except RuntimeError:  # (the main thread added a key meanwhile)
                    time.sleep(0.05)
        os._exit(124)

    threading.Thread(target=fire, daemon=True).start()


def progress(stage):
    """Rank side: '<seconds since start> <stage>' on rank 0's stderr and, in a self-launched run, in this rank's marker file."""
    _PARTIAL["stage"] = stage
    if os.environ.get("RANK", "0") == "0":  # (one line per stage on rank 0's stderr: a run that stalls says where)
        print(f"[bench {time.time() - _T0:7.1f}s] {stage}", file=sys.stderr, flush=True)
    pdir = os.environ.get("KNN355_PROGRESS_DIR")
    if not pdir:
        return
    try:
        with open(os.path.join(pdir, f"rank{os.environ.get('RANK', '0')}.log"), "a") as f:
            f.write(f"{time.time() - _T0:8.2f} {stage}\n")
    except OSError:
        pass


def read_progress(pdir, world):
    """Parent side: {rank: [stages]} from the marker files (a rank that never started has none)."""
    out = {}
    for r in range(world):
        try:
            with open(os.path.join(pdir, f"rank{r}.log")) as f:
                out[r] = [ln.split(None, 1)[1].strip() for ln in f if len(ln.split(None, 1)) == 2]
        except OSError:
            out[r] = []
    return out


def report_failed_launch(why, world, pdir, stderr_path):
    """One diagnosis on stderr: why, each rank's last marker, the stalled ranks by name, the child's stderr tail."""
    marks = read_progress(pdir, world)
    print(f"bench.py: the {world}-rank run failed: {why}", file=sys.stderr)
    for r in range(world):
        print(f"bench.py:   rank {r}: last progress marker: {marks[r][-1] if marks[r] else '(none: the rank never started)'}", file=sys.stderr)
    absent = [r for r in range(world) if not any(m.startswith("init_process_group") for m in marks[r])]  # never reached the rendezvous
    down = [r for r in range(world) if "process_group_up" not in marks[r]]
    if absent and len(absent) < world:
        print(f"bench.py: rank(s) {', '.join(map(str, absent))} stalled before init_process_group; the other ranks waited "
              f"KNN355_PG_TIMEOUT_S = {PG_TIMEOUT_S:.0f} s for them at the rendezvous and gave up", file=sys.stderr)
    elif down:
        print(f"bench.py: rank(s) {', '.join(map(str, down))} never got a process group (rendezvous at MASTER_ADDR:MASTER_PORT "
              f"incomplete after {PG_TIMEOUT_S:.0f} s)", file=sys.stderr)
    else:
        late = [r for r in range(world) if "timed_steps_done" not in marks[r]]
        if late:
            print(f"bench.py: every rank joined; rank(s) {', '.join(map(str, late))} did not finish the timed steps", file=sys.stderr)
    try:
        with open(stderr_path, errors="replace") as f:
            tail = f.read().splitlines()[-40:]
        print("bench.py: ---- tail of the ranks' stderr ----", file=sys.stderr)
        for ln in tail:
            print("  " + ln, file=sys.stderr)
    except OSError:
        pass


def self_launch(args):
    """`bench.py --gpus N` from a plain process: start the N ranks as a child torch.distributed.run.  This
    process has not initialised the GPU (importing torch and counting devices do not) and never will."""
    import signal
    import tempfile
    rehearse = os.environ.get("KNN355_REHEARSE_ONE_GPU", "0") == "1"
    if not (rehearse or args.launch_check):
        have = torch.cuda.device_count()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but this node shows {have} GPU(s) (torch.cuda.device_count())", file=sys.stderr)
            sys.exit(2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    pdir = tempfile.mkdtemp(prefix="knn355_bench_")
    env["KNN355_PROGRESS_DIR"] = pdir
    stderr_path = os.path.join(pdir, "stderr.log")
    t0 = time.time()
    with open(stderr_path, "w") as errf:
        # (a fresh child in its own session: on a timeout exactly the process group started HERE is killed)
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=errf, env=env, text=True, start_new_session=True)
        try:
            stdout, _ = proc.communicate(timeout=BENCH_TIMEOUT_S)
            timed_out = False
        except subprocess.TimeoutExpired:
            timed_out = True
            try:
                os.killpg(proc.pid, signal.SIGTERM)
                time.sleep(3)
                os.killpg(proc.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            stdout, _ = proc.communicate()
    lines = [ln for ln in (stdout or "").splitlines() if ln.startswith('{"metric"') or ln.startswith('{"launch_check"')]
    for ln in (stdout or "").splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)
    if timed_out or proc.returncode != 0 or len(lines) != 1:
        why = (f"no result within KNN355_BENCH_TIMEOUT_S = {BENCH_TIMEOUT_S:.0f} s (killed)" if timed_out else
               f"exit code {proc.returncode}, {len(lines)} result lines, after {time.time() - t0:.0f} s")
        report_failed_launch(why, args.gpus, pdir, stderr_path)
        sys.exit(124 if timed_out else (proc.returncode or 1))
    try:  # (the ranks' stderr of a good run still goes where stderr goes)
        with open(stderr_path, errors="replace") as f:
            sys.stderr.write(f.read())
    except OSError:
        pass
    import shutil
    shutil.rmtree(pdir, ignore_errors=True)
    print(lines[0], flush=True)
    sys.exit(0)


def bring_up(rank, world, backend, device=None):
    """Rank side: the process group under PG_TIMEOUT_S, then ONE tiny all-gather of the rank numbers (bounded by the same
    timeout) that proves the group.  Returns the ranks that gather saw.  KNN355_TEST_STALL_RANK / _S: a rank that sleeps in
    front of init_process_group (the fail-fast test of tests/test_sharded_cpu.py)."""
    import datetime
    import torch.distributed as dist
    progress("started")
    if os.environ.get("KNN355_TEST_STALL_RANK", "") == str(rank):
        progress("stalling (test hook)")
        time.sleep(float(os.environ.get("KNN355_TEST_STALL_S", "60")))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    timeout = datetime.timedelta(seconds=PG_TIMEOUT_S)
    progress(f"init_process_group({backend}) entered")
    if backend == "gloo":
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timeout)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=timeout)
    progress("process_group_up")
    mine = torch.tensor([rank], dtype=torch.int64, device=device if backend != "gloo" else "cpu")
    seen = torch.empty((world,), dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(seen, mine)
    seen = sorted(set(int(v) for v in seen.cpu().tolist()))
    if seen != list(range(world)):
        raise RuntimeError(f"bench.py: the first all-gather returned ranks {seen}, expected 0..{world - 1}")
    progress("first_collective_ok")
    return seen


def rccl_version():
    """RCCL's version as torch reports it (backend "nccl" IS RCCL on ROCm), e.g. "2.22.3"; None off a GPU build."""
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception:  # noqa: BLE001 -- a label, never a reason to fail the run
        return None


def launch_check():
    """`bench.py --gpus N --launch-check`: the bring-up alone (gloo on the CPU, no GPU work): what the fail-fast test runs,
    and a ten-second preflight of a node's rendezvous."""
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    seen = bring_up(rank, world, "gloo")
    dist.barrier()
    progress("timed_steps_done")
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"launch_check": True, "ranks_seen": len(seen), "backend": "gloo", "pg_timeout_s": PG_TIMEOUT_S}), flush=True)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)
    if args.launch_check:
        launch_check()
        return
    # stdout carries exactly ONE line (the JSON): everything libraries print while we run (RCCL's
    # version banner, for one) is sent to stderr instead
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if "WORLD_SIZE" not in os.environ:
        _PARTIAL["fd"] = real_stdout
        deadline_watch(float(os.environ.get("KNN355_BENCH_DEADLINE_S", "600")))
    try:
        line = run(args)
    finally:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        os.close(real_stdout)
    if line is not None:
        print(line, flush=True)


def run(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    rehearse = os.environ.get("KNN355_REHEARSE_ONE_GPU", "0") == "1"
    if rehearse:
        local_rank = 0
    progress("run: selecting the device")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    use_pg = world > 1 or os.environ.get("KNN355_FORCE_COLLECTIVE", "0") == "1"
    ranks_first_gather = [0]
    if use_pg:
        if not rehearse and torch.cuda.device_count() <= local_rank:
            raise RuntimeError(f"bench.py: rank {rank} wants GPU {local_rank}, this node shows {torch.cuda.device_count()}")
        ranks_first_gather = bring_up(rank, world, "gloo" if rehearse else "nccl", dev)

    from knn_for_homology_amd import _lib, faiss
    from knn_for_homology_amd.sharded import ShardedFlatIndex, shard_bounds
    L = _lib.lib()
    _lib.check(L.knn_init(local_rank))
    progress("library loaded, device initialised")

    d, k, nq = args.d, args.k, args.nq
    if args.scaling == "weak":
        args.nb_total = args.nb_per_gpu * world
    lo, hi = shard_bounds(args.nb_total, world, rank)
    nb_local = hi - lo
    # the queries first: one known row is PLANTED in every shard (rank r's local row nb_local // 2 + r is a copy of query
    # r mod nq), so the result of the timed steps can be checked on every rank whatever N is: query j must come back
    # with exactly the planted rows of the ranks r = j (mod nq) in front, ascending ids, score = <q, q> ~ 1
    q_host = np.random.default_rng(24).standard_normal((nq, d), dtype=np.float32)
    q = torch.from_numpy(q_host).to(dev)
    _lib.check(L.knn_normalize_l2_dev(q.data_ptr(), nq, d, None))
    plant_local = min(nb_local - 1, nb_local // 2 + rank)
    planted = {}  # query -> global ids of the rows planted for it, ascending
    for r in range(world):
        rlo, rhi = shard_bounds(args.nb_total, world, r)
        if rhi > rlo:
            planted.setdefault(r % nq, []).append(rlo + min(rhi - rlo - 1, (rhi - rlo) // 2 + r))

    # ---- resident inputs ---------------------------------------------------------
    index = ShardedFlatIndex(d, faiss.METRIC_INNER_PRODUCT, rank=rank, world=world, row_offset=lo)
    index.reserve(nb_local)
    gen = torch.Generator(device=dev)
    gen.manual_seed(23 + rank)
    chunk = 500_000
    cpu_rows = None
    if rank == 0 and world == 1 and not args.no_cpu:
        # the CPU leg's sample goes straight into memory first-touched by the threads that will scan it
        from oracle import cpu_scan as cs
        S = min(nb_local, cpu_sample_rows(args.cpu_sample_rows))
        cpu_rows = cs.Rows(S, d, host_cpu_share()[2])
    for i0 in range(0, nb_local, chunk):
        m = min(chunk, nb_local - i0)
        x = torch.randn((m, d), generator=gen, device=dev, dtype=torch.float32)
        _lib.check(L.knn_normalize_l2_dev(x.data_ptr(), m, d, None))
        if i0 <= plant_local < i0 + m:
            x[plant_local - i0] = q[rank % nq]
        index.add_dev(x)
        if cpu_rows is not None and i0 < cpu_rows.n:
            take = min(m, cpu_rows.n - i0)
            torch.from_numpy(cpu_rows.array[i0:i0 + take]).copy_(x[:take])
        del x
    torch.cuda.synchronize()
    progress("inputs_resident")

    in_flight = args.in_flight or (2 if world > 1 else 1)

    def step():
        # one search of the query batch, enqueued: consecutive steps alternate between the index's
        # two lanes (own stream and scratch memory each), so up to two searches are in flight and
        # the small launches at the ends of one hide behind the scan of the other
        if in_flight == 1:
            index.backend._turn = 0
        return index.submit(q, k)

    def fence():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    warm = [step() for _ in range(args.warmup)]
    fence()
    for pnd in warm:
        pnd.check()  # (a rank whose local scan failed raises here on every rank: sharded.ShardSearchError)
    del warm
    progress("warmup_done")
    index.collective_events = [] if use_pg else None  # HIP event pairs around every all-gather of the timed steps
    index.step_events = []  # a timing event behind every timed search: per-step durations (median beside the mean)
    pend = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pend.append(step())
    fence()
    elapsed = time.perf_counter() - t0
    progress("timed_steps_done")
    step_ms = None
    if index.step_events and len(index.step_events) >= 3:
        evs = index.step_events
        step_ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(len(evs) - 1))  # (completion to completion, K - 1 values)
    index.step_events = None
    for pnd in pend:  # (outside the timed region: the status rows of the K searches)
        pnd.check()
    D, I = pend[-1].result()
    del pend
    if use_pg:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    gather_ms = None
    if index.collective_events:
        gather_ms = float(np.mean([a.elapsed_time(b) for a, b in index.collective_events]))
    index.collective_events = None
    ranks_seen = len(ranks_first_gather) if use_pg else 1  # (what the first all-gather of the run returned, not the configured size)

    # ---- is the result right, and the same everywhere? (outside the timed region) --------------------------------
    # every planted row first for its query, with the score of a vector against itself; sorted best first; ids in range
    Ih, Dh = I.cpu().numpy(), D.cpu().numpy()
    planted_ok = True
    for j, ids in planted.items():
        m = len(ids)
        planted_ok &= Ih[j, :m].tolist() == ids and bool(np.all(np.abs(Dh[j, :m] - 1.0) < 1e-5))
    sorted_ok = bool((np.diff(Dh, axis=1) <= 0).all()) and bool(((Ih >= 0) & (Ih < args.nb_total)).all())
    # a 64-bit digest of (D bits, I) per rank, exchanged: the merge runs on every rank and must give the same bits
    digest = int(((I * 1000003 + D.view(torch.int32).to(torch.int64)) * torch.arange(1, I.numel() + 1, device=dev).view_as(I)).sum().item())
    digests, scan_ms_ranks = [digest], None
    if use_pg:
        buf_ms = (ctypes.c_float * 64)()
        n_ms = L.knn_scan_times(index.local._h, buf_ms, min(64, args.steps))
        mine = [buf_ms[i] for i in range(n_ms) if buf_ms[i] > 0]
        box = torch.tensor([float(digest & 0xFFFFFFFF), float((digest >> 32) & 0xFFFFFFFF), float(np.mean(mine)) if mine else -1.0,
                            1.0 if (planted_ok and sorted_ok) else 0.0], dtype=torch.float64, device=dev)
        allbox = torch.empty((world * 4,), dtype=torch.float64, device=dev)  # (flat: gloo wants the concatenation's own shape)
        dist.all_gather_into_tensor(allbox, box)
        ab = allbox.view(world, 4).cpu().numpy()
        digests = [int(r[0]) | (int(r[1]) << 32) for r in ab]
        scan_ms_ranks = [float(r[2]) for r in ab]
        planted_ok = bool(planted_ok and (ab[:, 3] == 1.0).all())
    verification = {"planted_rows_first_on_every_rank": bool(planted_ok), "sorted_and_in_range": sorted_ok,
                    "identical_on_all_ranks": len(set(digests)) == 1, "planted": {str(j): ids for j, ids in sorted(planted.items())},
                    "what": "one row per shard is a copy of a query: it must come back first with score 1 on every rank; the (D, I) "
                            "digests of all ranks must agree (checked on the last timed step, outside the timed region)"}
    if not (planted_ok and sorted_ok and verification["identical_on_all_ranks"]):
        raise RuntimeError(f"bench.py: the timed search returned a wrong result: {json.dumps(verification)}")

    # ---- scan kernel durations of the timed steps (hipEvents on the launch stream) --
    buf = (ctypes.c_float * 64)()
    n = L.knn_scan_times(index.local._h, buf, min(64, args.steps))
    scan_ms = [buf[i] for i in range(n) if buf[i] > 0]
    info = index.local.last_scan()
    seed = index.local.last_seed()
    passes = (nq + info["query_tile"] - 1) // info["query_tile"]
    # the timed kernel scans every row except the seed sample's (searched by a small launch of the same kernel just before)
    rows_kernel = nb_local - seed["sample_rows"]
    alg_bytes = passes * rows_kernel * d * 4 + nq * d * 4 + nq * k * 12
    avg_scan_ms = float(np.mean(scan_ms)) if scan_ms else None


    # the legs every rank takes part in (rank 0 runs them once its headline is assembled: a run cut at its deadline keeps it)
    def all_rank_legs():
        res = {}
        # the other multi-GPU split: Pfam-sized all-vs-all, rows replicated, queries split
        if not args.no_batch:
            progress("all_vs_all_query_sharded")
            a = query_sharded_all_vs_all(dev, L, _lib, faiss, rank, world, dist if use_pg else None)
            if a is not None:
                res["all_vs_all_query_sharded"] = a
        if world > 1 and not args.no_extras:
            progress("hnsw_replicas")
            hr = hnsw_replicas(dev, L, _lib, faiss, rank, world, dist)
            if hr is not None:
                res["hnsw_replicas"] = hr
        return res

    if rank != 0:
        all_rank_legs()
        dist.destroy_process_group()
        return None

    ms_per_step = 1e3 * elapsed / args.steps
    out = {
        "metric": "queries/sec, flat inner-product kNN, d=1024 fp32, k=100 (recall@k = 1.0: exact flat search)",
        "value": nq * args.steps / elapsed,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "ms_per_step_median": (step_ms[len(step_ms) // 2] if step_ms else None),
        "ms_per_step_spread": ({"min": step_ms[0], "p90": step_ms[int(0.9 * (len(step_ms) - 1))], "max": step_ms[-1],
                                "what": "HIP events behind consecutive timed searches, completion to completion (rank 0); `ms_per_step` "
                                        "and `value` are the mean over the timed region as the driver's contract defines them"} if step_ms else None),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": (f"BASELINE configs[3]: synthetic {args.nb_total}x{d} fp32 flat IP, k={k}, " if args.scaling == "strong" else
                         f"SURVEY 8(d) weak scaling: synthetic {args.nb_per_gpu} rows per GPU ({args.nb_total}x{d} in all) fp32 flat IP, k={k}, ")
                        + f"{nq} queries per step, DB row-sharded over {world} " + ("ranks on ONE GPU (rehearsal)" if rehearse and world > 1 else "GPU(s)")
                        + ((", gloo all-gather (one-GPU rehearsal of the RCCL path) of per-shard top-k keys + merge" if rehearse else
                            ", RCCL all-gather of per-shard top-k keys + merge") if world > 1 else ""),
            "nb_total": args.nb_total, "nb_per_gpu": nb_local, "d": d, "k": k, "queries_per_step": nq,
            "parallelism": f"db-row-shard x{world}", "searches_in_flight": in_flight,
        },
        "ranks_seen": ranks_seen,
        "verification": verification,
    }
    if scan_ms_ranks:
        ok = [v for v in scan_ms_ranks if v > 0]
        out["scan_ms_per_rank"] = {"min": min(ok) if ok else None, "max": max(ok) if ok else None, "all": scan_ms_ranks,
                                   "note": "mean HIP-event duration of each rank's scan launches (with two searches in flight a launch's "
                                           "events span the other lane's work too)"}
    sweep_ref = None if rehearse else shard_sweep_reference(nb_local, in_flight)  # (a rehearsal's times mean nothing: N ranks share one GPU over gloo)
    if sweep_ref is not None:
        # what ONE GPU needs for a shard of this size with no collective at all (builder-run sweep, committed): the part of a
        # multi-GPU step that is not the scan shows up as efficiency < 1
        out["efficiency_vs_shard_sweep"] = {"value": sweep_ref["ms"] / ms_per_step, "shard_sweep_ms": sweep_ref["ms"], "source": sweep_ref["source"]}
    if use_pg:
        out["collective"] = {"op": "all_gather_into_tensor", "backend": "gloo (one-GPU rehearsal)" if rehearse else "nccl (RCCL)",
                             "bytes_per_rank": nq * k * 8, "avg_ms": gather_ms,
                             "rccl_version": rccl_version(), "pg_timeout_s": PG_TIMEOUT_S,
                             "note": "HIP events on the lane's stream around the collective of every timed step (rank 0)"}
    basis = "HIP events around the scan launch on its stream"
    if avg_scan_ms and in_flight == 2:
        # two searches in flight: the two lanes' scan launches share the GPU, so the event pair around ONE launch spans
        # the other's work too (measured 11.1 ms per launch against 7.2 ms per step).  The step time is the honest
        # denominator there: one scan's bytes per step.
        avg_scan_ms = 1e3 * elapsed / args.steps
        basis = "ms_per_step (two searches in flight: per-launch event durations overlap)"
    if avg_scan_ms:
        achieved = alg_bytes / (avg_scan_ms * 1e-3) / 1e9
        traffic, traffic_source = None, None
        tfile = ROOT / "profiles" / "pmc_traffic.json"
        if tfile.exists():
            try:
                allrec = json.loads(tfile.read_text())
                rec = allrec.get(info["kernel"] + "_ip", {})
                # the counters were collected on the default workload, from the kernels of ONE source file: they are quoted
                # only for that workload and only while csrc/knn355.hip is byte for byte the file they were collected from
                same_source = allrec.get("_meta", {}).get("knn355_hip_sha256") == kernel_source_sha256()
                if rec.get("algorithmic_bytes_per_launch") == alg_bytes and same_source:
                    traffic = rec.get("hbm_bytes_per_launch")
                    traffic_source = ("profiles/pmc_traffic.json: builder-run rocprofv3 --pmc passes of this command on this source "
                                      "(FETCH_SIZE x2 + WRITE_SIZE, per launch), not measured in this run")
                elif rec.get("algorithmic_bytes_per_launch") == alg_bytes:
                    traffic_source = "profiles/pmc_traffic.json was collected from another revision of csrc/knn355.hip: not quoted"
            except Exception:
                traffic = None
        out["roofline"] = {
            "bound": "hbm", "kernel": info["kernel"], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
            "algorithmic_bytes_per_launch": alg_bytes, "avg_kernel_ms": avg_scan_ms, "duration_basis": basis,
            "launches_timed": len(scan_ms), "db_passes": passes, "grid": info["grid"],
            "rows_per_launch": rows_kernel, "seed_sample_rows": seed["sample_rows"],
            "mfma_tflops": 2.0 * nq * rows_kernel * d / (avg_scan_ms * 1e-3) / 1e12,
        }
        if rank == 0 and world == 1:
            # the same rows under a plain read kernel on THIS box (outside the timed region): boxes of a pool differ by
            # a few percent and no kernel reaches the data-sheet 8 TB/s -- `frac` stays priced against that
            ms_read, nbytes = ctypes.c_float(), ctypes.c_int64()
            _lib.check(L.knn_flat_read_rate(index.local._h, 5, ctypes.byref(ms_read), ctypes.byref(nbytes)))
            if ms_read.value > 0:
                read_gbs = nbytes.value / (ms_read.value * 1e-3) / 1e9
                out["roofline"]["box_read_rate"] = {
                    "GBs": read_gbs, "ms": ms_read.value, "bytes": nbytes.value, "frac_of_it": achieved / read_gbs,
                    "what": "knn_flat_read_rate: grid-stride 16-byte loads over the index's own rows, best of 15 launches"}

    _PARTIAL["out"] = out  # (from here on a run cut at its deadline still prints the headline and the legs that finished)
    progress("headline done; extras")
    out.update(all_rank_legs())
    if world == 1 and not args.no_extras:
        progress("shard_unit")
        out["shard_unit"] = shard_unit_step(dev, L, _lib, faiss, d, k, q)
        progress("sweep")
        out["sweep"] = nq_sweep(index, dev, L, _lib, d, k, nb_local)
        progress("host_buffers")
        out["host_buffers"] = host_buffer_step(index, q.cpu().numpy(), k)
    if world == 1 and not args.no_cpu and cpu_rows is not None:
        progress("cpu_baseline")
        out["cpu_baseline"] = cpu_baseline(cpu_rows, q.cpu().numpy(), k, args.nb_total)
        cpu_rows.close()
    if world == 1 and not (args.no_batch and args.no_extras):
        del index
        torch.cuda.empty_cache()
        L.knn_trim()
        if not args.no_batch:
            progress("batch")
            out["batch"] = batch_config(dev, L, _lib, faiss)
        if not args.no_extras:
            progress("hnsw")
            out["hnsw"] = hnsw_config(dev, L, _lib, faiss)
        progress("done")
        rd = real_data()
        if rd:
            out["real_data"] = rd
    if use_pg:
        dist.destroy_process_group()
    return json.dumps(out)


def real_data():
    """SURVEY 8(d), opportunistic and never required: if the reference's embedding files are on this box (paths.py resolves
    cath/data and pfam/full_sequences_data from $KNN355_PROJECT_ROOT / the git checkout / the current directory), the
    reference's own calls on them -- cath/search.py:13-26 on every *.npy of cath/data (k = 10 + self, cosine), the flat mode of
    pfam/proteins_search.py:49 on full_sequences.npy (k = 1000, cosine) -- timed end to end from numpy arrays."""
    try:
        from knn_for_homology_amd import paths
        from knn_for_homology_amd.cath import search as cath_search
        out = {}
        d = paths.cath_data()
        for npy in (sorted(d.glob("*.npy"))[:4] if d.is_dir() else []):
            x = np.load(npy).astype(np.float32)
            cath_search.search(x[: min(len(x), 2048)], hits=10)  # (warm-up)
            t0 = time.perf_counter()
            hits, scores = cath_search.search(x, hits=10)
            out[f"cath/data/{npy.name}"] = {"shape": list(x.shape), "hits": 10, "metric": "cosine", "s": time.perf_counter() - t0,
                                            "self_hit_dropped": bool((hits != np.arange(len(x))[:, None]).all())}
        f = paths.full_sequences_data() / "full_sequences.npy"
        if not f.is_file():
            f = paths.pfam_dir() / "full_sequences.npy"
        if f.is_file():
            from knn_for_homology_amd import faiss as kfaiss
            x = np.load(f).astype(np.float32)
            kfaiss.normalize_L2(x)
            t0 = time.perf_counter()
            idx = kfaiss.IndexFlat(x.shape[1], kfaiss.METRIC_INNER_PRODUCT)
            idx.add(x)
            D, I = idx.search(x, min(1000, len(x)))
            out[str(f.relative_to(paths.project_root()))] = {"shape": list(x.shape), "k": int(D.shape[1]), "metric": "cosine", "s": time.perf_counter() - t0}
        return out or None
    except Exception as e:  # noqa: BLE001 -- an extra: never a reason to lose the bench line
        return {"error": f"{type(e).__name__}: {e}"}


def kernel_source_sha256():
    import hashlib
    return hashlib.sha256((ROOT / "knn-for-homology_amd" / "csrc" / "knn355.hip").read_bytes()).hexdigest()


def shard_sweep_reference(rows, in_flight):
    """the committed one-GPU step time of a shard of `rows` rows (tools/shard_sweep.py), newest profile first"""
    for f in sorted((ROOT / "profiles").glob("r[0-9][0-9]_shard_sweep.json"), reverse=True):
        name = f.name
        try:
            for rec in json.loads(f.read_text())["rows"]:
                if rec.get("rows") == rows and rec.get("flags", 0) == 0:
                    key = "ms_lanes2" if in_flight == 2 else "ms_lanes1"
                    return {"ms": rec[key], "source": f"profiles/{name}: N={rec.get('N')}, {key}"}
        except Exception:
            continue
    return None


def query_sharded_all_vs_all(dev, L, _lib, faiss, rank, world, dist):
    """S-pfam: 200 000 clustered rows x 1024, cosine, k=100, all-vs-all.  Every rank holds all rows and
    answers its contiguous slice of the queries (QueryShardedFlatIndex): no collective on the data path, the results
    stay on the rank that computed them.  value = 200 000 / the slowest rank's time."""
    from knn_for_homology_amd.sharded import QueryShardedFlatIndex
    n, d, k, ncent = 200_000, 1024, 100, 2000
    g = torch.Generator(device=dev)
    g.manual_seed(21)  # the same rows on every rank
    cent = torch.randn((ncent, d), generator=g, device=dev)
    which = torch.randint(0, ncent, (n,), generator=g, device=dev)
    x = cent[which] + 0.35 * torch.randn((n, d), generator=g, device=dev)
    x[-1000:] = x[:1000]  # 0.5 % exact duplicates: the tie rule is on the path
    _lib.check(L.knn_normalize_l2_dev(x.data_ptr(), n, d, None))
    idx = QueryShardedFlatIndex(d, faiss.METRIC_INNER_PRODUCT, rank=rank, world=world)
    idx.reserve(n)
    idx.add_dev(x)
    lo, hi = idx.query_bounds(n)
    times = []
    for it in range(3):
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        D, I = idx.search_dev(x, k)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        if it >= 1:
            times.append(el)
    # every query finds itself (or its duplicate with the lower id) first
    first = I[:, 0]
    own = torch.arange(lo, hi, device=dev)
    ok = bool(((first == own) | (first == own - (n - 1000))).all().item())
    t = float(np.median(times))
    out = {"workload": f"S-pfam: {n}x{d} clustered, cosine k={k}, all-vs-all; rows replicated, queries split over {world} GPU(s), no collective",
           "value": n / t, "unit": "queries/s", "ms": 1e3 * t, "queries_per_rank": hi - lo, "self_hit_first_on_rank0": ok,
           "roofline": {"bound": "mfma", "achieved": 2.0 * n * n * d / t / 1e12 / world, "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s per GPU",
                        "frac": 2.0 * n * n * d / t / 1e12 / world / FP32_MFMA_PEAK_TF,
                        "note": "whole search per GPU (sample pass, scan, selection), not the kernel alone"}}
    box = box_mfma_rate(L, _lib) if rank == 0 else None
    if box:
        out["roofline"]["box_mfma_rate"] = dict(box, frac_of_it=out["roofline"]["achieved"] / box["tflops"])
    if world == 1:
        # one GPU and a whole-index self-search: the symmetric launch is available (only tiles on/above the diagonal)
        Ds = torch.empty((n, k), device=dev, dtype=torch.float32)
        Is = torch.empty((n, k), device=dev, dtype=torch.int64)
        ts = []
        for it in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _lib.check(L.knn_flat_search_self_dev(idx.backend.index._h, k, Ds.data_ptr(), Is.data_ptr()))
            torch.cuda.synchronize()
            if it >= 1:
                ts.append(time.perf_counter() - t0)
        same = bool(torch.equal(Is, I) and torch.equal(Ds.view(torch.int32), D.view(torch.int32)))
        out["self_search_symmetric"] = {"ms": 1e3 * float(np.median(ts)), "queries_per_s": n / float(np.median(ts)),
                                        "identical_to_plain_search": same}
    del idx, x, D, I
    torch.cuda.empty_cache()
    return out


def shard_unit_step(dev, L, _lib, faiss, d, k, q):
    """The per-GPU unit of the 8-GPU run on ONE GPU, measured in this run: 1.25 M x 1024 rows (10 M / 8 = the strong-scaling
    shard = the weak-scaling unit of SURVEY 8(d)), the bench's 32 queries, one search in flight and two.  No collective: what is
    left of a multi-GPU step once the all-gather is taken out (tools/shard_sweep.py sweeps N = 1, 2, 4, 8)."""
    from knn_for_homology_amd.sharded import ShardedFlatIndex
    nb = 1_250_000
    idx = ShardedFlatIndex(d, faiss.METRIC_INNER_PRODUCT, rank=0, world=1, row_offset=0)
    idx.reserve(nb)
    g = torch.Generator(device=dev)
    g.manual_seed(29)
    for i0 in range(0, nb, 250_000):
        x = torch.randn((250_000, d), generator=g, device=dev)
        _lib.check(L.knn_normalize_l2_dev(x.data_ptr(), 250_000, d, None))
        idx.add_dev(x)
        del x
    rec = {"rows": nb, "queries_per_step": int(q.shape[0]), "k": k}
    for lanes in (1, 2):
        best = None
        for _rep in range(3):
            for _ in range(5):
                if lanes == 1:
                    idx.backend._turn = 0
                idx.submit(q, k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(40):
                if lanes == 1:
                    idx.backend._turn = 0
                pend = idx.submit(q, k)
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / 40
            best = t if best is None else min(best, t)
        pend.result()
        rec[f"ms_in_flight_{lanes}"] = 1e3 * best
        rec[f"hbm_frac_in_flight_{lanes}"] = nb * d * 4 / best / 1e9 / HBM_PEAK_GBS
    rec["note"] = "best of 3 x 40 steps each; whole step (scan + selection), rows x d x 4 bytes per step against 8 TB/s"
    del idx
    torch.cuda.empty_cache()
    return rec


def nq_sweep(index, dev, L, _lib, d, k, nb):
    """nq in {1, 8, 32, 1024, 10 000} against the same resident database (search-only, device buffers)."""
    res = []
    rng = np.random.default_rng(25)
    for nq in (1, 8, 32, 1024, 10_000):
        qh = rng.standard_normal((nq, d), dtype=np.float32)
        q = torch.from_numpy(qh).to(dev)
        _lib.check(L.knn_normalize_l2_dev(q.data_ptr(), nq, d, None))
        reps = 2 if nq >= 10_000 else (3 if nq >= 1024 else 8)
        # One rank, a batch of many query tiles: the synchronous entry (what IndexFlat.search runs on device buffers) -- it may
        # take the statistical seed, whose verification flag it reads before returning, and with it the 256 x 256 tile; the
        # lanes of the sharded index are asynchronous and never do.  Several ranks: the sharded path, as in the timed steps.
        sync_entry = index.world == 1 and nq >= 1024
        run = (lambda: index.backend.search(q, k)) if sync_entry else (lambda: index.search_dev(q, k))
        run()
        torch.cuda.synchronize()
        times, scans = [], []
        for _ in range(reps):
            t0 = time.perf_counter()
            run()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
            scans.append(index.local.last_scan()["ms"])
        info = index.local.last_scan()
        t, sm = float(np.median(times)), float(np.median(scans))
        passes = (nq + info["query_tile"] - 1) // info["query_tile"]
        rows = nb - index.local.last_seed()["sample_rows"]  # rows the timed kernel scans (the seed sample has its own launch)
        by = passes * rows * d * 4 + nq * d * 4 + nq * k * 12
        fl = 2.0 * nq * rows * d
        if nq > 1024:
            # several launches (full query tiles + the remainder on a narrower build): last_scan() describes the last one only --
            # the whole search's time is the denominator, and the tile of the full launches is read from a search of the
            # queries in front of the remainder (not timed)
            if sync_entry and nq - nq % 256 > 0:
                index.backend.search(q[:nq - nq % 256].contiguous(), k)
                torch.cuda.synchronize()
                info = index.local.last_scan()
            passes = (nq + info["query_tile"] - 1) // info["query_tile"] if info["query_tile"] >= 128 else (nq + 127) // 128
            rec = {"nq": nq, "queries_per_s": nq / t, "ms": 1e3 * t, "kernel": info["kernel"] + " (+ the remainder's launch)", "kernel_ms": None,
                   "db_passes": passes, "hbm_frac": (passes * rows * d * 4 + nq * d * 4 + nq * k * 12) / t / 1e9 / HBM_PEAK_GBS,
                   "mfma_frac": fl / t / 1e12 / FP32_MFMA_PEAK_TF, "bound": "mfma", "basis": "whole search (all launches, selections included)"}
            if sync_entry:
                rec["entry"] = "synchronous (IndexFlat.search on device buffers): statistical seed allowed"
            res.append(rec)
            continue
        rec = {"nq": nq, "queries_per_s": nq / t, "ms": 1e3 * t, "kernel": info["kernel"], "kernel_ms": sm, "db_passes": passes,
               "hbm_frac": by / (sm * 1e-3) / 1e9 / HBM_PEAK_GBS, "mfma_frac": fl / (sm * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF}
        rec["bound"] = "hbm" if passes == 1 else "mfma"
        if sync_entry:
            rec["entry"] = "synchronous (IndexFlat.search on device buffers): statistical seed allowed"
        res.append(rec)
    return res


def host_buffer_step(index, q_host, k):
    """The same step through the host-pointer entry point (numpy in, numpy out): adds the H2D of the queries and
    the D2H of nq*k*12 bytes to every step (PCIe-inclusive; never `value`)."""
    idx = index.local
    idx.search(q_host, k)
    times = []
    for _ in range(10):
        t0 = time.perf_counter()
        idx.search(q_host, k)
        times.append(time.perf_counter() - t0)
    t = float(np.median(times))
    return {"ms_per_step": 1e3 * t, "queries_per_s": q_host.shape[0] / t,
            "note": "IndexFlat.search(numpy): query upload + scan + merge + result download, median of 10"}


def host_cpu_share():
    """CPUs this process may really use: the affinity mask, cut by the cgroup's cpu.max quota when there is one."""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = Path(path).read_text().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
            break
        except Exception:
            continue
    share = avail if quota is None else max(1, min(avail, int(quota + 0.5)))
    return avail, quota, share


def cpu_sample_rows(requested):
    """rows of the database the CPU leg scans: bounded by a quarter of the memory this process may still take"""
    free = None
    try:
        for ln in Path("/proc/meminfo").read_text().splitlines():
            if ln.startswith("MemAvailable:"):
                free = int(ln.split()[1]) * 1024
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            txt = Path(path).read_text().strip()
            if txt != "max":
                lim = int(txt)
                used = 0
                try:
                    used = int(Path(path).with_name("memory.current" if path.endswith("memory.max") else "memory.usage_in_bytes").read_text())
                except Exception:
                    pass
                free = min(free, lim - used) if free is not None else lim - used
            break
        except Exception:
            continue
    if free is None:
        return min(requested, 500_000)
    return int(max(100_000, min(requested, (free // 4) // 4096)))


def cpu_baseline(rows, q_host, k, nb_total):
    """The reference's CPU path (faiss-cpu 1.7.2 IndexFlat.search, /root/reference/seqvec_search/main.py:45) on the host
    cores of this box, on the first S database rows.  `value` = the FASTEST faithful variant:
      native_avx512_openmp  oracle/cpu_scan.c -- OpenMP over row ranges, AVX-512 4x4 register-blocked dots, per-thread
                            thresholds + candidate buffers, final merge; rows in memory first-touched by their thread
      blas_rows1024 / 8192  FAISS's blocked-sgemm algorithm restated with numpy's bundled OpenBLAS (sub-sample)
      faiss                 the real module, when importable
    and next to it the host's own read rate over the same rows (`dram_read_GBs`): no scan can be faster than that."""
    from oracle import knn_oracle as ko
    S, d = rows.n, rows.d
    nq = q_host.shape[0]
    scale = nb_total / S
    avail, quota, share = host_cpu_share()

    def timed(fn, min_passes=5, budget=8.0, max_passes=40):
        ts = []
        r = None
        while len(ts) < min_passes or (sum(ts) < budget and len(ts) < max_passes):
            t0 = time.perf_counter()
            r = fn()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), len(ts), float(min(ts)), float(max(ts)), r

    variants = {}
    # ---- the native scan at a few thread counts (the rows were first-touched with rows.threads threads) ----
    # (never more threads than the CPUs this process may use: an oversubscribed run's best pass is not a number the box
    # can repeat -- round 3 tried 2 x share and reported its thread count as "cores")
    cand = sorted({min(rows.threads, share), share, max(1, share // 2)}, reverse=True)
    best = None
    for th in cand:
        tm, n_, lo, hi, (Dc, Ic) = timed(lambda: rows.search(q_host, k, ko.METRIC_INNER_PRODUCT, threads=th), min_passes=3, budget=4.0, max_passes=20)
        rd = min(rows.read_seconds(th) for _ in range(3))
        # (the host is shared with other jobs: the BEST pass is what this CPU path can do, and the baseline gets the benefit of
        # the doubt; the median is reported beside it)
        rec = {"threads": th, "best_s_on_sample": lo, "median_s_on_sample": tm, "passes": n_, "max_s": hi, "queries_per_s": nq / (lo * scale),
               "queries_per_s_median": nq / (tm * scale), "scan_GBs": S * d * 4 / lo / 1e9, "gflops": 2.0 * nq * S * d / lo / 1e9,
               "dram_read_GBs": S * d * 4 / rd / 1e9, "times_dram_floor": lo / rd}
        variants[f"native_avx512_openmp_t{th}"] = rec
        if best is None or lo < best[0]:
            best = (lo, th, n_, rec, Dc, Ic)
    t_nat, th_nat, n_nat, rec_nat, Dc, Ic = best
    from oracle import cpu_scan as cs
    fma_peak = cs.fma_gflops(th_nat, 0.5)  # the same threads, nothing but vector FMAs out of registers
    t1 = timed(lambda: rows.search(q_host, k, ko.METRIC_INNER_PRODUCT, threads=1), min_passes=1, budget=0.0, max_passes=1)[0] if S <= 2_500_000 else None

    # ---- FAISS's own algorithm on numpy's BLAS, on a sub-sample (it is 50-100x slower) ----
    sub = rows.array[: min(S, 250_000)]
    sub_scale = nb_total / sub.shape[0]
    blas_threads = share
    limits = None
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        limits = threadpool_limits
        threadpool_limits(limits=share, user_api="blas")
        pools = [p["num_threads"] for p in threadpool_info() if p.get("user_api") == "blas"]
        if pools:
            blas_threads = max(pools)
    except Exception:
        pass
    tb, nb_, lo, hi, _ = timed(lambda: ko.faiss_flat_blas_restated(sub, q_host, k, ko.METRIC_INNER_PRODUCT, bs_y=1024), min_passes=3, budget=3.0)
    variants["blas_rows1024"] = {"queries_per_s": nq / (tb * sub_scale), "median_s_on_sample": tb, "passes": nb_, "min_s": lo, "max_s": hi,
                                 "sample_rows": int(sub.shape[0]), "threads": blas_threads,
                                 "what": "numpy/OpenBLAS sgemm, FAISS's blocking (4096 queries x 1024 rows) + per-row top-k"}
    tb8, nb8, lo, hi, _ = timed(lambda: ko.faiss_flat_blas_restated(sub, q_host, k, ko.METRIC_INNER_PRODUCT, bs_y=8192), min_passes=3, budget=3.0)
    variants["blas_rows8192"] = {"queries_per_s": nq / (tb8 * sub_scale), "median_s_on_sample": tb8, "passes": nb8, "min_s": lo, "max_s": hi,
                                 "sample_rows": int(sub.shape[0]), "threads": blas_threads,
                                 "what": "same with 8192-row blocks (amortises numpy's per-call overhead; not FAISS's blocking)"}
    if limits is not None:
        try:
            with limits(limits=1, user_api="blas"):
                t0 = time.perf_counter()
                ko.faiss_flat_blas_restated(sub[: sub.shape[0] // 4], q_host, k, ko.METRIC_INNER_PRODUCT, bs_y=1024)
                variants["blas_rows1024_one_thread"] = {"queries_per_s": nq / ((time.perf_counter() - t0) * 4.0 * sub_scale), "threads": 1,
                                                        "what": "a 32 x 1024 x 1024 sgemm per block is too small to thread: the fork/join of "
                                                                "every one of the ~10 k calls costs more than the block's arithmetic"}
        except Exception:
            pass
    # cores = the CPUs available to this process (affinity mask cut by the cgroup quota); threads_used is reported beside it
    kind, value, cores, t_used, n_used, what = "port", nq / (t_nat * scale), share, t_nat, n_nat, "native_avx512_openmp"
    threads_used = th_nat
    try:
        import faiss as real_faiss  # noqa: the site-packages module, not knn_for_homology_amd.faiss
        ref = real_faiss.IndexFlat(d, real_faiss.METRIC_INNER_PRODUCT)
        ref.add(rows.array)
        tf, nf, lo, hi, _ = timed(lambda: ref.search(q_host, k))
        variants["faiss"] = {"queries_per_s": nq / (tf * scale), "median_s_on_sample": tf, "passes": nf, "version": getattr(real_faiss, "__version__", "?")}
        kind, value, t_used, n_used, what = "reference", nq / (tf * scale), tf, nf, "faiss"
        threads_used = real_faiss.omp_get_max_threads()
        del ref
    except Exception:
        pass
    # same sample on the GPU: neighbours must agree (recall of the exact flat path)
    from knn_for_homology_amd import faiss
    subidx = faiss.IndexFlat(d, faiss.METRIC_INNER_PRODUCT)
    subidx.set_tuning(64, 0, 0)  # another instantiation, so the benched kernel's rocprof stats stay clean
    m = min(S, 500_000)
    subidx.add(rows.array[:m])
    Dg, Ig = subidx.search(q_host, k)
    Dm, Im = rows.search(q_host, k, ko.METRIC_INNER_PRODUCT, threads=th_nat) if m == S else cpu_scan_prefix(rows, m, q_host, k)
    recall = ko.recall_at_k(Ig, Im)
    return {"value": value, "unit": "queries/s", "cores": cores, "threads_used": threads_used, "kind": kind, "variant": what,
            "value_median": (rec_nat["queries_per_s_median"] if what == "native_avx512_openmp" else None),
            "sample": f"first {S} of {nb_total} database rows ({S * d * 4 / 1e9:.1f} GB in host memory), {nq} queries, k={k}; "
                      f"best of {n_used} passes ({t_used:.3f}s; the host is shared); value extrapolates linearly in the database size",
            "host": {"cpus_in_affinity_mask": avail, "cgroup_cpu_quota": quota, "cpus_available": share, "threads_used": threads_used, "avx512": bool(rows_has_avx512()),
                     "dram_read_GBs": rec_nat["dram_read_GBs"], "scan_over_dram_floor": rec_nat["times_dram_floor"],
                     "fma_peak_gflops": fma_peak, "scan_gflops": rec_nat["gflops"], "scan_over_compute_floor": fma_peak / rec_nat["gflops"],
                     "floors": "a scan of S rows for nq queries cannot take less than S*d*4 B / dram_read_GBs nor less than 2*nq*S*d flop / "
                               "fma_peak_gflops: with the box's CPU share the second floor is the higher one at nq = 32",
                     "single_thread_queries_per_s": (nq / (t1 * scale)) if t1 else None},
            "variants": variants, "gpu_recall_at_k_on_sample": recall}


def rows_has_avx512():
    from oracle import cpu_scan as cs
    return cs.lib().cpu_scan_has_avx512()


def cpu_scan_prefix(rows, m, q_host, k):
    """the native scan restricted to the first m rows (the GPU cross-check uses a prefix of the sample)"""
    from oracle import cpu_scan as cs
    return cs.flat_search(rows.array[:m], q_host, k, 0, threads=min(rows.threads, 16))


_BOX_MFMA = {}


def box_mfma_rate(L, _lib):
    """what THIS box's fp32 matrix pipes sustain (knn_mfma_rate: bare v_mfma_f32_32x32x2_f32 loops, random operands, two
    workgroups per CU, after 300 ms of load) -- measured once per run, outside every timed region; `frac` stays priced against
    the data sheet's 157.3 TFLOP/s"""
    if not _BOX_MFMA:
        tf, mhz = ctypes.c_float(), ctypes.c_float()
        try:
            _lib.check(L.knn_mfma_rate(300, ctypes.byref(tf), ctypes.byref(mhz)))
            _BOX_MFMA.update({"tflops": float(tf.value), "clock_mhz": float(mhz.value),
                              "what": "knn_mfma_rate: back-to-back fp32 MFMAs out of registers, best of 8 launches after 300 ms of load"})
        except Exception as e:  # noqa: BLE001 -- a measurement aid must not take the bench line down
            print("box_mfma_rate failed:", e, file=sys.stderr)
            return None
    return dict(_BOX_MFMA)


def batch_config(dev, L, _lib, faiss):
    """BASELINE configs[1]: CATH20-sized all-vs-all, L2, k=300 (+1 self hit)."""
    n, d, k = 14433, 1024, 301
    xh = np.random.default_rng(20).standard_normal((n, d), dtype=np.float32)
    x = torch.from_numpy(xh).to(dev)
    idx = faiss.IndexFlat(d, faiss.METRIC_L2)
    _lib.check(L.knn_flat_add_dev(idx._h, x.data_ptr(), n, None))
    D = torch.empty((n, k), device=dev, dtype=torch.float32)
    I = torch.empty((n, k), device=dev, dtype=torch.int64)
    # The chip raises its clock over the first ~100 ms of a burst of launches (profiles/r03_pmc_sq_summary.txt: CATH-sized
    # launches climbed from 1.97 to 2.32 GHz over the eight searches round 3 timed, Pfam-sized ones run at 2.38 GHz): the
    # searches are repeated for WARM_S seconds first, then 12 are timed back to back -- the steady state a caller that
    # searches file after file sees (cath/search.py:36-52 loops over the embedding files).
    WARM_S = 0.25
    times, scans = [], []
    t_warm = time.perf_counter()
    while time.perf_counter() - t_warm < WARM_S:
        _lib.check(L.knn_flat_search_dev(idx._h, x.data_ptr(), n, k, D.data_ptr(), I.data_ptr(), None))
    for it in range(12):
        t0 = time.perf_counter()
        _lib.check(L.knn_flat_search_dev(idx._h, x.data_ptr(), n, k, D.data_ptr(), I.data_ptr(), None))  # (synchronous)
        times.append(time.perf_counter() - t0)
        scans.append(idx.last_scan()["ms"])
    info = idx.last_scan()
    seed = idx.last_seed()
    t = float(np.median(times))
    sm = float(np.median(scans))
    flops = 2.0 * n * n * d                               # the whole search
    flops_kernel = 2.0 * n * (n - seed["sample_rows"]) * d  # what the timed launch computes (the seed sample has its own launch)
    # the same all-vs-all as a self-search of the index (what cath.search.search runs): only the score tiles on and
    # above the diagonal are multiplied
    ts, ss = [], []
    t_warm = time.perf_counter()
    while time.perf_counter() - t_warm < WARM_S:
        _lib.check(L.knn_flat_search_self_dev(idx._h, k, D.data_ptr(), I.data_ptr()))
    for it in range(12):
        t0 = time.perf_counter()
        _lib.check(L.knn_flat_search_self_dev(idx._h, k, D.data_ptr(), I.data_ptr()))  # (synchronous)
        ts.append(time.perf_counter() - t0)
        ss.append(idx.last_scan()["ms"])
    sinfo, sseed = idx.last_scan(), idx.last_seed()
    st_, sm_ = float(np.median(ts)), float(np.median(ss))
    tiles = (n + 127) // 128
    flops_sym = 2.0 * (tiles * (tiles + 1) // 2) * 128 * 128 * d  # what the symmetric launch multiplies (whole tiles)
    self_search = {"kernel": sinfo["kernel"], "ms": 1e3 * st_, "kernel_ms": sm_, "queries_per_s": n / st_, "seed": sseed,
                   "roofline": {"bound": "mfma", "achieved": flops_sym / (sm_ * 1e-3) / 1e12, "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                                "frac": flops_sym / (sm_ * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF,
                                "note": "flops the launch executes (tiles on and above the diagonal), not 2*n*n*d"},
                   "speedup_vs_plain_search": t / st_}
    # end to end as the reference times it (cath/search.py:42-46: copy + normalise + add + search), host numpy in/out
    from knn_for_homology_amd.cath.search import search as cath_search
    cath_search(xh, hits=300, metric=faiss.METRIC_L2)
    e2e = []
    for _ in range(5):
        t0 = time.perf_counter()
        cath_search(xh, hits=300, metric=faiss.METRIC_L2)
        e2e.append(time.perf_counter() - t0)
    box = box_mfma_rate(L, _lib)
    return {"workload": "BASELINE configs[1]: CATH20-sized 14433x1024 all-vs-all, L2, k=300 (+ self hit)",
            "value": n / t, "unit": "queries/s", "ms": 1e3 * t, "kernel": info["kernel"], "kernel_ms": sm,
            "seed": seed, "timing": f"{WARM_S} s of back-to-back searches first (clock ramp), then the median of 12",
            "roofline": {"bound": "mfma", "achieved": flops_kernel / (sm * 1e-3) / 1e12, "peak": FP32_MFMA_PEAK_TF,
                         "unit": "TFLOP/s", "frac": flops_kernel / (sm * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF,
                         "search_frac": flops / t / 1e12 / FP32_MFMA_PEAK_TF,
                         "box_mfma_rate": dict(box, frac_of_it=flops_kernel / (sm * 1e-3) / 1e12 / box["tflops"]) if box else None,
                         "note": "frac: the scan launch's own flops / its duration; search_frac: all 2*n*n*d flops / the whole "
                                 "device-resident search (sample pass, scan, final selection)"},
            "self_search": self_search,
            "end_to_end": {"ms": 1e3 * float(np.median(e2e)), "queries_per_s": n / float(np.median(e2e)),
                           "what": "cath.search.search(numpy fp32[14433,1024], hits=300, L2): H2D 59 MB + add + search + D2H 52 MB"}}


def hnsw_replicas(dev, L, _lib, faiss, rank, world, dist):
    """HNSW does not shard (replicas only): every rank builds the same graph (the construction is
    deterministic) over the S-pfam rows and answers its OWN 4096 queries; value = all ranks' queries / the slowest
    rank's time.  No collective on the data path."""
    n, d, k, nq = 200_000, 1024, 100, 4096
    g = torch.Generator(device=dev)
    g.manual_seed(21)
    cent = torch.randn((2000, d), generator=g, device=dev)
    which = torch.randint(0, 2000, (n,), generator=g, device=dev)
    x = cent[which] + 0.35 * torch.randn((n, d), generator=g, device=dev)
    _lib.check(L.knn_normalize_l2_dev(x.data_ptr(), n, d, None))
    idx = faiss.IndexHNSWFlat(d, 32, faiss.METRIC_INNER_PRODUCT)
    t0 = time.perf_counter()
    idx.add_dev(x)
    build_s = time.perf_counter() - t0
    idx.hnsw.efSearch = 256
    qh = x[torch.from_numpy(np.random.default_rng(26 + rank).choice(n, nq, replace=False)).to(dev)].cpu().numpy()
    idx.search(qh[:256], k)
    ts = []
    for _ in range(3):
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        D, I = idx.search(qh, k)
        el = time.perf_counter() - t0
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ts.append(float(t.item()))
    tmed = float(np.median(ts))
    self_first = float((I[:, 0] >= 0).mean())
    del idx, x
    torch.cuda.empty_cache()
    return {"workload": f"BASELINE configs[4] as {world} replicas: 200000x1024 clustered, IP, HNSW M=32 efSearch=256, k=100, {nq} queries per rank "
                        "(bf16 beam, fp32 re-score)",
            "queries_per_s": world * nq / tmed, "build_s_rank0": build_s, "ms": 1e3 * tmed, "results_found_rank0": self_first}


def hnsw_config(dev, L, _lib, faiss):
    """BASELINE configs[4]: Pfam-subset-sized HNSW, M=32, efSearch=256, k=100, recall@100 against the flat search."""
    n, d, k, nq = 200_000, 1024, 100, 4096
    g = torch.Generator(device=dev)
    g.manual_seed(21)
    cent = torch.randn((2000, d), generator=g, device=dev)
    which = torch.randint(0, 2000, (n,), generator=g, device=dev)
    x = cent[which] + 0.35 * torch.randn((n, d), generator=g, device=dev)
    _lib.check(L.knn_normalize_l2_dev(x.data_ptr(), n, d, None))
    xh = x.cpu().numpy()
    del x, cent
    qsel = np.random.default_rng(26).choice(n, nq, replace=False)
    qh = np.ascontiguousarray(xh[qsel])
    flat = faiss.IndexFlat(d, faiss.METRIC_INNER_PRODUCT)
    flat.add(xh)
    flat.search(qh, k)
    t0 = time.perf_counter()
    Dt, It = flat.search(qh, k)
    t_flat = time.perf_counter() - t0
    idx = faiss.IndexHNSWFlat(d, 32, faiss.METRIC_INNER_PRODUCT)
    t0 = time.perf_counter()
    idx.add(xh)
    build_s = time.perf_counter() - t0
    idx.hnsw.efSearch = 256
    idx.search(qh[:256], k)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        Dh, Ih = idx.search(qh, k)
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    recall = float(np.mean([len(np.intersect1d(a[a >= 0], b)) for a, b in zip(Ih, It)])) / k
    out = {"workload": "BASELINE configs[4]: 200000x1024 clustered (2000 centres + 0.35 noise), IP, HNSW M=32 efConstruction=40 efSearch=256, k=100; "
                       "the beam walks on bf16 copies of the rows (fp32 accumulation), every returned distance is re-scored in fp32 by the flat search's chain",
           "build_s": build_s, "queries_per_s": nq / t, "recall_at_100_vs_flat": recall, "nq": nq,
           "flat_queries_per_s_same_queries": nq / t_flat, "note": "host numpy in/out for both (IndexHNSWFlat.search / IndexFlat.search)"}
    del idx
    # the reference's own shape (pfam/proteins_search.py:30-31,49): M = 42, efSearch = 256, k = 1000 (FAISS walks with ef = max(efSearch, k));
    # its consumers read the first 300 hits (pfam/proteins.py:41,246): recall at that depth beside recall@1000
    kr = 1000
    ref = faiss.IndexHNSWFlat(d, 42, faiss.METRIC_INNER_PRODUCT)
    t0 = time.perf_counter()
    ref.add(xh)
    build_ref = time.perf_counter() - t0
    ref.hnsw.efSearch = 256
    ref.search(qh[:256], kr)
    t0 = time.perf_counter()
    Dr, Ir = ref.search(qh, kr)
    t_ref = time.perf_counter() - t0
    _, Itr = flat.search(qh, kr)

    def rec(depth):
        return float(np.mean([len(np.intersect1d(a[a >= 0], b)) for a, b in zip(Ir[:, :depth], Itr[:, :depth])])) / depth
    out["reference_shape"] = {"workload": "pfam/proteins_search.py hnsw mode: M=42, IP, efSearch=256, k=1000, same rows",
                              "build_s": build_ref, "queries_per_s": nq / t_ref, "nq": nq,
                              "recall_at_300_vs_flat": rec(300), "recall_at_1000_vs_flat": rec(1000), "recall_at_100_vs_flat": rec(100),
                              "note": "recall@N: the first N returned hits against the exact first N (the reference's consumers slice [:, :300])"}
    return out


if __name__ == "__main__":
    main()
