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
the other's scan (an 8-GPU shard on one GPU with the collective path: 1.11 -> 1.04 ms/step);
on one GPU the steps run one after the other (two in flight bring nothing there and stretch
the scan kernel's measured duration by the other lane's launches).

Also reported on the same JSON line (N=1 only, outside the timed region):
  roofline      -- the scan kernel against the HBM roofline (it reads each shard once per
                   step: P = ceil(nq / query_tile) = 1 pass)
  cpu_baseline  -- FAISS 1.7.2's flat CPU algorithm restated with numpy's BLAS
                   (oracle/knn_oracle.py: faiss_flat_blas_restated), timed on this box's
                   host cores on a bounded sample of the same database rows
  batch         -- BASELINE.json configs[1]: CATH20-sized all-vs-all (14433 x 1024, L2,
                   k = 300 + self hit) through the cath.search entry point's kernel path
"""
import argparse
import ctypes
import json
import os
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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nb-total", type=int, default=10_000_000)
    ap.add_argument("--nq", type=int, default=32)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--d", type=int, default=1024)
    ap.add_argument("--in-flight", type=int, default=0, choices=(0, 1, 2),
                    help="searches in flight: 2 = steps alternate between the index's two lanes (the all-gather and the small "
                         "launches of one search hide behind the other's scan), 1 = one after the other, 0 = 2 on several GPUs, 1 on one")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-batch", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=1_000_000)
    return ap.parse_args()


def main():
    # stdout carries exactly ONE line (the JSON): everything libraries print while we run (RCCL's
    # version banner, for one) is sent to stderr instead
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        line = run()
    finally:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        os.close(real_stdout)
    if line is not None:
        print(line, flush=True)


def run():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
        args.gpus = world
    # KNN355_REHEARSE_ONE_GPU=1: every rank uses GPU 0 and the collective runs over gloo (NCCL refuses two
    # ranks per device) -- a functional rehearsal of the multi-rank path on a one-GPU box, not a measurement
    rehearse = os.environ.get("KNN355_REHEARSE_ONE_GPU", "0") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    use_pg = world > 1 or os.environ.get("KNN355_FORCE_COLLECTIVE", "0") == "1"
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from knn_for_homology_amd import _lib, faiss
    from knn_for_homology_amd.sharded import ShardedFlatIndex, shard_bounds
    L = _lib.lib()
    _lib.check(L.knn_init(local_rank))

    d, k, nq = args.d, args.k, args.nq
    lo, hi = shard_bounds(args.nb_total, world, rank)
    nb_local = hi - lo

    # ---- resident inputs ---------------------------------------------------------
    index = ShardedFlatIndex(d, faiss.METRIC_INNER_PRODUCT, rank=rank, world=world, row_offset=lo)
    index.reserve(nb_local)
    gen = torch.Generator(device=dev)
    gen.manual_seed(23 + rank)
    chunk = 500_000
    first_rows = None
    for i0 in range(0, nb_local, chunk):
        m = min(chunk, nb_local - i0)
        x = torch.randn((m, d), generator=gen, device=dev, dtype=torch.float32)
        _lib.check(L.knn_normalize_l2_dev(x.data_ptr(), m, d, None))
        index.add_dev(x)
        if first_rows is None and rank == 0 and world == 1 and not args.no_cpu:
            first_rows = x[: min(m, args.cpu_sample_rows)].cpu().numpy()
        del x
    q_host = np.random.default_rng(24).standard_normal((nq, d), dtype=np.float32)
    q = torch.from_numpy(q_host).to(dev)
    _lib.check(L.knn_normalize_l2_dev(q.data_ptr(), nq, d, None))
    torch.cuda.synchronize()

    in_flight = args.in_flight or (2 if world > 1 else 1)

    def step():
        # one search of the query batch, enqueued: consecutive steps alternate between the index's
        # two lanes (own stream and scratch memory each), so up to two searches are in flight and
        # the small launches at the ends of one hide behind the scan of the other
        if in_flight == 1:
            index.backend._turn = 0
        return index.submit(q, k)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pending = step()
    fence()
    elapsed = time.perf_counter() - t0
    D, I = pending.result()
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- scan kernel durations of the timed steps (hipEvents on the launch stream) --
    buf = (ctypes.c_float * 64)()
    n = L.knn_scan_times(index.local._h, buf, min(64, args.steps))
    scan_ms = [buf[i] for i in range(n) if buf[i] > 0]
    info = index.local.last_scan()
    passes = (nq + info["query_tile"] - 1) // info["query_tile"]
    alg_bytes = passes * nb_local * d * 4 + nq * d * 4 + nq * k * 12
    avg_scan_ms = float(np.mean(scan_ms)) if scan_ms else None

    if rank != 0:
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
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"BASELINE configs[3]: synthetic {args.nb_total}x{d} fp32 flat IP, k={k}, "
                        f"{nq} queries per step, DB row-sharded over {world} GPU(s)"
                        + (", RCCL all-gather of per-shard top-k keys + merge" if world > 1 else ""),
            "nb_total": args.nb_total, "nb_per_gpu": nb_local, "d": d, "k": k, "queries_per_step": nq,
            "parallelism": f"db-row-shard x{world}", "searches_in_flight": in_flight,
        },
    }
    if avg_scan_ms:
        achieved = alg_bytes / (avg_scan_ms * 1e-3) / 1e9
        traffic = None
        tfile = ROOT / "profiles" / "pmc_traffic.json"
        if tfile.exists():
            try:
                rec = json.loads(tfile.read_text()).get(info["kernel"], {})
                # the counters were collected on the default workload: only quote them for it
                if rec.get("algorithmic_bytes_per_launch") == alg_bytes:
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out["roofline"] = {
            "bound": "hbm", "kernel": info["kernel"], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "algorithmic_bytes_per_launch": alg_bytes, "avg_kernel_ms": avg_scan_ms,
            "launches_timed": len(scan_ms), "db_passes": passes, "grid": info["grid"],
            "mfma_tflops": 2.0 * nq * nb_local * d / (avg_scan_ms * 1e-3) / 1e12,
        }

    if world == 1 and not args.no_cpu and first_rows is not None:
        out["cpu_baseline"] = cpu_baseline(first_rows, q.cpu().numpy(), k, args.nb_total, D, I, index)
    if world == 1 and not args.no_batch:
        del index
        torch.cuda.empty_cache()
        out["batch"] = batch_config(dev, L, _lib, faiss)
    if use_pg:
        dist.destroy_process_group()
    return json.dumps(out)


def cpu_baseline(sample_rows, q_host, k, nb_total, D_gpu, I_gpu, index):
    """FAISS's blocked-sgemm flat search restated in numpy, on the first S database rows."""
    from oracle import knn_oracle as ko
    # threads actually used: numpy's BLAS pool, capped to this process's CPU affinity
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = avail
    limiter = None
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        limiter = threadpool_limits(limits=avail, user_api="blas")
        pools = [p["num_threads"] for p in threadpool_info() if p.get("user_api") == "blas"]
        if pools:
            cores = max(pools)
    except Exception:
        pass
    S = sample_rows.shape[0]
    nq = q_host.shape[0]
    t_budget, reps, t_used = 12.0, 0, 0.0
    while t_used < t_budget and reps < 50:
        t0 = time.perf_counter()
        # database blocks of 8192 rows instead of FAISS's 1024: amortises numpy's per-call overhead
        Dc, Ic = ko.faiss_flat_blas_restated(sample_rows, q_host, k, ko.METRIC_INNER_PRODUCT, bs_y=8192)
        t_used += time.perf_counter() - t0
        reps += 1
    t_pass = t_used / reps
    # and on one thread (the reference's own note on its flat search is "single core", pfam/slices/slices_search.py:10)
    single = None
    try:
        from threadpoolctl import threadpool_limits as _limits
        with _limits(limits=1, user_api="blas"):
            t0 = time.perf_counter()
            ko.faiss_flat_blas_restated(sample_rows[: max(1, S // 8)], q_host, k, ko.METRIC_INNER_PRODUCT, bs_y=8192)
            t1 = (time.perf_counter() - t0) * 8.0
        single = nq / (t1 * nb_total / S)
    except Exception:
        pass
    # same sample on the GPU: neighbours must agree (recall of the exact flat path)
    from knn_for_homology_amd import faiss
    sub = faiss.IndexFlat(sample_rows.shape[1], faiss.METRIC_INNER_PRODUCT)
    sub.set_tuning(64, 0, 0)  # another instantiation, so the benched kernel's rocprof stats stay clean
    sub.add(sample_rows)
    Dg, Ig = sub.search(q_host, k)
    recall = ko.recall_at_k(Ig, Ic)
    qps_full = nq / (t_pass * nb_total / S)
    return {"value": qps_full, "unit": "queries/s", "cores": cores, "kind": "port",
            "sample": f"numpy/OpenBLAS restatement of FAISS 1.7.2 knn_inner_product_blas (sgemm blocks of 4096 "
                      f"queries x 8192 rows + per-row top-k) on the first {S} of {nb_total} database rows, {nq} queries, {reps} passes of "
                      f"{t_pass:.3f}s; value extrapolates linearly to the full database",
            "seconds_per_pass_on_sample": t_pass, "gpu_recall_at_k_on_sample": recall, "single_thread_value": single}


def batch_config(dev, L, _lib, faiss):
    """BASELINE configs[1]: CATH20-sized all-vs-all, L2, k=300 (+1 self hit)."""
    n, d, k = 14433, 1024, 301
    x = torch.from_numpy(np.random.default_rng(20).standard_normal((n, d), dtype=np.float32)).to(dev)
    idx = faiss.IndexFlat(d, faiss.METRIC_L2)
    _lib.check(L.knn_flat_add_dev(idx._h, x.data_ptr(), n, None))
    D = torch.empty((n, k), device=dev, dtype=torch.float32)
    I = torch.empty((n, k), device=dev, dtype=torch.int64)
    times, scans = [], []
    for it in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(L.knn_flat_search_dev(idx._h, x.data_ptr(), n, k, D.data_ptr(), I.data_ptr(), None))
        torch.cuda.synchronize()
        if it:
            times.append(time.perf_counter() - t0)
            scans.append(idx.last_scan()["ms"])
    info = idx.last_scan()
    t = float(np.median(times))
    sm = float(np.median(scans))
    flops = 2.0 * n * n * d
    return {"workload": "BASELINE configs[1]: CATH20-sized 14433x1024 all-vs-all, L2, k=300 (+ self hit)",
            "value": n / t, "unit": "queries/s", "ms": 1e3 * t, "kernel": info["kernel"], "kernel_ms": sm,
            "roofline": {"bound": "mfma", "achieved": flops / (sm * 1e-3) / 1e12, "peak": FP32_MFMA_PEAK_TF,
                         "unit": "TFLOP/s", "frac": flops / (sm * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF}}


if __name__ == "__main__":
    main()
