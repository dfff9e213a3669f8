#!/usr/bin/env python3
"""All of SURVEY.md section 8(d)'s input sets in one run on one MI355X:

  peaks   box-measured denominators: device read bandwidth (tools/micro/readbw), fp32 MFMA rate
          (tools/micro/mfmapeak) beside the data-sheet 8 TB/s / 157.3 TFLOP/s
  S-cath  14433 x 1024 all-vs-all, L2 and cosine, k = 11 and 301: search only (device-resident)
          and end to end through cath.search (host numpy in and out, PCIe included)
  S-pfam  200 000 x 1024 clustered (2000 centres + 0.35 noise, 0.5 % exact duplicate rows),
          all-vs-all cosine, k = 100 and 1000: scan, wall, host end to end; duplicates checked
  S-10M   10 M x 1024 generated on the device, cosine, k = 100, nq in {1, 8, 32, 1024, 10000}
  S-hnsw  S-pfam data, M in {32, 42}, efSearch 256, k = 100: build, search, recall@100 vs flat

Exactness is the GPU test-suite's job (tests/ hold the bit-exact comparisons); here the only result
checks are cheap invariants (duplicates adjacent, self hit first, recall of HNSW vs flat).
usage: measure_all.py [out.json] [sets ...]
"""
import json
import subprocess
import sys
import time
from pathlib import Path

import torch  # before the library: one shared HIP runtime
import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss, _lib  # noqa: E402
from knn_for_homology_amd.cath.search import search as cath_search  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
HBM_PEAK, MFMA_PEAK = 8000.0, 157.3
args = [a for a in sys.argv[1:]]
out_path = Path(args[0]) if args and args[0].endswith(".json") else None
sets = [a for a in args if not a.endswith(".json")] or ["peaks", "cath", "pfam", "10m", "hnsw"]
out = {"device": torch.cuda.get_device_name(0), "sets": {}}


def log(*a):
    print(*a, flush=True)


def dev_search(idx, q, k, reps=5):
    nq = q.shape[0]
    D = torch.empty((nq, k), device=dev, dtype=torch.float32)
    I = torch.empty((nq, k), device=dev, dtype=torch.int64)
    walls, scans = [], []
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(L.knn_flat_search_dev(idx._h, q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), None))
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
        scans.append(idx.last_scan()["ms"])
    info = idx.last_scan()
    wall, scan = float(np.median(walls[1:])), float(np.median(scans[1:]))
    seed = idx.last_seed()
    nb, d = idx.ntotal - seed["sample_rows"], idx.d  # rows the timed scan launch covers (a seed sample has its own small launch)
    passes = (nq + info["query_tile"] - 1) // info["query_tile"]
    if nq > 1024:
        # several launches (full 128-query tiles + the remainder on a narrower build): last_scan() describes the last one
        # only -- the whole search is the denominator
        scan, passes, info = 1e3 * wall, (nq + 127) // 128, dict(info, kernel="flat_scan_q128_d128 (+ the remainder's launch; whole search timed)")
    flops = 2.0 * nq * nb * d
    byts = passes * nb * d * 4 + nq * d * 4 + nq * k * 12
    return {"nq": nq, "nb": nb, "k": k, "kernel": info["kernel"], "grid": info["grid"], "scan_ms": scan, "wall_ms": 1e3 * wall,
            "qps_wall": nq / wall, "tflops_scan": flops / scan / 1e9, "gbps_alg_scan": byts / scan / 1e6, "db_passes": passes}, D, I


if "peaks" in sets:
    res = {"hbm_datasheet_gbps": HBM_PEAK, "mfma_fp32_datasheet_tflops": MFMA_PEAK}
    for name, cmd in (("readbw", [str(ROOT / "tools/micro/readbw"), "41"]), ("mfmapeak", [str(ROOT / "tools/micro/mfmapeak")])):
        try:
            txt = subprocess.run(cmd, capture_output=True, text=True, timeout=300).stdout
            res[name + "_output"] = txt.strip().splitlines()
        except Exception as e:  # binaries are built by `hipcc tools/micro/*.hip`; absent is not fatal
            res[name + "_output"] = [f"not run: {e}"]
    bw = [float(l.split()[-2]) for l in res.get("readbw_output", []) if "GB/s" in l]
    tf = [float(l.split("TFLOP/s")[0].split()[-1]) for l in res.get("mfmapeak_output", []) if "TFLOP/s" in l]
    res["hbm_read_box_gbps"] = max(bw) if bw else None
    res["mfma_fp32_box_tflops"] = max(tf) if tf else None
    out["sets"]["peaks"] = res
    log("peaks:", res["hbm_read_box_gbps"], "GB/s read,", res["mfma_fp32_box_tflops"], "TFLOP/s fp32 MFMA on this box")

if "cath" in sets:
    x = np.random.default_rng(20).standard_normal((14433, 1024), dtype=np.float32)
    rows = []
    for metric, name in ((faiss.METRIC_L2, "L2"), (faiss.METRIC_INNER_PRODUCT, "cosine")):
        xm = x.copy()
        if metric == faiss.METRIC_INNER_PRODUCT:
            faiss.normalize_L2(xm)
        idx = faiss.IndexFlat(1024, metric)
        idx.add(xm)
        xd = torch.from_numpy(xm).to(dev)
        for k in (11, 301):
            r, D, I = dev_search(idx, xd, k)
            assert (I[:, 0].cpu().numpy() == np.arange(14433)).all(), "self hit must come first"
            ts = []
            for _ in range(4):
                t0 = time.perf_counter()
                cath_search(x, hits=k - 1, metric=metric)
                ts.append(time.perf_counter() - t0)
            r.update(metric=name, frac_mfma_datasheet=r["tflops_scan"] / MFMA_PEAK, end_to_end_ms=1e3 * min(ts), qps_end_to_end=14433 / min(ts))
            rows.append(r)
            log(f"S-cath {name:6s} k={k:3d}: scan {r['scan_ms']:.3f} ms ({r['tflops_scan']:.1f} TFLOP/s, {100*r['frac_mfma_datasheet']:.1f} %), "
                f"search {r['wall_ms']:.3f} ms, cath.search end to end {r['end_to_end_ms']:.1f} ms")
        del idx, xd
    out["sets"]["S-cath"] = rows

if "pfam" in sets or "hnsw" in sets:
    n, d = 200_000, 1024
    cent = np.random.default_rng(21).standard_normal((2000, d)).astype(np.float32)
    rng = np.random.default_rng(22)
    lab = rng.integers(0, 2000, n)
    xp = cent[lab] + 0.35 * rng.standard_normal((n, d), dtype=np.float32)
    dup_dst = rng.choice(n, n // 200, replace=False)  # 0.5 % exact duplicates
    dup_src = (dup_dst + 1 + rng.integers(0, n - 1, dup_dst.size)) % n
    keep = ~np.isin(dup_src, dup_dst)
    dup_dst, dup_src = dup_dst[keep], dup_src[keep]
    xp[dup_dst] = xp[dup_src]
    faiss.normalize_L2(xp)

if "pfam" in sets:
    rows = []
    idx = faiss.IndexFlat(d, faiss.METRIC_INNER_PRODUCT)
    idx.add(xp)
    q = torch.from_numpy(xp[:16384]).to(dev)
    for k in (100, 1000):
        r, D, I = dev_search(idx, q, k, reps=3)
        t0 = time.perf_counter()
        Dh, Ih = idx.search_self(k)
        t_self = time.perf_counter() - t0
        # (twice: the second request of a result size class above 64 MB page-locks its block -- 0.2 s for 2.4 GB, once per
        # process -- and the third finds it in the pool: _lib.result_array)
        t_hosts = []
        for _ in range(3):
            t0 = time.perf_counter()
            Dh2, Ih2 = idx.search(xp, k)
            t_hosts.append(time.perf_counter() - t0)
            assert np.array_equal(Ih, Ih2) and np.array_equal(Dh, Dh2)
            del Dh2, Ih2
        t_host = min(t_hosts)
        # a duplicated row and its source tie on every score: they sit next to each other, lower id first
        a, b = np.minimum(dup_dst, dup_src), np.maximum(dup_dst, dup_src)
        first_two = np.sort(Ih[a, :2], axis=1)
        ok_dup = float(np.mean((first_two[:, 0] == a) & (first_two[:, 1] == b)))
        r.update(frac_mfma_datasheet=r["tflops_scan"] / MFMA_PEAK, all_vs_all_self_s=t_self, all_vs_all_host_s=t_host, all_vs_all_host_calls_s=[round(t, 4) for t in t_hosts],
                 qps_all_vs_all_host=n / t_host, duplicate_pairs=int(a.size), duplicate_pairs_adjacent_lower_id_first=ok_dup)
        rows.append(r)
        log(f"S-pfam k={k:4d}: 16384-query batch scan {r['scan_ms']:.2f} ms ({r['tflops_scan']:.1f} TFLOP/s, {100*r['frac_mfma_datasheet']:.1f} %); "
            f"200k x 200k host end to end {t_host:.3f} s (search_self {t_self:.3f} s); duplicate pairs in place: {100*ok_dup:.1f} %")
    out["sets"]["S-pfam"] = rows
    del idx, q

if "hnsw" in sets:
    rows = []
    flat = faiss.IndexFlat(d, faiss.METRIC_INNER_PRODUCT)
    flat.add(xp)
    nqh = 20000
    Dt, It = flat.search(xp[:nqh], 100)
    for M in (32, 42):
        idx = faiss.IndexHNSWFlat(d, M, faiss.METRIC_INNER_PRODUCT)
        t0 = time.perf_counter()
        idx.add(xp)
        tb = time.perf_counter() - t0
        idx.hnsw.efSearch = 256
        t0 = time.perf_counter()
        idx.search(xp[:nqh], 100)  # the first search uploads the level-0 lists and builds the coarse index
        t_first = time.perf_counter() - t0
        t0 = time.perf_counter()
        Dh, Ih = idx.search(xp[:nqh], 100)
        ts = time.perf_counter() - t0
        rec = float(np.mean([len(np.intersect1d(a[a >= 0], b)) for a, b in zip(Ih, It)])) / 100.0
        rows.append({"M": M, "efSearch": 256, "k": 100, "n": n, "build_s": tb, "nq": nqh, "first_search_s": t_first, "search_s": ts, "qps": nqh / ts, "recall_at_100_vs_flat": rec})
        log(f"S-hnsw M={M}: build {tb:.2f} s, {nqh} queries {ts:.3f} s ({nqh/ts:.0f} q/s), recall@100 vs flat {rec:.4f}")
        del idx
    out["sets"]["S-hnsw"] = rows
    del flat

if "10m" in sets:
    nb = 10_000_000
    idx = faiss.IndexFlat(1024, faiss.METRIC_INNER_PRODUCT)
    _lib.check(L.knn_flat_reserve(idx._h, nb))
    g = torch.Generator(device=dev)
    g.manual_seed(23)
    for i0 in range(0, nb, 1 << 20):
        m = min(1 << 20, nb - i0)
        x = torch.randn((m, 1024), generator=g, device=dev)
        _lib.check(L.knn_normalize_l2_dev(x.data_ptr(), m, 1024, None))
        torch.cuda.synchronize()
        _lib.check(L.knn_flat_add_dev(idx._h, x.data_ptr(), m, None))
        del x
    qh = np.random.default_rng(24).standard_normal((10000, 1024), dtype=np.float32)
    faiss.normalize_L2(qh)
    qd = torch.from_numpy(qh).to(dev)
    rows = []
    for nq in (1, 8, 32, 1024, 10000):
        r, D, I = dev_search(idx, qd[:nq].contiguous(), 100, reps=5 if nq <= 1024 else 2)
        r.update(frac_hbm_datasheet=r["gbps_alg_scan"] / HBM_PEAK, frac_mfma_datasheet=r["tflops_scan"] / MFMA_PEAK)
        rows.append(r)
        log(f"S-10M nq={nq:5d}: scan {r['scan_ms']:.3f} ms, search {r['wall_ms']:.3f} ms -> {r['qps_wall']:.0f} q/s; {r['gbps_alg_scan']:.0f} GB/s algorithmic "
            f"({100*r['frac_hbm_datasheet']:.1f} % of 8 TB/s), {r['tflops_scan']:.1f} TFLOP/s ({100*r['frac_mfma_datasheet']:.1f} %), {r['db_passes']} pass(es)")
    out["sets"]["S-10M"] = rows
    del idx

if out_path:
    out_path.parent.mkdir(parents=True, exist_ok=True)
    out_path.write_text(json.dumps(out, indent=1))
    log("wrote", out_path)
