#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/collect_profiles.sh (gpurun_out/) into the
committed summaries under profiles/: kernel stats CSV, per-kernel HBM traffic from the
PMC passes (FETCH_SIZE doubled for wide streaming reads on gfx950 as
MI355X_MICROARCH.md section HBM prescribes; units are KiB), and pmc_traffic.json which
bench.py reads for roofline.traffic.  usage: summarize_profiles.py <round-tag>"""
import collections
import csv
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = ROOT / "profiles"
out.mkdir(exist_ok=True)
go = ROOT / "gpurun_out"

stats = go / f"prof_{tag}" / f"{tag}_kernel_stats.csv"
shutil.copyfile(stats, out / f"{tag}_bench_kernel_stats.csv")
bench_json = go / f"bench_prof_{tag}.json"
if bench_json.exists():
    shutil.copyfile(bench_json, out / f"{tag}_bench_under_rocprof.json")


def short(name):
    # flat_scan_kernel<WM, WN, TM, TN, L2, NTDB, SYM>: a seed sample's pass is a small launch of the same instantiation;
    # the metric is part of the name (the bench's main config is inner product, its batch config L2)
    if "flat_scan_kernel<" in name:
        targs = name[name.index("flat_scan_kernel<") + len("flat_scan_kernel<"):].split(">")[0].split(", ")
        tile = {"4, 1, 2, 1": "flat_scan_q32_d256", "2, 2, 2, 1": "flat_scan_q64_d128", "2, 2, 2, 2": "flat_scan_q128_d128", "2, 2, 4, 4": "flat_scan_q256_d256"}[", ".join(targs[:4])]
        if len(targs) > 9 and targs[9] != "0":  # builds on 16-query blocks: 4 x 1 waves -> 48 queries, 2 x 2 waves -> 96
            tile = "flat_scan_q48_d256" if targs[0] == "4" else "flat_scan_q96_d128"
        if len(targs) > 8 and targs[8] != "0":  # difference builds (squared L2, fewer than 20 queries)
            tile = f"flat_scan_q32_d256_l2diff{targs[8]}"
        if len(targs) > 7 and targs[7] == "true":  # bf16 operands (HNSW's coarse entry scan)
            tile += "_bf16"
        return tile + ("_l2" if targs[4] == "true" else "_ip") + ("_sym" if len(targs) > 6 and targs[6] == "true" else "")
    return name.split("(")[0].replace("void ", "")[:60]


def by_grid(trace_csv, dest):
    """per (kernel, workgroups) rows of the kernel trace: a seed sample's pass and the main scan are launches of
    the SAME instantiation with different grids -- rocprofv3's --stats blends them into one average.  This is the
    table the headline roofline fraction is recomputable from: algorithmic bytes / mean_ms of the main grid."""
    import statistics
    groups = collections.defaultdict(list)
    for r in csv.DictReader(open(trace_csv)):
        name = r["Kernel_Name"]
        if not any(t in name for t in ("flat_scan_kernel", "select_topk", "init_level", "hamming_scan", "hnsw_beam", "beam_keys",
                                        "segment", "pair_", "l2diff")):
            continue
        wg = max(1, int(r["Workgroup_Size_X"]))
        groups[(short(name), name.split("(")[0].replace("void ", ""), int(r["Grid_Size_X"]) // wg, wg)].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    # One (kernel, grid) serves databases of different sizes in one bench run -- the 10 M-row step, the 1.25 M-row shard unit and
    # the sweep all launch the streaming build on 2 x CUs workgroups: launches whose durations lie a factor 2.5 apart are
    # different work and get a row each (duration_class 0 = the longest)
    split = {}
    for key, ts in groups.items():
        srt = sorted(ts, reverse=True)
        cls, cur = 0, [srt[0]]
        for t in srt[1:]:
            if t * 2.5 < cur[-1]:
                split[key + (cls,)] = cur
                cls, cur = cls + 1, []
            cur.append(t)
        split[key + (cls,)] = cur
    with open(dest, "w") as fh:
        fh.write("kernel,instantiation,workgroups,workgroup_size,duration_class,launches,mean_ms,median_ms,min_ms,max_ms,total_ms\n")
        for (sh, full, grid, wg, cls), ts in sorted(split.items(), key=lambda kv: -sum(kv[1])):
            fh.write(f'{sh},"{full}",{grid},{wg},{cls},{len(ts)},{statistics.mean(ts):.4f},{statistics.median(ts):.4f},{min(ts):.4f},{max(ts):.4f},{sum(ts):.3f}\n')
    return split


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


trace = go / f"prof_{tag}" / f"{tag}_kernel_trace.csv"
if trace.exists():
    groups = by_grid(trace, out / f"{tag}_bench_kernel_by_grid.csv")
    if bench_json.exists():
        try:
            b = json.loads(bench_json.read_text().strip().splitlines()[-1])
            rf = b["roofline"]
            for (sh, full, grid, wg, cls), ts in groups.items():
                if sh == rf["kernel"] + "_ip" and grid == rf["grid"] and cls == 0: # (class 0: the launches over the whole database)
                    import statistics
                    m = statistics.mean(ts)
                    print(f"headline check: {rf['algorithmic_bytes_per_launch']} B / mean {m:.4f} ms of the {len(ts)} longest grid-{grid} launches "
                          f"= {rf['algorithmic_bytes_per_launch'] / m / 1e6:.1f} GB/s = {rf['algorithmic_bytes_per_launch'] / m / 1e6 / 8000:.4f} of 8 TB/s "
                          f"(the line under rocprofv3 says {rf['frac']:.4f} from {rf['avg_kernel_ms']:.4f} ms)")
        except Exception as e:  # pragma: no cover
            print("headline check failed:", e)

fetch = per_kernel(go / f"pmc_fetch_{tag}" / "f_counter_collection.csv", "FETCH_SIZE")
write = per_kernel(go / f"pmc_write_{tag}" / "w_counter_collection.csv", "WRITE_SIZE")
traffic = {}
rows = []
for kname in sorted(set(fetch) | set(write)):
    f = fetch.get(kname, [])
    w = write.get(kname, [])
    # per launch: the bench's timed launches are the largest ones of that kernel
    fmax = max(f) if f else 0.0
    wmax = max(w) if w else 0.0
    hbm = (2.0 * fmax + wmax) * 1024.0
    rows.append((kname, len(f), fmax, wmax, hbm))
    if kname.startswith("flat_scan") or kname.startswith("select_topk"):
        traffic[kname] = {"hbm_bytes_per_launch": hbm, "fetch_size_kib_raw": fmax, "write_size_kib": wmax,
                          "note": "FETCH_SIZE x2 (gfx950 wide-read correction) + WRITE_SIZE, largest launch"}
with open(out / f"{tag}_pmc_hbm_traffic.csv", "w") as fh:
    fh.write("kernel,launches,FETCH_SIZE_KiB_raw_max,WRITE_SIZE_KiB_max,hbm_bytes_corrected\n")
    for r in rows:
        fh.write(",".join(str(x) for x in r) + "\n")
if bench_json.exists():
    try:
        b = json.loads(bench_json.read_text().strip().splitlines()[-1])
        kern = b["roofline"]["kernel"] + "_ip"
        if kern in traffic:
            traffic[kern]["algorithmic_bytes_per_launch"] = b["roofline"]["algorithmic_bytes_per_launch"]
            traffic[kern]["workload"] = b["config"]["workload"]
    except Exception as e:  # pragma: no cover
        print("could not attach workload to traffic:", e)
    try:
        bk = b["batch"]
        kern = bk["kernel"] + "_l2"
        if kern in traffic:
            n = 14433
            traffic[kern]["workload"] = bk["workload"]
            traffic[kern]["result_bytes"] = n * 301 * 12
            traffic[kern]["database_bytes"] = n * 1024 * 4
    except Exception as e:  # pragma: no cover
        print("could not attach the batch workload:", e)
# the counters belong to the kernels of ONE revision of the source: bench.py quotes them only while csrc/knn355.hip still
# hashes to this (a kernel change that keeps the byte count kept quoting stale counters)
import hashlib
traffic["_meta"] = {"knn355_hip_sha256": hashlib.sha256((ROOT / "knn-for-homology_amd" / "csrc" / "knn355.hip").read_bytes()).hexdigest(),
                    "round": tag, "collected_by": "tools/collect_profiles.sh + tools/summarize_profiles.py"}
(out / "pmc_traffic.json").write_text(json.dumps(traffic, indent=1))
print(open(out / f"{tag}_bench_kernel_stats.csv").read()[:1500])
print(json.dumps(traffic, indent=1))
