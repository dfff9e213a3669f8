#!/usr/bin/env python3
"""Reads a rocprofv3 kernel-trace CSV of `bench.py --nb-total <per-shard rows>` and prints the
per-step timeline: kernel durations in launch order and the idle gaps between them.
usage: shard_timeline.py <dir with *_kernel_trace.csv> [steps to show]"""
import csv
import glob
import sys

d = sys.argv[1]
show = int(sys.argv[2]) if len(sys.argv) > 2 else 2
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# a step starts at each main-pass scan launch
names = [r[2] for r in rows]
short = lambda n: n.split("(")[0][:70]
idx = [i for i, n in enumerate(names) if "flat_scan_kernel" in n and ", true>" not in n.replace("1, true", "")]
def _is_main(n):
    if "flat_scan_kernel<" not in n:
        return False
    targs = n[n.index("flat_scan_kernel<") + len("flat_scan_kernel<"):].split(">")[0].split(", ")
    if len(targs) > 7 and targs[7] == "true":  # (bf16 coarse scan of the bench's HNSW leg)
        return False
    return not (len(targs) > 6 and targs[6] == "true")


main = [i for i, n in enumerate(names) if _is_main(n)]
# the streaming steps only: the longest run of scans of one instantiation
from itertools import groupby
runs, pos = [], 0
for key, grp in groupby(main, key=lambda i: names[i]):
    g = list(grp)
    runs.append(g)
main = max(runs, key=len)
if len(main) < show + 3:
    print("too few steps", len(main)); sys.exit(1)
# take steps from the tail (steady state)
per = []
for a, b in zip(main[-show - 1:-1], main[-show:]):
    t0 = rows[a][0]
    print(f"--- step: {(rows[b][0]-t0)/1e3:.1f} us scan-start to scan-start")
    prev_end = None
    for i in range(a, b):
        s, e, n = rows[i]
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        print(f"   +{(s-t0)/1e3:8.1f} us  gap {gap:7.1f}  dur {(e-s)/1e3:8.1f}  {short(n)}")
        prev_end = e
steps = [(rows[b][0] - rows[a][0]) / 1e3 for a, b in zip(main[len(main)//2:-1], main[len(main)//2 + 1:])]
print(f"steady-state step period: mean {sum(steps)/len(steps):.1f} us over {len(steps)} steps")
