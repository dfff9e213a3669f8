#!/usr/bin/env python3
"""Average duration per (kernel, grid) from a rocprofv3 kernel-trace CSV.
usage: kernel_avg.py <dir or csv>"""
import csv
import glob
import os
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
src = sys.argv[1]
files = [src] if os.path.isfile(src) else glob.glob(src + "/**/*kernel_trace.csv", recursive=True)
acc = defaultdict(list)
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            nm = r["Kernel_Name"]
            for key in ("flat_scan_kernel", "merge_select_kernel", "merge_sort_kernel", "seed_thresholds_kernel"):
                if key in nm:
                    nm = nm[nm.index(key):].split("(")[0]
                    break
            else:
                nm = nm.split("(")[0][-60:]
            acc[(nm, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (nm, grid), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"{nm:58s} grid={grid:6d} n={len(v):5d} avg={sum(v)/len(v):9.1f} us  med={v2[len(v2)//2]:9.1f} us")
