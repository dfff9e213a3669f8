#!/usr/bin/env python3
"""Per-kernel timeline of the LAST hnsw search in a rocprofv3 kernel trace (tools/hnsw_beam_probe.py run under it).
usage: hnsw_timeline.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
beams = [i for i, r in enumerate(rows) if "hnsw_beam_kernel" in r[2]]
last = beams[-1]
# walk back to the coarse scan of that search
i0 = last
while i0 > 0 and "flat_scan_kernel" not in rows[i0][2]:
    i0 -= 1
i0 = max(0, i0 - 3)
t0 = rows[i0][0]
prev = None
for s, e, n in rows[i0:last + 4]:
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"+{(s - t0) / 1e3:9.1f} us  gap {gap:8.1f}  dur {(e - s) / 1e3:9.1f}  {n.split('(')[0][:80]}")
    prev = e
