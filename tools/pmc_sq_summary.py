#!/usr/bin/env python3
"""Summarises a rocprofv3 --pmc (SQ_* + GRBM_GUI_ACTIVE) counter_collection.csv for the scan
kernels: effective clock, MFMA pipe utilisation and where wave time goes.
usage: pmc_sq_summary.py <counter_collection.csv> [min_ms]"""
import csv
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
path = sys.argv[1]
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
disp = defaultdict(dict)
meta = {}
with open(path) as fh:
    for r in csv.DictReader(fh):
        if "flat_scan_kernel" not in r["Kernel_Name"]:
            continue
        d = int(r["Dispatch_Id"])
        disp[d][r["Counter_Name"]] = float(r["Counter_Value"])
        nm = r["Kernel_Name"]
        meta[d] = (nm[nm.index("flat_scan_kernel"):].split("(")[0], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size"]) // 256)
NUM_CU, SIMD = 256, 4
for d in sorted(disp):
    name, t0, t1, grid = meta[d]
    ms = (t1 - t0) / 1e6
    if ms < min_ms:
        continue
    c = disp[d]
    clk = c.get("GRBM_GUI_ACTIVE", 0) / 8 / (ms * 1e3)  # MHz; sum over 8 XCDs
    wc = c.get("SQ_WAVE_CYCLES", 0)
    line = f"{name:42s} grid={grid:5d} {ms:8.3f} ms  clk~{clk:5.0f} MHz"
    if wc:
        line += (f"  wait_any {c.get('SQ_WAIT_ANY',0)/wc:5.1%}  wait_inst {c.get('SQ_WAIT_INST_ANY',0)/wc:5.1%}"
                 f"  (lds {c.get('SQ_WAIT_INST_LDS',0)/wc:5.1%})  active {c.get('SQ_ACTIVE_INST_ANY',0)/wc:5.1%}")
    mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
    busy = c.get("SQ_BUSY_CU_CYCLES", 0)
    if mf:
        # MFMA_BUSY counts cycles per SIMD pipe; chip-wide capacity = cycles x CUs x 4 SIMDs
        cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8
        line += f"  mfma_busy/(cyc*CUs*4) {mf/(cyc*NUM_CU*SIMD):5.1%}  busy_cu/(cyc*CUs) {busy/(cyc*NUM_CU):5.2f}"
    print(line)
