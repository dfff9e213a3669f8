#!/usr/bin/env python3
"""One CATH-sized symmetric self-search (14433 x 1024 clustered, k = 301), device-resident, a few times -- for a kernel trace:
  rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/cath_once.py [metric] ;  python3 tools/cath_once.py <dir>
prints the last search's kernels in launch order with the idle gaps between them."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
if len(sys.argv) > 1 and not sys.argv[1].lstrip("-").isdigit():
    import csv, glob
    rows = []
    for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:90], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?")))
    rows.sort()
    last = max(i for i, r in enumerate(rows) if "select_topk" in r[2])
    # the last search: back from its final selection to the sample pass's scan (two scans back)
    scans = [i for i in range(last) if "flat_scan_kernel" in rows[i][2]]
    a = scans[-2]
    t0 = rows[a][0]
    prev = None
    for s, e, n, g in rows[a:last + 1]:
        print(f"+{(s - t0) / 1e3:8.1f} us  gap {((s - prev) / 1e3 if prev else 0):6.1f}  dur {(e - s) / 1e3:8.1f}  grid {g:>8}  {n}")
        prev = e
    print(f"whole search: {(rows[last][1] - t0) / 1e3:.1f} us")
    sys.exit(0)
import torch
from knn_for_homology_amd import faiss, _lib
L = _lib.lib(); dev = torch.device("cuda:0")
metric = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n, d, k = 14433, 1024, 301
g = torch.Generator(device=dev); g.manual_seed(20)
x = torch.randn((n, d), generator=g, device=dev)
idx = faiss.IndexFlat(d, metric)
_lib.check(L.knn_flat_add_dev(idx._h, x.data_ptr(), n, None))
D = torch.empty((n, k), device=dev, dtype=torch.float32); I = torch.empty((n, k), device=dev, dtype=torch.int64)
for rep in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _lib.check(L.knn_flat_search_self_dev(idx._h, k, D.data_ptr(), I.data_ptr()))
    torch.cuda.synchronize(); print(f"metric {metric}: {1e3 * (time.perf_counter() - t0):.3f} ms  {idx.last_scan()['kernel']}", flush=True)
