#!/usr/bin/env python3
"""Developer tool (trace build, see wg_timeline.py): per-workgroup stamps of one launch of a CATH20-sized plain search
(14433 x 1024, L2 k=301): KNN355_TRACE_LEVEL=1 stamps the sample pass, 0 the main scan."""
import ctypes, os, sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss, _lib
L = _lib.lib(); dev = torch.device("cuda:0")
n, d, k = 14433, 1024, 301
g = torch.Generator(device=dev); g.manual_seed(20)
x = torch.randn((n, d), generator=g, device=dev)
idx = faiss.IndexFlat(d, 1)
_lib.check(L.knn_flat_add_dev(idx._h, x.data_ptr(), n, None))
D = torch.empty((n, k), device=dev, dtype=torch.float32); I = torch.empty((n, k), device=dev, dtype=torch.int64)
SELF = len(sys.argv) > 1 and sys.argv[1] == "self"  # the symmetric whole-index self-search instead of the plain search
for _ in range(4):
    if SELF:
        _lib.check(L.knn_flat_search_self_dev(idx._h, k, D.data_ptr(), I.data_ptr()))
    else:
        _lib.check(L.knn_flat_search_dev(idx._h, x.data_ptr(), n, k, D.data_ptr(), I.data_ptr(), None))
torch.cuda.synchronize()
L.knn_dev_trace_read.restype = ctypes.c_int
buf = np.zeros((4096, 128), np.uint64)
grid = L.knn_dev_trace_read(buf.ctypes.data_as(ctypes.c_void_p), 4096)
counts = buf[:grid, 57:61].copy()  # (developer counters of the sparse epilogue, not stamps: see the end)
buf[:grid, 57:61] = 0
t = buf[:grid].astype(np.float64) / 100.0
t0 = t[:, 0][t[:, 0] > 0].min()
print(f"level {os.environ.get('KNN355_TRACE_LEVEL', '0')} grid {grid} {'self-search' if SELF else 'plain'} last scan {idx.last_scan()}")
start, end = t[:, 0] - t0, t[:, 63] - t0
print(f"start: min {start.min():.1f} med {np.median(start):.1f} p90 {np.percentile(start,90):.1f} max {start.max():.1f};  end: min {end.min():.1f} med {np.median(end):.1f} max {end.max():.1f} us")
ntiles = int(((t[:, 1:63:2] > 0).sum(1)).max())
for ti in range(min(ntiles, 14)):
    k_end, t_end = t[:, 1 + 2 * ti], t[:, 2 + 2 * ti]
    ok = k_end > 0
    prev = t[:, 2 * ti] if ti > 0 else t[:, 0]
    kd, ed = (k_end - prev)[ok], (t_end - k_end)[ok]
    f_end, b_end = t[:, 64 + 2 * ti], t[:, 65 + 2 * ti]
    print(f"        filter med {np.median((f_end - k_end)[ok]):6.1f} p90 {np.percentile((f_end - k_end)[ok],90):6.1f} | barrier wait med {np.median((b_end - f_end)[ok]):6.1f} p90 {np.percentile((b_end - f_end)[ok],90):6.1f} | compaction med {np.median((t_end - b_end)[ok]):6.1f} p90 {np.percentile((t_end - b_end)[ok],90):6.1f}")
    print(f"tile {ti:2d} ({int(ok.sum())} wgs): K loop med {np.median(kd):6.1f} p10 {np.percentile(kd,10):6.1f} p90 {np.percentile(kd,90):6.1f} max {kd.max():6.1f}   epilogue+compaction med {np.median(ed):6.1f} p90 {np.percentile(ed,90):6.1f} max {ed.max():6.1f}")
last = np.array([t[i, 2 * int((t[i, 1:63:2] > 0).sum())] for i in range(grid)])
print(f"flush: med {np.median(t[:,63]-last):.1f} max {(t[:,63]-last).max():.1f};  life med {np.median(end-start):.1f} max {(end-start).max():.1f}")

print(f"sparse epilogue (256 x 256 and 128 x 128 batch builds): tiles wave 0 filtered the dense way: {int(counts[:, 3].sum())}, tiles whose second direction went dense: {int(counts[:, 2].sum())}, "
      f"of {int((t[:, 1:63:2] > 0).sum())} tiles; largest per-lane survivor count of a workgroup, first direction: med {np.median(counts[:, 1]):.0f} max {int(counts[:, 1].max())}, second: med {np.median(counts[:, 0]):.0f} max {int(counts[:, 0].max())}")
# inside the sparse epilogue of tile 5 (slots 124.. = start, preloads, first direction's pass 1, its pass 2; then the "filter" stamp)
if (t[:, 124] > 0).any():
    ok = (t[:, 124] > 0) & (t[:, 64 + 10] > 0) & (t[:, 120] > 0) & (t[:, 122] > 0)
    seq = [("K loop end -> start (barrier)", t[:, 124] - t[:, 1 + 10]), ("preloads", t[:, 125] - t[:, 124]), ("pass 1, first direction", t[:, 126] - t[:, 125]),
           ("pass 2, first direction", t[:, 127] - t[:, 126]), ("second direction: pass 1 + barrier", t[:, 120] - t[:, 127]), ("ranks + barrier", t[:, 121] - t[:, 120]), ("global reservations + barrier", t[:, 122] - t[:, 121]), ("stores", t[:, 64 + 10] - t[:, 122])]
    print("tile 5 epilogue, medians / p90 in us: " + "; ".join(f"{nm} {np.median(v[ok]):.1f} / {np.percentile(v[ok], 90):.1f}" for nm, v in seq))
