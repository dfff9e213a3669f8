#!/usr/bin/env python3
"""Developer tool: per-workgroup timeline of one streaming scan launch (needs `make -C knn-for-homology_amd/csrc trace`
and KNN355_LIB=<repo>/knn-for-homology_amd/libknn355_trace.so).  Every workgroup stamps the 100-MHz wall clock at its
start, after each tile's K loop, after each tile's epilogue / compaction, and at its end.
usage: KNN355_LIB=... wg_timeline.py [rows] [flags]"""
import ctypes
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss, _lib  # noqa: E402

L = _lib.lib()
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(23)
q = torch.randn((32, 1024), generator=g, device=dev)
_lib.check(L.knn_normalize_l2_dev(q.data_ptr(), 32, 1024, None))
idx = faiss.IndexFlat(1024, 0)
_lib.check(L.knn_flat_reserve(idx._h, nb))
for i0 in range(0, nb, 500_000):
    m = min(500_000, nb - i0)
    x = torch.randn((m, 1024), generator=g, device=dev)
    _lib.check(L.knn_normalize_l2_dev(x.data_ptr(), m, 1024, None))
    _lib.check(L.knn_flat_add_dev(idx._h, x.data_ptr(), m, None))
    del x
idx.set_tuning(0, 0, flags)
D = torch.empty((32, 100), device=dev); I = torch.empty((32, 100), device=dev, dtype=torch.int64)
for _ in range(5):
    _lib.check(L.knn_flat_search_dev(idx._h, q.data_ptr(), 32, 100, D.data_ptr(), I.data_ptr(), None))
torch.cuda.synchronize()
L.knn_dev_trace_read.restype = ctypes.c_int
buf = np.zeros((2048, 128), np.uint64)
grid = L.knn_dev_trace_read(buf.ctypes.data_as(ctypes.c_void_p), 2048)
t = buf[:grid].astype(np.float64) / 100.0  # us
t0 = t[:, 0].min()
info = idx.last_scan()
print(f"rows {nb} flags {flags} grid {grid} kernel {info['kernel']} scan_ms {info['ms']:.4f} seed {idx.last_seed()}")
start, end = t[:, 0] - t0, t[:, 63] - t0
print(f"start: min {start.min():.1f} med {np.median(start):.1f} max {start.max():.1f} us;  end: min {end.min():.1f} med {np.median(end):.1f} max {end.max():.1f} us")
ntiles = int(((t[:, 1:63:2] > 0).sum(1)).max())
for ti in range(ntiles):
    k_end, t_end = t[:, 1 + 2 * ti], t[:, 2 + 2 * ti]
    ok = k_end > 0
    prev = t[:, 2 * ti] if ti > 0 else t[:, 0]
    kd, ed = (k_end - prev)[ok], (t_end - k_end)[ok]
    print(f"tile {ti:2d}: K loop med {np.median(kd):6.1f} (p10 {np.percentile(kd,10):6.1f} p90 {np.percentile(kd,90):6.1f} max {kd.max():6.1f}) us   "
          f"epilogue+compaction med {np.median(ed):5.1f} p90 {np.percentile(ed,90):5.1f} max {ed.max():6.1f} us   done at med {np.median(t_end[ok]-t0):7.1f} max {(t_end[ok]-t0).max():7.1f}")
last = np.array([t[i, 2 * int((t[i, 1:63:2] > 0).sum())] for i in range(grid)])
print(f"flush: med {np.median(t[:,63]-last):.1f} max {(t[:,63]-last).max():.1f} us;  workgroup life: med {np.median(end-start):.1f} min {(end-start).min():.1f} max {(end-start).max():.1f} us")

# where the workgroups ran: HW_ID bits cu_id[11:8] sh_id[12] se_id[15:13] (gfx9 layout), XCC_ID low bits
hw = buf[:grid, 62]
xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
hwid = (hw & np.uint64(0xFFFFFFFF)).astype(np.int64)
cu, sh, se = (hwid >> 8) & 0xF, (hwid >> 12) & 1, (hwid >> 13) & 7
place = xcc * 10000 + se * 1000 + sh * 100 + cu
life = end - start
print("life by XCC:", {int(x): (round(float(np.mean(life[xcc == x])), 1), int((xcc == x).sum())) for x in np.unique(xcc)})
from collections import Counter
cnt = Counter(place.tolist())
alone = np.array([cnt[int(pl)] == 1 for pl in place])
print(f"workgroups alone on their CU: {int(alone.sum())} (life med {np.median(life[alone]) if alone.any() else 0:.1f}); paired: {int((~alone).sum())} (life med {np.median(life[~alone]):.1f}, p90 {np.percentile(life[~alone],90):.1f}, max {life[~alone].max():.1f})")
pairs = {}
for i, pl in enumerate(place.tolist()):
    pairs.setdefault(pl, []).append(i)
diffs = [abs(life[v[0]] - life[v[1]]) for v in pairs.values() if len(v) == 2]
sums = [max(end[v[0]], end[v[1]]) for v in pairs.values() if len(v) == 2]
print(f"pairs on one CU: {len(diffs)}; |life difference| med {np.median(diffs):.1f} max {max(diffs):.1f}; the pair's later end: med {np.median(sums):.1f} p90 {np.percentile(sums,90):.1f} max {max(sums):.1f}")
slow = np.argsort(-life)[:12]
print("slowest:", [(int(i), round(float(life[i]), 1), int(place[i])) for i in slow])
print("distinct CUs used:", len(cnt), "by count:", Counter(cnt.values()))
