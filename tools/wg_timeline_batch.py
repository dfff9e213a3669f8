#!/usr/bin/env python3
"""Developer tool: per-workgroup stamps of ONE batch launch (16384 queries x nb clustered rows, cosine), the 256 x 256 tile
(flags 524288) beside the 128 x 128 one (flags 262144): K loop and epilogue time per tile, flush, workgroup life.
Needs `make -C knn-for-homology_amd/csrc trace` and KNN355_LIB=<repo>/knn-for-homology_amd/libknn355_trace.so.
usage: KNN355_LIB=... [NCH=chunks] wg_timeline_batch.py [rows] [k] [flags ...]   (stamps hold 30 tiles per workgroup: NCH=27 for 200 000 rows)"""
import ctypes
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from knn_for_homology_amd import faiss, _lib  # noqa: E402

L = _lib.lib()
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 61440
k = int(sys.argv[2]) if len(sys.argv) > 2 else 100
FLAGS = [int(a) for a in sys.argv[3:]] or [524288, 262144]
nq, d = 16384, 1024
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(21)
cent = torch.randn((2000, d), generator=g, device=dev)
x = cent[torch.randint(0, 2000, (nb,), generator=g, device=dev)] + 0.35 * torch.randn((nb, d), generator=g, device=dev)
_lib.check(L.knn_normalize_l2_dev(x.data_ptr(), nb, d, None))
q = cent[torch.randint(0, 2000, (nq,), generator=g, device=dev)] + 0.35 * torch.randn((nq, d), generator=g, device=dev)
_lib.check(L.knn_normalize_l2_dev(q.data_ptr(), nq, d, None))
idx = faiss.IndexFlat(d, 0)
_lib.check(L.knn_flat_add_dev(idx._h, x.data_ptr(), nb, None))
D = torch.empty((nq, k), device=dev); I = torch.empty((nq, k), device=dev, dtype=torch.int64)
L.knn_dev_trace_read.restype = ctypes.c_int
for flags in FLAGS:
    idx.set_tuning(0, int(os.environ.get('NCH', '0')), flags)
    for _ in range(4):
        _lib.check(L.knn_flat_search_dev(idx._h, q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), None))
    torch.cuda.synchronize()
    buf = np.zeros((4096, 128), np.uint64)
    grid = L.knn_dev_trace_read(buf.ctypes.data_as(ctypes.c_void_p), 4096)
    t = buf[:grid].astype(np.float64) / 100.0  # us
    t0 = t[:, 0].min()
    info = idx.last_scan()
    print(f"== flags {flags} rows {nb} k {k}: grid {grid} kernel {info['kernel']} chunks {info['nchunks']} scan_ms {info['ms']:.4f} seed {idx.last_seed()}")
    start, end = t[:, 0] - t0, t[:, 63] - t0
    print(f"start: min {start.min():.1f} med {np.median(start):.1f} max {start.max():.1f} us;  end: min {end.min():.1f} med {np.median(end):.1f} max {end.max():.1f} us")
    ntiles = min(31, int(((t[:, 1:63:2] > 0).sum(1)).max()))
    ksum = esum = 0.0
    for ti in range(ntiles):
        k_end, f_end, t_end = t[:, 1 + 2 * ti], t[:, 64 + 2 * ti], t[:, 2 + 2 * ti]
        ok = (k_end > 0) & (t_end > 0) & (f_end > 0)
        prev = t[:, 2 * ti] if ti > 0 else t[:, 0]
        kd, fd, cd = (k_end - prev)[ok], (f_end - k_end)[ok], (t_end - f_end)[ok]
        ksum += np.median(kd); esum += np.median(fd) + np.median(cd)
        if ti < 6 or ti % 5 == 0 or ti == ntiles - 1:
            print(f"tile {ti:2d}: prologue+K loop med {np.median(kd):6.1f} (p10 {np.percentile(kd,10):6.1f} p90 {np.percentile(kd,90):6.1f}) us   filter med {np.median(fd):5.1f} p90 {np.percentile(fd,90):5.1f}   "
                  f"barrier+compaction med {np.median(cd):6.1f} p90 {np.percentile(cd,90):6.1f} max {cd.max():7.1f}")
    # (256 x 256 builds, chunks of at most 30 tiles) inside the sparse epilogue of tile 5: slots 124.. = start, row norms / thresholds in registers, pass 1, pass 2
    if (t[:, 124] > 0).any():
        ok = t[:, 124] > 0
        names = ["K loop end -> epilogue start", "preloads", "pass 1", "pass 2"]
        prev = t[:, 1 + 2 * 5]
        parts = []
        for j, nm in enumerate(names):
            cur = t[:, 124 + j]
            if (cur[ok] > 0).all():
                parts.append(f"{nm} {np.median((cur - prev)[ok]):.2f}")
                prev = cur
        print("tile 5 epilogue (us, medians): " + ", ".join(parts))
    last = np.array([t[i, 2 * min(31, int((t[i, 1:63:2] > 0).sum()))] for i in range(grid)])
    life = end - start
    print(f"sum over {ntiles} tiles of medians: K loops {ksum:.1f} us, epilogues {esum:.1f} us ({100 * esum / (ksum + esum):.1f} %);  flush(+tiles beyond 31): med {np.median(t[:,63]-last):.1f} max {(t[:,63]-last).max():.1f};  "
          f"workgroup life med {np.median(life):.1f} min {life.min():.1f} max {life.max():.1f} us")
