#!/usr/bin/env python3
"""Do the scans of two searches in flight overlap on the device?  
  run:      rocprofv3 --kernel-trace -d <dir> -- python3 tools/shard_overlap.py run [lanes]
  analyse:  python3 tools/shard_overlap.py <dir>
The run: the 8-GPU-sized shard (1.25 M x 1024 rows, 32 queries, k = 100), 60 searches submitted back to back on `lanes`
lanes (default 2).  The analysis: per scan kernel its duration, how long it ran beside the previous scan and beside the
next one, and what else ran in between."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

if len(sys.argv) > 1 and sys.argv[1] == "run":
    import time
    import torch
    from knn_for_homology_amd import faiss, _lib
    from knn_for_homology_amd.sharded import ShardedFlatIndex
    lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    L = _lib.lib(); dev = torch.device("cuda:0")
    d, k, nb = 1024, 100, 1_250_000
    idx = ShardedFlatIndex(d, faiss.METRIC_INNER_PRODUCT, rank=0, world=1, row_offset=0)
    idx.reserve(nb)
    g = torch.Generator(device=dev); g.manual_seed(29)
    for i0 in range(0, nb, 250_000):
        x = torch.randn((250_000, d), generator=g, device=dev)
        _lib.check(L.knn_normalize_l2_dev(x.data_ptr(), 250_000, d, None))
        idx.add_dev(x)
    q = torch.randn((32, d), generator=g, device=dev)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(60):
            if lanes == 1:
                idx.backend._turn = 0
            pend = idx.submit(q, k)
        torch.cuda.synchronize()
        print(f"lanes {lanes}: {1e3 * (time.perf_counter() - t0) / 60:.4f} ms per search", flush=True)
    pend.result()
    sys.exit(0)

import csv, glob
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Stream_Id", "?")))
rows.sort()
scans = [i for i, r in enumerate(rows) if r[2].startswith("void flat_scan_kernel") or r[2].startswith("flat_scan_kernel")]
scans = scans[-50:]
print("scan   dur us | beside prev | start after prev start | others between (name dur, start offset)")
per = []
for a, b in zip(scans[:-1], scans[1:]):
    s0, e0 = rows[a][0], rows[a][1]
    s1, e1 = rows[b][0], rows[b][1]
    per.append((s1 - s0) / 1e3)
    others = [(rows[i][2][:28], (rows[i][1] - rows[i][0]) / 1e3, (rows[i][0] - s0) / 1e3, (rows[i][1] - s0) / 1e3) for i in range(a + 1, b)]
    print(f"{(e0 - s0) / 1e3:8.1f} | overlap with next {max(0, e0 - s1) / 1e3:6.1f} | next starts +{(s1 - s0) / 1e3:7.1f} | stream {rows[a][3]} | "
          + "; ".join(f"{n} {d:.1f} @{o:.1f}..{e:.1f}" for n, d, o, e in others))
per.sort()
print(f"scan start to scan start: median {per[len(per) // 2]:.1f} us, min {per[0]:.1f}, max {per[-1]:.1f}")
