#!/usr/bin/env python3
"""Vendor ceiling for the batch regime: the plain fp32 GEMM (torch.mm -> hipBLASLt/rocBLAS, no selection, scores
written to HBM) at the shapes the fused scan multiplies, next to the fused kernel's own rate.  PyTorch here is a
yardstick only -- nothing in the product calls it."""
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
dev = torch.device("cuda:0")
torch.backends.cuda.matmul.allow_tf32 = False
out = {}
for name, nq, nb in (("cath 14433x14433", 14433, 14433), ("pfam pass 16384x200000", 16384, 200000), ("aligned 16384x16384", 16384, 16384)):
    d = 1024
    a = torch.randn((nq, d), device=dev)
    b = torch.randn((nb, d), device=dev)
    c = torch.empty((nq, nb), device=dev)
    for _ in range(3):
        torch.mm(a, b.t(), out=c)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        torch.mm(a, b.t(), out=c)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    ms = ts[len(ts) // 2]
    out[name] = {"ms": ms, "tflops": 2.0 * nq * nb * d / (ms * 1e-3) / 1e12, "frac_of_157.3": 2.0 * nq * nb * d / (ms * 1e-3) / 1e12 / 157.3}
    print(name, out[name], flush=True)
    del a, b, c
print(json.dumps(out))
