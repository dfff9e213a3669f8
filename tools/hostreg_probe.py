#!/usr/bin/env python3
"""How fast is an upload from pageable numpy memory, and does hipHostRegister pay for one 59-MB upload?"""
import ctypes, time
import numpy as np
import torch
hip = ctypes.CDLL("libamdhip64.so")
x = np.random.default_rng(0).standard_normal((14433, 1024), dtype=np.float32)
nbytes = x.nbytes
d = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
torch.cuda.synchronize()
def t(fn, reps=7):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))
def plain():
    hip.hipMemcpy(ctypes.c_void_p(d.data_ptr()), ctypes.c_void_p(x.ctypes.data), ctypes.c_size_t(nbytes), 1)
def reg_copy():
    hip.hipHostRegister(ctypes.c_void_p(x.ctypes.data), ctypes.c_size_t(nbytes), 0)
    hip.hipMemcpy(ctypes.c_void_p(d.data_ptr()), ctypes.c_void_p(x.ctypes.data), ctypes.c_size_t(nbytes), 1)
    hip.hipHostUnregister(ctypes.c_void_p(x.ctypes.data))
def reg_only():
    hip.hipHostRegister(ctypes.c_void_p(x.ctypes.data), ctypes.c_size_t(nbytes), 0)
    hip.hipHostUnregister(ctypes.c_void_p(x.ctypes.data))
print(f"pageable hipMemcpy H2D {nbytes/1e6:.0f} MB: {t(plain):.2f} ms")
print(f"register + copy + unregister: {t(reg_copy):.2f} ms   (register + unregister alone: {t(reg_only):.2f} ms)")
out = np.empty((14433, 301), np.int64)
def d2h():
    hip.hipMemcpy(ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(d.data_ptr()), ctypes.c_size_t(out.nbytes), 2)
print(f"pageable hipMemcpy D2H {out.nbytes/1e6:.0f} MB: {t(d2h):.2f} ms")
