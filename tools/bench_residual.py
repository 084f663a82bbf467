#!/usr/bin/env python3
"""pm_lap_residual_error at 512^3, milliseconds per call including its read-back (development tool)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pysco_amd import _lib as L
n = 512
ctx = L.context(4)
x = torch.rand((n, n, n), device="cuda"); b = torch.rand((n, n, n), device="cuda")
out = C.c_float(0)
for _ in range(3): ctx.call("pm_lap_residual_error", L.ptr(x), L.ptr(b), n, C.byref(out))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): ctx.call("pm_lap_residual_error", L.ptr(x), L.ptr(b), n, C.byref(out))
print("residual_error ms", (time.perf_counter() - t0) / 50 * 1e3, out.value)
