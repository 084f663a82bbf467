#!/usr/bin/env python3
"""Time rocFFT decompositions of the 3-D r2c/c2r at n^3 (development tool)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pysco_amd import _lib as L
from tools.bench_kernels import timeit

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = L.context(n)
x = torch.randn((n, n, n), dtype=torch.float32, device="cuda")
xk = torch.empty((n, n, n // 2 + 1), dtype=torch.complex64, device="cuda")
t = timeit(lambda: ctx.call("pm_fft_r2c", L.ptr(x), L.ptr(xk)))
print(f"3-D r2c plan            {t:8.3f} ms")
t = timeit(lambda: ctx.call("pm_fft_c2r", L.ptr(xk), L.ptr(x)))
print(f"3-D c2r plan            {t:8.3f} ms")
t1 = timeit(lambda: ctx.call("pm_fft2d_r2c", L.ptr(x), L.ptr(xk), n))
print(f"2-D r2c batched         {t1:8.3f} ms")
t2 = timeit(lambda: ctx.call("pm_fft1d_x", L.ptr(xk), n * (n // 2 + 1), 0))
print(f"1-D c2c strided (x)     {t2:8.3f} ms   -> 2-D + 1-D forward {t1 + t2:8.3f} ms")
t3 = timeit(lambda: ctx.call("pm_fft1d_x", L.ptr(xk), n * (n // 2 + 1), 1))
t4 = timeit(lambda: ctx.call("pm_fft2d_c2r", L.ptr(xk), L.ptr(x), n))
print(f"1-D inverse {t3:8.3f} ms, 2-D c2r {t4:8.3f} ms -> inverse {t3 + t4:8.3f} ms")
from pysco_amd import fourier

k2, s, c0 = fourier.green_tables(n, 1)
phi = torch.empty_like(x)
t = timeit(lambda: ctx.call("pm_poisson_fft_fused", 1, 3, L.ptr(x), L.ptr(phi), L.hptr(k2), L.hptr(s), c0))
print(f"fused Poisson solve (5 passes)  {t:8.3f} ms")
t = timeit(lambda: ctx.call("pm_poisson_fft", 1, 3, L.ptr(x), L.ptr(phi), L.hptr(k2), L.hptr(s), c0))
print(f"rocFFT r2c + greens + c2r       {t:8.3f} ms")
