#!/usr/bin/env python3
"""Do the LDS-bound particle kernels and the FFT passes overlap when launched on two streams? (development tool)
    python tools/bench_overlap.py [ncells]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import zeldovich_device
from pysco_amd import _lib as L, fourier, utils

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
pos, vel = zeldovich_device(n, rms_cells=0.5)
utils.periodic_wrap(pos)
pos, vel = utils.reorder_particles(pos, vel)
acc = torch.zeros_like(pos)
rho = torch.empty((n, n, n), dtype=torch.float32, device="cuda")
rhs = torch.randn((n, n, n), dtype=torch.float32, device="cuda")
phi = torch.empty_like(rhs)
phi2 = torch.randn((n, n, n), dtype=torch.float32, device="cuda")
k2, s, c0 = fourier.green_tables(n, 1)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(sA):
    cA = L.Context(n)
with torch.cuda.stream(sB):
    cB = L.Context(n)


def dep():
    cA.call("pm_kick_drift_wrap_deposit", 3, L.ptr(pos), L.ptr(vel), L.ptr(acc), 1e-9, 1e-9, n**3, L.ptr(rho))


def gat():
    cA.call("pm_gradient_interp_kick", 5, 3, L.ptr(phi2), n, 0, L.ptr(pos), n**3, L.ptr(acc), L.ptr(vel), 1e-9)


def fft():
    cB.call("pm_poisson_fft_fused", 1, 3, L.ptr(rhs), L.ptr(phi), L.hptr(k2), L.hptr(s), c0)


def wall(fns, reps=5):
    for f in fns:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.default_stream())
    sA.wait_event(e0), sB.wait_event(e0)
    for _ in range(reps):
        for f in fns:
            f()
    ea, eb = torch.cuda.Event(), torch.cuda.Event()
    ea.record(sA), eb.record(sB)
    torch.cuda.default_stream().wait_event(ea), torch.cuda.default_stream().wait_event(eb)
    e1.record(torch.cuda.default_stream())
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, part in (("kick+drift+deposit", dep), ("gradient+gather+kick", gat)):
    a, b, ab = wall([part]), wall([fft]), wall([part, fft])
    print(f"{name:22s} {a:7.3f} ms   poisson_fft {b:7.3f} ms   both streams {ab:7.3f} ms   (sum {a + b:7.3f})")
