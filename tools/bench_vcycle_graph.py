#!/usr/bin/env python3
"""pm_lap_vcycle: plain launches against the replay of its captured hipGraph, same arrays (development tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pysco_amd import _lib as L

ctx = L.context(4)
for n in (32, 64, 128, 256):
    b = torch.rand((n, n, n), device="cuda") - 0.5
    x = torch.zeros((n, n, n), device="cuda")
    for mode, fused in ((0, 0), (1, 0), (0, 1), (1, 1)):
        ctx.call("pm_set_vcycle_graphs", mode)
        ctx.call("pm_set_fused_coarse", fused)
        for _ in range(5):
            ctx.call("pm_lap_vcycle", L.ptr(x), L.ptr(b), n, 2, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            ctx.call("pm_lap_vcycle", L.ptr(x), L.ptr(b), n, 2, 1)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        print(f"n={n:4d} graphs={mode} fused_coarse={fused}: {t / 200 * 1e3:.4f} ms per cycle (host issue {t_issue / 200 * 1e3:.4f})", flush=True)
