#!/usr/bin/env python3
"""Diagnostic: per-phase wall-clock shares of interp_tiled_kernel (needs the -DPM_STAMP build:
PMSTEP_LIB=pysco_amd/libpmstep_stamp.so python tools/stamp_interp.py)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import zeldovich_device
from pysco_amd import _lib as L
from pysco_amd import utils

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
pos, vel = zeldovich_device(n)
utils.periodic_wrap(pos)
pos, vel = utils.reorder_particles(pos, vel)
F = torch.randn((n, n, n, 3), dtype=torch.float32, device="cuda")
acc = torch.empty_like(pos)
ctx = L.context(n)
lib = L.load_library()
lib.pm_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
for sch, name in ((1, "NGP"), (3, "TSC")):
    ctx.call("pm_interp_kick", sch, L.ptr(F), L.ptr(pos), n**3, L.ptr(acc), L.ptr(vel), 1e-9)
    torch.cuda.synchronize()
    lib.pm_debug_stamps(None, 1)
    ctx.call("pm_interp_kick", sch, L.ptr(F), L.ptr(pos), n**3, L.ptr(acc), L.ptr(vel), 1e-9)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    lib.pm_debug_stamps(buf, 0)
    v = list(buf)
    nblocks = (n**3 + 2047) // 2048
    tot = sum(v[:5])
    names = ["anchor", "fill+barrier", "gather loop", "stragglers+barrier", "phase2 kick"]
    print(f"{name}: blocks {nblocks}, passes/block {v[8] / nblocks:.2f}, sum of block time {tot / 100e6 * 1e3:.1f} ms (100 MHz clock)")
    for nm, x in zip(names, v[:5]):
        print(f"   {nm:22s} {x / nblocks / 100:8.2f} us/block  {100 * x / tot:5.1f} %")
