#!/usr/bin/env python3
"""fp64 device mode, development timings: evolve a 512^3 run for some steps without re-sorting, then time the gather,
the deposit and the re-sort variants on that (aged, clustered) stream.  python tools/bench_real64_parts.py --steps 15"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ncells", type=int, default=512)
    ap.add_argument("--steps", type=int, default=15)
    args = ap.parse_args()
    import numpy as np
    import torch

    import bench
    from pysco_amd import _lib as L
    from pysco_amd import integration, real64, solver, utils

    n = args.ncells
    pos, vel = bench.zeldovich_device(n, rms_cells=0.5)
    utils.periodic_wrap(pos)
    pos, vel = utils.reorder_particles(pos, vel)
    pos, vel = pos.double(), vel.double()
    L._ctxs.clear()
    torch.cuda.empty_cache()
    param, tables = bench.make_param(n, n**3, "newton_fft")
    acc, phi, extra = solver.pm(pos, param, tables=tables)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timeit(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    ctx = L.context(n, 8)
    import ctypes as C

    def parts(tag, pos, vel):
        rho = torch.empty((n, n, n), dtype=torch.float64, device="cuda")
        force = torch.randn((n, n, n, 3), dtype=torch.float64, device="cuda")
        out = torch.empty_like(pos)
        v2 = vel.clone()
        ma, mv = C.c_double(0), C.c_double(0)
        td = timeit(lambda: ctx.call("pm_real64_deposit", L.TSC, L.ptr(pos), pos.shape[0], L.ptr(rho)))
        tg = timeit(lambda: ctx.call("pm_real64_interp_kick", L.TSC, L.ptr(force), L.ptr(pos), pos.shape[0], L.ptr(out), L.ptr(v2), 0.0,
                                     C.byref(ma), C.byref(mv)))
        print(f"{tag}: deposit {td:.2f} ms  gather {tg:.2f} ms  age {real64.stream_age(n)}  max rho {float(rho.max()):.0f} "
              f"empty cells {float((rho == 0).double().mean()):.3f}", flush=True)

    parts("fresh", pos, vel)
    for i in range(args.steps):
        param["nsteps"] = i + 1
        pos, vel, acc, phi, extra = integration.integrate(pos, vel, acc, phi, extra, tables, param, np.float32(3e38))
    del phi, extra
    parts(f"after {args.steps} in-place steps", pos, vel)
    for bits, tag in ((9, "cells (27 bits)"), (-1, "tiles"), (0, "full key")):
        po, vo, ao = torch.empty_like(pos), torch.empty_like(vel), torch.empty_like(acc)
        t = timeit(lambda: ctx.call("pm_real64_reorder_particles", pos.shape[0], L.ptr(pos), L.ptr(vel), L.ptr(acc), L.ptr(po), L.ptr(vo),
                                    L.ptr(ao), None, bits), reps=2)
        print(f"re-sort to {tag}: {t:.2f} ms", flush=True)
        parts(f"  sorted to {tag}", po, vo)
        del po, vo, ao


if __name__ == "__main__":
    main()
