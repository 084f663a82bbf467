#!/usr/bin/env python3
"""Step time as a function of the age of the Morton sort (development tool): one sort, then `--steps` leapfrog
steps, ms per step in windows of 5.    python tools/bench_sort_age.py --ncells 512 --steps 60"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ncells", type=int, default=512)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--rms-cells", type=float, default=0.5)
    ap.add_argument("--vel-factor", type=float, default=0.02)
    args = ap.parse_args()
    import torch

    from bench import make_param, zeldovich_device
    from pysco_amd import _lib as L
    from pysco_amd import integration, solver, utils

    n = args.ncells
    pos, vel = zeldovich_device(n, rms_cells=args.rms_cells, vel_factor=args.vel_factor)
    utils.periodic_wrap(pos)
    pos0 = None
    pos, vel = utils.reorder_particles(pos, vel)
    pos0 = pos.clone()
    param, tables = make_param(n, n**3)
    acc, phi, extra = solver.pm(pos, param, tables=tables)
    torch.cuda.synchronize()
    for w in range(0, args.steps, 5):
        L.profile_start()
        t0 = time.perf_counter()
        for _ in range(5):
            pos, vel, acc, phi, extra = integration.integrate(pos, vel, acc, phi, extra, tables, param, np.float32(1))
        torch.cuda.synchronize()
        d = (pos - pos0).abs()
        d = torch.minimum(d, 1 - d) * n
        wall = time.perf_counter() - t0
        stages = L.profile_stop()
        frac, complete = utils.straggler_fraction(n)
        print(f"steps {w + 1:3d}-{w + 5:3d}: {1e3 * wall / 5:7.2f} ms/step   "
              f"rms drift since the sort {float((d * d).sum(1).mean().sqrt()):.3f} cells, max {float(d.max()):.2f}; "
              f"stragglers {frac:.3f} complete={complete}; "
              + ", ".join(f"{k[3:]} {ms / 5:.2f}" for k, (c, ms) in sorted(stages.items(), key=lambda kv: -kv[1][1])[:4]))


if __name__ == "__main__":
    main()
