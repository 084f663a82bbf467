#!/usr/bin/env python3
"""Secondary BASELINE configs (development / reporting tool, not the driver's bench):
    python tools/bench_configs.py --ncells 512 --config fr|mond|newton_mg|newton_fft [--steps 3]
Under torch.distributed.run (one rank per GPU; PMSTEP_DIST_BACKEND=gloo rehearses several ranks on one GPU) the
run is slab-decomposed exactly as bench.py's.
One leapfrog step = integration.integrate (dt selection, kick/drift/deposit, solve, gather/kick).
Prints ms per step, multigrid cycle counts where the solver records them, and the per-stage table."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ncells", type=int, default=512)
    ap.add_argument("--config", default="fr")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--boxlen", type=float, default=0.0, help="override the box size (Mpc/h)")
    ap.add_argument("--scan-boxlen", default="", help="comma list: run one pm() per value, report finiteness only")
    args = ap.parse_args()
    import torch

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group(os.environ.get("PMSTEP_DIST_BACKEND", "nccl"))
    from bench import make_param, zeldovich_device
    from pysco_amd import _lib as L
    from pysco_amd import integration, solver, utils

    n = args.ncells
    param, tables = make_param(n, n**3)
    param.update({"w0": -1.0, "wa": 0.0, "Npre": 2, "Npost": 1, "epsrel": 1e-2})
    if args.config == "fr":
        param.update({"theory": "fr", "fR_n": 1, "fR_logfR0": 5, "linear_newton_solver": "multigrid",
                      "boxlen": 1000.0 * (n / 32)})  # the reference itself returns NaN below ~31 Mpc/h cells
    elif args.config == "mond":
        param.update({"theory": "mond", "mond_function": "simple", "mond_g0": 1.2, "mond_alpha": 1,
                      "mond_scale_factor_exponent": 0, "linear_newton_solver": "multigrid"})
    elif args.config == "newton_mg":
        param.update({"linear_newton_solver": "multigrid"})
    elif args.config != "newton_fft":
        raise SystemExit("config: fr | mond | newton_mg | newton_fft")
    if args.boxlen > 0:
        param["boxlen"] = args.boxlen
    utils.set_units(param)
    pos, vel = zeldovich_device(n)
    utils.periodic_wrap(pos)
    if world > 1:
        from pysco_amd import slab

        sl = slab.get(n)
        lo, hi = sl.x_lo * n * n, (sl.x_lo + sl.nx) * n * n
        pos, vel = sl.migrate(pos[lo:hi].clone(), vel[lo:hi].clone())
    pos, vel = utils.reorder_particles(pos, vel)
    if args.scan_boxlen:
        for b in [float(x) for x in args.scan_boxlen.split(",")]:
            q = dict(param, boxlen=b)
            utils.set_units(q)
            acc, phi, extra = solver.pm(pos, q, tables=tables)
            ok = bool(torch.isfinite(acc).all()) and bool(torch.isfinite(extra).all() if extra.numel() else True)
            print(f"boxlen {b:9.1f}: finite={ok}  fR_q={q.get('fR_q')}  max|acc|={float(acc.abs().nan_to_num(0).max()):.3e}"
                  f"  u in [{float(extra.nan_to_num(0).min()) if extra.numel() else 0:.3f}, "
                  f"{float(extra.nan_to_num(0).max()) if extra.numel() else 0:.3f}]")
        return
    t0 = time.perf_counter()
    acc, phi, extra = solver.pm(pos, param, tables=tables)
    torch.cuda.synchronize()
    print(f"first pm() (cold start of the iterative solves): {1e3 * (time.perf_counter() - t0):.1f} ms")
    for _ in range(args.warmup):
        pos, vel, acc, phi, extra = integration.integrate(pos, vel, acc, phi, extra, tables, param, np.float32(1))
    torch.cuda.synchronize()
    L.profile_start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pos, vel, acc, phi, extra = integration.integrate(pos, vel, acc, phi, extra, tables, param, np.float32(1))
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    stages = L.profile_stop()
    assert bool(torch.isfinite(pos).all()) and bool(torch.isfinite(acc).all())
    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    print(json.dumps({"config": args.config, "ncells": n, "ranks": world, "ms_per_step": 1e3 * wall / args.steps,
                      "particles_per_s": n**3 * args.steps / wall}))
    tot = sum(ms for _, ms in stages.values())
    for name, (calls, ms) in sorted(stages.items(), key=lambda kv: -kv[1][1])[:18]:
        print(f"  {name:34s} calls/step {calls / args.steps:8.1f}  ms/step {ms / args.steps:9.3f}  avg {ms / calls:8.4f} ms")
    print(f"  (sum of stage events {tot / args.steps:.1f} ms/step)")
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
