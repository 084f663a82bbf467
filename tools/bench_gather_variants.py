#!/usr/bin/env python3
"""Same-process, same-arrays comparison of library variants on the step's two particle kernels (development tool):
the fused kick + drift + deposit (halo buffer form) and the re-bucketing gather, ping-ponging between two sets of
particle arrays exactly as the product loop does, with dt ~ 0 so that every variant sees the same particles.

    python tools/bench_gather_variants.py --ncells 1024 base NOLDS NOFILL ...    (tags of pysco_amd/_ab/lib<tag>.so;
                                                                                   "base" = pysco_amd/libpmstep.so)
Different libraries in ONE process share the arrays, hence the physical placement that moves a kernel by 10 % from
process to process."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ncells", type=int, default=1024)
    ap.add_argument("--rms-cells", type=float, default=0.5)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--scheme", type=int, default=3)
    ap.add_argument("--order", type=int, default=5)
    ap.add_argument("--evolve", type=int, default=0, help="take this many steps of the product loop first (a clustered, evolved state)")
    ap.add_argument("--resort", action="store_true", help="full Morton sort of the evolved stream before measuring")
    ap.add_argument("tags", nargs="+")
    args = ap.parse_args()
    import torch

    import bench
    from pysco_amd import _lib as L
    from pysco_amd import solver, utils

    n = args.ncells
    npart = n**3
    pos, vel = bench.zeldovich_device(n, rms_cells=args.rms_cells)
    utils.periodic_wrap(pos)
    pos, vel = utils.reorder_particles(pos, vel)
    L._ctxs.clear()
    torch.cuda.empty_cache()
    param, tables = bench.make_param(n, npart)
    acc, phi, extra = solver.pm(pos, param, tables=tables)
    if args.evolve:
        from pysco_amd import integration
        from pysco_amd import main as pm_main

        param["rebucket_straggler_fraction"] = pm_main.REBUCKET_FRACTION
        pm_main.reserve_step_buffers(pos)
        L.profile_start()
        for i in range(args.evolve):
            param["nsteps"] += 1
            pos, vel, acc, phi, extra = integration.integrate(pos, vel, acc, phi, extra, tables, param, np.float32(3e38))
            pos, vel, acc = pm_main.reorder_policy(pos, vel, acc, param)
            L.profile_mark()
        st, per = L.profile_stop(per_step=True)
        print("evolved", args.evolve, "steps; last step stages [ms]:", {k[3:30]: round(v, 3) for k, v in sorted(per[-1].items(), key=lambda kv: -kv[1])[:3]},
              "dt", param.get("_dt_last"), flush=True)
        from pysco_amd import mesh as _mesh

        _mesh._rebucket_pool.clear()
        # how clustered: particles per occupied cell, fraction of empty cells
        c = (pos * n).to(torch.int32).clamp_(0, n - 1)
        cell = (c[:, 0].long() * n + c[:, 1].long()) * n + c[:, 2].long()
        del c
        cnt = torch.bincount(cell, minlength=n**3)
        del cell
        print(f"cells empty {float((cnt == 0).float().mean()):.3f}, max per cell {int(cnt.max())}, particles in cells with >= 8: "
              f"{float(cnt[cnt >= 8].sum()) / npart:.3f}", flush=True)
        del cnt
        c = (pos * n).to(torch.int32).clamp_(0, n - 1)
        tile = ((c[:, 0] >> 3).long() * (n >> 4) + (c[:, 1] >> 4).long()) * (n >> 4) + (c[:, 2] >> 4).long()
        del c
        tc = torch.bincount(tile, minlength=n**3 // 2048)
        del tile
        for ch in (4096, 8192, 16384):
            extra = int(torch.clamp((tc + ch - 1) // ch - 1, min=0).sum())
            print(f"chunk {ch}: {extra} surplus workgroups on {tc.numel()} tiles ({extra / tc.numel():.3f})")
        q = torch.tensor([0.1, 0.25, 0.5, 0.75, 0.9, 0.99], device=tc.device)
        print("particles per tile quantiles", torch.quantile(tc.float(), q).tolist(), "max", int(tc.max()),
              "tiles under 512:", float((tc < 512).float().mean()), "particles in tiles over 4096:", float(tc[tc > 4096].sum()) / npart, flush=True)
        del tc
        if args.resort:
            pos, vel, acc = utils.reorder_particles(pos, vel, acc)
            print("re-sorted (Morton)", flush=True)
    L._ctxs.clear()
    torch.cuda.empty_cache()
    sets = [(pos, vel, acc), tuple(torch.empty_like(pos) for _ in range(3))]
    rho = torch.empty((n, n, n), dtype=torch.float32, device="cuda")
    base_lib = L.load_library()
    khalo = torch.empty((int(base_lib.pm_khalo_floats(n)),), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def open_lib(tag):
        path = os.path.join(ROOT, "pysco_amd", "libpmstep.so" if tag == "base" else f"_ab/lib{tag}.so")
        lib = C.CDLL(path)
        for name, argtypes in L.SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = list(argtypes)
            fn.restype = L._RESTYPES.get(name, C.c_int)
        h = L._vp()
        assert lib.pm_ctx_create(C.byref(h), n, L._vp(stream)) == 0
        return lib, h

    tiny = float(np.float32(1e-12))
    cur = 0
    results = {t: ([], []) for t in args.tags}
    for rnd in range(args.rounds):
        for tag in args.tags:
            lib, h = open_lib(tag)  # (one context at a time: each holds GBs of list / re-sort buffers at 1024^3)
            tk, tg = [], []
            for rep in range(args.reps + 1):
                a, b = sets[cur], sets[cur ^ 1]
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                ev[0].record()
                rc = lib.pm_kick_drift_wrap_deposit_khalo(h, args.scheme, L.ptr(a[0]), L.ptr(a[1]), L.ptr(a[2]), tiny, tiny, npart,
                                                          L.ptr(rho), L.ptr(khalo))
                assert rc == 0, lib.pm_last_error()
                ev[1].record()
                rc = lib.pm_gradient_interp_kick_rebucket(h, args.order, args.scheme, L.ptr(phi), L.ptr(a[0]), L.ptr(a[1]), npart,
                                                          tiny, L.ptr(b[0]), L.ptr(b[1]), L.ptr(b[2]), None, None)
                assert rc == 0, lib.pm_last_error()
                ev[2].record()
                torch.cuda.synchronize()
                if "NOPOSOUT" in tag or "NOSTORE" in tag or "NOLOOP" in tag:
                    pass  # (the output set is incomplete: keep reading the same input set)
                else:
                    cur ^= 1
                if rep > 0:  # (the first call of a library on a set looks its ranges up from scratch)
                    tk.append(ev[0].elapsed_time(ev[1]))
                    tg.append(ev[1].elapsed_time(ev[2]))
            lib.pm_ctx_destroy(h)
            results[tag][0].extend(tk)
            results[tag][1].extend(tg)
            print(f"round {rnd} {tag:12s} kick+deposit {np.median(tk):7.3f} ms   gather {np.median(tg):7.3f} ms "
                  f"(min {min(tg):.3f} max {max(tg):.3f})", flush=True)
    print("== medians over all rounds")
    for tag in args.tags:
        print(f"{tag:12s} kick+deposit {np.median(results[tag][0]):7.3f}   gather {np.median(results[tag][1]):7.3f}")


if __name__ == "__main__":
    main()
