#!/usr/bin/env python3
"""Per-entry-point timing of libpmstep on synthetic sorted particles (development tool).
    python tools/bench_kernels.py --ncells 512 [--only deposit,interp]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, reps=5):
    import torch

    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ncells", type=int, default=512)
    ap.add_argument("--rms-cells", type=float, default=0.5)
    ap.add_argument("--only", default="")
    ap.add_argument("--drift-cells", type=float, default=0.0, help="rms displacement applied AFTER the sort")
    args = ap.parse_args()
    import torch

    from bench import zeldovich_device
    from pysco_amd import _lib as L
    from pysco_amd import fourier, mesh, utils

    n = args.ncells
    pos, vel = zeldovich_device(n, rms_cells=args.rms_cells)
    utils.periodic_wrap(pos)
    pos, vel = utils.reorder_particles(pos, vel)
    npart = n**3
    if args.drift_cells > 0:
        pos += (args.drift_cells / n) * torch.randn_like(pos)
        utils.periodic_wrap(pos)
        c = (pos * n).to(torch.int32).clamp_(0, n - 1)
        tile = (c[:, 0] >> 3) * (n >> 4) ** 2 + (c[:, 1] >> 4) * (n >> 4) + (c[:, 2] >> 4)
        del c
        frac = float((tile[1:] != tile[:-1]).float().mean())
        del tile
        print(f"drift {args.drift_cells} cells: {frac:.4f} of consecutive particles change tile "
              f"(sorted stream: {1.0 / 2048 * (n**3 / npart):.4f})")
    only = set(args.only.split(",")) if args.only else None
    ctx = L.context(n)
    rho = torch.empty((n, n, n), dtype=torch.float32, device="cuda")
    F = torch.randn((n, n, n, 3), dtype=torch.float32, device="cuda")
    acc = torch.zeros_like(pos)

    def want(k):
        return only is None or k in only

    rows = []
    if want("deposit"):
        for name, sch in (("NGP", 1), ("CIC", 2), ("TSC", 3)):
            for hint in (1, 2, 0):
                t = timeit(lambda: ctx.call("pm_deposit", sch, L.ptr(pos), npart, L.ptr(rho), hint))
                rows.append((f"deposit {name} sorted_hint={hint}", t, (12 * npart + 4 * n**3) / t / 1e6))
        t = timeit(lambda: rho.zero_())
        rows.append(("memset rho", t, 4 * n**3 / t / 1e6))
        perm = torch.randperm(npart, device="cuda")
        pos_shuf = pos[perm].contiguous()
        for hint in (1, 2, 0):
            t = timeit(lambda: ctx.call("pm_deposit", 3, L.ptr(pos_shuf), npart, L.ptr(rho), hint), reps=2)
            rows.append((f"deposit TSC SHUFFLED sorted_hint={hint}", t, (12 * npart + 4 * n**3) / t / 1e6))
        del perm, pos_shuf
    if want("interp"):
        for name, sch in (("NGP", 1), ("CIC", 2), ("TSC", 3)):
            t = timeit(lambda: ctx.call("pm_interp", sch, 3, L.ptr(F), L.ptr(pos), npart, L.ptr(acc)))
            rows.append((f"interp {name}", t, (24 * npart + 12 * n**3) / t / 1e6))
        for mode in (1, 2):
            ctx.call("pm_set_gather_mode", mode)
            for name, sch in (("NGP", 1), ("CIC", 2), ("TSC", 3)):
                t = timeit(lambda: ctx.call("pm_interp_kick", sch, L.ptr(F), L.ptr(pos), npart, L.ptr(acc), L.ptr(vel), 1e-9))
                rows.append((f"interp_kick {name} mode={mode} (fused kick+max)", t, (48 * npart + 12 * n**3) / t / 1e6))
        ctx.call("pm_set_gather_mode", 1)
        t = timeit(lambda: ctx.call("pm_interp_kick", 3, L.ptr(F), L.ptr(pos), npart, L.ptr(acc), None, 0.0))
        rows.append(("interp_kick TSC (tiled, no kick)", t, (24 * npart + 12 * n**3) / t / 1e6))
    if want("fusedgather"):
        phi = torch.randn((n, n, n), dtype=torch.float32, device="cuda")
        for order in (3, 5, 7):
            t = timeit(lambda: ctx.call("pm_gradient_interp_kick", order, 3, L.ptr(phi), n, 0, L.ptr(pos), npart,
                                        L.ptr(acc), L.ptr(vel), 1e-9))
            rows.append((f"gradient+interp_kick TSC order {order}", t, (48 * npart + 4 * n**3) / t / 1e6))
        del phi
    if want("gradient"):
        phi = torch.randn((n, n, n), dtype=torch.float32, device="cuda")
        for order in (2, 3, 5, 7):
            t = timeit(lambda: ctx.call("pm_gradient", order, L.ptr(phi), L.ptr(F)))
            rows.append((f"gradient order {order}", t, 16 * n**3 / t / 1e6))
    if want("fft"):
        phi = torch.randn((n, n, n), dtype=torch.float32, device="cuda")
        xk = torch.empty((n, n, n // 2 + 1), dtype=torch.complex64, device="cuda")
        t = timeit(lambda: ctx.call("pm_fft_r2c", L.ptr(phi), L.ptr(xk)))
        rows.append(("fft r2c", t, 24 * n**3 / t / 1e6))
        t = timeit(lambda: ctx.call("pm_fft_c2r", L.ptr(xk), L.ptr(phi)))
        rows.append(("fft c2r", t, 24 * n**3 / t / 1e6))
        k2, s, c0 = fourier.green_tables(n, 1)
        t = timeit(lambda: ctx.call("pm_greens", 1, 3, L.ptr(xk), L.hptr(k2), L.hptr(s), c0))
        rows.append(("greens compensated", t, 16 * n * n * (n // 2 + 1) / t / 1e6))
    if want("kdwdep"):
        for name, sch in (("CIC", 2), ("TSC", 3)):
            t = timeit(lambda: ctx.call("pm_kick_drift_wrap_deposit", sch, L.ptr(pos), L.ptr(vel), L.ptr(acc), 1e-9, 1e-9,
                                        npart, L.ptr(rho)))
            rows.append((f"kick_drift_wrap + deposit {name}", t, (60 * npart + 4 * n**3) / t / 1e6))
        ctx.call("pm_tile_lists", None, 0)
    if want("kdwdepk"):
        from pysco_amd import fourier as F2
        kh = torch.empty((int(L.load_library().pm_khalo_floats(n)),), dtype=torch.float32, device="cuda")
        t = timeit(lambda: ctx.call("pm_kick_drift_wrap_deposit_khalo", 3, L.ptr(pos), L.ptr(vel), L.ptr(acc), 1e-9, 1e-9,
                                    npart, L.ptr(rho), L.ptr(kh)))
        rows.append(("kick_drift_wrap + deposit TSC, k-halo buffer", t, (60 * npart + 4 * n**3) / t / 1e6))
        ctx.call("pm_tile_lists", None, 0)
        k2, s_, c0 = F2.green_tables(n, 1)
        phi = torch.empty_like(rho)
        t = timeit(lambda: ctx.call("pm_poisson_fft_fused", 1, 3, L.ptr(rho), L.ptr(phi), L.hptr(k2), L.hptr(s_), c0))
        rows.append(("poisson_fft_fused", t, 48 * n**3 / t / 1e6))
        t = timeit(lambda: ctx.call("pm_poisson_fft_fused_khalo", 1, 3, L.ptr(rho), L.ptr(kh), L.ptr(phi), L.hptr(k2),
                                    L.hptr(s_), c0))
        rows.append(("poisson_fft_fused_khalo", t, 48 * n**3 / t / 1e6))
        del kh, phi
    if want("stream"):
        t = timeit(lambda: ctx.call("pm_kick_drift_wrap", L.ptr(pos), L.ptr(vel), L.ptr(acc), 1e-9, 1e-9, npart))
        rows.append(("kick_drift_wrap", t, 60 * npart / t / 1e6))
        t = timeit(lambda: ctx.call("pm_axpy", L.ptr(vel), L.ptr(acc), 1e-9, 3 * npart))
        rows.append(("axpy", t, 36 * npart / t / 1e6))
    for name, t, gbs in rows:
        print(f"{name:42s} {t:10.3f} ms   {gbs:9.1f} GB/s (algorithmic)")


if __name__ == "__main__":
    main()
