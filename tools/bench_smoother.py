#!/usr/bin/env python3
"""Red-black sweep and V-cycle timings, single-pass sweep on / off (development tool).  python tools/bench_smoother.py --ncells 512"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ncells", type=int, default=512)
    args = ap.parse_args()
    import torch

    from pysco_amd import _lib as L
    from pysco_amd import laplacian, multigrid

    n = args.ncells
    b = torch.randn((n, n, n), device="cuda")
    b -= b.mean()
    x = torch.zeros_like(b)
    ctx = L.context(4)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timeit(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    param = {"Npre": 2, "Npost": 1, "ncoarse": n.bit_length() - 1}
    for fused in (0, 1):
        ctx.call("pm_set_fused_sweep", fused)
        for sweeps in (1, 2, 3):
            t = timeit(lambda: laplacian.smoothing(x, b, sweeps))
            print(f"fused={fused} {sweeps} sweep(s): {t:.3f} ms  ({12.0 * n**3 * sweeps / t / 1e6:.0f} GB/s algorithmic)")
        t = timeit(lambda: multigrid.V_cycle(x, b, param))
        print(f"fused={fused} V-cycle: {t:.3f} ms")
    ctx.call("pm_set_fused_sweep", 1)
    # f(R): scalaron sweeps and the FAS V-cycle (pm_fr_smoothing / pm_fr_vcycle), single-pass on / off
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    bb = torch.rand((n, n, n), generator=g, device="cuda") * 0.5 + 0.2
    x0 = torch.rand((n, n, n), generator=g, device="cuda") * 0.2 + 0.9
    rhs = (torch.rand((n, n, n), generator=g, device="cuda") - 0.5) * 1e-3
    for fr_n in (1, 2):
        for fused in (0, 1):
            ctx.call("pm_set_fused_sweep", fused)
            for r, tag in ((None, "no rhs"), (rhs, "rhs")):
                xx = x0.clone()
                t = timeit(lambda: ctx.call("pm_fr_smoothing", fr_n, L.ptr(xx), L.ptr(bb), -0.1, L.ptr(r), n, 2))
                print(f"f(R) n={fr_n} fused={fused} 2 sweeps, {tag}: {t / 2:.3f} ms per sweep  ({12.0 * n**3 * 2 / t / 1e6:.0f} GB/s algorithmic)"
                      f"  nan: {int(torch.isnan(xx).sum())}")
            xx = x0.clone()
            t = timeit(lambda: ctx.call("pm_fr_vcycle", fr_n, L.ptr(xx), L.ptr(bb), -0.1, n, 2, 1), reps=4)
            print(f"f(R) n={fr_n} fused={fused} FAS V-cycle: {t:.3f} ms  nan: {int(torch.isnan(xx).sum())}")
    ctx.call("pm_set_fused_sweep", 1)


if __name__ == "__main__":
    main()
