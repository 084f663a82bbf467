#!/usr/bin/env python3
"""Three separately allocated particle arrays whose allocations are padded by different amounts: does the speed of the
streaming kick + drift + wrap depend on the spacing of the allocations?  (development tool)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from pysco_amd import _lib as L

    n = 1024
    npart = n**3
    S = npart * 12
    ctx = L.context(4)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def t_of(pos, vel, acc):
        ts = []
        for rep in range(3):
            e0.record()
            ctx.call("pm_kick_drift_wrap", L.ptr(pos), L.ptr(vel), L.ptr(acc), 0.0, 0.0, npart)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return min(ts[1:])

    MiB = 1 << 20
    for rep in range(2):
        for pads in ((0, 0, 0), (0, 64 * MiB, 128 * MiB), (0, 257 * MiB, 515 * MiB), (0, 1024 * MiB, 2048 * MiB),
                     (0, 1500 * MiB, 3100 * MiB), (0, 4096 * MiB, 8192 * MiB), (0, 5000 * MiB, 11000 * MiB)):
            bufs = [torch.zeros((S + p,), dtype=torch.uint8, device="cuda") for p in pads]
            arrs = [b[:S].view(torch.float32).view(npart, 3) for b in bufs]
            print(f"pads {[p // MiB for p in pads]} MiB  ptrs {[hex(b.data_ptr()) for b in bufs]}: {t_of(*arrs):.3f} ms", flush=True)
            del bufs, arrs
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
