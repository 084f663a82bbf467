#!/usr/bin/env python3
"""Is the placement effect a property of single allocations?  Read bandwidth (pm_max_abs) and read+write bandwidth
(pm_scale) of each of several 12.9 GB arrays, then the 3-array kick + drift + wrap on triples of them.  (development tool)"""
import ctypes as C
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from pysco_amd import _lib as L

    n = 1024
    npart = n**3
    ctx = L.context(4)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timeit(fn):
        ts = []
        for _ in range(3):
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return min(ts[1:])

    arrays = [torch.zeros((npart, 3), dtype=torch.float32, device="cuda") for _ in range(9)]
    out = C.c_float(0)
    for i, a in enumerate(arrays):
        tr = timeit(lambda: ctx.call("pm_max_abs", L.ptr(a), a.numel(), C.byref(out)))
        tw = timeit(lambda: ctx.call("pm_scale", L.ptr(a), 1.0, a.numel()))
        print(f"array {i} at {a.data_ptr():#x}: read {tr:.3f} ms ({12.0 * npart / tr / 1e6:.0f} GB/s)  read+write {tw:.3f} ms "
              f"({24.0 * npart / tw / 1e6:.0f} GB/s)", flush=True)
    for trip in ((0, 1, 2), (3, 4, 5), (6, 7, 8), (0, 3, 6), (1, 4, 7), (2, 5, 8), (0, 4, 8), (2, 4, 6)):
        t = timeit(lambda: ctx.call("pm_kick_drift_wrap", L.ptr(arrays[trip[0]]), L.ptr(arrays[trip[1]]), L.ptr(arrays[trip[2]]),
                                    0.0, 0.0, npart))
        print(f"kick+drift+wrap on {trip}: {t:.3f} ms ({60.0 * npart / t / 1e6:.0f} GB/s)", flush=True)


if __name__ == "__main__":
    main()
