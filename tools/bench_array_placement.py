#!/usr/bin/env python3
"""Does the speed of a streaming particle kernel depend on WHICH allocations the arrays are?  (development tool)
Several [Np,3] arrays allocated at different moments of the process's life; the plain kick + drift + wrap timed on
every (pos, vel, acc) combination of interest.    python tools/bench_array_placement.py --ncells 1024"""
import argparse
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ncells", type=int, default=1024)
    args = ap.parse_args()
    import torch

    from pysco_amd import _lib as L

    n = args.ncells
    npart = n**3
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

    def new(tag):
        a = torch.zeros((npart, 3), dtype=torch.float32, device="cuda")
        print(f"  {tag}: {a.data_ptr():#x}")
        return a

    print("early allocations (fresh device memory):")
    A = [new(f"A{i}") for i in range(6)]
    for trip in ((0, 1, 2), (3, 4, 5), (0, 2, 4), (5, 3, 1)):
        print(f"A{trip}: {t_of(*[A[i] for i in trip]):.3f} ms")
    print("churn: allocate 150 GB in odd sizes, free every other block, empty the cache")
    junk = [torch.empty((int(1.37e9 * (1 + 0.02 * i)),), dtype=torch.uint8, device="cuda") for i in range(70)]
    del junk[::2]
    torch.cuda.empty_cache()
    B = [new(f"B{i}") for i in range(3)]
    del junk
    torch.cuda.empty_cache()
    C = [new(f"C{i}") for i in range(3)]
    print(f"B: {t_of(*B):.3f} ms   C: {t_of(*C):.3f} ms   A(0,1,2) again: {t_of(A[0], A[1], A[2]):.3f} ms")
    print(f"mixed A0 B1 C2: {t_of(A[0], B[1], C[2]):.3f} ms")
    # one arena, arrays carved at 2 MiB-aligned offsets
    del B, C
    torch.cuda.empty_cache()
    S = npart * 12
    arena = torch.empty((3 * S + (4 << 21),), dtype=torch.uint8, device="cuda")
    off = (-arena.data_ptr()) % (1 << 21)
    V = [arena[off + k * (S + (1 << 21)): off + k * (S + (1 << 21)) + S].view(torch.float32).view(npart, 3) for k in range(3)]
    print(f"arena {arena.data_ptr():#x}: {t_of(*V):.3f} ms")
    del arena, V
    torch.cuda.empty_cache()
    for gib in (13, 16, 24, 30):
        ar = [torch.empty((gib << 30,), dtype=torch.uint8, device="cuda") for _ in range(3)]
        V = [a[:S].view(torch.float32).view(npart, 3) for a in ar]
        print(f"three allocations of {gib} GiB, arrays at their starts: {t_of(*V):.3f} ms")
        V = [a[(gib << 30) - S:].view(torch.float32).view(npart, 3) for a in ar]
        print(f"   ... arrays at their ends: {t_of(*V):.3f} ms")
        del ar, V
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
