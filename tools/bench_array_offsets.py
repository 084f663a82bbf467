#!/usr/bin/env python3
"""Does the speed of the streaming particle kernels depend on WHERE the three particle arrays sit relative to one
another?  (development tool)  pos / vel / acc are carved out of one allocation at base + k * (S + d)
for a sweep of skews d and the plain kick + drift + wrap (x, v, a read; x, v written; 60 B per particle) is timed.
    python tools/bench_array_offsets.py --ncells 1024"""
import argparse
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
    S = npart * 12
    maxd = 64 << 20
    buf = torch.empty((3 * (S + maxd) + (1 << 21),), dtype=torch.uint8, device="cuda")
    base = (-buf.data_ptr()) % (1 << 21)  # 2 MiB-aligned start
    ctx = L.context(4)
    print(f"S = {S} bytes = {S / 2**30:.3f} GiB per array; buffer at {buf.data_ptr():#x}")

    def view(off):
        return buf[base + off: base + off + S].view(torch.float32).view(npart, 3)

    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for d in (0, 256, 1024, 4096, 8192, 16384, 32768, 65536, 1 << 17, 1 << 18, 1 << 19, 1 << 20, (1 << 20) + 4096,
              (1 << 21) + 8192, 3 << 20, (5 << 20) + 12288, 16 << 20, (33 << 20) + 256 * 5, 64 << 20):
        pos, vel, acc = view(0), view(S + d), view(2 * (S + d))
        pos.uniform_(0, 1)
        vel.zero_()
        acc.zero_()
        ts = []
        for rep in range(4):
            e0.record()
            ctx.call("pm_kick_drift_wrap", L.ptr(pos), L.ptr(vel), L.ptr(acc), 0.0, 0.0, npart)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = min(ts[1:])
        print(f"skew {d:>10d} B: {t:7.3f} ms  {60.0 * npart / t / 1e9:7.1f} GB/s   (runs {' '.join(f'{x:.2f}' for x in ts)})", flush=True)


if __name__ == "__main__":
    main()
