"""One rank of a G-way x-slab split, alone on the GPU: the fused kick + drift + deposit of its particles into its density
slab with the plain flush (memset + atomics) and with the halo buffer (pm_kick_drift_wrap_deposit_slab_khalo), and
the FFT's z pass with and without the buffer to add.

    python tools/bench_slab_khalo.py [n=1024] [G=2]"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from pysco_amd import _lib as L  # noqa: E402
from pysco_amd import utils  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
G = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nx, W = n // G, 2
pos, vel = bench.zeldovich_device(n)
utils.periodic_wrap(pos)
keep = pos[:, 0] < 1.0 / G
pos, vel = pos[keep].contiguous(), vel[keep].contiguous()
pos, vel = utils.reorder_particles(pos, vel)
acc = torch.zeros_like(pos)
npart = pos.shape[0]
ctx = L.context(n)
rho = torch.empty((nx + 2 * W, n, n), dtype=torch.float32, device="cuda")
khalo = torch.zeros(int(L.load_library().pm_khalo_floats_slab(n, nx)), dtype=torch.float32, device="cuda")
leave = torch.empty(max(4096, npart // 16), dtype=torch.int64, device="cuda")
count = C.c_int(0)
dt = float(np.float32(1e-6))


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def k1(buf):
    args = [L.TSC, L.ptr(pos), L.ptr(vel), L.ptr(acc), 0.0, dt, npart, L.ptr(rho), 0.0, 1.0 / G, L.ptr(leave), leave.numel(),
            C.byref(count)]
    if buf is None:
        ctx.call("pm_kick_drift_wrap_deposit_slab", *args)
    else:
        ctx.call("pm_kick_drift_wrap_deposit_slab_khalo", *args, L.ptr(buf))


own = rho[W:W + nx]
ctx.call("pm_set_slab", nx + 2 * W, -W)
ctx.call("pm_set_tile_window", 0, nx)
try:
    t_plain = timed(lambda: k1(None))
    t_halo = timed(lambda: k1(khalo))
finally:
    ctx.call("pm_set_slab", n, 0)
    ctx.call("pm_set_tile_window", 0, n)
# (after the last call, the one with the buffer: every particle's unit of mass is in the slab or in the buffer)
mass = float(rho.sum(dtype=torch.float64)) + float(khalo.sum(dtype=torch.float64))
print(f"mass in the slab + buffer / particles - 1 = {mass / npart - 1:.2e}")
if "--single" in sys.argv:  # the same particles through the single-GPU entry points (full mesh, the other tiles empty)
    rho_full = torch.empty((n, n, n), dtype=torch.float32, device="cuda")
    kh_full = torch.zeros(int(L.load_library().pm_khalo_floats(n)), dtype=torch.float32, device="cuda")
    t_sp = timed(lambda: ctx.call("pm_kick_drift_wrap_deposit", L.TSC, L.ptr(pos), L.ptr(vel), L.ptr(acc), 0.0, dt, npart,
                                  L.ptr(rho_full)))
    t_sh = timed(lambda: ctx.call("pm_kick_drift_wrap_deposit_khalo", L.TSC, L.ptr(pos), L.ptr(vel), L.ptr(acc), 0.0, dt,
                                  npart, L.ptr(rho_full), L.ptr(kh_full)))
    print(f"single-GPU entry points, same particles: plain {t_sp:.2f} ms, halo buffer {t_sh:.2f} ms")
    del rho_full, kh_full
own = own.contiguous()
t_z = timed(lambda: ctx.call("pm_fftp_z_forward", L.ptr(own), nx))
t_zh = timed(lambda: ctx.call("pm_fftp_z_forward_khalo", L.ptr(own), nx, L.ptr(khalo)))
print(f"n={n}, 1 of {G} slabs ({npart / 1e6:.0f} M particles): kick + deposit, plain flush {t_plain:.2f} ms, halo buffer "
      f"{t_halo:.2f} ms; z pass {t_z:.2f} ms, adding the buffer {t_zh:.2f} ms; together {t_plain + t_z:.2f} -> "
      f"{t_halo + t_zh:.2f} ms")
