"""What a slab rank's fused kick + deposit would gain from the halo buffer WITHOUT touching the FFT's z pass: owners
store their cells, halo cells go to the buffer, and pm_khalo_apply adds the buffer into the density with one atomic per
non-zero entry -- against the plain form (memset + every cell of every tile flushed with atomics), which is what
pm_kick_drift_wrap_deposit_slab does today.  One GPU, the single-GPU entry points (same kernels).

    python tools/bench_khalo_apply.py [n=1024]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from pysco_amd import _lib as L  # noqa: E402
from pysco_amd import utils  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pos, vel = bench.zeldovich_device(n)
utils.periodic_wrap(pos)
pos, vel = utils.reorder_particles(pos, vel)
acc = torch.zeros_like(pos)
npart = pos.shape[0]
ctx = L.context(n)
rho = torch.empty((n, n, n), dtype=torch.float32, device="cuda")
khalo = torch.empty(int(L.load_library().pm_khalo_floats(n)), dtype=torch.float32, device="cuda")
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


def plain():
    ctx.call("pm_kick_drift_wrap_deposit", L.TSC, L.ptr(pos), L.ptr(vel), L.ptr(acc), 0.0, dt, npart, L.ptr(rho))


def halo_only():
    ctx.call("pm_kick_drift_wrap_deposit_khalo", L.TSC, L.ptr(pos), L.ptr(vel), L.ptr(acc), 0.0, dt, npart, L.ptr(rho), L.ptr(khalo))


def halo_applied():
    halo_only()
    ctx.call("pm_khalo_apply", L.ptr(rho), L.ptr(khalo))


t_plain = timed(plain)
ref = rho.clone()
t_halo = timed(halo_only)
t_applied = timed(halo_applied)
err = float((rho - ref).abs().max() / ref.abs().max())
print(f"n={n}: plain flush {t_plain:.2f} ms; halo buffer {t_halo:.2f} ms (+ apply: {t_applied:.2f} ms); "
      f"max |rho_applied - rho_plain| / max = {err:.2e}")
