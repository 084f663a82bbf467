#!/usr/bin/env python3
"""Headline benchmark: particles pushed per second for one complete PM step
(= one integration.integrate() call including its solver.pm()) -- Newtonian, TSC, FFT Poisson,
order-5 gradient, Np = N^3, fp32, device-resident state (BASELINE.json / SURVEY.md section 8d).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--ncells 1024]

Prints ONE JSON line (rank 0).  `value` = particles * K / wall time of the K timed steps (max over
ranks, barrier + synchronize on both sides).  `roofline` describes the dominant C-ABI stage,
timed with HIP events on the launch stream inside the timed region; `cpu_baseline` times the CPU
oracle (oracle/, a restatement of the reference's Numba path -- NOT PySCo itself) on a bounded
sample of the same workload on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md); 6.29e12 measured float4 copy
HBM_COPY = 6.3e12    # what a float4 copy kernel sustains (torch add / fill: 6.0-6.9 TB/s here)
B_ALG_PER_PARTICLE = 200.0  # fused-ideal bytes per particle-step, fp32, Np = N^3


def synth_tables():
    """Monotone stand-ins for tables[0]=lna(t), [1]=t(lna), [3]=D1(lna): Einstein-de Sitter, a = (1 + t)^(2/3),
    tabulated on a grid uniform in ln a that is long enough for hundreds of steps of 10 % growth."""
    from scipy.interpolate import interp1d

    lna = np.linspace(np.log(0.02), np.log(0.5) + 400 * np.log(1.1), 60000)
    a = np.exp(lna)
    t = a**1.5 - 1.0
    return [interp1d(t, lna, fill_value="extrapolate"), interp1d(lna, t, fill_value="extrapolate"), None,
            interp1d(lna, a, fill_value="extrapolate")] + [None] * 9


CONFIGS = {
    # name: (default mesh size, param overrides) -- BASELINE.json configs; newton_fft is the headline (configs 2 / 3)
    "newton_fft": (1024, {}),
    "newton_mg": (128, {"linear_newton_solver": "multigrid"}),  # config 1: examples/param.ini semantics
    "fr": (512, {"theory": "fr", "fR_n": 1, "fR_logfR0": 5, "linear_newton_solver": "multigrid"}),  # config 4
    "mond": (512, {"theory": "mond", "mond_function": "simple", "mond_g0": 1.2, "mond_alpha": 1,
                   "mond_scale_factor_exponent": 0, "linear_newton_solver": "multigrid"}),  # config 5 (on one GPU)
}


def make_param(n, npart, config="newton_fft"):
    ncoarse = int(np.log2(n))
    p = {"theory": "newton", "mass_scheme": "TSC", "npart": npart, "ncoarse": ncoarse, "Om_m": 0.25733,
         "Om_lambda": 0.742589237, "H0": 72.0, "boxlen": 100.0 * (n / 128), "aexp": 0.5, "aexp_old": 0.49,
         "linear_newton_solver": "fft", "gradient_stencil_order": 5, "save_power_spectrum": "no", "nthreads": 1,
         "Courant_factor": 1.0, "max_aexp_stepping": 10, "integrator": "leapfrog", "write_snapshot": False,
         "nsteps": 0, "n_reorder": 50}
    p.update({"w0": -1.0, "wa": 0.0, "Npre": 2, "Npost": 1, "epsrel": 1e-2})
    p.update(CONFIGS[config][1])
    if config == "fr":
        p["boxlen"] = 1000.0 * (n / 32)  # 31 Mpc/h cells: below that the reference's cubic root itself returns NaN
    tables = synth_tables()
    p["t"] = float(tables[1](np.log(p["aexp"])))
    if config != "newton_fft":  # (the oracle and the f(R) / MOND prefactors need the unit system)
        from pysco_amd import utils as _u

        _u.set_units(p)
    return p, tables


def zeldovich_device(n, seed=42, rms_cells=0.5, vel_factor=0.02):
    """Synthetic ICs built on the GPU with torch (input synthesis only, not the product path):
    cell-centred lattice + Zel'dovich displacement from a k^-2 Gaussian field truncated at k_Nyq/2,
    rms |psi| = rms_cells cells, v = vel_factor * psi.  Returns (pos, vel) float32 [n^3, 3] on cuda."""
    import torch

    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    kf = torch.fft.fftfreq(n, d=1.0 / n, device=dev)
    kz = torch.fft.rfftfreq(n, d=1.0 / n, device=dev)
    k2 = kf[:, None, None] ** 2 + kf[None, :, None] ** 2 + kz[None, None, :] ** 2
    k2[0, 0, 0] = 1.0
    amp = torch.where(k2.sqrt() <= n / 4, 1.0 / k2, torch.zeros_like(k2)).sqrt()
    amp[0, 0, 0] = 0.0
    delta_k = torch.complex(torch.randn(k2.shape, generator=g, device=dev), torch.randn(k2.shape, generator=g, device=dev))
    delta_k *= amp / k2
    del amp, k2
    npart = n**3
    vel = torch.empty((npart, 3), dtype=torch.float32, device=dev)
    for ax, kk in enumerate((kf[:, None, None], kf[None, :, None], kz[None, None, :])):
        psi = torch.fft.irfftn(1j * kk * delta_k, s=(n, n, n))
        vel[:, ax] = psi.reshape(-1)
        del psi
    del delta_k
    rms = float((vel.double() ** 2).sum(dim=1).mean().sqrt())
    vel *= (rms_cells / n) / rms
    lat = (torch.arange(n, device=dev, dtype=torch.float32) + 0.5) / n
    pos = vel.clone()
    pos.view(n, n, n, 3)[..., 0] += lat[:, None, None]
    pos.view(n, n, n, 3)[..., 1] += lat[None, :, None]
    pos.view(n, n, n, 3)[..., 2] += lat[None, None, :]
    vel *= vel_factor
    return pos, vel


# algorithmic HBM bytes per launch of each C-ABI stage (compulsory traffic)
def alg_bytes(n, npart):
    cells = float(n) ** 3
    spec = float(n) * n * (n // 2 + 1) * 8.0
    return {
        "pm_kick_drift_wrap": 60.0 * npart,          # read x,v,a; write x,v
        "pm_deposit": 12.0 * npart + 4.0 * cells,    # read x; write rho
        "pm_kick_drift_wrap_deposit": 60.0 * npart + 4.0 * cells,  # the two above in one kernel (x read once)
        "pm_kick_drift_wrap_deposit_khalo": 60.0 * npart + 4.0 * cells,  # same work, k-halo cells via a side buffer
        "pm_linear_operator": 8.0 * cells,           # rhs scaling (fused into the FFT in the ideal count)
        "pm_fft_r2c": 24.0 * cells,                  # 3 axis passes, read+write
        "pm_greens": 2.0 * spec,                     # read+write half spectrum (0 in the fused ideal)
        "pm_fft_c2r": 24.0 * cells,
        "pm_poisson_fft_fused": 48.0 * cells,       # r2c + Green's function + c2r in five LDS-FFT passes
        "pm_poisson_fft_fused_khalo": 48.0 * cells,  # the same, the z pass also folding the k-halo buffer in
        "pm_gradient": 16.0 * cells,                 # read phi, write F(3)
        "pm_interp": 24.0 * npart + 12.0 * cells,    # read x, F; write a
        "pm_interp_kick": 48.0 * npart + 12.0 * cells,  # gather + second kick + CFL maxima: read x,v,F; write v,a
        "pm_gradient_interp_kick": 48.0 * npart + 4.0 * cells,  # the same reading phi instead of F (gradient fused)
        # ... and writing the arrays out of place in tile order: the same algorithmic work (the 12 B/particle of the
        # re-written position are this design's overhead, not compulsory traffic)
        "pm_gradient_interp_kick_rebucket": 48.0 * npart + 4.0 * cells,
        # "the kick re-buckets": the gather stays in place and decides, the fused kick writes the new stream
        "pm_gradient_interp_kick_codes": 48.0 * npart + 4.0 * cells,
        "pm_kick_drift_wrap_deposit_rebucket": 60.0 * npart + 4.0 * cells,
        "pm_axpy": 36.0 * npart,                     # second kick: read v,a; write v
        "pm_max_abs": 12.0 * npart,                  # CFL reductions (two per step)
        # the fp64 device mode (8-byte elements; its gradient is a pass of its own: read phi, write F(3))
        "pm_real64_kick_drift_wrap": 120.0 * npart,
        "pm_real64_deposit": 24.0 * npart + 8.0 * cells,
        "pm_real64_kick_drift_wrap_deposit": 120.0 * npart + 8.0 * cells,
        "pm_real64_scale_add": 16.0 * cells,
        "pm_real64_poisson_fft": 96.0 * cells,
        "pm_real64_gradient": 32.0 * cells,
        "pm_real64_interp_kick": 96.0 * npart + 24.0 * cells,
        "pm_real64_max_abs": 24.0 * npart,
    }


def alg_bytes_level(stage):
    """Algorithmic bytes of one call of a per-level multigrid stage "name@n" (every logical
    array read once and written once): None if the stage is not one."""
    name, _, size = stage.partition("@")
    if not size:
        return None
    c = float(int(size)) ** 3
    per_cell = {"pm_lap_gauss_seidel": 12.0, "pm_cubic_gauss_seidel": 12.0, "pm_quartic_gauss_seidel": 12.0,  # x, b -> x
                "pm_lap_restrict_residual": 8.5, "pm_lap_operator": 8.0, "pm_lap_residual_error": 8.0,
                "pm_cubic_operator": 12.0, "pm_quartic_operator": 12.0, "pm_cubic_residual_error": 8.0,
                "pm_quartic_residual_error": 8.0, "pm_restriction": 4.5, "pm_mond_rhs": 8.0,
                "pm_prolongation": 8.0 * 8 + 4.0,  # (size = the coarse grid's: 8 fine cells read + written, 1 coarse read)
                # one V-cycle: (Npre + Npost) sweeps of 12 B + restricted residual 16 B + prolongation 12 B per cell,
                # summed over the levels (x 8/7), Npre + Npost = 3
                "pm_lap_vcycle": (3 * 12.0 + 16.0 + 12.0) * 8.0 / 7.0,
                # one FAS V-cycle of the scalaron (multigrid.py:297-376; SURVEY.md section 8d, "73-82 B/cell" per level):
                # (Npre + Npost) = 3 sweeps of u, b, rhs -> u (16 B), the restriction of u (4.5 B), the coarse right-hand
                # side L(R u) - R(residual) and the correction u += P(e_c - R u): the upper figure of SURVEY's range,
                # 82 B/cell on a level, x 8/7 over the levels
                "pm_fr_vcycle": 82.0 * 8.0 / 7.0}.get(name)
    return None if per_cell is None else per_cell * c


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cpu_n, budget_s=15.0, config="newton_fft"):
    """CPU oracle (C restatement + FFT) timed on leapfrog steps of the same workload at cpu_n^3: first on one
    core (the parity checker's serial build, numpy.fft), then on all the cores this process may use (OpenMP
    build: atomic-add deposit as the reference's parallel TSC, scipy.fft with workers).  The reported value is
    the multi-core one.  Bounded: as many steps as fit in about `budget_s` seconds per leg."""
    from oracle import pysco_oracle as orc

    pos_d, vel_d = zeldovich_device(cpu_n)
    pos0, vel0 = pos_d.cpu().numpy(), vel_d.cpu().numpy()
    del pos_d, vel_d
    orc.periodic_wrap(pos0)
    # a one-GPU box's CPU share is 16 cores; more threads than that only fight over the atomics
    cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    orc.set_num_threads(cores)

    def leg(omp):
        orc.use_openmp(omp)
        pos, vel = pos0.copy(), vel0.copy()
        param, tables = make_param(cpu_n, cpu_n**3, config)
        acc, phi, extra = orc.pm(pos, param, tables=tables)
        steps, t0 = 0, time.perf_counter()
        while True:
            pos, vel, acc, phi, extra = orc.integrate(pos, vel, acc, phi, extra, tables, param, np.float32(1))
            steps += 1
            dt = time.perf_counter() - t0
            if dt > budget_s or steps >= 200 or (steps >= 2 and dt + dt / steps > 1.3 * budget_s):
                return steps, dt

    # The second half of BASELINE.json's metric, AFTER stepping: the same K leapfrog steps on both sides from the same
    # particles, then the binned, window-deconvolved P(k) (TSC deposit -> r2c -> fourier.py:22-100) of each side's own
    # particles with each side's own estimator; and the HIP estimator against a float64 evaluation of the same
    # pipeline on the same (device-evolved) particles (<= 1e-4 per bin).
    import torch

    from pysco_amd import fourier, integration, mesh, solver

    K = 10
    orc.use_openmp(True)
    try:
        pos_c, vel_c = pos0.copy(), vel0.copy()
        par_c, tables = make_param(cpu_n, cpu_n**3, config)
        acc_c, phi_c, ex_c = orc.pm(pos_c, par_c, tables=tables)
        for _ in range(K):
            pos_c, vel_c, acc_c, phi_c, ex_c = orc.integrate(pos_c, vel_c, acc_c, phi_c, ex_c, tables, par_c, np.float32(3e38))
        _, pk_cpu, nm = orc.fourier_grid_to_Pk(orc.fft_3D_real(orc.TSC(pos_c, cpu_n)), 3)
    finally:
        orc.use_openmp(False)
    # the device side runs the very loop the headline times: Morton-sorted stream, prediction + incremental re-sort in
    # every step (param["rebucket_straggler_fraction"] as pysco_amd.main sets it), the order policy after every step;
    # int32 tags follow the particles so that the two sides are compared particle by particle
    from pysco_amd import main as pm_main
    from pysco_amd import utils as pm_utils

    pos_g, vel_g = torch.from_numpy(pos0).cuda(), torch.from_numpy(vel0).cuda()
    par_g, tables = make_param(cpu_n, cpu_n**3, config)
    order = []
    pos_g, vel_g = pm_utils.reorder_particles(pos_g, vel_g, _order_out=order)
    par_g["particle_id"] = torch.arange(cpu_n**3, dtype=torch.int32, device="cuda")[order[0]].contiguous()
    par_g["rebucket_straggler_fraction"] = pm_main.REBUCKET_FRACTION
    acc_g, phi_g, ex_g = solver.pm(pos_g, par_g, tables=tables)
    pk_stragglers = []
    for _ in range(K):
        par_g["nsteps"] += 1
        pos_g, vel_g, acc_g, phi_g, ex_g = integration.integrate(pos_g, vel_g, acc_g, phi_g, ex_g, tables, par_g,
                                                                 np.float32(3e38))
        pos_g, vel_g, acc_g = pm_main.reorder_policy(pos_g, vel_g, acc_g, par_g)
        pk_stragglers.append(pm_utils.straggler_fraction(cpu_n)[0])
    _, pk_gpu, _ = fourier.fourier_grid_to_Pk(fourier.fft_3D_real(mesh.TSC(pos_g, cpu_n)), 3)
    inv = torch.argsort(par_g["particle_id"].long())
    pos_gh = pos_g[inv].cpu().numpy()  # (back in the CPU side's order)
    del inv
    dx = np.abs(pos_gh - pos_c)
    max_dx = float(np.minimum(dx, 1 - dx).max())
    del pos_g, vel_g, acc_g, phi_g, ex_g, dx
    _, pk_64, _ = orc.pk_fp64(pos_gh, cpu_n, 3)
    ok = nm > 0
    pk_rel_err = float(np.max(np.abs(pk_gpu[ok] - pk_cpu[ok]) / np.abs(pk_cpu[ok])))
    pk_rel_err_fp64 = float(np.max(np.abs(pk_gpu[ok].astype(np.float64) - pk_64[ok]) / np.abs(pk_64[ok])))

    s1, t1 = leg(False)
    try:
        sn, tn = leg(True)
        threads = orc.num_threads()
    finally:
        orc.use_openmp(False)
    return {"value": cpu_n**3 * sn / tn, "unit": "particles/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
            "cores_visible": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count(),
            "value_1core": cpu_n**3 * s1 / t1,
            "pk_rel_err": {"value": pk_rel_err, "vs_fp64": pk_rel_err_fp64, "bins": int(ok.sum()), "steps": K,
                           "max_dx": max_dx, "device_loop": "sorted stream, prediction + re-bucket every step",
                           "device_stragglers_max": None if not pk_stragglers or pk_stragglers[0] is None
                           else float(max(pk_stragglers)),
                           "what": f"after {K} leapfrog steps of the {cpu_n}^3 sample on both sides: max over bins of "
                                   "|P_hip - P_cpu| / P_cpu (each side's particles and estimator; TSC, deconvolved, "
                                   "p = 3); vs_fp64: the HIP estimator against a float64 evaluation of the same "
                                   "pipeline on the device-evolved particles; max_dx: largest position difference "
                                   "between the two sides (box units)"},
            "sample": f"leapfrog steps at {cpu_n}^3 particles / {cpu_n}^3 mesh ({config}: TSC, order 5) with the CPU "
                      f"restatement of PySCo's Numba path (oracle/pm_oracle.c): {sn} steps in {tn:.1f} s on {threads} "
                      f"threads (OpenMP, atomic deposit, scipy.fft workers), {s1} steps in {t1:.1f} s on 1 core "
                      "(serial build, numpy.fft)"}


def launch_ranks(n):
    """python bench.py --gpus N without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py
    <the same arguments>` as a child (one rank per GPU, rendezvous on 127.0.0.1) and return its exit code.  The ranks
    inherit stdout, so rank 0's JSON line is this command's output."""
    import socket
    import subprocess

    with socket.socket() as sock:  # a free port for the rendezvous
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC: what RCCL needs on this driver)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


# the short legs the default invocation runs after the headline, each in a child process of its own (a clean device,
# and a failure there cannot take the headline with it): the other BASELINE.json configs on one GPU and the headline's
# "clustered" workload (rms displacement 3 cells)
SECONDARY = [
    ("newton_fft_512", ["--config", "newton_fft", "--ncells", "512", "--steps", "20", "--warmup", "5"]),   # config 2 in fp32
    ("newton_fft_512_f64", ["--config", "newton_fft", "--ncells", "512", "--dtype", "f64", "--steps", "10", "--warmup", "3"]),  # config 2 as stated
    ("newton_mg_128", ["--config", "newton_mg", "--steps", "20", "--warmup", "5"]),                        # config 1
    ("fr_512", ["--config", "fr", "--steps", "8", "--warmup", "3"]),                                       # config 4
    ("mond_512", ["--config", "mond", "--steps", "8", "--warmup", "3"]),                                   # config 5 on one GPU
    ("newton_fft_1024_clustered", ["--config", "newton_fft", "--rms-cells", "3.0", "--steps", "10", "--warmup", "3"]),
]


def run_secondary(budget_s=420.0):
    """Runs the SECONDARY legs as children (`bench.py --leg NAME ...`: no CPU baseline, no children of their own) and
    returns {name: compact result}.  Bounded: a leg that would start after `budget_s` seconds is skipped and says so."""
    import subprocess

    out, t0 = {}, time.perf_counter()
    for name, leg_args in SECONDARY:
        if time.perf_counter() - t0 > budget_s:
            out[name] = {"skipped": "time budget of the default run spent"}
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-cpu-baseline", "--leg", name] + leg_args
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            out[name] = json.loads(line[-1]) if r.returncode == 0 and line else {
                "failed": r.returncode, "stderr_tail": r.stderr[-400:]}
        except subprocess.TimeoutExpired:
            out[name] = {"failed": "timeout (300 s)"}
    return out


XGMI_PEAK = 7 * 153.0e9  # B/s per GPU and direction: 7 links x 153 GB/s


def comm_report(dist, slab, n, world, steps):
    """N > 1: what the step's exchanges moved, per rank and step, and what the all-to-all of this rank's share of the
    half-spectrum achieves on its own -- the collective timed alone, same tensor the slab FFT exchanges per kz chunk --
    against the xGMI peak of one GPU."""
    import torch

    cs = dict(slab.comm_stats)
    nh = n // 2 + 1
    chunk = torch.zeros((n, (nh + 3) // 4, n // world), dtype=torch.complex64, device="cuda")  # [ky][kz chunk][x local]
    reps = 5
    slab.all_to_all(chunk)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        slab.all_to_all(chunk)
    torch.cuda.synchronize()
    dist.barrier()
    dt = (time.perf_counter() - t0) / reps
    sent = chunk.numel() * chunk.element_size() * (world - 1) / world
    tw = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(tw, op=dist.ReduceOp.MAX)
    dt = float(tw.item())
    for k in cs:
        slab.comm_stats[k] = cs[k]
    return {"backend": dist.get_backend(), "alltoall_bytes_per_rank_per_step": cs["a2a_bytes"] / steps,
            "alltoall_calls_per_step": cs["a2a_calls"] / steps, "ring_bytes_per_rank_per_step": cs["ring_bytes"] / steps,
            "alltoall_alone": {"bytes_sent_per_rank": sent, "ms": round(dt * 1e3, 4), "GBps_per_rank": round(sent / dt / 1e9, 2),
                               "xgmi_peak_GBps": XGMI_PEAK / 1e9, "frac_of_xgmi_peak": round(sent / dt / XGMI_PEAK, 4),
                               "what": "all_to_all_single of one kz chunk of this rank's half-spectrum, timed alone "
                                       f"({reps} calls, max over ranks)"},
            "alltoall_ms_per_step_at_that_rate": round(cs["a2a_bytes"] / steps / max(sent / dt, 1.0) * 1e3, 3)}


def compact(out):
    """What a secondary leg reports: step time, throughput, the dominant stage against the roofline, V-cycles,
    the straggler series in brief."""
    r = out["roofline"]
    ss = out.get("straggler_series") or []
    return {"workload": out["config"]["workload"], "ms_per_step": round(out["ms_per_step"], 4), "value": out["value"],
            "unit": out["unit"], "steps": out["steps"], "warmup": out["warmup"], "dtype": out["dtype"],
            "dominant_stage": r["kernel"], "frac": r["frac"], "achieved_GBps": r["achieved"], "step_frac": r["step_frac"],
            "vcycles_per_step": out["config"].get("vcycles_per_step", {}),
            "stages": [{k: st[k] for k in ("stage", "calls_per_step", "avg_ms", "alg_GBps")} for st in r["stages"][:5]],
            "ms_first_last": [out["per_step"][0]["wall_ms"], out["per_step"][-1]["wall_ms"]] if out.get("per_step") else None,
            "stragglers_first_last_max": [ss[0], ss[-1], max(ss)] if ss else None}


def cpu_baseline_f64(cpu_n, budget_s=12.0):
    """The fp64 CPU path (oracle/pysco_oracle_f64.py: the C restatement with REAL = double, OpenMP, numpy's float64 FFT)
    timed on leapfrog steps of the same workload at cpu_n^3, and the fp64 device mode held against it after the same
    steps: positions and P(k) (<= 1e-10)."""
    import torch

    from oracle import pysco_oracle as o32
    from oracle import pysco_oracle_f64 as o64
    from pysco_amd import integration, real64, solver

    pos_d, vel_d = zeldovich_device(cpu_n)
    pos0, vel0 = pos_d.double().cpu().numpy(), vel_d.double().cpu().numpy()
    del pos_d, vel_d
    o64.periodic_wrap(pos0)
    par_c, tables = make_param(cpu_n, cpu_n**3)
    o32.set_units(par_c)
    par_g = dict(par_c)
    pc, vc = pos0.copy(), vel0.copy()
    ac, _ = o64.pm(pc, par_c)
    steps, t0 = 0, time.perf_counter()
    while True:
        pc, vc, ac, _, _ = o64.integrate(pc, vc, ac, tables, par_c, 3e38)
        steps += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or steps >= 10:
            break
    pg, vg = torch.from_numpy(pos0).cuda(), torch.from_numpy(vel0).cuda()
    ag, phig, exg = solver.pm(pg, par_g, tables=tables)
    for _ in range(steps):
        pg, vg, ag, phig, exg = integration.integrate(pg, vg, ag, phig, exg, tables, par_g, 3e38)
    dx = np.abs(pg.cpu().numpy() - pc)
    max_dx = float(np.minimum(dx, 1 - dx).max())
    dens = torch.empty((cpu_n,) * 3, dtype=torch.float64, device="cuda")
    L_ = __import__("pysco_amd._lib", fromlist=["x"])
    L_.context(cpu_n, 8).call("pm_real64_deposit", L_.TSC, L_.ptr(pg), pg.shape[0], L_.ptr(dens))
    par_g["MAS_index"] = 3
    _, pk_g, _ = real64.power_spectrum(dens, par_g)
    _, pk_c, _ = o64.pk_of_density(o64.deposit("tsc", pc, cpu_n), 3)
    pk_c = pk_c * (par_g["boxlen"] / cpu_n**2) ** 3
    pk_rel = float(np.max(np.abs(pk_g / pk_c - 1)))
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return {"value": cpu_n**3 * steps / dt, "unit": "particles/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
            "pk_rel_err": {"value": pk_rel, "max_dx": max_dx, "steps": steps,
                           "what": f"fp64 device mode vs fp64 CPU path after {steps} leapfrog steps at {cpu_n}^3: max over "
                                   "bins of |P_dev / P_cpu - 1| and the largest position difference (box units)"},
            "sample": f"{steps} leapfrog steps at {cpu_n}^3 in {dt:.1f} s with the fp64 CPU path (oracle/pm_oracle.c, REAL = double, "
                      "OpenMP; numpy float64 FFT)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="newton_fft", choices=sorted(CONFIGS),
                    help="BASELINE.json workload: newton_fft (headline, default), newton_mg (config 1), fr (4), mond (5)")
    ap.add_argument("--ncells", type=int, default=0, help="mesh cells per side (default: the config's)")
    ap.add_argument("--rms-cells", type=float, default=0.5)
    ap.add_argument("--cpu-ncells", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-placement", action="store_true", help="skip pysco_amd.main.tune_placement (development)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"],
                    help="f64: the fp64 device mode of the Newtonian FFT step (BASELINE.json config 2 as stated: float64 "
                         "arrays, pm_real64_* entry points, 400 B per particle-step)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="headline only: skip the short legs of the other BASELINE configs / workloads (profiling runs)")
    ap.add_argument("--full-series", action="store_true", help="per_step for EVERY timed step (long runs), not the first / last five")
    ap.add_argument("--leg", default="", help=argparse.SUPPRESS)  # a secondary leg run as a child: compact line, no children
    ap.add_argument("--rebucket-fraction", type=float, default=None,
                    help="param['rebucket_straggler_fraction'] of the timed loop (default: pysco_amd.main's; "
                         "negative: off, i.e. the reference's in-place step + a full sort every n_reorder steps)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        # not started by a launcher: start one rank per GPU ourselves -- BEFORE anything touches the GPU (this process
        # never does) -- and pass rank 0's line and the launcher's exit code on
        raise SystemExit(launch_ranks(args.gpus))

    import torch
    dev_index = local_rank % max(1, torch.cuda.device_count())  # (== local_rank on a full node)
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist

        backend = os.environ.get("PMSTEP_DIST_BACKEND", "nccl")  # "gloo": rehearsal of this path on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    from pysco_amd import _lib as L
    from pysco_amd import integration, real64, solver, utils
    from pysco_amd import main as pm_main

    n = args.ncells or CONFIGS[args.config][0]
    npart = n**3
    pos, vel = zeldovich_device(n, rms_cells=args.rms_cells)
    utils.periodic_wrap(pos)
    if world > 1:  # x-slab decomposition: keep the lattice planes of this rank, then hand strays to their owners
        from pysco_amd import slab

        if os.environ.get("PMSTEP_SLAB_REBUCKET") == "0":  # development A/B: ranks never re-sort their streams
            slab._SLAB_REBUCKET = False

        sl = slab.get(n)
        lo, hi = sl.x_lo * n * n, (sl.x_lo + sl.nx) * n * n
        pos, vel = pos[lo:hi].clone(), vel[lo:hi].clone()
        torch.cuda.empty_cache()
        pos, vel = sl.migrate(pos, vel)
    pos, vel = utils.reorder_particles(pos, vel)  # Morton-sorted once before timing
    f64 = args.dtype == "f64"
    if f64:
        if world > 1 or args.config != "newton_fft":
            raise SystemExit("--dtype f64: the fp64 device mode is the single-GPU Newtonian FFT step")
        pos, vel = pos.double(), vel.double()  # (float64 arrays: solver.pm / integrate dispatch to the fp64 mode)
    L._ctxs.clear()  # drop the sort workspace
    if world > 1:
        slab._slabs.clear()
    torch.cuda.empty_cache()
    param, tables = make_param(n, npart, args.config)
    # the product's own loop body (pysco_amd.main.run_nbody): integrate() + the particle-order policy, with its
    # incremental re-sorts and any full re-sort INSIDE the timed region
    rebucket = pm_main.REBUCKET_FRACTION if args.rebucket_fraction is None else args.rebucket_fraction
    if world == 1 and rebucket >= 0:  # (fp64 mode: the key switches on the tile re-sort inside integrate, real64.integrate)
        param["rebucket_straggler_fraction"] = rebucket
    acc, phi, extra = solver.pm(pos, param, tables=tables)
    t_far = np.float32(3.0e38)  # no snapshot inside the run: every step takes its own dt
    if "rebucket_straggler_fraction" in param and not f64:
        # the product loop's set-up (pysco_amd.main.run_nbody): particle arrays and the gather's output set in the
        # best-placed allocations it can find, no allocation inside the step loop
        if npart >= 2**24 and not args.no_placement:
            pos, vel, acc = pm_main.tune_placement(pos, vel, acc)
        pm_main.reserve_step_buffers(pos)

    resort_at = int(os.environ.get("PMSTEP_EXP_RESORT_AT", "0"))
    if os.environ.get("PMSTEP_KICK_REBUCKETS") == "0":  # development A/B: the re-sort inside the gather (round 2's scheme)
        integration._KICK_REBUCKETS = False
    if os.environ.get("PMSTEP_VCYCLE_GRAPHS") == "0":  # development A/B
        from pysco_amd import multigrid as _mg

        _mg._VCYCLE_GRAPHS = False

    def step(pos, vel, acc, phi, extra):
        param["nsteps"] += 1
        pos, vel, acc, phi, extra = integration.integrate(pos, vel, acc, phi, extra, tables, param, t_far)
        if world == 1 and f64:  # (the fp64 mode: a sort to cell resolution once the gather reports an aged stream)
            pos, vel, acc = pm_main.reorder_policy(pos, vel, acc, param)
        if world == 1 and not f64:
            pos, vel, acc = pm_main.reorder_policy(pos, vel, acc, param)
            if resort_at and param["nsteps"] == resort_at:  # development: does a fresh full sort change the step time?
                pos, vel, acc = pm_main._reorder(pos, vel, acc, param)
        return pos, vel, acc, phi, extra

    for _ in range(args.warmup):
        pos, vel, acc, phi, extra = step(pos, vel, acc, phi, extra)

    if dist is not None:
        dist.barrier()
        for k in slab.comm_stats:
            slab.comm_stats[k] = 0
    torch.cuda.synchronize()
    L.profile_start()
    t0 = time.perf_counter()
    dt_series = []
    straggler_frac, step_events = [], [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    passes_per_chunk = []
    step_events[0].record()
    prof = None
    if os.environ.get("PMSTEP_HOST_PROFILE"):  # development: where does the host spend the timed loop
        import cProfile

        prof = cProfile.Profile()
        prof.enable()
    for i in range(args.steps):
        pos, vel, acc, phi, extra = step(pos, vel, acc, phi, extra)
        if world > 1:
            straggler_frac.append(utils.straggler_fraction(n)[0])  # of this rank's stream, as the gather found it
        if world == 1:
            if f64:  # (direct-served share, tile passes per chunk) of the gather; None right after a re-sort
                age = real64.stream_age(n)
                straggler_frac.append(None if age is None else age[0])
                passes_per_chunk.append(None if age is None else round(age[1], 2))
            else:
                straggler_frac.append(utils.straggler_fraction(n)[0])
            dt_series.append((float(param.get("_dt_last", 0.0)), float(param.get("_dt_pred", 0.0) or 0.0)))
        L.profile_mark()
        step_events[i + 1].record()
    if prof is not None:
        prof.disable()
        import pstats

        pstats.Stats(prof, stream=sys.stderr).sort_stats("cumulative").print_stats(25)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    wall = time.perf_counter() - t0
    stages, per_step = L.profile_stop(per_step=True)
    if dist is not None:  # the step is over when the slowest rank is
        tw = torch.tensor([wall], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())

    comm = None
    if dist is not None:
        comm = comm_report(dist, slab, n, world, args.steps)
    ms_per_step = wall / args.steps * 1e3
    value = npart * args.steps / wall
    assert bool(torch.isfinite(pos).all()) and bool(torch.isfinite(vel).all())

    ab = alg_bytes(n, npart / world)
    for k in ab:  # per-rank share of the grid stages in slab mode
        if k not in ("pm_kick_drift_wrap", "pm_deposit", "pm_interp", "pm_interp_kick", "pm_gradient_interp_kick",
                     "pm_gradient_interp_kick_rebucket", "pm_gradient_interp_kick_codes", "pm_kick_drift_wrap_deposit_rebucket",
                     "pm_kick_drift_wrap_deposit", "pm_kick_drift_wrap_deposit_khalo",
                     "pm_axpy", "pm_max_abs"):
            ab[k] /= world
    if world > 1:
        cells, spec = float(n) ** 3 / world, float(n) * n * (n // 2 + 1) * 8.0 / world
        ab.update({"pm_deposit": 12.0 * npart / world + 4.0 * cells, "pm_interp_kick": 48.0 * npart / world + 12.0 * cells,
                   "pm_fft2d_r2c": 16.0 * cells, "pm_fft2d_c2r": 16.0 * cells, "pm_fft1d_x": 2.0 * spec,
                   "pm_greens_slab": 2.0 * spec, "pm_gradient_slab": 16.0 * cells,
                   "pm_fftp_forward": 16.0 * cells, "pm_fftp_inverse": 16.0 * cells, "pm_fftp_xgreen": 2.0 * spec,
                   # pipelined form: z passes whole, y / x passes per kz chunk (4 chunks)
                   "pm_fftp_z_forward": 8.0 * cells, "pm_fftp_z_forward_khalo": 8.0 * cells, "pm_fftp_z_inverse": 8.0 * cells,
                   "pm_fftp_y_forward": 2.0 * spec / 4, "pm_fftp_y_inverse": 2.0 * spec / 4,
                   "pm_fftp_xgreen_kz": 2.0 * spec / 4,
                   "pm_gradient_interp_kick_rebucket_slab": 48.0 * npart / world + 4.0 * cells,
                   "pm_tile_lists_codes": 13.0 * npart / world, "pm_tile_lists": 12.0 * npart / world,
                   "pm_kick_drift_wrap_slab": 60.0 * npart / world,
                   "pm_kick_drift_wrap_deposit_slab": 60.0 * npart / world + 4.0 * cells,
                   "pm_kick_drift_wrap_deposit_slab_khalo": 60.0 * npart / world + 4.0 * cells})
    rank_spread = None
    if dist is not None:  # every rank's stage table: max / min over ranks of the per-call average
        gathered = [None] * world
        dist.all_gather_object(gathered, {k: v[1] / max(1, v[0]) for k, v in stages.items()})
        rank_spread = {k: [round(min(g.get(k, 0.0) for g in gathered), 4), round(max(g.get(k, 0.0) for g in gathered), 4)]
                       for k in stages}
    # the solves issued as one C call each (pm_lap_solve / pm_fr_solve: V-cycles + residual norms until the reference's
    # stopping rule says stop): algorithmic bytes of a call = its average number of cycles x (one V-cycle + one norm)
    solve_bytes = {}
    for name, (calls, _ms) in stages.items():
        base, _, size = name.partition("@")
        if base in ("pm_lap_solve", "pm_fr_solve") and size:
            key = "_vcycles_linear" if base == "pm_lap_solve" else "_vcycles_fas"
            per_cycle = alg_bytes_level(("pm_lap_vcycle@" if base == "pm_lap_solve" else "pm_fr_vcycle@") + size) + 8.0 * float(int(size)) ** 3
            total_calls = max(1, sum(c for nm, (c, _m) in stages.items() if nm.partition("@")[0] == base))
            # (cycles counted over warm-up + timed steps + the first pm(); calls over the timed steps: scale by the step counts)
            cyc = param.get(key, 0) * args.steps / max(1, args.steps + args.warmup + 1) / total_calls
            solve_bytes[name] = per_cycle * cyc
    table = []
    for name, (calls, ms) in sorted(stages.items(), key=lambda kv: -kv[1][1]):
        avg = ms / calls
        per_call = solve_bytes.get(name, ab.get(name, alg_bytes_level(name)))
        gbs = per_call / (avg * 1e-3) / 1e9 if per_call else None
        table.append({"stage": name, "calls_per_step": calls / args.steps, "avg_ms": round(avg, 4),
                      "alg_GBps": None if gbs is None else round(gbs, 1)})
    dom = table[0]
    achieved = dom["alg_GBps"] or 0.0
    b_alg = B_ALG_PER_PARTICLE * (2.0 if f64 else 1.0)  # (200 B per particle-step in fp32, 400 in fp64)
    # HBM bytes the PMC counters saw for the dominant stage's kernels, per launch of the stage, averaged over the
    # timed steps of the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (profiles/, tools/pmc_summary.py).
    # Only a summary collected with THIS command line (size, step counts, policy) describes this run: else null.
    traffic = None
    pmc_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r03_pmc_summary.json")
    if world == 1 and os.path.exists(pmc_path):
        pmc = json.load(open(pmc_path))
        same = (pmc.get("ncells") == n and pmc.get("steps") == args.steps and pmc.get("warmup") == args.warmup
                and pmc.get("rebucket_fraction") == param.get("rebucket_straggler_fraction")
                and pmc.get("rms_cells") == args.rms_cells)
        if same:
            traffic = pmc.get("stages", {}).get(dom["stage"], {}).get("hbm_bytes_per_call")
    roofline = {"bound": "hbm", "kernel": dom["stage"], "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": round(achieved / (HBM_PEAK / 1e9), 4), "traffic": traffic,
                "alg_bytes_per_launch": solve_bytes.get(dom["stage"], ab.get(dom["stage"], alg_bytes_level(dom["stage"]))),
                "step_frac": round(b_alg * npart / (wall / args.steps) / (HBM_PEAK * world), 4),
                "step_alg_bytes_per_particle": b_alg, "stages": table}
    # the same two figures against what a float4 copy reaches on this GPU (SURVEY.md section 8d asks for both
    # denominators), and the dominant stage's HBM-side rate from the counters
    roofline["peak_copy"] = HBM_COPY / 1e9
    roofline["frac_of_copy"] = round(achieved / (HBM_COPY / 1e9), 4)
    roofline["step_frac_of_copy"] = round(b_alg * npart / (wall / args.steps) / (HBM_COPY * world), 4)
    if traffic:
        roofline["traffic_GBps"] = round(traffic / (dom["avg_ms"] * 1e-3) / 1e9, 1)
    if dom["stage"].startswith("pm_gradient_interp_kick_rebucket"):
        # (the stage also does the re-sort: the position written again in tile order and the destination byte read
        # are compulsory traffic of THAT job, 13 B per particle; `frac` above counts the gather's bytes only)
        with_resort = roofline["alg_bytes_per_launch"] + 13.0 * npart / world
        roofline["frac_with_resort_bytes"] = round(with_resort / (dom["avg_ms"] * 1e-3) / HBM_PEAK, 4)

    solver_name = {"newton_fft": "Newtonian TSC+FFT", "newton_mg": "Newtonian TSC + multigrid", "fr": "f(R) n=1 FAS multigrid",
                   "mond": "QUMOND simple, multigrid"}[args.config]
    cycles = {k[len("_vcycles_"):]: round(param.get(k, 0) / max(1, args.steps + args.warmup + 1), 2)
              for k in ("_vcycles_linear", "_vcycles_fas") if k in param}
    out = {"metric": f"particles pushed/sec (one PM step), {solver_name}, {'fp64' if f64 else 'fp32'}", "value": value,
           "unit": "particles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": f"{args.config}: {n}^3 mesh / {n}^3 particles, TSC, "
                                  + ("FFT Poisson (compensated), " if args.config == "newton_fft" else
                                     f"multigrid Npre 2 Npost 1 epsrel 1e-2 (V-cycles per solve-step: {cycles}), ")
                                  + f"order-5 gradient, leapfrog KDK, Zel'dovich ICs rms {args.rms_cells} cell, "
                                  "Morton-sorted once, then the product loop's order policy inside the timed steps ("
                                  + (f"incremental tile re-sort above {param['rebucket_straggler_fraction']:g} stragglers"
                                     if "rebucket_straggler_fraction" in param and not f64 else
                                     "fp64 device mode: steps over 2048-particle chunks of the sorted stream (dense tiles through LDS, "
                                     f"sparse ones through L2), tile re-sort inside the step above {real64.RESORT_PASSES_PER_CHUNK:g} "
                                     "tile passes per chunk" if f64 else
                                     ("every rank's stream re-bucketed by its gather every step"
                                      + (", halo buffer of the fused kick + deposit on the slab" if slab._SLAB_KHALO else "")
                                      if world > 1 and slab._SLAB_REBUCKET else "in-place steps"))
                                  + f", full sort above {pm_main.REORDER_FRACTION:g})",
                      "ncells_1d": n, "npart": npart, "vcycles_per_step": cycles,
                      "parallelism": "1 GPU" if world == 1 else f"{world} x-slabs, RCCL all-to-all FFT transposes + ring halos"},
           "roofline": roofline}
    # the first and last five timed steps: straggler fraction (particles outside the tile the sort filed them
    # under) and milliseconds per C-ABI stage, so a drift of the step time with the age of the sort shows
    show = sorted(set(list(range(min(5, args.steps))) + list(range(max(0, args.steps - 5), args.steps))))
    if args.full_series:
        show = list(range(args.steps))
    out["per_step"] = [{"step": i + 1, "stragglers": None if not straggler_frac or straggler_frac[i] is None
                        else round(straggler_frac[i], 5),
                        "wall_ms": round(step_events[i].elapsed_time(step_events[i + 1]), 3),
                        "ms": {k[3:]: round(v, 3) for k, v in sorted(per_step[i].items(), key=lambda kv: -kv[1])[:3]}}
                       for i in show]
    if comm is not None:
        out["comm"] = comm
        out["stage_ms_min_max_over_ranks"] = rank_spread
        probe = getattr(slab.get(n), "placement_probe_ms", None)  # (rank 0's; every rank runs its own search)
        if probe:
            out["placement_probe_ms"] = probe
    if pm_main.last_placement_probe_ms:  # in-place pass over each candidate particle array [ms], ascending: the first six are in use
        out["placement_probe_ms"] = pm_main.last_placement_probe_ms
    if straggler_frac and any(f is not None for f in straggler_frac):  # every timed step: stragglers found, dt taken, dt guessed for the next
        out["straggler_series"] = [None if f is None else round(f, 5) for f in straggler_frac]
        if passes_per_chunk:
            out["passes_per_chunk_series"] = passes_per_chunk  # fp64 mode: the gather's tile passes per 2048-particle chunk (None: re-sorted)
        if dt_series:
            out["dt_series"] = [[float(f"{a:.6g}"), float(f"{b:.6g}")] for a, b in dt_series]
    if not args.no_cpu_baseline and rank == 0 and world == 1 and f64:
        del pos, vel, acc, phi, extra
        torch.cuda.empty_cache()
        out["cpu_baseline"] = cpu_baseline_f64(min(128, n))
        out["pk_rel_err"], out["pk_steps"] = out["cpu_baseline"]["pk_rel_err"]["value"], out["cpu_baseline"]["pk_rel_err"]["steps"]
    elif not args.no_cpu_baseline and rank == 0 and world == 1:
        del pos, vel, acc, phi, extra
        torch.cuda.empty_cache()
        out["cpu_baseline"] = cpu_baseline(min(args.cpu_ncells, n) if args.config == "newton_fft" else min(128, n),
                                           config=args.config)
        # the P(k) half of the metric as top-level keys (BASELINE.json: "... + P(k) rel-err")
        pk = out["cpu_baseline"]["pk_rel_err"]
        out["pk_rel_err"], out["pk_rel_err_fp64"], out["pk_steps"] = pk["value"], pk["vs_fp64"], pk["steps"]
    if args.leg:
        out = compact(out)
    elif (rank == 0 and world == 1 and not args.no_secondary and args.config == "newton_fft" and not args.ncells
          and args.rms_cells == 0.5):
        # the default invocation: everything this process holds on the device goes first, then the other workloads
        pos = vel = acc = phi = extra = None
        L._ctxs.clear()
        from pysco_amd import mesh as _mesh

        for pool in ("_rebucket_pool", "_pool"):
            if hasattr(_mesh, pool) and hasattr(getattr(_mesh, pool), "clear"):
                getattr(_mesh, pool).clear()
        torch.cuda.empty_cache()
        out["secondary"] = run_secondary()
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
