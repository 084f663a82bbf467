"""The fp64 device mode of the Newtonian FFT step (BASELINE.json config 2): solver.pm and
integration.integrate for float64 particle arrays, through the pm_real64_* entry points of include/pmstep.h.

The reference is float32 end to end (mesh.py:2467, fourier.py:21) -- it has no such mode.  This one evaluates the same
formulas in double (arrays, Green's-function tables and every scalar: dt, half_dt, the density scaling), so that the
device can be held against a CPU evaluation of the very same arithmetic to 1e-10, which
separates rounding from algorithm.  solver.pm / integration.integrate dispatch here when `position` is float64
(a torch.float64 CUDA tensor: device-resident, or a NumPy float64 array: uploaded and downloaded per call).

Covered: theory = newton, linear_newton_solver = fft / fft_7pt, mass_scheme = TSC / CIC (NGP with param["allow_ngp"]),
gradient_stencil_order 2 / 3 / 5 / 7, the leapfrog integrator, P(k) output.  Anything else raises NotImplementedError:
there is no silent fall-back to float32.  Citations: /root/reference/pysco/.
"""
from __future__ import annotations

import ctypes as C
import logging

import numpy as np

from . import _lib as L
from . import iostream, utils


RESORT_PASSES_PER_CHUNK = 2.6  # default of param["resort_passes_per_chunk"]: a sorted stream makes ~2 (integrate)


def is_float64(x) -> bool:
    if L.is_device_array(x):
        return str(x.dtype) == "torch.float64"
    return isinstance(x, np.ndarray) and x.dtype == np.float64


def _dev(x):
    torch = L.require_gpu()
    return L.to_device(x, torch.float64)


def _kfreq(n: int) -> np.ndarray:
    i = np.arange(n)
    return np.where(i >= n // 2, i - n, i).astype(np.float64)


def green_tables(n: int, solver: str):
    """fourier.py:502-544 (compensated: g = c0 (s_i s_j s_k)^(-2p) / (k2_i + k2_j + k2_k)) and :555-595 (seven-point:
    g = c0 / (s_i + s_j + s_k)), per-axis tables in float64."""
    k = _kfreq(n)
    if solver == "fft":
        return L.GREEN_COMPENSATED, np.ascontiguousarray(k**2), np.ascontiguousarray(np.sinc(k / n)), -0.25 / np.pi**2
    return L.GREEN_7PT, np.zeros(n), np.ascontiguousarray(np.sin(np.pi / n * k) ** 2), -0.25 / n**2


def _scheme(param):
    s = param["mass_scheme"].casefold()
    if s not in ("cic", "tsc") and not (s == "ngp" and param.get("allow_ngp", False)):
        raise NotImplementedError(f"{param['mass_scheme']=}, should be 'CIC' or 'TSC'")
    return {"ngp": L.NGP, "cic": L.CIC, "tsc": L.TSC}[s]


def _check(param) -> str:
    solver = param["linear_newton_solver"].casefold()
    if param["theory"].casefold() != "newton" or solver not in ("fft", "fft_7pt"):
        raise NotImplementedError("the fp64 device mode covers theory = newton with linear_newton_solver = fft / fft_7pt "
                                  f"({param['theory']=}, {param['linear_newton_solver']=}); use float32 arrays otherwise")
    if param["gradient_stencil_order"] not in (2, 3, 5, 7):
        raise ValueError(f"gradient_order={param['gradient_stencil_order']}, should be 2, 3, 5 or 7")
    return solver


def power_spectrum(density, param):
    """(k, P(k), Nmodes) of a density grid: fourier.fourier_grid_to_Pk(fft_3D_real(density), MAS_index) with the
    normalisation of solver.py:134-140.  `density` (a float64 CUDA tensor) may be overwritten."""
    n = density.shape[0]
    ctx = L.context(n, 8)
    sinc = np.ascontiguousarray(np.sinc(_kfreq(n) / n))
    ks, ps, nm = (np.zeros(n, dtype=np.float64) for _ in range(3))
    ctx.call("pm_real64_pk", int(param["MAS_index"]), L.ptr(density), L.hptr(sinc), L.hptr(ks), L.hptr(ps), L.hptr(nm))
    kmax = int(2 * (n // 2) / 3)
    with np.errstate(invalid="ignore", divide="ignore"):
        k, pk, nmodes = ks[1:kmax] / nm[1:kmax], ps[1:kmax] / nm[1:kmax], nm[1:kmax]
    return k * (2 * np.pi / param["boxlen"]), pk * (param["boxlen"] / n**2) ** 3, nmodes


def pm(position, param, potential=None, additional_field=None, tables=(), _kick=None, _density=None):
    """solver.pm (solver.py:30-215) in float64.  Returns (acceleration, potential, additional_field): the kinds of the
    input (NumPy in, NumPy out); `_kick = (velocity, half_dt)` (private) applies the second kick of the leapfrog in
    the gather's pass and leaves max|a|, max|v| in param["_max_f64"]."""
    torch = L.require_gpu()
    solver = _check(param)
    scheme = _scheme(param)
    n = 2 ** int(param["ncoarse"])
    param["MAS_index"] = {L.NGP: 1, L.CIC: 2, L.TSC: 3}[scheme]
    param["parametrized_mu_z"] = 1.0
    pos_d = _dev(position)
    npart = pos_d.shape[0]
    ctx = L.context(n, 8)
    if _density is not None:  # (private: deposited by integrate()'s fused kick + drift + deposit)
        density = _density
    else:
        density = torch.empty((n, n, n), dtype=torch.float64, device=pos_d.device)
        ctx.call("pm_real64_deposit", scheme, L.ptr(pos_d), npart, L.ptr(density))
    if n**3 != param["npart"]:
        ctx.call("pm_real64_scale_add", L.ptr(density), float(n**3 / param["npart"]), 0.0, n**3)
    save = param["save_power_spectrum"].casefold()
    if save not in ("yes", "z_out", "no"):
        raise NotImplementedError(f"SAVE_POWER_SPECTRUM={save!r}, should be 'yes', 'z_out' or 'no'")
    param["save_pk"] = save == "yes" or (save == "z_out" and bool(param["write_snapshot"]))
    if param["save_pk"]:  # (of the density itself: the same spectrum solver.fft derives from the scaled source)
        k, pk, nmodes = power_spectrum(density.clone(), param)
        iostream.write_power_spectrum_to_ascii_file(k, pk, nmodes, param)
    # solver.rhs_poisson (solver.py:381-449): f1 rho - f1.  The constant only feeds the k = 0 mode, which the Green's
    # function zeroes (fourier.py:491, 544, 595), and the factor commutes with the transform: both ride on the Green's
    # prefactor instead of a pass over the grid (1e-16 relative; as the fp32 step does)
    f1 = 1.5 * param["aexp"] * param["Om_m"]
    kind, k2, s, c0 = green_tables(n, solver)
    phi = torch.empty_like(density)
    ctx.call("pm_real64_poisson_fft", kind, int(param["MAS_index"]), L.ptr(density), L.ptr(phi), L.hptr(k2), L.hptr(s), float(c0 * f1))
    del density
    force = torch.empty((n, n, n, 3), dtype=torch.float64, device=pos_d.device)
    ctx.call("pm_real64_gradient", int(param["gradient_stencil_order"]), L.ptr(phi), L.ptr(force))
    acc = torch.empty_like(pos_d)
    ma, mv = C.c_double(0.0), C.c_double(0.0)
    if _kick is not None:
        vel_d, half_dt = _kick
        ctx.call("pm_real64_interp_kick", scheme, L.ptr(force), L.ptr(pos_d), npart, L.ptr(acc), L.ptr(vel_d), float(half_dt),
                 C.byref(ma), C.byref(mv))
        param["_max_f64"] = (ma.value, mv.value)
    else:
        ctx.call("pm_real64_interp_kick", scheme, L.ptr(force), L.ptr(pos_d), npart, L.ptr(acc), None, 0.0, C.byref(ma), None)
        param.pop("_max_f64", None)
    del force
    empty = torch.empty(0, dtype=torch.float64, device=pos_d.device)
    return L.like_input(acc, position), L.like_input(phi, position), L.like_input(empty, position)


def max_abs(x) -> float:
    t = _dev(x)
    out = C.c_double(0.0)
    L.context(4, 8).call("pm_real64_max_abs", L.ptr(t), t.numel(), C.byref(out))
    return out.value


def integrate(position, velocity, acceleration, potential, additional_field, tables, param, t_snap_next=0.0):
    """integration.integrate (integration.py:17-118) with the leapfrog of :192-264, in float64.  Device tensors are
    updated in place and handed back (the reference's semantics); NumPy arrays are uploaded, stepped and returned."""
    if param["integrator"].casefold() != "leapfrog":
        raise NotImplementedError("the fp64 device mode steps with the leapfrog integrator")
    _check(param)
    host = not L.is_device_array(position)
    pos_d, vel_d, acc_d = _dev(position), _dev(velocity), _dev(acceleration)
    dx = 0.5 ** param["ncoarse"]
    cached = param.pop("_max_f64", None)  # (left by the last step's gather: the arrays it describes are these)
    amax, vmax = cached if (cached and not host) else (max_abs(acc_d), max_abs(vel_d))
    dt1 = param["Courant_factor"] * np.sqrt(dx / amax)                      # integration.py:267-295
    dt2 = param["Courant_factor"] * dx / vmax                               # :298-326
    f = 1.0 + 0.01 * param["max_aexp_stepping"]
    dt3 = float(utils.table_eval(tables[1], np.log(f * param["aexp"])) - utils.table_eval(tables[1], np.log(param["aexp"])))
    dt = min(dt1, dt2, dt3)
    if param["t"] + dt > t_snap_next:
        dt = float(t_snap_next) - param["t"]
        param["write_snapshot"] = True
    else:
        param["write_snapshot"] = False
    logging.info(f"Conditions: velocity {dt1=}, acceleration {dt2=}, scale factor {dt3=}")
    half_dt = 0.5 * dt
    n = 2 ** int(param["ncoarse"])
    # The product loop's order policy (main.run_nbody sets the key: the caller takes new arrays back).  When the last
    # gather on a stream one drift old made more tile passes per chunk than pays (a sorted stream makes ~2; a step that
    # sorted says ~2 whatever the next drift will do, so every ninth step in a row goes unsorted to take the measure
    # again), the drifted particles are sorted to tile resolution HERE, before the deposit and the gather of this step
    # read them -- after the step the order would already be one drift old when it is first used -- and only position
    # and velocity move (the old acceleration is dead after the kick).
    resort = False
    if not host and param.get("rebucket_straggler_fraction") is not None:
        if param.get("_f64_reserved") != pos_d.shape[0]:
            # what a re-sort needs, taken once and outside the steps that sort: the sort workspace and two spare particle
            # arrays in the loop's pool (device allocations of several GB cost ~0.15 s)
            from . import mesh

            L.context(n, 8).call("pm_real64_reserve_sort", pos_d.shape[0])
            torch = L.require_gpu()
            mesh.pool_give(torch.empty_like(pos_d), torch.empty_like(vel_d))
            param["_f64_reserved"] = pos_d.shape[0]
        age = stream_age(n)
        if age is not None and not param.get("_f64_sorted_last", False):
            param["_f64_aged_passes"] = age[1]
        aged, streak = param.get("_f64_aged_passes"), int(param.get("_f64_sort_streak", 0))
        resort = aged is not None and aged > float(param.get("resort_passes_per_chunk", RESORT_PASSES_PER_CHUNK)) and streak < 8
        param["_f64_sorted_last"], param["_f64_sort_streak"] = resort, (streak + 1 if resort else 0)
    density = None
    if resort or host or n < 16:  # kick + drift + wrap (then the sort), the deposit in pm()
        L.context(n, 8).call("pm_real64_kick_drift_wrap", L.ptr(pos_d), L.ptr(vel_d), L.ptr(acc_d), float(half_dt), float(dt),
                             pos_d.shape[0])
    else:  # ... or all of it in one pass over the particles
        density = L.require_gpu().empty((n, n, n), dtype=pos_d.dtype, device=pos_d.device)
        L.context(n, 8).call("pm_real64_kick_drift_wrap_deposit", _scheme(param), L.ptr(pos_d), L.ptr(vel_d), L.ptr(acc_d),
                             float(half_dt), float(dt), pos_d.shape[0], L.ptr(density))
    if resort:
        ids = param.get("particle_id")
        order = [] if ids is not None and L.is_device_array(ids) else None
        pos_d, vel_d = reorder_particles(pos_d, vel_d, _order_out=order, _key_bits=-1, _ncells=n, _pool=True)
        if order:
            param["particle_id"] = ids[order[0]]
        param["_resorted_f64"] = int(param.get("_resorted_f64", 0)) + 1
    param["t"] += dt
    param["aexp_old"] = param["aexp"]
    param["aexp"] = float(np.exp(utils.table_eval(tables[0], param["t"])))
    utils.set_units(param)
    acc_d, potential, additional_field = pm(pos_d, param, tables=tables, _kick=(vel_d, half_dt), _density=density)
    if host:
        param.pop("_max_f64", None)
        return pos_d.cpu().numpy(), vel_d.cpu().numpy(), acc_d.cpu().numpy(), potential.cpu().numpy(), additional_field.cpu().numpy()
    return pos_d, vel_d, acc_d, potential, additional_field


def reorder_particles(position, velocity=None, acceleration=None, _order_out=None, _key_bits=0, _ncells=4, _pool=False):
    """utils.reorder_particles (utils.py:1019-1116) for float64 arrays: stable Morton sort, in double all the way
    (utils.reorder_particles dispatches here; nothing is rounded to float32).  `_key_bits` (private): order to that many
    bits per axis only; -1: to the mesh's tiles (all the kernels care about: a quarter of the sort's passes).  The mesh
    is the context's: `_ncells`."""
    host = not L.is_device_array(position)
    pos = _dev(position)
    vel = _dev(velocity) if velocity is not None else None
    acc = _dev(acceleration) if acceleration is not None else None
    torch = L.require_gpu()
    from . import mesh

    # (_pool, private: outputs from / inputs back to the step loop's pool of spare particle arrays -- no device
    # allocation inside the loop; the caller no longer reads the input arrays)
    new = mesh._pool_take if _pool else torch.empty_like
    po = new(pos)
    vo = new(vel) if vel is not None else None
    ao = new(acc) if acc is not None else None
    order = None
    if _order_out is not None:
        order = torch.empty((pos.shape[0],), dtype=torch.int64, device=pos.device)
        _order_out.append(order)
    L.context(_ncells, 8).call("pm_real64_reorder_particles", pos.shape[0], L.ptr(pos), L.ptr(vel), L.ptr(acc), L.ptr(po), L.ptr(vo),
                         L.ptr(ao), L.ptr(order), int(_key_bits))

    if _pool and not host:
        mesh.pool_give(pos, vel, acc)

    def back(t, like):
        return t.cpu().numpy() if host else L.like_input(t, like)

    if acceleration is not None:
        return back(po, position), back(vo, velocity), back(ao, acceleration)
    if velocity is not None:
        return back(po, position), back(vo, velocity)
    return back(po, position)


def stream_age(ncells_1d: int):
    """(direct, passes) as the last gather found the stream (pm_real64_stream_age): the share of particles it served one
    by one through L2 (too few of their chunk in their tile) and its tile passes per 2048-particle chunk.  A sorted
    stream: the few particles of near-empty tiles, about two passes per chunk; both grow as the in-place steps let the
    order age; a shuffled stream: ~1 and 0.  None before any gather.  The fp64 mode's counterpart of
    utils.straggler_fraction, for main.reorder_policy."""
    aged, npart, passes, chunks = C.c_int64(0), C.c_int64(-1), C.c_int64(0), C.c_int64(0)
    L.context(ncells_1d, 8).call("pm_real64_stream_age", C.byref(aged), C.byref(npart), C.byref(passes), C.byref(chunks))
    if npart.value <= 0:
        return None
    return aged.value / npart.value, passes.value / max(chunks.value, 1)
