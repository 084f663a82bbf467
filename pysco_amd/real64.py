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


def pm(position, param, potential=None, additional_field=None, tables=(), _kick=None):
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
    f1 = 1.5 * param["aexp"] * param["Om_m"]
    ctx.call("pm_real64_scale_add", L.ptr(density), float(f1), float(-f1), n**3)  # solver.rhs_poisson: f1 rho - f1
    kind, k2, s, c0 = green_tables(n, solver)
    phi = torch.empty_like(density)
    ctx.call("pm_real64_poisson_fft", kind, int(param["MAS_index"]), L.ptr(density), L.ptr(phi), L.hptr(k2), L.hptr(s), float(c0))
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
    L.context(n, 8).call("pm_real64_kick_drift_wrap", L.ptr(pos_d), L.ptr(vel_d), L.ptr(acc_d), float(half_dt), float(dt),
                         pos_d.shape[0])
    param["t"] += dt
    param["aexp_old"] = param["aexp"]
    param["aexp"] = float(np.exp(utils.table_eval(tables[0], param["t"])))
    utils.set_units(param)
    acc_d, potential, additional_field = pm(pos_d, param, tables=tables, _kick=(vel_d, half_dt))
    if host:
        param.pop("_max_f64", None)
        return pos_d.cpu().numpy(), vel_d.cpu().numpy(), acc_d.cpu().numpy(), potential.cpu().numpy(), additional_field.cpu().numpy()
    return pos_d, vel_d, acc_d, potential, additional_field
