"""CPU oracle of the fp64 device mode -- TEST INFRASTRUCTURE, NOT PRODUCT.

The reference is float32 end to end (mesh.py:2467, fourier.py:21), so an fp64 tolerance can only be stated between
the build's own fp64 paths (fp64-GPU vs fp64-CPU <= 1e-10).  This is the
CPU one: the SAME formulas as pysco_oracle.py -- pm_oracle.c compiled with REAL = double (libpm_oracle_f64.so), numpy's
float64 FFT, every scalar a Python float -- for the Newtonian FFT step (TSC / CIC / NGP deposit, compensated or 7-point
Green's function, gradient of order 2 / 3 / 5 / 7, inverse interpolation, kick-drift-kick, dt criteria, P(k)).

Parity status: the fp32 build of the same C file is PINNED against the reference's golden vectors
(tests/test_oracle_golden.py); this build differs from it by `typedef double REAL` alone, and
tests/test_oracle_f64.py checks that it agrees with the pinned fp32 oracle to fp32 rounding.

Citations are relative to /root/reference/pysco/.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import pysco_oracle as o32

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib64 = None


def _lib():
    global _lib64
    if _lib64 is None:
        o32.build()
        lib = C.CDLL(os.path.join(_HERE, "_build", "libpm_oracle_f64.so"))
        lib.pmo_max_abs.restype = C.c_double
        _lib64 = lib
    return _lib64


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _c64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def deposit(scheme: str, position, n: int) -> np.ndarray:  # mesh.py:2240 (NGP), 2284 (CIC), 2468 (TSC)
    pos = _c64(position)
    rho = np.empty((n,) * 3, dtype=np.float64)
    fn = {"ngp": _lib().pmo_ngp, "cic": _lib().pmo_cic, "tsc": _lib().pmo_tsc}[scheme.casefold()]
    fn(_p(pos), C.c_int64(pos.shape[0]), C.c_int(n), _p(rho))
    return rho


def gather(scheme: str, grid, position) -> np.ndarray:  # mesh.py:2639 / 2752 / 2960
    g, pos = _c64(grid), _c64(position)
    out = np.empty_like(pos)
    n = g.shape[0]
    if scheme.casefold() == "ngp":
        _lib().pmo_inv_ngp_vec(_p(g), C.c_int(n), _p(pos), C.c_int64(pos.shape[0]), _p(out))
    elif scheme.casefold() == "cic":
        _lib().pmo_inv_cic_vec(_p(g), C.c_int(n), _p(pos), C.c_int64(pos.shape[0]), _p(out))
    else:
        _lib().pmo_inv_tsc(_p(g), C.c_int(n), C.c_int(3), _p(pos), C.c_int64(pos.shape[0]), _p(out))
    return out


def derivative(a, order: int) -> np.ndarray:  # mesh.py:2072 -> 639 / 681 / 725 / 782
    if order not in (2, 3, 5, 7):
        raise ValueError(f"gradient_order={order}, should be 2, 3, 5 or 7")
    a = _c64(a)
    n = a.shape[0]
    out = np.empty((n, n, n, 3), dtype=np.float64)
    _lib().pmo_derivative(_p(a), None, C.c_double(0), C.c_int(0), C.c_int(n), C.c_int(order), _p(out))
    return out


def max_abs(x) -> float:  # utils.py:410-430
    x = _c64(x)
    return float(_lib().pmo_max_abs(_p(x), C.c_int64(x.size)))


def periodic_wrap(x) -> None:  # utils.py:1120-1149
    assert x.dtype == np.float64 and x.flags.c_contiguous
    _lib().pmo_periodic_wrap(_p(x), C.c_int64(x.size))


def _kfreq(n: int) -> np.ndarray:
    i = np.arange(n)
    return np.where(i >= n // 2, i - n, i).astype(np.float64)


def green_tables(n: int, solver: str):
    """Per-axis tables and prefactor of the Green's functions (fourier.py:502-544 compensated, 555-595 seven-point):
    g = c0 / (k2[i] + k2[j] + k2[k]) [* (s[i] s[j] s[k])^(-2p)], or c0 / (s[i] + s[j] + s[k])."""
    k = _kfreq(n)
    if solver == "fft":
        return 1, k**2, np.sinc(k / n), -0.25 / np.pi**2
    f = np.sin(np.pi / n * k) ** 2
    return 2, np.zeros(n), f, -0.25 / n**2


def poisson_fft(rhs, n: int, solver: str, p: int) -> np.ndarray:  # solver.py:453-522
    xk = np.ascontiguousarray(np.fft.rfftn(rhs))
    kind, k2, s, c0 = green_tables(n, solver)
    xr = xk.view(np.float64)
    _lib().pmo_greens(_p(xr), C.c_int(n), C.c_int(kind), C.c_int(int(p)), _p(_c64(k2)), _p(_c64(s)), C.c_double(c0))
    return np.ascontiguousarray(np.fft.irfftn(xk, s=(n, n, n), axes=(0, 1, 2)))


def pm(position, param):
    """solver.pm (solver.py:30-215) for theory = newton, linear_newton_solver = fft / fft_7pt, in float64.
    Returns (acceleration, potential)."""
    n = 2 ** int(param["ncoarse"])
    scheme = param["mass_scheme"].casefold()
    solver = param["linear_newton_solver"].casefold()
    if param["theory"].casefold() != "newton" or solver not in ("fft", "fft_7pt"):
        raise NotImplementedError("the fp64 mode covers the Newtonian FFT step")
    p = {"ngp": 1, "cic": 2, "tsc": 3}[scheme]
    density = deposit(scheme, position, n)
    if n**3 != param["npart"]:
        density *= n**3 / param["npart"]
    f1 = 1.5 * param["aexp"] * param["Om_m"]
    density *= f1           # utils.linear_operator_inplace: f1 * x + f2 (solver.py:381-449, utils.py:644-684)
    density += -f1
    potential = poisson_fft(density, n, solver, p)
    force = derivative(potential, int(param["gradient_stencil_order"]))
    return gather(scheme, force, position), potential


def integrate(position, velocity, acceleration, tables, param, t_snap_next=0.0):
    """integration.integrate (integration.py:17-264), leapfrog, float64.  Arrays are updated in place and returned."""
    dx = 0.5 ** param["ncoarse"]
    dt1 = param["Courant_factor"] * np.sqrt(dx / max_abs(acceleration))          # integration.py:267-295
    dt2 = param["Courant_factor"] * dx / max_abs(velocity)                      # :298-326
    f = 1.0 + 0.01 * param["max_aexp_stepping"]
    dt3 = float(tables[1](np.log(f * param["aexp"])) - tables[1](np.log(param["aexp"])))  # :329-358
    dt = min(dt1, dt2, dt3)
    if param["t"] + dt > t_snap_next:
        dt = t_snap_next - param["t"]
        param["write_snapshot"] = True
    else:
        param["write_snapshot"] = False
    half_dt = 0.5 * dt
    velocity += (-half_dt) * acceleration      # utils.add_vector_scalar_inplace: y += a * x (utils.py:220-298)
    position += dt * velocity
    param["t"] += dt
    param["aexp_old"] = param["aexp"]
    param["aexp"] = float(np.exp(tables[0](param["t"])))
    o32.set_units(param)
    periodic_wrap(position)
    acceleration, potential = pm(position, param)
    velocity += (-half_dt) * acceleration
    return position, velocity, acceleration, potential, dt


def pk(position, n: int, p: int = 3):
    """The binned, window-deconvolved P(k) of the TSC density (fourier.py:22-100) in float64: pysco_oracle.pk_fp64
    from float64 positions (no rounding of the positions to fp32 first)."""
    x = np.asarray(position, dtype=np.float64) * n
    i = np.minimum(x.astype(np.int64), n - 1)
    d = x - 0.5 - i
    w = np.stack([0.5 * (0.5 - d) ** 2, 0.75 - d * d, 0.5 * (0.5 + d) ** 2], axis=0)
    rho = np.zeros(n**3, dtype=np.float64)
    for a in range(3):
        ia = (i[:, 0] + a - 1) % n
        for b in range(3):
            ib = (i[:, 1] + b - 1) % n
            wab = w[a, :, 0] * w[b, :, 1]
            for c in range(3):
                ic = (i[:, 2] + c - 1) % n
                rho += np.bincount((ia * n + ib) * n + ic, weights=wab * w[c, :, 2], minlength=n**3)
    return pk_of_density(rho.reshape(n, n, n), p)


def pk_of_density(rho, p: int = 3):
    n = rho.shape[0]
    dk = np.fft.rfftn(rho)
    kf = _kfreq(n)
    kz = np.arange(n // 2 + 1, dtype=np.float64)
    sx, sz = np.sinc(kf / n), np.sinc(kz / n)
    wk = (sx[:, None] * sx[None, :])[:, :, None] * sz[None, None, :]
    power = (dk.real**2 + dk.imag**2) * wk ** (-2.0 * p)
    knorm = np.sqrt(((kf**2)[:, None] + (kf**2)[None, :])[:, :, None] + (kz**2)[None, None, :])
    kidx = (knorm + 0.5).astype(np.int64).ravel()
    knorm, power = knorm.ravel().copy(), power.ravel().copy()
    knorm[0], power[0] = 0.0, 0.0
    nm = np.bincount(kidx, minlength=n).astype(np.float64)
    nm[0] -= 1
    ks = np.bincount(kidx, weights=knorm, minlength=n)
    ps = np.bincount(kidx, weights=power, minlength=n)
    kmax = int(2 * (n // 2) / 3)
    with np.errstate(invalid="ignore", divide="ignore"):
        return ks[1:kmax] / nm[1:kmax], ps[1:kmax] / nm[1:kmax], nm[1:kmax]
