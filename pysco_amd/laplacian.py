"""7-point Laplacian: operator, residuals, red-black SOR smoother (reference laplacian.py).

Only the functions the V-cycle uses (laplacian.py:12-226, 327-381, 502-533, 765-797, 844-1055);
the `_half` and Knebe variants are dead code in the reference's cycle and are not mirrored.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from . import mesh


def _dev(x):
    return L.to_device(x, L.require_gpu().float32)


def operator(x):
    """Laplacian(x) (laplacian.py:12)."""
    torch = L.require_gpu()
    xd = _dev(x)
    out = torch.empty_like(xd)
    L.context(4).call("pm_lap_operator", L.ptr(xd), None, xd.shape[0], 0, L.ptr(out))
    return L.like_input(out, x)


def residual(x, b):
    """b - Laplacian(x) (laplacian.py:63)."""
    torch = L.require_gpu()
    xd, bd = _dev(x), _dev(b)
    out = torch.empty_like(xd)
    L.context(4).call("pm_lap_operator", L.ptr(xd), L.ptr(bd), xd.shape[0], 1, L.ptr(out))
    return L.like_input(out, x)


def restrict_residual(x, b):
    """restriction(residual(x, b)) in one pass (laplacian.py:125)."""
    torch = L.require_gpu()
    xd, bd = _dev(x), _dev(b)
    n = xd.shape[0]
    out = torch.empty((n // 2,) * 3, dtype=torch.float32, device=xd.device)
    L.context(4).call("pm_lap_restrict_residual", L.ptr(xd), L.ptr(bd), n, L.ptr(out))
    return L.like_input(out, x)


def residual_error(x, b) -> np.float32:
    """sqrt(sum residual^2) (laplacian.py:327)."""
    xd, bd = _dev(x), _dev(b)
    out = C.c_float(0)
    L.context(4).call("pm_lap_residual_error", L.ptr(xd), L.ptr(bd), xd.shape[0], C.byref(out))
    return np.float32(out.value)


def truncation_error(x) -> np.float32:
    """|| R(L x) - L(R x) || (laplacian.py:502)."""
    xd = _dev(x)
    RLx = mesh.restriction(operator(xd))
    LRx = operator(mesh.restriction(xd))
    out = C.c_float(0)
    L.context(4).call("pm_diff_norm", L.ptr(RLx), L.ptr(LRx), 1.0, RLx.numel(), C.byref(out))
    return np.float32(out.value)


def initialise_potential(b):
    """First guess -h^2/6 * b (laplacian.py:765)."""
    torch = L.require_gpu()
    bd = _dev(b)
    h = np.float32(1.0 / bd.shape[0])
    minus_h2_over_six = np.float32(-float(h * h) / 6.0)
    out = torch.empty_like(bd)
    L.context(4).call("pm_linear_operator", L.ptr(out), L.ptr(bd), float(minus_h2_over_six), 0.0, bd.numel())
    return L.like_input(out, b)


def gauss_seidel(x, b, f_relax) -> None:
    """One red + one black over-relaxed sweep, in place (laplacian.py:844)."""
    bd = _dev(b)
    ctx = L.context(4)
    if L.is_device_array(x):
        ctx.call("pm_lap_gauss_seidel", L.ptr(x), L.ptr(bd), x.shape[0], float(np.float32(f_relax)))
    else:
        xd = _dev(x)
        ctx.call("pm_lap_gauss_seidel", L.ptr(xd), L.ptr(bd), xd.shape[0], float(np.float32(f_relax)))
        x[...] = xd.cpu().numpy()


def smoothing(x, b, n_smoothing: int) -> None:
    """n_smoothing sweeps with omega = 1.25 (laplacian.py:1026)."""
    if L.is_device_array(x):
        L.context(4).call("pm_lap_smoothing", L.ptr(x), L.ptr(_dev(b)), x.shape[0], int(n_smoothing))
    else:
        xd = _dev(x)
        for _ in range(int(n_smoothing)):
            gauss_seidel(xd, b, np.float32(1.25))
        x[...] = xd.cpu().numpy()
