"""f(R) Hu-Sawicki n = 1: the scalaron's depressed-cubic operator u^3 + p u + q h^2 = 0 and its
non-linear red-black Gauss-Seidel smoother with an fp64 analytic root per cell (reference cubic.py)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from . import mesh


def _dev(x):
    return L.to_device(x, L.require_gpu().float32)


def operator(x, b, q):
    """u^3 + p u + q h^2 with p = h^2 b - (1/6) sum u_nb^2 (cubic.py:23)."""
    torch = L.require_gpu()
    xd, bd = _dev(x), _dev(b)
    out = torch.empty_like(xd)
    L.context(4).call("pm_cubic_operator", L.ptr(xd), L.ptr(bd), float(np.float32(q)), None, xd.shape[0], L.ptr(out))
    return L.like_input(out, x)


def residual_with_rhs(x, b, q, rhs):
    """rhs - operator(x, b, q) (cubic.py:90)."""
    torch = L.require_gpu()
    xd, bd, rd = _dev(x), _dev(b), _dev(rhs)
    out = torch.empty_like(xd)
    L.context(4).call("pm_cubic_operator", L.ptr(xd), L.ptr(bd), float(np.float32(q)), L.ptr(rd), xd.shape[0],
                      L.ptr(out))
    return L.like_input(out, x)


def initialise_potential(b, q):
    """Root of the cubic with all neighbours set to zero (cubic.py:217)."""
    torch = L.require_gpu()
    bd = _dev(b)
    out = torch.empty_like(bd)
    L.context(4).call("pm_cubic_initialise_potential", L.ptr(bd), float(np.float32(q)), bd.shape[0], L.ptr(out))
    return L.like_input(out, b)


def _gs(x, b, q, f_relax, rhs) -> None:
    bd = _dev(b)
    rd = None if rhs is None else _dev(rhs)
    ctx = L.context(4)
    args = (L.ptr(bd), float(np.float32(q)), L.ptr(rd))
    if L.is_device_array(x):
        ctx.call("pm_cubic_gauss_seidel", L.ptr(x), *args, x.shape[0], float(np.float32(f_relax)))
    else:
        xd = _dev(x)
        ctx.call("pm_cubic_gauss_seidel", L.ptr(xd), *args, xd.shape[0], float(np.float32(f_relax)))
        x[...] = xd.cpu().numpy()


def gauss_seidel(x, b, q, f_relax) -> None:  # cubic.py:269
    _gs(x, b, q, f_relax, None)


def gauss_seidel_with_rhs(x, b, q, rhs, f_relax) -> None:  # cubic.py:446
    _gs(x, b, q, f_relax, rhs)


def _smooth(x, b, q, n_smoothing, rhs) -> None:
    if L.is_device_array(x) and L.is_device_array(b) and (rhs is None or L.is_device_array(rhs)):
        L.context(4).call("pm_fr_smoothing", 1, L.ptr(x), L.ptr(b), float(np.float32(q)), L.ptr(rhs), x.shape[0], int(n_smoothing))
        return
    for _ in range(int(n_smoothing)):
        _gs(x, b, q, np.float32(1.25), rhs)


def smoothing(x, b, q, n_smoothing: int) -> None:  # cubic.py:1065
    _smooth(x, b, q, n_smoothing, None)


def smoothing_with_rhs(x, b, q, n_smoothing: int, rhs) -> None:  # cubic.py:1100
    _smooth(x, b, q, n_smoothing, rhs)


def residual_error(x, b, q) -> np.float32:  # cubic.py:844
    xd, bd = _dev(x), _dev(b)
    out = C.c_float(0)
    L.context(4).call("pm_cubic_residual_error", L.ptr(xd), L.ptr(bd), float(np.float32(q)), xd.shape[0], C.byref(out))
    return np.float32(out.value)


def truncation_error(x, b, q) -> np.float32:  # cubic.py:1021
    xd, bd = _dev(x), _dev(b)
    RLx = mesh.restriction(operator(xd, bd, q))
    LRx = operator(mesh.restriction(xd), mesh.restriction(bd), q)
    out = C.c_float(0)
    L.context(4).call("pm_diff_norm", L.ptr(RLx), L.ptr(LRx), 4.0, RLx.numel(), C.byref(out))
    return np.float32(out.value)
