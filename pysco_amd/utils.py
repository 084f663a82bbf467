"""Element-wise helpers, reductions, periodic wrap, unit system and Morton re-ordering.

Mirror of the reference's utils.py functions that sit on the per-step path (utils.py:167-298,
410-430, 644-755, 1019-1149).  In-place functions mutate device tensors directly; NumPy inputs
are round-tripped through the GPU and written back in place.
"""
from __future__ import annotations

import numpy as np

from . import _lib as L

# CODATA 2018, the values astropy.constants exposes as G, pc (utils.py:11)
_G = 6.6743e-11
_PC = 3.0856775814913674e16


def set_units(param) -> None:
    """Box units -> km, kg, s conversion factors (utils.py:167-192); pure host scalars."""
    mpc_to_km = 1e3 * _PC
    g = _G * 1e-9
    H0 = param["H0"] / mpc_to_km
    rhoc = 3.0 * H0**2 / (8.0 * np.pi * g)
    param["unit_l"] = param["aexp"] * param["boxlen"] * 100.0 / H0
    param["unit_t"] = param["aexp"] ** 2 / H0
    param["unit_d"] = param["Om_m"] * rhoc / param["aexp"] ** 3
    param["mpart"] = param["unit_d"] * param["unit_l"] ** 3 / param["npart"]


# max|acc|, max|vel| left in a context by the fused gather+kick kernel (pm_interp_kick); consumed by
# the next integrate()'s dt criteria instead of two more passes over the particle arrays.
_fused_maxabs = None


def _tensor_key(t):
    return (t.data_ptr(), t._version, t.numel())


def register_fused_maxabs(ctx, acc, vel) -> None:
    global _fused_maxabs
    _fused_maxabs = None if acc is None else {
        "ctx": ctx, "acc": _tensor_key(acc), "vel": None if vel is None else _tensor_key(vel), "values": None}


def _lookup_fused_maxabs(x):
    import ctypes as C

    f = _fused_maxabs
    if f is None or not L.is_device_array(x):
        return None
    key = _tensor_key(x)
    which = 0 if key == f["acc"] else (1 if key == f["vel"] else None)
    if which is None:
        return None
    if f["values"] is None:
        a, v = C.c_float(0), C.c_float(0)
        f["ctx"].call("pm_last_maxabs", C.byref(a), C.byref(v))
        f["values"] = (np.float32(a.value), np.float32(v.value))
        ms = C.c_double(0)
        f["ctx"].call("pm_last_acc_meanabs", C.byref(ms))  # (came with the same read-back)
        f["meanabs"] = float(ms.value)
    return f["values"][which]


def fused_acc_meanabs(acceleration):
    """Mean |a_x| of a sample of `acceleration` if the fused gather that wrote it took one (pmstep.h:
    pm_last_acc_meanabs), else None.  Costs nothing beyond the read-back max_abs(acceleration) does anyway."""
    if _lookup_fused_maxabs(acceleration) is None or _tensor_key(acceleration) != _fused_maxabs["acc"]:
        return None
    ms = _fused_maxabs.get("meanabs")
    return ms if ms and np.isfinite(ms) and ms > 0 else None


def _inplace(y, fn):
    """Run fn(device_tensor) and write the result back if y was a NumPy array."""
    global _fused_maxabs
    torch = L.require_gpu()
    if L.is_device_array(y):
        if not y.is_contiguous() or y.dtype != torch.float32:
            raise ValueError("device arrays must be contiguous float32")
        f = _fused_maxabs
        if f is not None and y.data_ptr() in (f["acc"][0], f["vel"][0] if f["vel"] else None):
            _fused_maxabs = None  # the array is about to change under the cached maxima
        fn(y)
    else:
        t = L.to_device(y, torch.float32)
        fn(t)
        y[...] = t.cpu().numpy().reshape(y.shape)


def _ctx():
    return L.context(4)  # element-wise kernels do not depend on the grid size


def add_vector_scalar_inplace(y, x, a) -> None:
    """y += a*x (utils.py:264)."""
    torch = L.require_gpu()
    xd = L.to_device(x, torch.float32)
    ctx = _ctx()
    _inplace(y, lambda t: ctx.call("pm_axpy", L.ptr(t), L.ptr(xd), float(np.float32(a)), t.numel()))


def prod_vector_scalar_inplace(y, a) -> None:
    """y *= a (utils.py:410)."""
    ctx = _ctx()
    _inplace(y, lambda t: ctx.call("pm_scale", L.ptr(t), float(np.float32(a)), t.numel()))


def linear_operator(x, f1, f2):
    """f1*x + f2 into a new array (utils.py:644)."""
    torch = L.require_gpu()
    xd = L.to_device(x, torch.float32)
    out = torch.empty_like(xd)
    ctx = _ctx()
    ctx.call("pm_linear_operator", L.ptr(out), L.ptr(xd), float(np.float32(f1)),
                                       float(np.float32(f2)), xd.numel())
    return L.like_input(out, x)


def linear_operator_inplace(x, f1, f2) -> None:
    """x = f1*x + f2 (utils.py:684)."""
    ctx = _ctx()
    _inplace(x, lambda t: ctx.call("pm_linear_operator", L.ptr(t), L.ptr(t), float(np.float32(f1)),
                                                               float(np.float32(f2)), t.numel()))


def linear_operator_vectors_inplace(x, f1, y, f2) -> None:
    """x = f1*x + f2*y (utils.py:721)."""
    torch = L.require_gpu()
    yd = L.to_device(y, torch.float32)
    ctx = _ctx()
    _inplace(x, lambda t: ctx.call("pm_linear_operator_vectors", L.ptr(t), float(np.float32(f1)),
                                                                       L.ptr(yd), float(np.float32(f2)), t.numel()))


def max_abs(x) -> np.float32:
    """max |x| (utils.py:220); NaN propagates as in np.max(np.abs(x))."""
    import ctypes as C

    cached = _lookup_fused_maxabs(x)
    if cached is not None:
        return cached
    torch = L.require_gpu()
    xd = L.to_device(x, torch.float32)
    ctx = _ctx()
    out = C.c_float(0)
    ctx.call("pm_max_abs", L.ptr(xd), xd.numel(), C.byref(out))
    return np.float32(out.value)


def periodic_wrap(position) -> None:
    """Wrap positions into [0,1) with the reference's -2^-25 guard (utils.py:1120)."""
    ctx = _ctx()
    _inplace(position, lambda t: ctx.call("pm_periodic_wrap", L.ptr(t), t.numel()))


def straggler_fraction(ncells_1d: int):
    """(fraction, complete) of the most recent straggler scan on the n-cell context: the share of particles that
    were outside the tile the sorted order files them under, and whether the per-tile lists held them all.
    (None, True) before any scan.  No reference counterpart; main.run_nbody uses it to re-sort on demand."""
    import ctypes as C

    s, npart, complete = C.c_int64(-1), C.c_int64(0), C.c_int(1)
    L.context(ncells_1d).call("pm_tile_lists_info", C.byref(s), C.byref(npart), C.byref(complete))
    if s.value < 0 or npart.value <= 0:
        return None, True
    return s.value / npart.value, bool(complete.value)


def reorder_particles(position, velocity=None, acceleration=None, _order_out=None):
    """Sort particles by Morton key (utils.py:1019).  Unlike the reference's per-thread partial
    argsort (utils.py:1080-1116) this is always a full, stable sort.  `_order_out` (private): a list that
    receives the permutation (int64 device tensor: new[i] = old[order[i]])."""
    from . import real64

    if real64.is_float64(position):  # the fp64 device mode: sorted in double, nothing rounded to float32
        return real64.reorder_particles(position, velocity, acceleration, _order_out=_order_out)
    torch = L.require_gpu()
    pos = L.to_device(position, torch.float32)
    vel = L.to_device(velocity, torch.float32) if velocity is not None else None
    acc = L.to_device(acceleration, torch.float32) if acceleration is not None else None
    ctx = _ctx()
    po = torch.empty_like(pos)
    vo = torch.empty_like(vel) if vel is not None else None
    ao = torch.empty_like(acc) if acc is not None else None
    order = None
    if _order_out is not None:
        order = torch.empty((pos.shape[0],), dtype=torch.int64, device=pos.device)
        _order_out.append(order)
    ctx.call("pm_reorder_particles", pos.shape[0], L.ptr(pos), L.ptr(vel), L.ptr(acc), L.ptr(po),
                                         L.ptr(vo), L.ptr(ao), L.ptr(order))
    if acceleration is not None:
        return L.like_input(po, position), L.like_input(vo, velocity), L.like_input(ao, acceleration)
    if velocity is not None:
        return L.like_input(po, position), L.like_input(vo, velocity)
    return L.like_input(po, position)


# ---- background tables ------------------------------------------------------------------------------------
_table_fast = {}


def table_eval(table, x):
    """table(x) for one of the background interpolators (cosmotable.py: scipy interp1d objects) at a scalar x.
    The step evaluates seven of them between the read-back of the CFL maxima and the next launch, with the GPU idle:
    a linear, extrapolating interp1d is evaluated here with the very operations of interp1d._call_linear on
    float64 scalars (same bits, ~1 us instead of ~15); anything else is called as it is."""
    if np.ndim(x) != 0:
        return table(x)
    entry = _table_fast.get(id(table))
    if entry is None or entry[0] is not table:
        fast = None
        try:
            if (getattr(table, "_kind", None) == "linear" and getattr(table, "_extrapolate", False)
                    and table.x.ndim == 1 and table.x.dtype == np.float64 and table._y.ndim == 2
                    and table._y.shape[1] == 1 and table._y.dtype == np.float64 and table.x.shape[0] >= 2):
                fast = (table.x.tolist(), table._y[:, 0].tolist())
        except Exception:
            fast = None
        entry = _table_fast[id(table)] = (table, fast)
        if len(_table_fast) > 64:
            _table_fast.clear()
    fast = entry[1]
    if fast is None:
        return table(x)
    import bisect

    xs, ys = fast
    xv = float(x)
    if xv != xv:
        return table(x)
    hi = min(max(bisect.bisect_left(xs, xv), 1), len(xs) - 1)
    lo = hi - 1
    slope = (ys[hi] - ys[lo]) / (xs[hi] - xs[lo])
    return np.float64(slope * (xv - xs[lo]) + ys[lo])

