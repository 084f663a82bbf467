"""Mass assignment, inverse interpolation and finite-difference gradients on the GPU.

Host-side mirror of the reference's mesh.py entry points (same names, argument meaning and
errors); the bodies are the HIP kernels in csrc/pm_particles.hip and csrc/pm_grid.hip.
Arrays may be NumPy (uploaded, result returned as NumPy) or torch CUDA tensors (zero-copy).
"""
from __future__ import annotations

import numpy as np

from . import _lib as L


def _deposit(scheme: int, position, ncells_1d: int, sorted_hint: bool = True):
    torch = L.require_gpu()
    pos = L.to_device(position, torch.float32)
    if pos.ndim != 2 or pos.shape[1] != 3:
        raise ValueError("position must have shape [N_part, 3]")
    ctx = L.context(ncells_1d)
    rho = torch.empty((ncells_1d,) * 3, dtype=torch.float32, device=pos.device)
    ctx.call("pm_deposit", scheme, L.ptr(pos), pos.shape[0], L.ptr(rho), int(sorted_hint))
    return L.like_input(rho, position)


def NGP(position, ncells_1d: int):
    """Nearest-grid-point density [n,n,n] (reference mesh.py:2240)."""
    return _deposit(L.NGP, position, ncells_1d)


def CIC(position, ncells_1d: int):
    """Cloud-in-cell density (reference mesh.py:2284)."""
    return _deposit(L.CIC, position, ncells_1d)


def TSC(position, ncells_1d: int):
    """Triangular-shaped-cloud density (reference mesh.py:2468)."""
    return _deposit(L.TSC, position, ncells_1d)


def TSC_seq(position, ncells_1d: int):
    """Same result as TSC (reference mesh.py:2363 is its single-threaded twin)."""
    return _deposit(L.TSC, position, ncells_1d)


def _interp(scheme: int, ncomp: int, grid, position):
    torch = L.require_gpu()
    g = L.to_device(grid, torch.float32)
    pos = L.to_device(position, torch.float32)
    n = g.shape[0]
    want = (n, n, n, 3) if ncomp == 3 else (n, n, n)
    if tuple(g.shape) != want:
        raise ValueError(f"grid must have shape {want}")
    ctx = L.context(n)
    out = torch.empty((pos.shape[0], 3) if ncomp == 3 else (pos.shape[0],), dtype=torch.float32, device=pos.device)
    if ncomp == 3 and n >= 16:  # LDS-tiled gather (fast on Morton-sorted streams, same result otherwise)
        from . import utils

        ctx.call("pm_interp_kick", scheme, L.ptr(g), L.ptr(pos), pos.shape[0], L.ptr(out), None, 0.0)
        utils.register_fused_maxabs(None, None, None)
    else:
        ctx.call("pm_interp", scheme, ncomp, L.ptr(g), L.ptr(pos), pos.shape[0], L.ptr(out))
    return L.like_input(out, position)


def interp_vec_kick(scheme_name: str, grid, position, velocity, half_dt):
    """Fused solver.py:205-209 + integration.py:262 on device tensors: returns acc = inv*_vec(grid,
    position) and applies velocity += -half_dt*acc in place; the maxima for the next step's CFL
    criteria are kept in the context (utils.max_abs picks them up)."""
    from . import utils

    torch = L.require_gpu()
    n = grid.shape[0]
    scheme = {"cic": L.CIC, "tsc": L.TSC, "ngp": L.NGP}[scheme_name]
    ctx = L.context(n)
    out = torch.empty_like(position)
    ctx.call("pm_interp_kick", scheme, L.ptr(grid), L.ptr(position), position.shape[0], L.ptr(out),
             L.ptr(velocity), float(np.float32(half_dt)))
    utils.register_fused_maxabs(ctx, out, velocity)
    return out


def derivative_interp_vec_kick(scheme_name: str, potential, order: int, position, velocity=None, half_dt=0.0,
                               nx_phi=None, x0_phi=0):
    """mesh.derivative (mesh.py:2072) + inv*_vec (+ the second kick when `velocity` is given) in one
    kernel, without materialising the force grid; bit-identical to derivative() followed by
    interp_vec_kick().  `potential` may be a slab [nx_phi, n, n] starting at global plane x0_phi."""
    from . import utils

    torch = L.require_gpu()
    n = potential.shape[1]
    scheme = {"cic": L.CIC, "tsc": L.TSC, "ngp": L.NGP}[scheme_name]
    if order not in (2, 3, 5, 7):
        raise ValueError(f"{order=}, should be 2, 3, 5 or 7")
    ctx = L.context(n)
    out = torch.empty_like(position)
    ctx.call("pm_gradient_interp_kick", int(order), scheme, L.ptr(potential),
             int(potential.shape[0] if nx_phi is None else nx_phi), int(x0_phi), L.ptr(position), position.shape[0],
             L.ptr(out), None if velocity is None else L.ptr(velocity), float(np.float32(half_dt)))
    utils.register_fused_maxabs(ctx, out, velocity)
    return out


def invNGP(grid, position):  # reference mesh.py:2600
    return _interp(L.NGP, 1, grid, position)


def invCIC(grid, position):  # reference mesh.py:2679
    return _interp(L.CIC, 1, grid, position)


def invTSC(grid, position):  # reference mesh.py:2827
    return _interp(L.TSC, 1, grid, position)


def invNGP_vec(grid, position):  # reference mesh.py:2639
    return _interp(L.NGP, 3, grid, position)


def invCIC_vec(grid, position):  # reference mesh.py:2752
    return _interp(L.CIC, 3, grid, position)


def invTSC_vec(grid, position):  # reference mesh.py:2960
    return _interp(L.TSC, 3, grid, position)


def derivative(a, gradient_order: int):
    """+grad(a) as [n,n,n,3]; order 2 (forward), 3, 5, 7 (central) -- reference mesh.py:2072."""
    if gradient_order not in (2, 3, 5, 7):
        raise ValueError(f"gradient_order={gradient_order}, should be 2, 3, 5 or 7")
    torch = L.require_gpu()
    phi = L.to_device(a, torch.float32)
    n = phi.shape[0]
    ctx = L.context(n)
    out = torch.empty((n, n, n, 3), dtype=torch.float32, device=phi.device)
    ctx.call("pm_gradient", int(gradient_order), L.ptr(phi), L.ptr(out))
    return L.like_input(out, a)


def derivative_fR(a, b, f, fR_n: int, gradient_order: int):
    """grad(a) + f*grad(b**(fR_n+1)) -- reference mesh.py:2112."""
    if gradient_order not in (2, 3, 5, 7):
        raise ValueError(f"gradient_order={gradient_order}, should be 2, 3, 5 or 7")
    if fR_n not in (1, 2):
        raise NotImplementedError(f"Only f(R) with n = 1 and 2, currently fR_n={fR_n}")
    torch = L.require_gpu()
    phi = L.to_device(a, torch.float32)
    u = L.to_device(b, torch.float32)
    n = phi.shape[0]
    ctx = L.context(n)
    out = torch.empty((n, n, n, 3), dtype=torch.float32, device=phi.device)
    ctx.call("pm_gradient_fr", int(gradient_order), int(fR_n), float(np.float32(f)), L.ptr(phi),
                                   L.ptr(u), L.ptr(out))
    return L.like_input(out, a)


# ---- multigrid transfer operators (reference mesh.py:14-551) ----------------------------------------


def _restrict(x, scale: float):
    torch = L.require_gpu()
    xd = L.to_device(x, torch.float32)
    nf = xd.shape[0]
    ctx = L.context(4)
    out = torch.empty((nf // 2,) * 3, dtype=torch.float32, device=xd.device)
    ctx.call("pm_restriction", L.ptr(xd), nf, scale, L.ptr(out))
    return L.like_input(out, x)


def restriction(x):
    """Average of the 8 children: [n]^3 -> [n/2]^3 (mesh.py:14)."""
    return _restrict(x, 0.125)


def minus_restriction(x):
    """-restriction(x) (mesh.py:63)."""
    return _restrict(x, -0.125)


def prolongation(x):
    """Tri-linear interpolation [n]^3 -> [2n]^3 (mesh.py:206)."""
    torch = L.require_gpu()
    xd = L.to_device(x, torch.float32)
    nc = xd.shape[0]
    ctx = L.context(4)
    y = torch.empty((2 * nc,) * 3, dtype=torch.float32, device=xd.device)
    ctx.call("pm_prolongation", L.ptr(y), L.ptr(xd), nc, 0)
    return L.like_input(y, x)


def add_prolongation(y, x) -> None:
    """y += prolongation(x), in place (mesh.py:334)."""
    torch = L.require_gpu()
    xd = L.to_device(x, torch.float32)
    ctx = L.context(4)
    if L.is_device_array(y):
        ctx.call("pm_prolongation", L.ptr(y), L.ptr(xd), xd.shape[0], 1)
    else:
        yd = L.to_device(y, torch.float32)
        ctx.call("pm_prolongation", L.ptr(yd), L.ptr(xd), xd.shape[0], 1)
        y[...] = yd.cpu().numpy()


# The second set of particle arrays of the re-bucketing gather: exactly six [Np, 3] arrays circulate (three hold the
# particles, three are written by the next gather), no device allocation inside the step loop.  main.tune_placement
# fills the pool with arrays whose placement it has measured; without it the arrays come from torch's allocator.
_rebucket_pool = []


def _pool_take(like):
    for i, t in enumerate(_rebucket_pool):
        if t.shape == like.shape and t.dtype == like.dtype and t.device == like.device:
            return _rebucket_pool.pop(i)
    return L.require_gpu().empty_like(like)


def pool_give(*tensors) -> None:
    """Hand arrays the caller no longer reads (the inputs of a re-bucketing step) back for the next step's outputs."""
    for t in tensors:
        if t is not None and L.is_device_array(t) and t.is_contiguous() and len(_rebucket_pool) < 6:
            if all(t.data_ptr() != u.data_ptr() for u in _rebucket_pool):
                _rebucket_pool.append(t)


def derivative_interp_vec_kick_rebucket(scheme_name: str, potential, order: int, position, velocity, half_dt, ids=None):
    """derivative_interp_vec_kick with the kick, writing the particle arrays out of place in tile order
    (pmstep.h: pm_gradient_interp_kick_rebucket): returns (acceleration, position, velocity, ids) as NEW
    tensors holding the same particles tile after tile -- the incremental form of utils.reorder_particles
    (utils.py:1019), for 12 bytes per particle on top of the gather.  Needs the tile lists the fused kick +
    drift + deposit left prepared for `position`.  `ids`: optional int32 tags carried along."""
    from . import utils

    torch = L.require_gpu()
    n = potential.shape[1]
    scheme = {"cic": L.CIC, "tsc": L.TSC, "ngp": L.NGP}[scheme_name]
    if order not in (0, 2, 3, 5, 7):  # 0: `potential` is the force grid [n, n, n, 3] (interp_vec_kick's input)
        raise ValueError(f"{order=}, should be 2, 3, 5 or 7")
    ctx = L.context(n)
    # (same order as integration's pool_give: an array keeps its role -- position, velocity or acceleration -- from
    # step to step, which is what main.tune_placement's choice of arrays relies on)
    pos_out, vel_out, acc = _pool_take(position), _pool_take(velocity), _pool_take(position)
    ids_out = None if ids is None else torch.empty_like(ids)
    if ids is not None and (ids.dtype != torch.int32 or not ids.is_contiguous() or ids.shape[0] != position.shape[0]):
        raise ValueError("ids must be a contiguous int32 tensor with one entry per particle")
    ctx.call("pm_gradient_interp_kick_rebucket", int(order), scheme, L.ptr(potential), L.ptr(position), L.ptr(velocity),
             position.shape[0], float(np.float32(half_dt)), L.ptr(pos_out), L.ptr(vel_out), L.ptr(acc), L.ptr(ids),
             L.ptr(ids_out))
    utils.register_fused_maxabs(ctx, acc, vel_out)
    return acc, pos_out, vel_out, ids_out


def derivative_interp_vec_kick_codes(scheme_name: str, potential, order: int, position, velocity, half_dt):
    """derivative_interp_vec_kick with the kick, in place (velocity kicked, acceleration returned, nothing permuted),
    which also decides where the NEXT fused kick + drift + deposit files every particle (pmstep.h:
    pm_gradient_interp_kick_codes; "the kick re-buckets").  Needs the tile lists the fused kick left prepared."""
    from . import utils

    n = potential.shape[1]
    scheme = {"cic": L.CIC, "tsc": L.TSC, "ngp": L.NGP}[scheme_name]
    if order not in (0, 2, 3, 5, 7):
        raise ValueError(f"{order=}, should be 2, 3, 5 or 7")
    ctx = L.context(n)
    acc = _pool_take(position)
    ctx.call("pm_gradient_interp_kick_codes", int(order), scheme, L.ptr(potential), L.ptr(position), L.ptr(velocity),
             position.shape[0], float(np.float32(half_dt)), L.ptr(acc))
    utils.register_fused_maxabs(ctx, acc, velocity)
    return acc


def kick_rebucket_ready(position, ncells_1d: int) -> bool:
    """True if the last gather left its decisions for exactly these positions (pm_rebucket_kick_ready)."""
    import ctypes as C

    ready = C.c_int(0)
    L.context(ncells_1d).call("pm_rebucket_kick_ready", L.ptr(position), position.shape[0], C.byref(ready))
    return bool(ready.value)


def kick_drift_wrap_deposit_rebucket(scheme: int, position, velocity, acceleration, half_dt, dt, density, khalo, ids=None):
    """The fused first kick + drift + wrap + deposit writing the new position and velocity out of place, tile after tile,
    as the last gather decided (pmstep.h: pm_kick_drift_wrap_deposit_rebucket).  Returns (position, velocity, ids) as NEW
    tensors; the inputs are only read."""
    torch = L.require_gpu()
    n = density.shape[0]
    pos_out, vel_out = _pool_take(position), _pool_take(velocity)
    ids_out = None if ids is None else torch.empty_like(ids)
    L.context(n).call("pm_kick_drift_wrap_deposit_rebucket", int(scheme), L.ptr(position), L.ptr(velocity), L.ptr(acceleration),
                      float(np.float32(half_dt)), float(np.float32(dt)), position.shape[0], L.ptr(density), L.ptr(khalo),
                      L.ptr(pos_out), L.ptr(vel_out), L.ptr(ids), L.ptr(ids_out))
    return pos_out, vel_out, ids_out

