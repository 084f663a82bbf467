"""Multigrid drivers: linear V-cycles for the Poisson equation and Full-Approximation-Scheme
V-cycles for the f(R) scalaron (reference multigrid.py:23-138, 142-579).

The recursion over levels and the data-dependent stopping rule stay on the host exactly as in the
reference; every level operation is a HIP kernel.  The reference's F and W cycles
(multigrid.py:583-858) are never called by linear()/FAS() and are not mirrored; its non-f(R) FAS
branch (laplacian_reformulated) is unreachable from solver.pm() and raises here.
"""
from __future__ import annotations

import ctypes as C
import logging

import numpy as np

from . import _lib as L
from . import cubic, laplacian, mesh, quartic, utils


def _is_fr_field(param) -> bool:
    return bool(param["compute_additional_field"]) and "fr" == param["theory"].casefold()


def linear(x, b, param):
    """Solve Laplacian(x) = b to epsrel x truncation error with V-cycles (multigrid.py:23)."""
    THEORY = param["theory"].casefold()
    if param["compute_additional_field"] and "fr" == THEORY:
        raise ValueError("Linear should not be used for scalaron field")
    mond_pot = (not param["compute_additional_field"]) and "mond" == THEORY
    if ("tolerance" not in param) or (param["nsteps"] % 3) == 0:
        logging.info("Compute Truncation error")
        tolerance = param["epsrel"] * laplacian.truncation_error(x)
        if mond_pot:
            param["tolerance_mond"] = tolerance
        else:
            param["tolerance"] = tolerance
    tolerance = param["tolerance_mond"] if mond_pot else param["tolerance"]
    logging.info("Start linear Multigrid")
    n = x.shape[0]
    if (_NATIVE_VCYCLE and _NATIVE_SOLVE and L.is_device_array(x) and L.is_device_array(b) and n >= 8 and (n & (n - 1)) == 0
            and param["ncoarse"] == int(np.log2(n)) and str(x.dtype) == "torch.float32"):
        # the loop below issued from C, one call (pmstep.h: pm_lap_solve): same kernels, same comparisons, same count
        ctx = L.context(4)
        if getattr(ctx, "_vcycle_graphs", True) != _VCYCLE_GRAPHS:
            ctx.call("pm_set_vcycle_graphs", int(_VCYCLE_GRAPHS))
            ctx._vcycle_graphs = _VCYCLE_GRAPHS
        cycles, res = C.c_int(0), C.c_float(0)
        ctx.call("pm_lap_solve", L.ptr(x), L.ptr(b), n, int(param["Npre"]), int(param["Npost"]), float(np.float32(tolerance)), 0,
                 C.byref(cycles), C.byref(res))
        param["_vcycles_linear"] = param.get("_vcycles_linear", 0) + cycles.value
        logging.info(f"residual_error={res.value} {tolerance=} after {cycles.value} V-cycles")
        return x
    residual_err = 1e30
    while residual_err > tolerance:
        V_cycle(x, b, param)
        param["_vcycles_linear"] = param.get("_vcycles_linear", 0) + 1  # (bookkeeping for bench.py, not a parameter)
        residual_error_tmp = laplacian.residual_error(x, b)
        logging.info(f"{residual_error_tmp=} {tolerance=}")
        if residual_error_tmp < tolerance or residual_err / residual_error_tmp < 2:
            break
        residual_err = residual_error_tmp
    return x


_NATIVE_VCYCLE = True  # tests flip this to compare with the recursion written out below
_NATIVE_SOLVE = True   # ... and this for the stopping loops written out in linear() / FAS() (pm_lap_solve / pm_fr_solve otherwise)
_VCYCLE_GRAPHS = True  # False: pm_lap_vcycle issues plain launches instead of replaying captured hipGraphs


def V_cycle(x, b, param, nlevel: int = 0) -> None:
    """One linear V-cycle, in place on x (multigrid.py:474)."""
    n = x.shape[0]
    if (_NATIVE_VCYCLE and n >= 8 and (n & (n - 1)) == 0 and (param["ncoarse"] - nlevel) == int(np.log2(n))
            and x.dtype == (np.float32 if not L.is_device_array(x) else L.require_gpu().float32)):
        # the same sequence of kernels issued from C in one call (pmstep.h: pm_lap_vcycle)
        torch = L.require_gpu()
        xd, bd = L.to_device(x, torch.float32), L.to_device(b, torch.float32)
        ctx = L.context(4)
        if getattr(ctx, "_vcycle_graphs", True) != _VCYCLE_GRAPHS:
            ctx.call("pm_set_vcycle_graphs", int(_VCYCLE_GRAPHS))
            ctx._vcycle_graphs = _VCYCLE_GRAPHS
        ctx.call("pm_lap_vcycle", L.ptr(xd), L.ptr(bd), n, int(param["Npre"]), int(param["Npost"]))
        if not L.is_device_array(x):
            x[...] = xd.cpu().numpy()
        return
    laplacian.smoothing(x, b, param["Npre"])
    res_c = laplacian.restrict_residual(x, b)
    x_corr_c = laplacian.initialise_potential(res_c)
    if nlevel >= (param["ncoarse"] - 3):
        laplacian.smoothing(x_corr_c, res_c, param["Npre"])
    else:
        V_cycle(x_corr_c, res_c, param, nlevel + 1)
    del res_c
    mesh.add_prolongation(x, x_corr_c)
    del x_corr_c
    laplacian.smoothing(x, b, param["Npost"])


# ---- dispatch helpers (multigrid.py:142-470); only the f(R) n = 1 branch is reachable from pm() --------


def _fr(param):
    """(module, q) of the scalaron operator: cubic for n = 1, quartic for n = 2."""
    if not _is_fr_field(param):
        raise NotImplementedError(
            "FAS for non-f(R) theories (laplacian_reformulated) is never reached from solver.pm() and is not built")
    if param["fR_n"] == 1:
        return cubic, np.float32(param["fR_q"])
    if param["fR_n"] == 2:
        return quartic, np.float32(param["fR_q"])
    raise NotImplementedError(f"Only f(R) with n = 1 and 2, currently {param['fR_n']=}")


def truncation_error(x, param, b=None) -> np.float32:  # multigrid.py:142
    mod, q = _fr(param)
    return mod.truncation_error(x, b, q)


def normalisation_residual(param) -> np.float32:  # multigrid.py:192
    return np.float32(4)


def residual_error(x, b, param) -> np.float32:  # multigrid.py:231
    mod, q = _fr(param)
    return mod.residual_error(x, b, q)


def restrict_residual(x, b, param, rhs=None):  # multigrid.py:282
    mod, q = _fr(param)
    if rhs is None or len(rhs) == 0:
        return mesh.minus_restriction(mod.operator(x, b, q))
    return mesh.restriction(mod.residual_with_rhs(x, b, q, rhs))


def smoothing(x, b, n_smoothing, param, rhs=None) -> None:  # multigrid.py:350
    mod, q = _fr(param)
    if rhs is None or len(rhs) == 0:
        mod.smoothing(x, b, q, n_smoothing)
    else:
        mod.smoothing_with_rhs(x, b, q, n_smoothing, rhs)


def operator(x, param, b=None):  # multigrid.py:420
    mod, q = _fr(param)
    return mod.operator(x, b, q)


def FAS(x, b, param):
    """Non-linear multigrid (Full Approximation Scheme) for the scalaron (multigrid.py:88)."""
    if ("tolerance_FAS" not in param) or (param["nsteps"] % 3) == 0:
        logging.info("Compute FAS Truncation error")
        param["tolerance_FAS"] = param["epsrel"] * truncation_error(x, param, b)
    tolerance = param["tolerance_FAS"]
    logging.info("Start Full-Approximation Storage Multigrid")
    n = x.shape[0]
    if (_NATIVE_VCYCLE and _NATIVE_SOLVE and _is_fr_field(param) and param["fR_n"] in (1, 2) and n >= 8 and (n & (n - 1)) == 0
            and param["ncoarse"] == int(np.log2(n)) and L.is_device_array(x) and L.is_device_array(b)):
        cycles, res = C.c_int(0), C.c_float(0)  # (pmstep.h: pm_fr_solve -- the loop below, issued from C)
        L.context(4).call("pm_fr_solve", int(param["fR_n"]), L.ptr(x), L.ptr(b), float(np.float32(param["fR_q"])), n,
                          int(param["Npre"]), int(param["Npost"]), float(np.float32(tolerance)), 0, C.byref(cycles), C.byref(res))
        param["_vcycles_fas"] = param.get("_vcycles_fas", 0) + cycles.value
        logging.info(f"residual_error={res.value} {tolerance=} after {cycles.value} FAS V-cycles")
        return x
    residual_err = 1e30
    while residual_err > tolerance:
        V_cycle_FAS(x, b, param)
        param["_vcycles_fas"] = param.get("_vcycles_fas", 0) + 1
        residual_error_tmp = residual_error(x, b, param)
        logging.info(f"{residual_error_tmp=} {tolerance=}")
        if residual_error_tmp < tolerance or residual_err / residual_error_tmp < 2:
            break
        residual_err = residual_error_tmp
    return x


def V_cycle_FAS(x, b, param, nlevel: int = 0, rhs=None) -> None:
    """One FAS V-cycle, in place on x (multigrid.py:521)."""
    n = x.shape[0]
    if (_NATIVE_VCYCLE and nlevel == 0 and (rhs is None or len(rhs) == 0) and _is_fr_field(param)
            and param["fR_n"] in (1, 2) and n >= 8 and (n & (n - 1)) == 0 and param["ncoarse"] == int(np.log2(n))
            and L.is_device_array(x) and L.is_device_array(b)):
        # the same sequence of kernels issued from C in one call (pmstep.h: pm_fr_vcycle)
        L.context(4).call("pm_fr_vcycle", int(param["fR_n"]), L.ptr(x), L.ptr(b), float(np.float32(param["fR_q"])), n,
                          int(param["Npre"]), int(param["Npost"]))
        return
    smoothing(x, b, param["Npre"], param, rhs)
    res_c = restrict_residual(x, b, param, rhs)
    x_c = mesh.restriction(x)
    x_corr_c = x_c.clone() if L.is_device_array(x_c) else x_c.copy()
    b_c = mesh.restriction(b)
    L_c = operator(x_c, param, b_c)
    utils.linear_operator_vectors_inplace(res_c, normalisation_residual(param), L_c, np.float32(1))
    del L_c
    if nlevel >= (param["ncoarse"] - 3):
        smoothing(x_corr_c, b_c, param["Npre"], param, res_c)
    else:
        V_cycle_FAS(x_corr_c, b_c, param, nlevel + 1, res_c)
    del res_c, b_c
    utils.add_vector_scalar_inplace(x_corr_c, x_c, np.float32(-1))
    del x_c
    mesh.add_prolongation(x, x_corr_c)
    del x_corr_c
    smoothing(x, b, param["Npost"], param, rhs)
