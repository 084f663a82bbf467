"""Time integration: drop-in for the reference's integration.integrate() (integration.py:17-358).

dt selection and the integrator dispatch stay on the host (scalars); the particle passes are HIP
kernels.  With torch CUDA tensors the particle state never leaves the GPU between steps.
"""
from __future__ import annotations

import logging
from typing import List

import numpy as np

from . import _lib as L
from . import solver, utils


def predict_next_step(dt, dt1, dt2, tables, param, t_snap_next, meanabs) -> None:
    """Guesses for the incremental re-sort, left in param["_dt_pred"] (the NEXT step's dt) and
    param["_acc_scale_pred"] (growth of the accelerations over the step about to be taken); `meanabs`: mean |a_x| of
    the current accelerations (utils.fused_acc_meanabs) or None.  Shared by integrate() and the slab step."""
    # Guess of the NEXT step's dt for the incremental re-sort (pmstep.h: pm_set_predicted_dt): the three criteria
    # again -- the scale-factor one at the time this step ends (known exactly), the two CFL ones extrapolated
    # with their own trends -- and the next snapshot's clipping.  It anticipates the step on which the binding
    # criterion changes, which an extrapolation of dt itself gets wrong by tens of per cent (3 % of the particles
    # misfiled, one 60 ms step at 1024^3).
    # First, how much the accelerations grow over this step (the filing uses scale * a_old for the a_new it
    # cannot know, pmstep.h: pm_set_predicted_acc_scale): the ratio of the mean |a_x| over the last step, to the power of the
    # ratio of the step lengths.  In the linear regime a_new / a_old is the same number for every particle and
    # this removes most of the misfilings (tools/exp_predict_tiles.py: 0.42 % -> 0.08 % at the longest steps).
    ms = meanabs
    ms_last, dt_between = param.get("_acc_meanabs_last"), param.get("_dt_last")
    scale = grown = 1.0  # (grown: a_n / a_{n-1}, over the last step)
    if ms and ms_last and dt_between and dt_between > 0 and dt > 0:
        grown = float(ms / ms_last)
        scale = grown ** min(2.0, float(dt) / dt_between)
        scale = min(2.0, max(0.5, scale)) if np.isfinite(scale) else 1.0
        grown = min(2.0, max(0.5, grown)) if np.isfinite(grown) else 1.0
    param["_acc_meanabs_last"], param["_acc_scale_pred"] = ms, scale
    try:
        t_end = param["t"] + dt
        a_end = float(np.exp(utils.table_eval(tables[0], t_end)))
        f = 1.0 + 0.01 * param["max_aexp_stepping"]
        dt3_next = float(utils.table_eval(tables[1], np.log(f * a_end)) - utils.table_eval(tables[1], np.log(a_end)))
        # (each CFL criterion with its own trend over the step about to be taken, h = dt, from its change over
        # the last one, h_last: max|a| -- dt1 ~ max|a|^-1/2 -- grows geometrically in time; max|v| -- dt2 ~
        # 1 / max|v| -- by what the kicks add, i.e. in proportion to the time and to the accelerations: extrapolating
        # dt2's RATIO instead undershoots by 4-10 % on the steps after the velocity criterion takes over, where h
        # shrinks quickly, and time alone overshoots as much.
        # Both within [0.5, 1.5] of the present value.)
        last = param.get("_dt_cfl_last") or (float(dt1), float(dt2))
        param["_dt_cfl_last"] = (float(dt1), float(dt2))
        h_last = param.get("_dt_last")
        r = min(2.0, float(dt) / h_last) if (h_last and h_last > 0) else 1.0
        clamp = lambda x: min(1.5, max(0.5, x)) if np.isfinite(x) else 1.0  # noqa: E731
        f1 = clamp((float(dt1) / last[0]) ** r) if last[0] > 0 and dt1 > 0 else 1.0
        # (a step h adds h (a_before + a_after) / 2 to a velocity: the next increment of max|v| over the last one)
        u, ul = 1.0 / float(dt2), 1.0 / last[1]
        u_next = u + (u - ul) * r * grown * (1.0 + scale) / (1.0 + grown)
        f2 = clamp(u / u_next) if u_next > 0 else 1.5
        guess = min(float(dt1) * f1, float(dt2) * f2, dt3_next)
        if t_end < t_snap_next:
            guess = min(guess, float(t_snap_next) - t_end)
        param["_dt_pred"] = guess if np.isfinite(guess) and guess > 0 else None
    except (KeyError, IndexError, TypeError, ValueError, ArithmeticError) as err:
        # (tables that cannot be evaluated ahead, a degenerate criterion: fall back to the trend of dt -- the guess
        # only decides where particles are FILED, never what is computed)
        logging.debug(f"no prediction of the next step ({err!r})")
        param["_dt_pred"] = None


def integrate(position, velocity, acceleration, potential, additional_field, tables: List, param,
              t_snap_next=np.float32(0)):
    """One step: dt = min(free-fall, CFL, scale-factor criterion), then leapfrog or euler."""
    from . import real64, slab

    if real64.is_float64(position):  # the fp64 device mode (Newtonian FFT step, leapfrog)
        if slab.active():
            raise NotImplementedError("the fp64 device mode is single-GPU")
        return real64.integrate(position, velocity, acceleration, potential, additional_field, tables, param, t_snap_next)
    if slab.active():  # slab-decomposed step: same logic with all-reduced maxima + particle migration
        return slab.get(2 ** int(param["ncoarse"])).integrate(position, velocity, acceleration, potential,
                                                              additional_field, tables, param, t_snap_next)
    dt1 = dt_CFL_maxacc(acceleration, param)
    dt2 = dt_CFL_maxvel(velocity, param)
    dt3 = dt_weak_variation(tables[1], param)
    dt = np.min([dt1, dt2, dt3])
    if (param["t"] + dt) > t_snap_next:
        dt = t_snap_next - param["t"]
        param["write_snapshot"] = True
    else:
        param["write_snapshot"] = False
    if param.get("rebucket_straggler_fraction") is not None:
        predict_next_step(dt, dt1, dt2, tables, param, t_snap_next,
                          utils.fused_acc_meanabs(acceleration) if L.is_device_array(acceleration) else None)
    logging.info(f"Conditions: velocity {dt1=}, acceleration {dt2=}, scale factor {dt3=}")
    INTEGRATOR = param["integrator"].casefold()
    if INTEGRATOR == "leapfrog":
        return leapfrog(position, velocity, acceleration, potential, additional_field, dt, tables, param)
    if INTEGRATOR == "euler":
        return euler(position, velocity, acceleration, potential, additional_field, dt, tables, param)
    raise NotImplementedError("ERROR: Integrator must be 'leapfrog' or 'euler'")


def _advance_clock(dt, tables, param) -> None:
    param["t"] += dt
    param["aexp_old"] = param["aexp"]
    param["aexp"] = np.exp(utils.table_eval(tables[0], param["t"]))
    utils.set_units(param)


def euler(position, velocity, acceleration, potential, additional_field, dt, tables, param):
    """Drift, wrap, kick with the old acceleration, then pm (integration.py:121-189)."""
    utils.add_vector_scalar_inplace(position, velocity, dt)
    _advance_clock(dt, tables, param)
    utils.periodic_wrap(position)
    utils.add_vector_scalar_inplace(velocity, acceleration, -dt)
    acceleration, potential, additional_field = solver.pm(position, param, potential, additional_field, tables)
    return position, velocity, acceleration, potential, additional_field


_PREDICT = True       # incremental re-sort: file particles under the tile they are expected to be in NEXT step
_KICK_REBUCKETS = True  # ... decided by the gather (in place), carried out by the next fused kick + deposit, which
                        # rewrites position and velocity anyway (pmstep.h: pm_gradient_interp_kick_codes); False: the
                        # gather itself writes the three arrays out of place (pm_gradient_interp_kick_rebucket)
_KR_HOLD_STEPS = 16     # after a gather whose mover lists overflowed (pm_rebucket_kick_ready says no): that many steps
                        # are re-bucketed by the gather itself (which has no lists to overflow) before the lists are tried again
_FUSE_DEPOSIT = True  # tests flip this to compare with the separate kick-drift-wrap and deposit
_USE_KHALO = True     # ... and this for the plain flush (every tile cell through global atomics)
_khalo_buffers = {}


def _khalo_buffer(torch, ncells_1d, device):
    """The halo side buffer of pm_kick_drift_wrap_deposit_khalo (pmstep.h; 0.58 n^3 floats), kept per size."""
    key = (str(device), ncells_1d)
    buf = _khalo_buffers.get(key)
    if buf is None:
        _khalo_buffers.clear()
        buf = _khalo_buffers[key] = torch.empty((int(L.load_library().pm_khalo_floats(ncells_1d)),), dtype=torch.float32,
                                                device=device)
    return buf


def leapfrog(position, velocity, acceleration, potential, additional_field, dt, tables, param):
    """Kick-drift-kick (integration.py:192-264).  On device tensors the first kick, the drift and
    the periodic wrap are one fused pass over the particle arrays."""
    half_dt = np.float32(0.5 * dt)
    acceleration_in = acceleration
    kick_rebucketed = False
    fused = L.is_device_array(position) and L.is_device_array(velocity) and L.is_device_array(acceleration)
    density = None
    if fused:
        from . import slab

        ncells_1d = 2 ** int(param["ncoarse"])
        scheme = {"ngp": L.NGP, "cic": L.CIC, "tsc": L.TSC}.get(param["mass_scheme"].casefold())  # (mesh.py:2240-2595)
        if (_FUSE_DEPOSIT and scheme is not None and ncells_1d >= 16 and not slab.active()
                and 0 < position.shape[0] < 2**32):
            # the deposit of pm() rides on the streaming update (pmstep.h: pm_kick_drift_wrap_deposit)
            torch = L.require_gpu()
            density = torch.empty((ncells_1d,) * 3, dtype=torch.float32, device=position.device)
            if param.get("rebucket_straggler_fraction") is not None:
                # guess of the next step's dt, for the tile the re-bucketing gather files every particle under:
                # this step's, scaled by how it changed from the last one (pmstep.h: pm_set_predicted_dt)
                dt_last = param.get("_dt_last")
                ratio = min(1.5, max(0.5, float(dt) / dt_last)) if (dt_last and dt_last > 0) else 1.0
                param["_dt_last"] = float(dt)
                guess = param.get("_dt_pred") or float(dt) * ratio  # (integrate()'s estimate; dt's trend without it)
                L.context(ncells_1d).call("pm_set_predicted_dt", float(np.float32(guess)) if _PREDICT and dt > 0 else 0.0)
                L.context(ncells_1d).call("pm_set_predicted_acc_scale", float(np.float32(param.get("_acc_scale_pred", 1.0))))
            # the halo side buffer pays off when the fused FFT solve consumes it (its z pass adds it while loading
            # the density); every other solver needs the completed density: for those, the plain flush
            khalo_consumer = (param["theory"].casefold() in ("newton", "parametrized")
                              and param["linear_newton_solver"].casefold() in ("fft", "fft_7pt")
                              and 32 <= ncells_1d <= 1024 and solver._USE_FUSED_FFT
                              and param["save_power_spectrum"].casefold() == "no")
            use_khalo = bool(_USE_KHALO and (khalo_consumer or _USE_KHALO == "always"))
            from . import mesh as _mesh

            kr_ready = (param.get("rebucket_straggler_fraction") is not None and _KICK_REBUCKETS
                        and _mesh.kick_rebucket_ready(position, ncells_1d))
            if (not kr_ready and _KICK_REBUCKETS and param.get("_kr_decided") and param.get("rebucket_straggler_fraction") is not None):
                # the last gather decided, and its decisions cannot be used (a mover list overflowed: a cluster on a tile
                # face): this kick runs in place, and this step's gather and the next few re-bucket the old way
                logging.warning("the re-bucketing kick's mover lists overflowed: re-bucketing inside the gather for "
                                f"{_KR_HOLD_STEPS} steps")
                param["_kr_hold"] = _KR_HOLD_STEPS
            param["_kr_decided"] = False
            if kr_ready:
                # "the kick re-buckets": the last gather decided where every particle goes; this pass, which rewrites
                # position and velocity anyway, writes them out of place in that order.  The arrays read become the
                # next step's output arrays.
                khalo = _khalo_buffer(torch, ncells_1d, position.device) if use_khalo else None
                position_in, velocity_in = position, velocity
                position, velocity, ids = _mesh.kick_drift_wrap_deposit_rebucket(
                    scheme, position, velocity, acceleration, half_dt, dt, density, khalo, param.get("particle_id"))
                _mesh.pool_give(position_in, velocity_in)
                kick_rebucketed = True
                if ids is not None:
                    param["particle_id"] = ids
                if use_khalo:
                    density = (density, khalo)
            elif use_khalo:
                # the tiles' halo cells go to a side buffer and the own cells to the density, all with plain
                # stores instead of memory-side atomics; solver.pm hands the buffer to the FFT solve (or has
                # it added to the density)
                khalo = _khalo_buffer(torch, ncells_1d, position.device)
                L.context(ncells_1d).call("pm_kick_drift_wrap_deposit_khalo", scheme, L.ptr(position), L.ptr(velocity),
                                          L.ptr(acceleration), float(half_dt), float(np.float32(dt)),
                                          position.shape[0], L.ptr(density), L.ptr(khalo))
                density = (density, khalo)
            else:
                L.context(ncells_1d).call("pm_kick_drift_wrap_deposit", scheme, L.ptr(position), L.ptr(velocity),
                                          L.ptr(acceleration), float(half_dt), float(np.float32(dt)),
                                          position.shape[0], L.ptr(density))
        else:
            L.context(4).call("pm_kick_drift_wrap", L.ptr(position), L.ptr(velocity), L.ptr(acceleration),
                              float(half_dt), float(np.float32(dt)), position.shape[0])
        _advance_clock(dt, tables, param)
    else:
        utils.add_vector_scalar_inplace(velocity, acceleration, -half_dt)
        utils.add_vector_scalar_inplace(position, velocity, dt)
        _advance_clock(dt, tables, param)
        utils.periodic_wrap(position)
    logging.info(f"{param['t']=} {param['aexp']=}")
    if fused:  # second kick fused into the interpolation kernel
        rebucket = _rebucket_request(density, param, position, kick_rebucketed)
        acceleration, potential, additional_field = solver.pm(position, param, potential, additional_field, tables,
                                                              _kick=(velocity, half_dt), _density=density,
                                                              _rebucket=rebucket)
        if rebucket is not None and rebucket.get("mode") == "codes" and rebucket.get("done"):
            # the gather stayed in place (a new acceleration array from the pool, the old one goes back): nothing permuted
            from . import mesh

            param["_kr_decided"] = True  # (the next kick checks that the decisions could be used)

            if acceleration_in.data_ptr() != acceleration.data_ptr():
                mesh.pool_give(acceleration_in)
        elif rebucket is not None and "position" in rebucket:
            # the gather wrote the arrays out of place, tile after tile (pmstep.h: pm_gradient_interp_kick_rebucket);
            # the arrays it read are stale from here on: they become the next step's output arrays
            from . import mesh

            mesh.pool_give(position, velocity, acceleration_in)
            position, velocity = rebucket["position"], rebucket["velocity"]
            if rebucket["ids"] is not None:
                param["particle_id"] = rebucket["ids"]
    else:
        acceleration, potential, additional_field = solver.pm(position, param, potential, additional_field, tables)
        utils.add_vector_scalar_inplace(velocity, acceleration, -half_dt)
    return position, velocity, acceleration, potential, additional_field


_aging_warned = False


def _warn_if_aging(param) -> None:
    """In-place steps (the reference's semantics) leave the particle stream in the order of its last sort; the fused
    kernels serve whoever has left its tile since from lists, more slowly.  Say so once when that share gets expensive
    ("Particle order": 24 ms instead of 13.5 for the fused kick + deposit at 4 %, 1024^3)."""
    global _aging_warned
    if _aging_warned:
        return
    frac, _ = utils.straggler_fraction(2 ** int(param["ncoarse"]))
    if frac is not None and frac > 0.03:
        _aging_warned = True
        logging.warning(f"{100 * frac:.1f} % of the particles are outside the tile their last sort filed them under: "
                        "re-sort (utils.reorder_particles) or set param['rebucket_straggler_fraction'] = 0 to let every "
                        "step's gather keep the stream tile-sorted (integrate() then returns new, permuted tensors)")


def _rebucket_request(density, param, position, kick_rebucketed=False):
    """Incremental re-sort policy (no reference counterpart; the reference re-sorts every n_reorder steps,
    main.py:146-150).  With param["rebucket_straggler_fraction"] = f set (main.run_nbody sets it to 0: every step), a
    step whose fused kick + drift + deposit found more than f of the particles outside the tile they are filed under has
    its gather write position, velocity and acceleration out of place in tile order: integrate() then returns NEW
    tensors (same particles, permuted; param["particle_id"], an optional int32 tensor, follows them)."""
    thr = param.get("rebucket_straggler_fraction")
    if thr is None and density is not None:
        _warn_if_aging(param)
    if thr is None or density is None:
        return None
    frac, complete = utils.straggler_fraction(2 ** int(param["ncoarse"]))
    if frac is None or not complete:
        return None
    # f <= 0: every step -- also one that found nobody out of place: the particles are filed for the NEXT drift, and a
    # stream left as it is hands the next step every tile crosser of that drift (a perfect prediction at 32^3 used
    # to be punished that way every other step)
    if float(thr) > 0.0 and frac <= float(thr):
        return None
    hold = int(param.get("_kr_hold", 0) or 0)
    if hold > 0 and not kick_rebucketed:
        param["_kr_hold"] = hold - 1
    elif _KICK_REBUCKETS and _PREDICT and float(thr) <= 0.0 and (param.get("_dt_pred") or 0.0) > 0.0:
        return {"mode": "codes"}  # the gather decides, the next fused kick carries it out (every step: f <= 0 only)
    if kick_rebucketed:  # (this step's fused kick already re-sorted, and left no counts for a re-bucketing gather)
        return None
    return {"ids": param.get("particle_id")}


def dt_CFL_maxacc(acceleration, param) -> np.float32:
    """Free-fall criterion C*sqrt(dx/max|a|) (integration.py:267-295)."""
    dx = np.float32(0.5 ** param["ncoarse"])
    max_acc = utils.max_abs(acceleration)
    return np.float32(param["Courant_factor"]) * np.sqrt(dx / max_acc)


def dt_CFL_maxvel(velocity, param) -> np.float32:
    """Courant criterion C*dx/max|v| (integration.py:298-326)."""
    dx = np.float32(0.5 ** param["ncoarse"])
    max_vel = utils.max_abs(velocity)
    return np.float32(param["Courant_factor"]) * dx / max_vel


def dt_weak_variation(func_t_a, param) -> np.float32:
    """Step giving a max_aexp_stepping % change of the scale factor (integration.py:329-358)."""
    aexp_factor = 1.0 + 0.01 * param["max_aexp_stepping"]
    return np.float32(utils.table_eval(func_t_a, np.log(aexp_factor * param["aexp"]))
                      - utils.table_eval(func_t_a, np.log(param["aexp"])))
