"""The N-body section of the reference's driver (main.py:116-156) on top of solver.pm / integration.integrate:
first force evaluation, step loop up to the last output redshift, re-ordering of the particle stream every
`n_reorder` steps, snapshot hand-off.  Background tables and initial conditions are inputs here (the reference
builds them with astropy / its IC generator, main.py:104-112; both are outside this library's scope).

The re-ordering is more than a locality nicety on the GPU: the tile-range kernels serve every particle from LDS
whatever the order is (per-tile straggler lists), but the lists -- and with them the time per step -- grow with the
age of the sort; `n_reorder` bounds it exactly as it bounds cache misses in the reference.
"""
from __future__ import annotations

import logging
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import _lib as L
from . import integration, solver, utils


REBUCKET_FRACTION = 0.0    # default of param["rebucket_straggler_fraction"] in this loop (device arrays): every step
                           # that found a straggler re-buckets (measured at 1024^3: 47.7 ms per step sustained against
                           # 48.5 / 49.7 with 0.02 / 0.04, where in-place and re-bucketing steps alternate)
REORDER_FRACTION = 0.10    # default of param["reorder_straggler_fraction"]


def reserve_step_buffers(position) -> None:
    """The re-bucketing gather writes position, velocity and acceleration into a second set of arrays.  Allocating
    them once up front (they go back to torch's caching allocator at once and are handed out again by the steps)
    keeps the device allocations -- 0.8 s per 12.9 GB array at 1024^3 -- out of the step loop."""
    if L.is_device_array(position):
        torch = L.require_gpu()
        spare = [torch.empty_like(position) for _ in range(3)]
        del spare


def reorder_policy(position, velocity, acceleration, param):
    """What the loop does about the particle order after a step (main.py:146-150 re-sorts every n_reorder steps;
    on the GPU the order is a performance matter only: every kernel serves every particle whatever it is).
    Device arrays: integration.integrate keeps the stream tile-sorted on its own (incremental re-sort inside the
    gather, param["rebucket_straggler_fraction"]); a full Morton sort happens only when the stream has aged past
    param["reorder_straggler_fraction"] anyway or the straggler lists overflowed (unsorted input), never on the
    fixed cadence.  NumPy arrays or no incremental re-sort: the reference's cadence, plus the aged test."""
    device = L.is_device_array(position)
    frac, complete = (None, True)
    if device:
        frac, complete = utils.straggler_fraction(2 ** int(param["ncoarse"]))
    aged = (not complete) or (frac is not None and frac > float(param.get("reorder_straggler_fraction", REORDER_FRACTION)))
    incremental = device and param.get("rebucket_straggler_fraction") is not None
    cadence = (not incremental) and param.get("n_reorder") and (param["nsteps"] % param["n_reorder"]) == 0
    if cadence or aged:
        logging.info(f"Reordering particles ({frac=})")
        position, velocity, acceleration = _reorder(position, velocity, acceleration, param)
    return position, velocity, acceleration


def _reorder(position, velocity, acceleration, param):
    """utils.reorder_particles, with the optional tags of param["particle_id"] following the permutation."""
    ids = param.get("particle_id") if L.is_device_array(position) else None
    order = [] if ids is not None else None
    position, velocity, acceleration = utils.reorder_particles(position, velocity, acceleration, _order_out=order)
    if ids is not None:
        param["particle_id"] = ids[order[0]]
    return position, velocity, acceleration


def run_nbody(position, velocity, param, tables: List, z_out: Sequence[float],
              on_snapshot: Optional[Callable] = None, on_step: Optional[Callable] = None):
    """main.py:116-156.  `param` needs what solver.pm / integration.integrate need plus `n_reorder`; `aexp`, `t`
    and the units must be set (main.py:106-113).  `on_snapshot(position, velocity, param)` replaces
    iostream.write_snapshot_particles; `on_step(param)` is called after every step.
    Returns (position, velocity, acceleration, potential, additional_field)."""
    if "nsteps" not in param:
        param["nsteps"] = 0
    if L.is_device_array(position):
        param.setdefault("rebucket_straggler_fraction", REBUCKET_FRACTION)
        if param["rebucket_straggler_fraction"] is not None:
            reserve_step_buffers(position)
    acceleration, potential, additional_field = solver.pm(position, param, tables=tables)
    if L.is_device_array(position):  # an unsorted input stream (initial conditions in lattice order) is sorted once
        frac, complete = utils.straggler_fraction(2 ** int(param["ncoarse"]))
        if not complete or (frac is not None and frac > float(param.get("reorder_straggler_fraction", REORDER_FRACTION))):
            position, velocity, acceleration = _reorder(position, velocity, acceleration, param)
    aexp_out = 1.0 / (np.array(z_out, dtype=np.float64) + 1)
    aexp_out.sort()
    t_out = tables[1](np.log(aexp_out))
    logging.info(f"{aexp_out=}")
    if "i_snap" not in param:
        param["i_snap"] = 1
    else:
        param["i_snap"] += 1
    while param["aexp"] < aexp_out[-1]:
        param["nsteps"] += 1
        position, velocity, acceleration, potential, additional_field = integration.integrate(
            position, velocity, acceleration, potential, additional_field, tables, param,
            t_out[param["i_snap"] - 1])
        position, velocity, acceleration = reorder_policy(position, velocity, acceleration, param)
        if param["write_snapshot"]:
            if on_snapshot is not None:
                on_snapshot(position, velocity, param)
            param["i_snap"] += 1
        if on_step is not None:
            on_step(param)
        logging.info(f"{param['nsteps']=} {param['aexp']=} z = {1.0 / param['aexp'] - 1}")
    return position, velocity, acceleration, potential, additional_field
