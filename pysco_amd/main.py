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


def L_is_device(x) -> bool:
    return L.is_device_array(x)


def run_nbody(position, velocity, param, tables: List, z_out: Sequence[float],
              on_snapshot: Optional[Callable] = None, on_step: Optional[Callable] = None):
    """main.py:116-156.  `param` needs what solver.pm / integration.integrate need plus `n_reorder`; `aexp`, `t`
    and the units must be set (main.py:106-113).  `on_snapshot(position, velocity, param)` replaces
    iostream.write_snapshot_particles; `on_step(param)` is called after every step.
    Returns (position, velocity, acceleration, potential, additional_field)."""
    if "nsteps" not in param:
        param["nsteps"] = 0
    acceleration, potential, additional_field = solver.pm(position, param, tables=tables)
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
        # besides the reference's fixed cadence: re-sort when more than `reorder_straggler_fraction` of the
        # particles (default 10 %) have left the tile the last sort filed them under -- the cost of serving them
        # has then grown to about that of the sort (one step's worth); particle order is not an observable
        frac, complete = (None, True)
        if L_is_device(position):
            frac, complete = utils.straggler_fraction(2 ** int(param["ncoarse"]))
        aged = (not complete) or (frac is not None and frac > float(param.get("reorder_straggler_fraction", 0.10)))
        if (param["nsteps"] % param["n_reorder"]) == 0 or aged:
            logging.info(f"Reordering particles ({frac=})")
            position, velocity, acceleration = utils.reorder_particles(position, velocity, acceleration)
        if param["write_snapshot"]:
            if on_snapshot is not None:
                on_snapshot(position, velocity, param)
            param["i_snap"] += 1
        if on_step is not None:
            on_step(param)
        logging.info(f"{param['nsteps']=} {param['aexp']=} z = {1.0 / param['aexp'] - 1}")
    return position, velocity, acceleration, potential, additional_field
