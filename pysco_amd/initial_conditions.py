"""Initial conditions: drop-in for the reference's initial_conditions.generate() (initial_conditions.py:25-213).

Rows f2 / f3 of SURVEY.md section 8: restart from a snapshot this library (or PySCo) wrote -- `initial_conditions` an
integer -- in either format, with the snapshot's parameters taking over the run's (initial_conditions.py:79-107); and
first / second / third order Lagrangian perturbation theory from a power spectrum table (`1LPT`, `2LPT`, `3LPT`:
pysco_amd/lpt.py, grid operations on the GPU).  RayGal and Gadget inputs are not built.
"""
from __future__ import annotations

import logging

from . import iostream


def generate(param, tables):
    """(position, velocity) float32 [Np, 3] host arrays."""
    ic = param["initial_conditions"]
    if isinstance(ic, (int,)) and not isinstance(ic, bool):
        i_restart = int(ic)
        param["initial_conditions"] = i_restart
        fmt = param["output_snapshot_format"].casefold()
        directory = f"{param['base']}/output_{i_restart:05d}"
        if fmt == "parquet":
            position, velocity = iostream.read_snapshot_particles_parquet(f"{directory}/particles_{param['extra']}.parquet")
            param_filename = f"{directory}/param_{param['extra']}_{i_restart:05d}.txt"
            restart = iostream.read_param_file(param_filename)
            logging.warning(f"Parameter file read at ...{param_filename=}")
            for key in restart.index:  # (the reference's `is not "nthreads"` test is always true: every key is taken)
                param[key] = restart[key]
        elif fmt == "hdf5":
            filename = f"{directory}/particles_{param['extra']}.h5"
            position, velocity = iostream.read_snapshot_particles_hdf5(filename)
            for key, value in iostream.read_snapshot_attributes_hdf5(filename).items():
                param[key] = value
        else:
            raise ValueError(f"{param['output_snapshot_format']=}, should be 'parquet' or 'hdf5'")
        return position, velocity
    if isinstance(ic, str) and "lpt" in ic.casefold():
        from . import lpt

        return lpt.generate(param, tables)
    raise NotImplementedError(f"{param['initial_conditions']=}: this library starts from a snapshot number or nLPT")
