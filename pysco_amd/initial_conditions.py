"""Initial conditions: drop-in for the reference's initial_conditions.generate() (initial_conditions.py:25-213).

Row f2 of SURVEY.md section 8: restart from a snapshot this library (or PySCo) wrote -- `initial_conditions` an
integer -- in either format, with the snapshot's parameters taking over the run's (initial_conditions.py:79-107).
Row f3 (Lagrangian perturbation theory from a power spectrum) is served by pysco_amd.lpt when present.
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
        try:
            from . import lpt
        except ImportError as exc:  # pragma: no cover
            raise NotImplementedError("LPT initial conditions (SURVEY.md section 8 row f3) are not built yet; start "
                                      "from a snapshot (initial_conditions = <snapshot number>)") from exc
        return lpt.generate(param, tables)
    raise NotImplementedError(f"{param['initial_conditions']=}: this library starts from a snapshot number or nLPT")
