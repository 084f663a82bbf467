"""The one piece of the reference's I/O layer the per-step path calls: the ascii P(k) writer
(reference iostream.py:268-304).  Snapshots, param.ini parsing etc. stay with the caller."""
from __future__ import annotations

import logging

import numpy as np


def write_power_spectrum_to_ascii_file(k, Pk, Nmodes, param) -> None:
    output_pk = f"{param['base']}/power/pk_{param['extra']}_{param['nsteps']:05d}.dat"
    logging.warning(f"Write P(k) in {output_pk}")
    np.savetxt(
        f"{output_pk}",
        np.c_[k, Pk, Nmodes],
        header=f"aexp = {param['aexp']}\nboxlen = {param['boxlen']} Mpc/h \nnpart = {param['npart']} \n"
        "k [h/Mpc] P(k) [Mpc/h]^3 Nmodes",
    )
