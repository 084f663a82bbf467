#!/opt/conda/bin/python3.9
"""A small particle snapshot written with the genuine h5py (3.3.0, HDF5 1.10, this container's conda environment)
exactly the way the reference writes one (iostream.py:259-265: two datasets + every parameter as a root attribute),
with the parameter container of a typical run.  Commits tests/golden/snapshot_h5py.h5 (data only).

    /opt/conda/bin/python3.9 -W ignore oracle/gen_golden_h5.py"""
import os

import h5py
import numpy as np

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "snapshot_h5py.h5")
rng = np.random.default_rng(7)
position = rng.random((257, 3), dtype=np.float32)
velocity = rng.normal(size=(257, 3)).astype(np.float32)
param = {"nthreads": 1, "theory": "newton", "fR_logfR0": 5, "fR_n": 1, "mond_function": "simple", "mond_g0": 1.2,
         "mond_scale_factor_exponent": 0, "mond_alpha": 1, "parametrized_mu0": -0.1, "H0": 72, "Om_m": 0.25733,
         "T_cmb": 2.726, "N_eff": 3.044, "w0": -1.0, "wa": 0.0, "boxlen": 100, "ncoarse": 7, "npart": 2097152,
         "z_start": 49, "seed": 42, "position_ICS": "center", "fixed_ICS": False, "paired_ICS": False,
         "dealiased_ICS": False, "power_spectrum_file": "examples/pk_lcdmw7v2.dat", "initial_conditions": "2LPT",
         "base": "examples/boxlen100_n128_lcdmw7v2_00000/", "output_snapshot_format": "HDF5",
         "z_out": "[10, 5, 2, 1, 0.5, 0]", "save_power_spectrum": "yes", "integrator": "leapfrog", "mass_scheme": "TSC",
         "n_reorder": 50, "Courant_factor": 1.0, "max_aexp_stepping": 10, "linear_newton_solver": "multigrid",
         "gradient_stencil_order": 5, "Npre": 2, "Npost": 1, "epsrel": 0.01, "verbose": 1, "write_snapshot": True,
         "extra": "newton_multigrid_ncoarse7", "Om_r": 8.07417e-05, "Om_lambda": 0.742589258, "aexp": np.float64(0.0909),
         "unit_l": 3.9e24, "unit_t": np.float64(3.5e15), "unit_d": 7.1e-27, "mpart": np.float64(2.1e41), "nsteps": 17,
         "t": np.float64(-12.25), "i_snap": 1, "aexp_old": 0.0871, "MAS_index": 3, "save_pk": True,
         "parametrized_mu_z": np.float32(1.0), "compute_additional_field": False, "tolerance": np.float32(1e-3)}
with h5py.File(OUT, "w") as h5f:
    h5f.create_dataset("position", data=position)
    h5f.create_dataset("velocity", data=velocity)
    for key, item in param.items():
        h5f.attrs[key] = item
print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(param), "attributes; h5py", h5py.__version__, h5py.version.hdf5_version)
