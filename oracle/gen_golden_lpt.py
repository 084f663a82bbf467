#!/usr/bin/env python3
"""Golden vectors for the Lagrangian-perturbation-theory initial conditions (initial_conditions.py:25-213, 402-445,
531-728, 976-1039, 1635-1856; fourier.py:730-905) by EXECUTING THE REFERENCE SOURCE, same harness as gen_golden.py
(Numba decorators as identities, NumPy FFT fallback).  The reference's module needs Python >= 3.11 for one call,
math.cbrt (integer cube root of npart); this interpreter is 3.10, so math.cbrt is provided as numpy.cbrt (same
integers).  Build container only; commits tests/golden/lpt_*.npz (numbers only: the power spectrum table is written
by this script, not the reference's data file).

    python oracle/gen_golden_lpt.py"""
import math
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "refshim"))
sys.path.insert(1, "/root/reference/pysco")
sys.path.insert(2, ROOT)

import numpy as np  # noqa: E402

if not hasattr(math, "cbrt"):
    math.cbrt = lambda x: float(np.cbrt(x))

import pandas as pd  # noqa: E402

import initial_conditions as ref_ic  # noqa: E402  (the unmodified reference modules)
import utils as ref_utils  # noqa: E402

from pysco_amd import cosmotable  # noqa: E402  (astropy-free tables, pinned by tests/test_cosmotable.py)

OUT = os.path.join(ROOT, "tests", "golden")


def power_spectrum_table():
    """A smooth CDM-like spectrum on 600 log-spaced wavenumbers [h/Mpc], (Mpc/h)^3."""
    k = np.logspace(-4, 2, 600)
    q = k / 0.02
    pk = 8.0e6 * q / (1.0 + 1.5 * q + q * q) ** 1.6
    return k, pk


CASES = {
    "1lpt_center_n16": dict(npart=16**3, initial_conditions="1LPT", position_ICS="center", fixed_ICS=False, paired_ICS=False, seed=42),
    "2lpt_center_n16": dict(npart=16**3, initial_conditions="2LPT", position_ICS="center", fixed_ICS=False, paired_ICS=False, seed=7),
    "2lpt_edge_fixed_paired_n16": dict(npart=16**3, initial_conditions="2LPT", position_ICS="edge", fixed_ICS=True, paired_ICS=True, seed=3),
    "2lpt_center_n32": dict(npart=32**3, initial_conditions="2LPT", position_ICS="center", fixed_ICS=False, paired_ICS=False, seed=11),
    "3lpt_center_n16": dict(npart=16**3, initial_conditions="3LPT", position_ICS="center", fixed_ICS=False, paired_ICS=False, seed=5),
    "3lpt_edge_n32": dict(npart=32**3, initial_conditions="3LPT", position_ICS="edge", fixed_ICS=False, paired_ICS=False, seed=13),
    # Orszag's 3/2 rule for the products (initial_conditions.py:1008, 1034-1038, 1859-1928: pad / trim)
    "2lpt_dealiased_n16": dict(npart=16**3, initial_conditions="2LPT", position_ICS="center", fixed_ICS=False, paired_ICS=False, seed=17,
                               dealiased_ICS=True),
    "3lpt_dealiased_n16": dict(npart=16**3, initial_conditions="3LPT", position_ICS="edge", fixed_ICS=False, paired_ICS=False, seed=19,
                               dealiased_ICS=True),
    "2lpt_dealiased_n32": dict(npart=32**3, initial_conditions="2LPT", position_ICS="center", fixed_ICS=True, paired_ICS=False, seed=23,
                               dealiased_ICS=True),
}
ONLY = [a for a in sys.argv[1:] if a in CASES]  # regenerate a subset: python oracle/gen_golden_lpt.py 3lpt_center_n16 ...

k_tab, pk_tab = power_spectrum_table()
for tag, case in CASES.items():
    if ONLY and tag not in ONLY:
        continue
    with tempfile.TemporaryDirectory() as tmp:
        pk_file = os.path.join(tmp, "pk.dat")
        np.savetxt(pk_file, np.c_[k_tab, pk_tab])
        param = dict(theory="newton", H0=72.0, Om_m=0.25733, T_cmb=2.726, N_eff=3.044, w0=-1.0, wa=0.0, boxlen=100.0,
                     z_start=49.0, dealiased_ICS=False, power_spectrum_file=pk_file, base=tmp, extra="golden",
                     output_snapshot_format="parquet", nthreads=1)
        param.update(case)
        tables = cosmotable.generate(dict(param))  # (the Series below gets Om_r / Om_lambda from its own call)
        p = pd.Series(param, dtype=object)
        cosmotable.generate(p)
        p["aexp"] = 1.0 / (1 + p["z_start"])
        ref_utils.set_units(p)
        os.makedirs(os.path.join(tmp, "output_00000"))
        dk = ref_ic.generate_density_fourier(p)
        with np.errstate(all="ignore"):
            pos, vel = ref_ic.generate(p, tables)
        n = int(round(case["npart"] ** (1 / 3)))
        lna = np.log(1.0 / (1 + p["z_start"]))
        scalars = {"Hz": float(tables[2](lna)), "d1_z0": float(tables[3](0)), "d1": float(tables[3](lna)), "f1": float(tables[4](lna)),
                   "d2": float(tables[5](lna)), "f2": float(tables[6](lna)), "unit_t": float(p["unit_t"]), "aexp": float(p["aexp"]),
                   "d3a": float(tables[7](lna)), "f3a": float(tables[8](lna)), "d3b": float(tables[9](lna)),
                   "f3b": float(tables[10](lna)), "d3c": float(tables[11](lna)), "f3c": float(tables[12](lna))}
        out = {"k_table": k_tab, "pk_table": pk_tab, "density_k_half": np.ascontiguousarray(dk[:, :, : n // 2 + 1]),
               "position": pos.reshape(-1, 3), "velocity": vel.reshape(-1, 3),
               "param_json": np.array(repr({k: v for k, v in case.items()})),
               "scalars_json": np.array(repr(scalars))}
        path = os.path.join(OUT, f"lpt_{tag}.npz")
        np.savez_compressed(path, **out)
        print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB): rms displacement "
              f"{np.sqrt(((pos.reshape(n, n, n, 3) - (np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing='ij'), -1) + (0.5 if case['position_ICS'] == 'center' else 0.0)) / n + 0.5) % 1 - 0.5)** 2).sum(-1).mean() * n:.3f} cells, |v|max {np.abs(vel).max():.3e}")
