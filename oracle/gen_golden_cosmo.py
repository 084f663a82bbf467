#!/opt/conda/bin/python3.9
"""Golden vectors for the background cosmology table (cosmotable.py:18-110, 113-242) by EXECUTING THE REFERENCE
SOURCE with the genuine astropy (4.3.1, the copy in this container's conda environment; the main interpreter has
none).  Build container only; commits tests/golden/cosmotable_*.npz (numbers only).

    /opt/conda/bin/python3.9 -W ignore oracle/gen_golden_cosmo.py

astropy 4.3.1 predates NumPy 1.24 and names a few functions NumPy has since removed in its dispatch tables (never
called here); they are given placeholders so that the package imports.  Nothing of the reference is modified."""
import os
import sys
import tempfile

import numpy as np

for _nm in ("asscalar", "alen", "msort", "product", "cumproduct", "sometrue", "alltrue", "rank"):
    if not hasattr(np, _nm):
        setattr(np, _nm, lambda *a, **k: None)
for _nm, _t in (("float", float), ("int", int), ("bool", bool), ("object", object), ("complex", complex), ("str", str)):
    if not hasattr(np, _nm):
        setattr(np, _nm, _t)

import pandas as pd  # noqa: E402

sys.path.insert(0, "/root/reference/pysco")
import cosmotable  # noqa: E402  (the unmodified reference module)

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
CASES = {
    "lcdm": dict(theory="newton", H0=72.0, Om_m=0.25733, T_cmb=2.726, N_eff=3.044, w0=-1.0, wa=0.0),
    "w0wa": dict(theory="newton", H0=67.0, Om_m=0.31, T_cmb=2.7255, N_eff=3.044, w0=-0.9, wa=0.2),
    "parametrized": dict(theory="parametrized", H0=70.0, Om_m=0.3, T_cmb=2.726, N_eff=3.044, w0=-1.0, wa=0.0,
                         parametrized_mu0=-0.1),
    "norad": dict(theory="newton", H0=70.0, Om_m=0.3, T_cmb=0.0, N_eff=3.044, w0=-1.0, wa=0.0),
}
NAMES = ["lna_of_t", "t_of_lna", "H", "d1", "f1", "d2", "f2", "d3a", "f3a", "d3b", "f3b", "d3c", "f3c"]

for tag, c in CASES.items():
    with tempfile.TemporaryDirectory() as tmp:
        param = pd.Series(dict(c, base=tmp, extra="golden"))
        tables = cosmotable.generate(param)
        txt = np.loadtxt(os.path.join(tmp, "evolution_table_pysco.txt"))
    out = {"Om_r": np.float64(param["Om_r"]), "Om_lambda": np.float64(param["Om_lambda"]),
           "param_json": np.array(repr({k: v for k, v in c.items()}))}
    # every interpolator sampled at fixed abscissae inside its range, plus its knots thinned out
    lna = np.linspace(np.log(1.0 / 201), 0.0, 2001)
    out["lna"] = lna
    for k, name in enumerate(NAMES):
        if k == 0:
            t = tables[1](lna)
            out["t"] = t
            out[name] = tables[0](t)
        else:
            out[name] = tables[k](lna)
    out["table_rows"] = txt[::250]  # aexp, H/H0, t_supercomoving, dplus1, f1, ... as the reference writes them
    path = os.path.join(OUT, f"cosmotable_{tag}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)  Om_r={out['Om_r']:.6e} Om_lambda={out['Om_lambda']:.9f}")
