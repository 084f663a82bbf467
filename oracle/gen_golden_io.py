#!/usr/bin/env python3
"""Golden vectors for the parameter-file reader and the snapshot side files (iostream.py:13-69, 157-165) by
EXECUTING THE REFERENCE SOURCE (same harness as gen_golden.py).  Build container only; commits
tests/golden/param_io.json (inputs written by this script + the reference's outputs; no reference text).

    python oracle/gen_golden_io.py"""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "refshim"))
sys.path.insert(1, "/root/reference/pysco")

import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402

import iostream  # noqa: E402  (the unmodified reference module)

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "param_io.json")

SYNTHETIC = """# a parameter file written for the parity test: every syntax case the reader has
nthreads = 4   # integer with a comment
theory= newton # no space before '='
fR_logfR0 = 5.5
boolean_upper = True
boolean_lower = false # lower case
boolean_mixed = TRUE

npart = 64**3 # an expression
z_out = [10, 5, 0.5, 0]  # a list stays a string
base = some/dir_00000/ # a path
empty_value =
linear_newton_solver = fft_7pt
epsrel = 1e-2
negative = -3
name with spaces = hello world
"""


def typed(series):
    return {str(k): [type(v).__name__, v if isinstance(v, (bool, int, float, str)) else repr(v)] for k, v in series.items()}


out = {"synthetic_text": SYNTHETIC}
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "p.ini")
    open(path, "w").write(SYNTHETIC)
    out["synthetic_parsed"] = typed(iostream.read_param_file(path))
    out["example_parsed"] = typed(iostream.read_param_file("/root/reference/examples/param.ini"))
    # the parameter side file of a Parquet snapshot (iostream.py:157-165) and what the reader makes of it again
    p = pd.Series({"theory": "newton", "aexp": 0.02, "nsteps": 12, "t": np.float64(-7.25), "boxlen": 100, "npart": 32768,
                   "z_out": "[10, 0]", "write_snapshot": True, "base": tmp, "extra": "newton_fft_ncoarse5",
                   "unit_t": 4.3e17, "i_snap": 2, "output_snapshot_format": "parquet", "Om_m": np.float32(0.25)})
    out["side_file_param"] = typed(p.drop("base"))
    os.makedirs(os.path.join(tmp, "output_00002"))
    pos = np.arange(24, dtype=np.float32).reshape(8, 3) / 24
    iostream.write_snapshot_particles(pos, -pos, p)
    side = os.path.join(tmp, "output_00002", f"param_{p['extra']}_00002.txt")
    out["side_file_text"] = open(side).read().replace(tmp, "BASE")
    back = iostream.read_param_file(side)
    out["side_file_parsed"] = typed(back.drop("base"))
    import pyarrow.parquet as pq

    t = pq.read_table(os.path.join(tmp, "output_00002", f"particles_{p['extra']}.parquet"))
    out["parquet_columns"] = t.column_names
    out["parquet_types"] = [str(x) for x in t.schema.types]
json.dump(out, open(OUT, "w"), indent=1)
print("wrote", OUT, os.path.getsize(OUT), "bytes")
print(json.dumps(out["synthetic_parsed"], indent=0)[:1500])
