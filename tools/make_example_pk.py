#!/usr/bin/env python3
"""Writes examples/pk_example.dat: the linear matter power spectrum table [k in h/Mpc, P in (Mpc/h)^3 at z = 0] that
examples/param.ini points to.  An analytic CDM-like shape (turn-over at k ~ 0.02 h/Mpc, slope -2.7 beyond), enough for
the example run; a production run brings its own table (CAMB / CLASS output in the same two-column format)."""
import os

import numpy as np

k = np.logspace(-4, 2, 2000)
q = k / 0.02
pk = 1.2e5 * q / (1.0 + 1.5 * q + q * q) ** 1.85
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "pk_example.dat")
np.savetxt(out, np.c_[k, pk], fmt="%.6e")
print("wrote", out)
