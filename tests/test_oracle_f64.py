"""The fp64 build of the CPU oracle (oracle/pysco_oracle_f64.py: pm_oracle.c with REAL = double) against the pinned
fp32 one: same formulas, so the two agree to fp32 rounding -- on a PM force, a leapfrog step and P(k)."""
import numpy as np
import pytest

from oracle import pysco_oracle as o32
from oracle import pysco_oracle_f64 as o64
from tests.helpers import synth_tables, zeldovich_particles


def _param(n, scheme="TSC", solver="fft", order=5):
    p = {"theory": "newton", "mass_scheme": scheme, "npart": n**3, "ncoarse": int(np.log2(n)), "Om_m": 0.25733,
         "Om_lambda": 0.742589237, "H0": 72.0, "boxlen": 100.0, "aexp": 0.5, "aexp_old": 0.49,
         "linear_newton_solver": solver, "gradient_stencil_order": order, "save_power_spectrum": "no", "nthreads": 1,
         "Courant_factor": 1.0, "max_aexp_stepping": 10, "integrator": "leapfrog", "write_snapshot": False, "nsteps": 0}
    tables = synth_tables()
    p["t"] = float(tables[1](np.log(p["aexp"])))
    o32.set_units(p)
    return p, tables


@pytest.mark.parametrize("scheme,solver,order", [("TSC", "fft", 5), ("CIC", "fft_7pt", 3), ("TSC", "fft", 7), ("CIC", "fft", 2)])
def test_f64_oracle_is_the_f32_oracle_in_double(scheme, solver, order):
    n = 32
    pos, vel = zeldovich_particles(n, seed=4, rms_cells=0.7, vel_factor=0.3)
    par32, tables = _param(n, scheme, solver, order)
    par64 = dict(par32)
    a32, phi32, _ = o32.pm(pos.copy(), par32, tables=tables)
    a64, phi64 = o64.pm(pos.astype(np.float64), par64)
    s = np.abs(a64).max()
    assert np.abs(a32 - a64).max() <= 3e-5 * s
    assert np.abs(phi32 - phi64).max() <= 3e-5 * np.abs(phi64).max()
    # three leapfrog steps
    p32, v32 = pos.copy(), vel.copy()
    p64, v64 = pos.astype(np.float64), vel.astype(np.float64)
    phi, ex = phi32, np.empty(0, np.float32)
    for _ in range(3):
        p32, v32, a32, phi, ex = o32.integrate(p32, v32, a32, phi, ex, tables, par32, np.float32(3e38))
        p64, v64, a64, _, _ = o64.integrate(p64, v64, a64, tables, par64, 3e38)
    assert abs(par32["t"] - par64["t"]) < 1e-5
    d = np.abs(p32 - p64)
    assert np.minimum(d, 1 - d).max() <= 1e-5
    assert np.abs(v32 - v64).max() <= 1e-4 * np.abs(v64).max()
    k32, pk32, nm32 = o32.fourier_grid_to_Pk(o32.fft_3D_real(o32.TSC(p32, n)), 3)
    k64, pk64, nm64 = o64.pk(p64, n, 3)
    assert np.array_equal(nm32, nm64)
    assert np.abs(pk32 - pk64).max() <= 1e-3 * np.abs(pk64).max()


def test_f64_wrap_and_max():
    x = np.array([-0.2, 1.3, 0.8, -1e-9, 1.0, 0.0, -3e-8], dtype=np.float64)
    o64.periodic_wrap(x)
    assert np.array_equal(x, np.array([0.8, 1.3 - 1.0, 0.8, 0.0, 0.0, 0.0, 1.0 - 3e-8]))
    assert o64.max_abs(np.array([[1.0, -7.5], [3.0, 2.0]])) == 7.5
