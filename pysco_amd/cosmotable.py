"""Background cosmology tables: drop-in for the reference's cosmotable.generate() (cosmotable.py:18-110) without
astropy.  Host-side, run once per simulation; the 13 interpolators it returns are what integration.integrate
(tables[0], [1]), solver.initialise_potential (tables[3]) and the initial conditions (tables[2..12]) consume.

The reference delegates the background to astropy.cosmology.Flatw0waCDM (un-pinned third-party dependency; 4.3.1
in this image's conda environment, which is what the golden vectors of tests/golden/cosmotable_*.npz were made
with, oracle/gen_golden_cosmo.py).  Restated here from its published definitions:

    Ogamma0 = (4 sigma_SB / c^3) Tcmb0^4 / rho_crit0,      rho_crit0 = 3 H0^2 / (8 pi G)     (CODATA 2018)
    Onu0    = 7/8 (4/11)^(4/3) Neff Ogamma0                (massless neutrinos, astropy's default m_nu = 0)
    Ode0    = 1 - Om0 - Ogamma0 - Onu0                     (flat)
    E(z)^2  = Om0 (1+z)^3 + (Ogamma0 + Onu0) (1+z)^4 + Ode0 (1+z)^(3 (1 + w0 + wa)) exp(-3 wa z / (1+z))

and the growth factors D1, D2, D3a, D3b, D3c with their logarithmic rates from the ODE system of
cosmotable.py:113-242 (Rampf & Buchert 2012 initial conditions at a = 1e-8), integrated with the same
scipy.integrate.solve_ivp call (RK45, rtol = atol = 1e-13, 100000 output points).
"""
from __future__ import annotations

import logging
from typing import List

import numpy as np
from scipy.integrate import cumulative_trapezoid, solve_ivp
from scipy.interpolate import interp1d

# CODATA 2018 (astropy.constants of astropy >= 4.0)
_SIGMA_SB = 5.6703744191844314e-08  # W m^-2 K^-4
_C = 299792458.0                    # m / s
_G = 6.6743e-11                     # m^3 kg^-1 s^-2
_PC = 3.0856775814913674e16         # m
_NU_PER_PHOTON = 0.22710731766      # 7/8 (4/11)^(4/3), as astropy rounds it


class FlatW0WaCDM:
    """The subset of astropy.cosmology.Flatw0waCDM the reference uses (cosmotable.py:48-58, 276-283)."""

    def __init__(self, H0, Om0, Tcmb0, Neff, w0, wa):
        self.H0, self.Om0, self.Tcmb0, self.Neff, self.w0, self.wa = (float(H0), float(Om0), float(Tcmb0), float(Neff),
                                                                      float(w0), float(wa))
        h0_si = self.H0 * 1e3 / (1e6 * _PC)
        rho_crit0 = 3.0 * h0_si**2 / (8.0 * np.pi * _G)          # kg / m^3
        self.Ogamma0 = 4.0 * _SIGMA_SB / _C**3 * self.Tcmb0**4 / rho_crit0
        self.Onu0 = _NU_PER_PHOTON * self.Neff * self.Ogamma0 if self.Tcmb0 > 0 else 0.0
        self.Ode0 = 1.0 - self.Om0 - self.Ogamma0 - self.Onu0

    def _de_scale(self, z):
        zp1 = 1.0 + np.asarray(z, dtype=np.float64)
        return zp1 ** (3.0 * (1.0 + self.w0 + self.wa)) * np.exp(-3.0 * self.wa * (zp1 - 1.0) / zp1)

    def efunc(self, z):
        zp1 = 1.0 + np.asarray(z, dtype=np.float64)
        return np.sqrt(zp1**3 * ((self.Ogamma0 + self.Onu0) * zp1 + self.Om0) + self.Ode0 * self._de_scale(z))

    def Om(self, z):
        return self.Om0 * (1.0 + np.asarray(z, dtype=np.float64)) ** 3 / self.efunc(z) ** 2

    def Ogamma(self, z):
        return self.Ogamma0 * (1.0 + np.asarray(z, dtype=np.float64)) ** 4 / self.efunc(z) ** 2

    def Onu(self, z):
        return self.Onu0 * (1.0 + np.asarray(z, dtype=np.float64)) ** 4 / self.efunc(z) ** 2

    def Ode(self, z):
        return self.Ode0 * self._de_scale(z) / self.efunc(z) ** 2


def _cosmology(param) -> FlatW0WaCDM:
    return FlatW0WaCDM(param["H0"], param["Om_m"], param["T_cmb"], param["N_eff"], param["w0"], param["wa"])


def generate(param) -> List[interp1d]:
    """[lna(t), t(lna), H(lna), D1, f1, D2, f2, D3a, f3a, D3b, f3b, D3c, f3c] as interp1d objects with
    fill_value="extrapolate"; sets param["Om_r"], param["Om_lambda"] and writes
    {base}/evolution_table_pysco.txt exactly like cosmotable.py:47-110."""
    cosmo = _cosmology(param)
    param["Om_r"] = cosmo.Ogamma0 + cosmo.Onu0
    param["Om_lambda"] = cosmo.Ode0

    a_start = 1.0 / (1 + 200)
    lna = np.linspace(np.log(a_start), 0, 100_000)
    a = np.exp(lna)
    dlna = lna[1] - lna[0]
    E_array = cosmo.efunc(1.0 / a - 1)
    t_supercomoving = cumulative_trapezoid(dlna / (a**2 * E_array), initial=0)  # dt = dlna / (a^2 E)
    t_supercomoving -= t_supercomoving[-1]
    growth = compute_growth_functions(cosmo, param)
    growth = growth[:, growth[0] > lna[0]]
    lna_g = growth[0]

    logging.warning(f"Write table in: {param['base']}/evolution_table_pysco_{param['extra']}.txt")
    np.savetxt(f"{param['base']}/evolution_table_pysco.txt",
               np.c_[(a, E_array, t_supercomoving) + tuple(np.interp(lna, lna_g, g) for g in growth[1:])],
               header="aexp, H/H0, t_supercomoving, dplus1, f1, dplus2, f2, dplus3a, f3a, dplus3b, f3b, dplus3c, f3c")
    return [interp1d(t_supercomoving, lna, fill_value="extrapolate"),
            interp1d(lna, t_supercomoving, fill_value="extrapolate"),
            interp1d(lna, param["H0"] * E_array, fill_value="extrapolate")] + [
                interp1d(lna_g, g, fill_value="extrapolate") for g in growth[1:]]


def compute_growth_functions(cosmo: FlatW0WaCDM, param) -> np.ndarray:
    """Rows: lna, D1, f1, D2, f2, D3a, f3a, D3b, f3b, D3c, f3c on 100000 points of ln a in [ln 1e-8, 0]
    (cosmotable.py:113-242; `parametrized` theory: mu(a) = 1 + mu0 Ode(a) / Ode0 multiplies the source, :293-373)."""
    a_eq = (cosmo.Ogamma0 + cosmo.Onu0) / cosmo.Om0
    if (cosmo.Ogamma0 + cosmo.Onu0) == 0:
        a_eq = 2e-7
    d1 = 3.0 / 5 * a_eq  # matter-dominated start (Rampf & Buchert 2012)
    y0 = [d1, 0, -3.0 / 7 * d1**2, 0, -1.0 / 3.0 * d1**3, 0, 10.0 / 21.0 * d1**3, 0, -1.0 / 7.0 * d1**3, 0]
    span = (np.log(1e-8), np.log(1.0))
    mu0 = param["parametrized_mu0"] if param["theory"].casefold() == "parametrized" else None
    sol = solve_ivp(_growth_rhs, span, y0, t_eval=np.linspace(span[0], span[1], 100_000), rtol=1e-13, atol=1e-13,
                    args=(cosmo, mu0))
    D, dD = sol.y[0::2], sol.y[1::2]
    out = np.empty((11, sol.t.size))
    out[0] = sol.t
    out[1::2] = D
    out[2::2] = dD / D
    return out


def _growth_rhs(lna, y, cosmo: FlatW0WaCDM, mu0):
    """d/dlna of (D1, D1', D2, D2', D3a, D3a', D3b, D3b', D3c, D3c'):  D'' + gamma D' = beta * source, with
    beta = 3/2 mu Om(a), gamma = 1/2 (1 - 3 w(a) Ode(a) - Or(a)); the transverse third-order term D3c has its own
    form (cosmotable.py:276-311)."""
    a = np.exp(lna)
    z = 1.0 / a - 1
    Ode_z = cosmo.Ode(z)
    mu = 1.0 if mu0 is None else 1 + (mu0 * Ode_z / cosmo.Ode0)
    beta = 1.5 * mu * cosmo.Om(z)
    gamma = 0.5 * (1.0 - 3.0 * Ode_z * (cosmo.w0 + cosmo.wa * (1.0 - a)) - (cosmo.Ogamma(z) + cosmo.Onu(z)))
    D1, dD1, D2, dD2, D3a, dD3a, D3b, dD3b, D3c, dD3c = y
    return np.array([
        dD1, -gamma * dD1 + beta * D1,
        dD2, -gamma * dD2 + beta * (D2 - D1**2),
        dD3a, -gamma * dD3a + beta * (D3a - 2.0 * D1**3),
        dD3b, -gamma * dD3b + beta * (D3b - 2.0 * D1 * (D2 - D1**2)),
        dD3c, (1 - gamma) * dD3c + D2 * dD1 - D1 * dD2 - beta * D1**3,
    ])
