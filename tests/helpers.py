"""Shared test helpers: synthetic inputs reproducible from seeds alone (no reference needed)."""
import json

import numpy as np
from scipy.interpolate import interp1d


def synth_tables():
    """Same tables as oracle/gen_golden.py:synth_tables (tables[0]=lna(t), [1]=t(lna), [3]=D1(lna))."""
    t = np.linspace(-0.95, 0.0, 4000)
    a = (1.0 + t) ** (2.0 / 3.0)
    lna = np.log(a)
    t0 = interp1d(t, lna, fill_value="extrapolate")
    t1 = interp1d(lna, t, fill_value="extrapolate")
    d1 = interp1d(lna, a, fill_value="extrapolate")
    return [t0, t1, None, d1] + [None] * 9


def param_from_json(s) -> dict:
    return json.loads(str(s))


def zeldovich_particles(n: int, seed: int = 42, rms_cells: float = 0.5, vel_factor: float = 0.02):
    """BASELINE.md section 4 workload: cell-centred lattice displaced by a Zel'dovich field from a
    k^-2 Gaussian spectrum truncated at k_Nyq/2, rms |psi| = rms_cells cells; v = vel_factor*psi.
    Returns (pos[Np,3] f4 wrapped to [0,1), vel[Np,3] f4)."""
    rng = np.random.default_rng(seed)
    kf = np.fft.fftfreq(n, d=1.0 / n)
    kz = np.fft.rfftfreq(n, d=1.0 / n)
    KX, KY, KZ = np.meshgrid(kf, kf, kz, indexing="ij")
    k2 = KX**2 + KY**2 + KZ**2
    k2[0, 0, 0] = 1.0
    amp = np.where(np.sqrt(k2) <= n / 4, 1.0 / k2, 0.0)  # sqrt(P) ~ k^-1 -> delta_k ~ k^-1; psi_k = i k/k^2 delta_k
    amp[0, 0, 0] = 0.0
    delta_k = (rng.normal(size=k2.shape) + 1j * rng.normal(size=k2.shape)) * np.sqrt(amp)
    psi = np.empty((n, n, n, 3), dtype=np.float64)
    for ax, K in enumerate((KX, KY, KZ)):
        psi[..., ax] = np.fft.irfftn(1j * K / k2 * delta_k, s=(n, n, n), axes=(0, 1, 2))
    psi = psi.reshape(-1, 3)
    psi *= (rms_cells / n) / np.sqrt((psi**2).sum(axis=1).mean())
    g = (np.arange(n, dtype=np.float64) + 0.5) / n
    x0 = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    pos = np.mod(x0 + psi, 1.0).astype(np.float32)
    pos[pos >= 1.0] = 0.0
    vel = (vel_factor * psi).astype(np.float32)
    return np.ascontiguousarray(pos), np.ascontiguousarray(vel)
