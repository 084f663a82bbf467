"""CPU oracle for PySCo's per-step PM path -- TEST INFRASTRUCTURE, NOT PRODUCT.

A restatement (C kernels in pm_oracle.c + numpy for the k-space tables and the FFT) of the
reference's algorithm.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the product (pysco_amd/) never does.

Parity status: PINNED -- every function here is checked in tests/test_oracle_golden.py against
golden vectors captured by oracle/gen_golden.py from the reference source itself (executed in the
build container; the reference has no tests or fixtures of its own).

Citations are relative to /root/reference/pysco/.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
from typing import List, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f32 = np.float32
_fp = C.POINTER(C.c_float)


def build(force: bool = False) -> None:
    """Compile the C restatement (gcc, a second or two)."""
    out = os.path.join(_HERE, "_build", "libpm_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("pm_oracle.c", "pm_oracle_mg.c", "Makefile")]
    if force or not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])


_libs = {}


def _lib(omp: bool = False):
    key = "omp" if omp else "seq"
    if key not in _libs:
        build()
        name = "libpm_oracle_omp.so" if omp else "libpm_oracle.so"
        lib = C.CDLL(os.path.join(_HERE, "_build", name))
        for fn in ("pmo_max_abs", "pmo_lap_residual_error", "pmo_diff_norm", "pmo_cubic_residual_error",
                   "pmo_cubic_solution", "pmo_quartic_solution", "pmo_mond_nu"):
            getattr(lib, fn).restype = C.c_float
        lib.pmo_num_threads.restype = C.c_int
        _libs[key] = lib
    return _libs[key]


_use_omp = False


def use_openmp(flag: bool) -> None:
    """bench.py's cpu_baseline leg switches to the OpenMP build; parity tests use the serial one."""
    global _use_omp
    _use_omp = bool(flag)


def set_num_threads(t: int) -> None:
    """Thread count of the OpenMP build (the OpenMP runtime may have read its environment long ago)."""
    _lib(True).pmo_set_num_threads(C.c_int(int(t)))


def num_threads() -> int:
    return int(_lib(_use_omp).pmo_num_threads())


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _c32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


# ------------------------------------------------------------------ mesh.py


def NGP(position, ncells_1d: int) -> np.ndarray:  # mesh.py:2240
    pos = _c32(position)
    rho = np.empty((ncells_1d,) * 3, dtype=np.float32)
    _lib(_use_omp).pmo_ngp(_p(pos), C.c_int64(pos.shape[0]), C.c_int(ncells_1d), _p(rho))
    return rho


def CIC(position, ncells_1d: int) -> np.ndarray:  # mesh.py:2284
    pos = _c32(position)
    rho = np.empty((ncells_1d,) * 3, dtype=np.float32)
    _lib(_use_omp).pmo_cic(_p(pos), C.c_int64(pos.shape[0]), C.c_int(ncells_1d), _p(rho))
    return rho


def TSC(position, ncells_1d: int) -> np.ndarray:  # mesh.py:2363 / 2468
    pos = _c32(position)
    rho = np.empty((ncells_1d,) * 3, dtype=np.float32)
    _lib(_use_omp).pmo_tsc(_p(pos), C.c_int64(pos.shape[0]), C.c_int(ncells_1d), _p(rho))
    return rho


TSC_seq = TSC


def invNGP_vec(grid, position) -> np.ndarray:  # mesh.py:2639
    g, pos = _c32(grid), _c32(position)
    out = np.empty_like(pos)
    _lib(_use_omp).pmo_inv_ngp_vec(_p(g), C.c_int(g.shape[0]), _p(pos), C.c_int64(pos.shape[0]), _p(out))
    return out


def invCIC_vec(grid, position) -> np.ndarray:  # mesh.py:2752
    g, pos = _c32(grid), _c32(position)
    out = np.empty_like(pos)
    _lib(_use_omp).pmo_inv_cic_vec(_p(g), C.c_int(g.shape[0]), _p(pos), C.c_int64(pos.shape[0]), _p(out))
    return out


def invTSC_vec(grid, position) -> np.ndarray:  # mesh.py:2960
    g, pos = _c32(grid), _c32(position)
    out = np.empty_like(pos)
    _lib(_use_omp).pmo_inv_tsc(_p(g), C.c_int(g.shape[0]), C.c_int(3), _p(pos), C.c_int64(pos.shape[0]), _p(out))
    return out


def invTSC(grid, position) -> np.ndarray:  # mesh.py:2827
    g, pos = _c32(grid), _c32(position)
    out = np.empty(pos.shape[0], dtype=np.float32)
    _lib(_use_omp).pmo_inv_tsc(_p(g), C.c_int(g.shape[0]), C.c_int(1), _p(pos), C.c_int64(pos.shape[0]), _p(out))
    return out


def derivative(a, order: int) -> np.ndarray:  # mesh.py:2072 -> 639/681/725/782
    if order not in (2, 3, 5, 7):
        raise ValueError(f"gradient_order={order}, should be 2, 3, 5 or 7")
    a = _c32(a)
    n = a.shape[0]
    out = np.empty((n, n, n, 3), dtype=np.float32)
    _lib(_use_omp).pmo_derivative(_p(a), None, C.c_float(0), C.c_int(0), C.c_int(n), C.c_int(order), _p(out))
    return out


def derivative_fR(a, b, f, fR_n: int, order: int) -> np.ndarray:  # mesh.py:2112
    if order not in (2, 3, 5, 7):
        raise ValueError(f"gradient_order={order}, should be 2, 3, 5 or 7")
    if fR_n not in (1, 2):
        raise NotImplementedError(f"Only f(R) with n = 1 and 2, currently fR_n={fR_n}")
    a, b = _c32(a), _c32(b)
    n = a.shape[0]
    out = np.empty((n, n, n, 3), dtype=np.float32)
    _lib(_use_omp).pmo_derivative(_p(a), _p(b), C.c_float(f), C.c_int(fR_n), C.c_int(n), C.c_int(order), _p(out))
    return out


# ------------------------------------------------------------------ utils.py


def add_vector_scalar_inplace(y, x, a) -> None:  # utils.py:264
    assert y.dtype == np.float32 and y.flags.c_contiguous
    x = _c32(x)
    _lib(_use_omp).pmo_axpy(_p(y), _p(x), C.c_float(a), C.c_int64(y.size))


def prod_vector_scalar_inplace(y, a) -> None:  # utils.py:410
    assert y.dtype == np.float32 and y.flags.c_contiguous
    _lib(_use_omp).pmo_scale(_p(y), C.c_float(a), C.c_int64(y.size))


def linear_operator(x, f1, f2) -> np.ndarray:  # utils.py:644
    x = _c32(x)
    out = np.empty_like(x)
    _lib(_use_omp).pmo_linear_operator(_p(x), C.c_float(f1), C.c_float(f2), _p(out), C.c_int64(x.size))
    return out


def linear_operator_inplace(x, f1, f2) -> None:  # utils.py:684
    assert x.dtype == np.float32 and x.flags.c_contiguous
    _lib(_use_omp).pmo_linear_operator(_p(x), C.c_float(f1), C.c_float(f2), _p(x), C.c_int64(x.size))


def linear_operator_vectors_inplace(x, f1, y, f2) -> None:  # utils.py:721
    assert x.dtype == np.float32 and x.flags.c_contiguous
    y = _c32(y)
    _lib(_use_omp).pmo_linear_operator_vectors(_p(x), C.c_float(f1), _p(y), C.c_float(f2), C.c_int64(x.size))


def max_abs(x) -> np.float32:  # utils.py:220
    x = _c32(x)
    return np.float32(_lib(_use_omp).pmo_max_abs(_p(x), C.c_int64(x.size)))


def periodic_wrap(position) -> None:  # utils.py:1120
    assert position.dtype == np.float32 and position.flags.c_contiguous
    _lib(_use_omp).pmo_periodic_wrap(_p(position), C.c_int64(position.size))


def positions_to_keys(position) -> np.ndarray:  # morton.py:113
    pos = _c32(position)
    keys = np.empty(pos.shape[0], dtype=np.int64)
    _lib(_use_omp).pmo_morton_keys(_p(pos), C.c_int64(pos.shape[0]), _p(keys))
    return keys


def reorder_particles(position, velocity=None, acceleration=None):  # utils.py:1019 (nthreads == 1)
    arg = np.argsort(positions_to_keys(position), kind="stable")
    if acceleration is not None:
        return position[arg], velocity[arg], acceleration[arg]
    if velocity is not None:
        return position[arg], velocity[arg]
    return position[arg]


def set_units(param) -> None:  # utils.py:167-192 (G, pc: CODATA 2018 as astropy.constants)
    G, pc = 6.6743e-11, 3.0856775814913674e16
    mpc_to_km = 1e3 * pc
    g = G * 1e-9
    H0 = param["H0"] / mpc_to_km
    rhoc = 3.0 * H0**2 / (8.0 * np.pi * g)
    param["unit_l"] = param["aexp"] * param["boxlen"] * 100.0 / H0
    param["unit_t"] = param["aexp"] ** 2 / H0
    param["unit_d"] = param["Om_m"] * rhoc / param["aexp"] ** 3
    param["mpart"] = param["unit_d"] * param["unit_l"] ** 3 / param["npart"]


# ------------------------------------------------------------------ fourier.py


def fft_3D_real(x, threads: int = 1) -> np.ndarray:  # fourier.py:104 (NumPy fallback :129-130)
    if _use_omp:  # cpu_baseline timing: the multi-threaded stand-in for the reference's FFTW threads (:133-147)
        import scipy.fft

        return np.ascontiguousarray(scipy.fft.rfftn(x.astype(np.float64), workers=num_threads()).astype(np.complex64))
    return np.ascontiguousarray(np.fft.rfftn(x).astype(np.complex64))


def ifft_3D_real(x, threads: int = 1) -> np.ndarray:  # fourier.py:251 (NumPy fallback :276-277)
    if _use_omp:
        import scipy.fft

        return np.ascontiguousarray(scipy.fft.irfftn(x.astype(np.complex128), s=(x.shape[0],) * 3,
                                                     workers=num_threads()).astype(np.float32))
    return np.ascontiguousarray(np.fft.irfftn(x).astype(np.float32))


def _kfreq(n: int) -> np.ndarray:
    """kx = i - n if i >= n//2 else i, as fp32 (fourier.py:523-527)."""
    i = np.arange(n)
    return np.where(i >= n // 2, i - n, i).astype(np.float32)


def _apply_greens(x: np.ndarray, kind: int, p: int, k2: np.ndarray, s: np.ndarray, c0: float) -> None:
    assert x.dtype == np.complex64 and x.flags.c_contiguous
    n = x.shape[0]
    k2, s = _c32(k2), _c32(s)
    _lib(_use_omp).pmo_greens(_p(x), C.c_int(n), C.c_int(kind), C.c_int(p), _p(k2), _p(s), C.c_double(c0))


def inverse_laplacian(x) -> None:  # fourier.py:460-491
    n = x.shape[0]
    i = np.arange(n)
    k = np.where(i > n // 2, n - i, i).astype(np.float32)  # note: '>' here, '>=' in the others
    c0 = float(np.float32(-0.25 / np.pi**2))
    _apply_greens(x, 0, 0, k**2, np.ones(n, np.float32), c0)


def inverse_laplacian_compensated(x, p: int) -> None:  # fourier.py:502-544
    n = x.shape[0]
    k = _kfreq(n)
    h = np.float32(1.0 / n)
    c0 = float(np.float32(-0.25 / np.pi**2))
    _apply_greens(x, 1, int(p), k**2, np.sinc(k * h).astype(np.float32), c0)


def inverse_laplacian_7pt(x) -> None:  # fourier.py:555-595
    n = x.shape[0]
    k = _kfreq(n)
    pi_h = np.float32(np.pi / n)
    h = np.float32(1.0 / n)
    minus_h2_inv4 = float(-(np.float32(0.25) * h**2))
    # Numba types math.sin(f4) as f4: the table, its sums and the quotient are all fp32.
    f = (np.sin(pi_h * k) ** 2).astype(np.float32)
    _apply_greens(x, 2, 0, np.zeros(n, np.float32), f, minus_h2_inv4)


def _grad_inverse_laplacian(x, p: int) -> np.ndarray:
    """-i k_j / (2 pi k^2) [W(k)^(-2p)] x, fp32 arithmetic in the reference's order (fourier.py:606-719); p = 0: no
    compensation.  (0 - i c)(a + i b) = c b - i c a exactly, then each component times k_j."""
    assert x.dtype == np.complex64
    n = x.shape[0]
    nh = n // 2 + 1
    k = _kfreq(n)
    kz = np.arange(nh, dtype=np.float32)
    invtwopi = np.float32(0.5 / np.pi)
    with np.errstate(all="ignore"):
        k2 = ((k**2)[:, None] + (k**2)[None, :])[:, :, None] + (kz**2)[None, None, :]
        c = invtwopi / k2
        if p:
            h = np.float32(1.0 / n)
            w = (np.sinc(k * h)[:, None] * np.sinc(k * h)[None, :])[:, :, None] * np.sinc(kz * h)[None, None, :]
            c = (w.astype(np.float32) ** (-2 * p)).astype(np.float32) * c
        re, im = c * x.imag, -(c * x.real)
        out = np.empty((n, n, nh, 3), dtype=np.complex64)
        for m, km in enumerate((k[:, None, None], k[None, :, None], kz[None, None, :])):
            out[..., m] = (re * km) + 1j * (im * km)
    out[0, 0, 0, :] = 0
    return out


def gradient_inverse_laplacian(x) -> np.ndarray:  # fourier.py:606-657
    return _grad_inverse_laplacian(x, 0)


def gradient_inverse_laplacian_compensated(x, p: int) -> np.ndarray:  # fourier.py:660-719
    return _grad_inverse_laplacian(x, int(p))


def ifft_3D_real_grad(x, threads: int = 1) -> np.ndarray:  # fourier.py:343-396 (NumPy fallback :370-371)
    return np.ascontiguousarray(np.fft.irfftn(x, axes=(0, 1, 2)).astype(np.float32))


def fft_force(rhs, param, pk_out: list | None = None) -> np.ndarray:  # solver.py:524-566
    MAS_index = param["MAS_index"]
    rhs_fourier = fft_3D_real(rhs)
    force = gradient_inverse_laplacian(rhs_fourier) if MAS_index == 0 else gradient_inverse_laplacian_compensated(
        rhs_fourier, MAS_index)
    if param.get("save_pk", False):
        k, Pk, Nmodes = fourier_grid_to_Pk(rhs_fourier, MAS_index)
        Pk = Pk * np.float32((param["boxlen"] / len(rhs) ** 2) ** 3 / (1.5 * param["aexp"] * param["Om_m"]) ** 2
                             / param["parametrized_mu_z"] ** 2)
        k = k * np.float32(2 * np.pi / param["boxlen"])
        if pk_out is not None:
            pk_out.append((k, Pk, Nmodes))
    return ifft_3D_real_grad(force)


def fourier_grid_to_Pk(density_k, p: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:  # fourier.py:22-100
    assert density_k.dtype == np.complex64 and density_k.flags.c_contiguous
    n = density_k.shape[0]
    k = _kfreq(n)
    prefactor = np.float32(1.0 / n)
    sincs = np.sinc(k * prefactor).astype(np.float32)
    ks = np.zeros(n, np.float32)
    ps = np.zeros(n, np.float32)
    nm = np.zeros(n, np.float32)
    _lib(False).pmo_pk(_p(density_k), C.c_int(n), C.c_int(int(p)), _p(sincs), _p(ks), _p(ps), _p(nm))
    kmax = int(2 * (n // 2) / 3)
    with np.errstate(invalid="ignore", divide="ignore"):
        return ks[1:kmax] / nm[1:kmax], ps[1:kmax] / nm[1:kmax], nm[1:kmax]


def pk_fp64(position, ncells_1d: int, p: int = 3):
    """The P(k) pipeline -- TSC deposit (mesh.py:2468), r2c, binned and window-deconvolved spectrum
    (fourier.py:22-100) -- evaluated in float64 from the same fp32 positions: the "truth" the fp32 paths are held
    against at benchmark sizes (<= 1e-4 per bin).  Same formulas, same truncations to cell
    indices, every product and sum in fp64."""
    n = int(ncells_1d)
    x = np.float32(np.asarray(position, dtype=np.float32) * np.float32(n)).astype(np.float64)  # as the kernels form it
    i = np.minimum(x.astype(np.int64), n - 1)
    d = x - 0.5 - i
    w = np.stack([0.5 * (0.5 - d) ** 2, 0.75 - d * d, 0.5 * (0.5 + d) ** 2], axis=0)  # [3, Np, 3]: cells i-1, i, i+1
    rho = np.zeros(n**3, dtype=np.float64)
    for a in range(3):
        ia = (i[:, 0] + a - 1) % n
        for b in range(3):
            ib = (i[:, 1] + b - 1) % n
            wab = w[a, :, 0] * w[b, :, 1]
            for c in range(3):
                ic = (i[:, 2] + c - 1) % n
                rho += np.bincount((ia * n + ib) * n + ic, weights=wab * w[c, :, 2], minlength=n**3)
    import scipy.fft

    dk = scipy.fft.rfftn(rho.reshape(n, n, n), workers=num_threads())
    del rho
    kf = np.where(np.arange(n) >= n // 2, np.arange(n) - n, np.arange(n)).astype(np.float64)
    kz = np.arange(n // 2 + 1, dtype=np.float64)
    sx, sz = np.sinc(kf / n), np.sinc(kz / n)
    wk = (sx[:, None] * sx[None, :])[:, :, None] * sz[None, None, :]
    power = (dk.real**2 + dk.imag**2) * wk ** (-2.0 * p)
    del dk, wk
    knorm = np.sqrt(((kf**2)[:, None] + (kf**2)[None, :])[:, :, None] + (kz**2)[None, None, :])
    kidx = (knorm + 0.5).astype(np.int64).ravel()
    knorm, power = knorm.ravel(), power.ravel()
    knorm[0], power[0] = 0.0, 0.0
    nm = np.bincount(kidx, minlength=n).astype(np.float64)
    nm[0] -= 1  # the k = 0 mode is skipped (fourier.py:80-82)
    ks = np.bincount(kidx, weights=knorm, minlength=n)
    ps = np.bincount(kidx, weights=power, minlength=n)
    kmax = int(2 * (n // 2) / 3)
    with np.errstate(invalid="ignore", divide="ignore"):
        return ks[1:kmax] / nm[1:kmax], ps[1:kmax] / nm[1:kmax], nm[1:kmax]


# ------------------------------------------------------------------ solver.py (Newtonian FFT path)


def rhs_poisson(density, additional_field, param) -> None:  # solver.py:381-449 (linear branch)
    f1 = np.float32(1.5 * param["aexp"] * param["Om_m"] * param["parametrized_mu_z"])
    linear_operator_inplace(density, f1, -f1)


def fft(rhs, param, pk_out: list | None = None) -> np.ndarray:  # solver.py:453-522
    MAS_index = param["MAS_index"]
    rhs_fourier = fft_3D_real(rhs)
    solver = param["linear_newton_solver"].casefold()
    if param.get("save_pk", False):
        k, Pk, Nmodes = fourier_grid_to_Pk(rhs_fourier, MAS_index)
        Pk = Pk * np.float32(
            (param["boxlen"] / len(rhs) ** 2) ** 3
            / (1.5 * param["aexp"] * param["Om_m"]) ** 2
            / param["parametrized_mu_z"] ** 2
        )
        k = k * np.float32(2 * np.pi / param["boxlen"])
        if pk_out is not None:
            pk_out.append((k, Pk, Nmodes))
    if solver == "fft":
        if MAS_index == 0:
            inverse_laplacian(rhs_fourier)
        else:
            inverse_laplacian_compensated(rhs_fourier, MAS_index)
    elif solver == "fft_7pt":
        inverse_laplacian_7pt(rhs_fourier)
    else:
        raise NotImplementedError(f"{solver=}, should be 'fft' or 'fft_7pt'")
    return ifft_3D_real(rhs_fourier)


# ------------------------------------------------------------------ solver.pm / integration (oracle side)


def pm(position, param, potential=None, additional_field=None, tables=(), pk_out: list | None = None):
    """solver.py:30-215.  `param` is any mutable mapping (dict or pd.Series)."""
    empty = np.empty(0, dtype=np.float32)
    potential = empty if potential is None else potential
    additional_field = empty if additional_field is None else additional_field
    n = 2 ** int(param["ncoarse"])
    scheme = param["mass_scheme"].casefold()
    theory = param["theory"].casefold()
    if scheme == "cic":
        param["MAS_index"] = 2
        density = CIC(position, n)
    elif scheme == "tsc":
        param["MAS_index"] = 3
        density = TSC(position, n)
    else:
        raise NotImplementedError(f"{param['mass_scheme']=}, should be 'CIC' or 'TSC'")

    if theory == "parametrized":  # solver.py:95-110
        evo = param["aexp"] ** (-3 * (1 + param["w0"] + param["wa"])) * np.exp(-3 * param["wa"] * (1 - param["aexp"]))
        oml = param["Om_lambda"] * evo / (
            param["Om_m"] * param["aexp"] ** (-3) + param["Om_r"] * param["aexp"] ** (-4) + param["Om_lambda"] * evo
        )
        param["parametrized_mu_z"] = np.float32(1 + param["parametrized_mu0"] * oml / param["Om_lambda"])
    else:
        param["parametrized_mu_z"] = np.float32(1)

    if n**3 != param["npart"]:
        prod_vector_scalar_inplace(density, np.float32(n**3 / param["npart"]))

    save = param["save_power_spectrum"].casefold()
    if save == "yes":
        param["save_pk"] = True
    elif save == "z_out":
        param["save_pk"] = bool(param["write_snapshot"])
    elif save == "no":
        param["save_pk"] = False
    else:
        raise NotImplementedError(f"SAVE_POWER_SPECTRUM={save!r}, should be 'yes', 'z_out' or 'no'")

    solver = param["linear_newton_solver"].casefold()
    if param["save_pk"] and solver == "multigrid":  # solver.py:134-140
        dk = fft_3D_real(density)
        k, Pk, Nmodes = fourier_grid_to_Pk(dk, param["MAS_index"])
        Pk = Pk * np.float32((param["boxlen"] / len(density) ** 2) ** 3)
        k = k * np.float32(2 * np.pi / param["boxlen"])
        if pk_out is not None:
            pk_out.append((k, Pk, Nmodes))

    param["compute_additional_field"] = True
    additional_field = get_additional_field(additional_field, density, param, tables)
    param["compute_additional_field"] = False
    rhs_poisson_full(density, additional_field, param)
    rhs = density

    if solver == "multigrid":
        potential = initialise_potential(potential, rhs, param, tables)
        potential = mg_linear(potential, rhs, param)
    elif solver in ("fft", "fft_7pt"):
        potential = fft(rhs, param, pk_out)
    elif solver == "full_fft":
        # The reference's full_fft branch calls the two-argument fft_force with three (solver.py:181, 200) and dies
        # with TypeError: so does the restatement, unless param["full_fft_force"] asks for what the branch spells out:
        # the potential is left alone and the force comes from fft_force (solver.py:524).
        if not param.get("full_fft_force", False):
            raise TypeError("fft_force() takes 2 positional arguments but 3 were given")
    else:
        raise NotImplementedError(f"{param['linear_newton_solver']=}, should be multigrid, fft, fft_7pt or full_fft")

    if theory == "fr":
        half_c2 = fr_half_c2(param)
        if solver == "full_fft":  # solver.py:181-189: force += half_c2 grad(u^(n+1)) (mesh.add_derivative_fR)
            force = fft_force(rhs, param, pk_out)
            force = force + derivative_fR(np.zeros_like(additional_field), additional_field, half_c2, param["fR_n"],
                                          param["gradient_stencil_order"])
        else:
            force = derivative_fR(potential, additional_field, half_c2, param["fR_n"], param["gradient_stencil_order"])
    elif solver == "full_fft":
        force = fft_force(rhs, param, pk_out)
    else:
        force = derivative(potential, param["gradient_stencil_order"])

    acc = invCIC_vec(force, position) if scheme == "cic" else invTSC_vec(force, position)
    return acc, potential, additional_field


def get_additional_field(additional_field, density, param, tables):  # solver.py:285-378
    theory = param["theory"].casefold()
    if theory in ("newton", "parametrized"):
        return np.empty(0, dtype=np.float32)
    if theory == "fr":
        return _fr_additional_field(additional_field, density, param, tables)
    if theory == "mond":
        return _mond_additional_field(additional_field, density, param, tables)
    raise NotImplementedError(f"{param['theory']=}, should be 'newton', 'fr', 'parametrized' or 'mond'")


def rhs_poisson_full(density, additional_field, param) -> None:  # solver.py:381-449
    if param["compute_additional_field"] is False and param["theory"].casefold() == "mond":
        _mond_rhs(density, additional_field, param)
    else:
        rhs_poisson(density, additional_field, param)


def fr_half_c2(param):  # solver.py:167-179
    c = 299792458.0
    Rbar = 3 * param["Om_m"] * param["aexp"] ** (-3) + 12 * param["Om_lambda"]
    Rbar0 = 3 * param["Om_m"] + 12 * param["Om_lambda"]
    fR_a = -param["aexp"] ** 2 * ((Rbar0 / Rbar) ** (param["fR_n"] + 1)) * 10.0 ** (-param["fR_logfR0"])
    return np.float32(0.5 * (-fR_a) * (c * 1e-3 * param["unit_t"] / (param["unit_l"] * param["aexp"])) ** 2)


def dt_CFL_maxacc(acceleration, param):  # integration.py:267-295
    dx = np.float32(0.5 ** param["ncoarse"])
    return np.float32(param["Courant_factor"]) * np.sqrt(dx / max_abs(acceleration))


def dt_CFL_maxvel(velocity, param):  # integration.py:298-326
    dx = np.float32(0.5 ** param["ncoarse"])
    return np.float32(param["Courant_factor"]) * dx / max_abs(velocity)


def dt_weak_variation(func_t_a, param):  # integration.py:329-358
    aexp_factor = 1.0 + 0.01 * param["max_aexp_stepping"]
    return np.float32(func_t_a(np.log(aexp_factor * param["aexp"])) - func_t_a(np.log(param["aexp"])))


def integrate(position, velocity, acceleration, potential, additional_field, tables, param,
              t_snap_next=np.float32(0)):  # integration.py:17-118
    dt1 = dt_CFL_maxacc(acceleration, param)
    dt2 = dt_CFL_maxvel(velocity, param)
    dt3 = dt_weak_variation(tables[1], param)
    dt = np.min([dt1, dt2, dt3])
    if (param["t"] + dt) > t_snap_next:
        dt = t_snap_next - param["t"]
        param["write_snapshot"] = True
    else:
        param["write_snapshot"] = False
    integ = param["integrator"].casefold()
    if integ == "leapfrog":  # integration.py:192-264
        half_dt = np.float32(0.5 * dt)
        add_vector_scalar_inplace(velocity, acceleration, -half_dt)
        add_vector_scalar_inplace(position, velocity, dt)
        param["t"] += dt
        param["aexp_old"] = param["aexp"]
        param["aexp"] = np.exp(tables[0](param["t"]))
        set_units(param)
        periodic_wrap(position)
        acceleration, potential, additional_field = pm(position, param, potential, additional_field, tables)
        add_vector_scalar_inplace(velocity, acceleration, -half_dt)
    elif integ == "euler":  # integration.py:121-189
        add_vector_scalar_inplace(position, velocity, dt)
        param["t"] += dt
        param["aexp_old"] = param["aexp"]
        param["aexp"] = np.exp(tables[0](param["t"]))
        set_units(param)
        periodic_wrap(position)
        add_vector_scalar_inplace(velocity, acceleration, -dt)
        acceleration, potential, additional_field = pm(position, param, potential, additional_field, tables)
    else:
        raise NotImplementedError("ERROR: Integrator must be 'leapfrog' or 'euler'")
    return position, velocity, acceleration, potential, additional_field


# ------------------------------------------------------------------ multigrid pieces (laplacian.py, mesh.py)


def restriction(x) -> np.ndarray:  # mesh.py:14
    x = _c32(x)
    n = x.shape[0]
    out = np.empty((n // 2,) * 3, np.float32)
    _lib(_use_omp).pmo_restriction(_p(x), C.c_int(n), C.c_float(0.125), _p(out))
    return out


def minus_restriction(x) -> np.ndarray:  # mesh.py:63
    x = _c32(x)
    n = x.shape[0]
    out = np.empty((n // 2,) * 3, np.float32)
    _lib(_use_omp).pmo_restriction(_p(x), C.c_int(n), C.c_float(-0.125), _p(out))
    return out


def prolongation(x) -> np.ndarray:  # mesh.py:206
    x = _c32(x)
    nc = x.shape[0]
    y = np.empty((2 * nc,) * 3, np.float32)
    _lib(_use_omp).pmo_prolongation(_p(y), _p(x), C.c_int(nc), C.c_int(0))
    return y


def add_prolongation(y, x) -> None:  # mesh.py:334
    assert y.dtype == np.float32 and y.flags.c_contiguous
    x = _c32(x)
    _lib(_use_omp).pmo_prolongation(_p(y), _p(x), C.c_int(x.shape[0]), C.c_int(1))


def lap_operator(x) -> np.ndarray:  # laplacian.py:12
    x = _c32(x)
    out = np.empty_like(x)
    _lib(_use_omp).pmo_lap_operator(_p(x), None, C.c_int(x.shape[0]), C.c_int(0), _p(out))
    return out


def lap_residual(x, b) -> np.ndarray:  # laplacian.py:63
    x, b = _c32(x), _c32(b)
    out = np.empty_like(x)
    _lib(_use_omp).pmo_lap_operator(_p(x), _p(b), C.c_int(x.shape[0]), C.c_int(1), _p(out))
    return out


def lap_restrict_residual(x, b) -> np.ndarray:  # laplacian.py:125
    x, b = _c32(x), _c32(b)
    n = x.shape[0]
    out = np.empty((n // 2,) * 3, np.float32)
    _lib(_use_omp).pmo_lap_restrict_residual(_p(x), _p(b), C.c_int(n), _p(out))
    return out


def lap_residual_error(x, b) -> np.float32:  # laplacian.py:327
    x, b = _c32(x), _c32(b)
    return np.float32(_lib(False).pmo_lap_residual_error(_p(x), _p(b), C.c_int(x.shape[0])))


def _diff_norm(a, b, f) -> np.float32:
    a, b = _c32(a), _c32(b)
    return np.float32(_lib(False).pmo_diff_norm(_p(a), _p(b), C.c_float(f), C.c_int64(a.size)))


def lap_truncation_error(x) -> np.float32:  # laplacian.py:502
    return _diff_norm(restriction(lap_operator(x)), lap_operator(restriction(x)), 1.0)


def lap_initialise_potential(b) -> np.ndarray:  # laplacian.py:765
    b = _c32(b)
    out = np.empty_like(b)
    _lib(_use_omp).pmo_lap_initialise_potential(_p(b), C.c_int(b.shape[0]), _p(out))
    return out


def lap_gauss_seidel(x, b, f_relax) -> None:  # laplacian.py:844
    assert x.dtype == np.float32 and x.flags.c_contiguous
    b = _c32(b)
    _lib(_use_omp).pmo_lap_gauss_seidel(_p(x), _p(b), C.c_int(x.shape[0]), C.c_float(f_relax))


def lap_smoothing(x, b, n_smoothing) -> None:  # laplacian.py:1026
    for _ in range(int(n_smoothing)):
        lap_gauss_seidel(x, b, np.float32(1.25))


# ------------------------------------------------------------------ cubic.py (f(R), n = 1)


def cubic_operator(x, b, q, fr_n: int = 1) -> np.ndarray:  # cubic.py:23 / quartic.py:20
    x, b = _c32(x), _c32(b)
    out = np.empty_like(x)
    _lib(_use_omp).pmo_cubic_operator(_p(x), _p(b), C.c_float(q), None, C.c_int(x.shape[0]), C.c_int(fr_n), _p(out))
    return out


def cubic_residual_with_rhs(x, b, q, rhs, fr_n: int = 1) -> np.ndarray:  # cubic.py:90 / quartic.py:86
    x, b, rhs = _c32(x), _c32(b), _c32(rhs)
    out = np.empty_like(x)
    _lib(_use_omp).pmo_cubic_operator(_p(x), _p(b), C.c_float(q), _p(rhs), C.c_int(x.shape[0]), C.c_int(fr_n), _p(out))
    return out


def solution_quartic_equation(p, q) -> np.float32:  # quartic.py:157
    return np.float32(_lib(False).pmo_quartic_solution(C.c_float(p), C.c_float(q)))


def quartic_initialise_potential(b, q) -> np.ndarray:  # quartic.py:214
    b = _c32(b)
    out = np.empty_like(b)
    _lib(_use_omp).pmo_quartic_initialise_potential(_p(b), C.c_float(q), C.c_int(b.shape[0]), _p(out))
    return out


def solution_cubic_equation(p, d1) -> np.float32:  # cubic.py:162
    return np.float32(_lib(False).pmo_cubic_solution(C.c_float(p), C.c_float(d1)))


def cubic_initialise_potential(b, q) -> np.ndarray:  # cubic.py:217
    b = _c32(b)
    out = np.empty_like(b)
    _lib(_use_omp).pmo_cubic_initialise_potential(_p(b), C.c_float(q), C.c_int(b.shape[0]), _p(out))
    return out


def cubic_gauss_seidel(x, b, q, f_relax, rhs=None, fr_n: int = 1) -> None:  # cubic.py:269/:446, quartic.py:270/:448
    assert x.dtype == np.float32 and x.flags.c_contiguous
    b = _c32(b)
    r = None if rhs is None else _c32(rhs)
    _lib(_use_omp).pmo_cubic_gauss_seidel(_p(x), _p(b), C.c_float(q), None if r is None else _p(r),
                                          C.c_int(x.shape[0]), C.c_int(fr_n), C.c_float(f_relax))


def cubic_smoothing(x, b, q, n_smoothing, rhs=None, fr_n: int = 1) -> None:  # cubic.py:1065 / :1100
    for _ in range(int(n_smoothing)):
        cubic_gauss_seidel(x, b, q, np.float32(1.25), rhs, fr_n)


def cubic_residual_error(x, b, q, fr_n: int = 1) -> np.float32:  # cubic.py:844 / quartic.py:844
    x, b = _c32(x), _c32(b)
    return np.float32(_lib(False).pmo_cubic_residual_error(_p(x), _p(b), C.c_float(q), C.c_int(x.shape[0]),
                                                           C.c_int(fr_n)))


def cubic_truncation_error(x, b, q, fr_n: int = 1) -> np.float32:  # cubic.py:1021 / quartic.py:1022
    RLx = restriction(cubic_operator(x, b, q, fr_n))
    LRx = cubic_operator(restriction(x), restriction(b), q, fr_n)
    return _diff_norm(RLx, LRx, 4.0)


# ------------------------------------------------------------------ mond.py

_MOND_KIND = {"simple": 0, "n": 1, "beta": 2, "gamma": 3, "delta": 4}


def mond_nu(y, kind: str, alpha=1.0) -> np.float32:  # mond.py:16-162
    return np.float32(_lib(False).pmo_mond_nu(C.c_float(y), C.c_int(_MOND_KIND[kind]), C.c_float(alpha)))


def mond_rhs(potential, out, g0, kind: str, alpha=1.0) -> None:  # mond.py:171-932
    assert out.dtype == np.float32 and out.flags.c_contiguous
    pot = _c32(potential)
    _lib(_use_omp).pmo_mond_rhs(_p(pot), _p(out), C.c_int(pot.shape[0]), C.c_float(g0), C.c_int(_MOND_KIND[kind]),
                                C.c_float(alpha))


# ------------------------------------------------------------------ multigrid.py / solver.py orchestration


def initialise_potential(potential, rhs, param, tables):  # solver.py:218-282
    if len(potential) == 0:
        if param["compute_additional_field"] and param["theory"].casefold() == "fr":
            if param["fR_n"] == 1:
                return cubic_initialise_potential(rhs, param["fR_q"])
            if param["fR_n"] == 2:
                return quartic_initialise_potential(rhs, param["fR_q"])
            raise NotImplementedError(f"Only f(R) with n = 1 and 2, currently {param['fR_n']=}")
        return lap_initialise_potential(rhs)
    if not param["compute_additional_field"]:
        scaling = (param["aexp"] * tables[3](np.log(param["aexp"]))
                   / (param["aexp_old"] * tables[3](np.log(param["aexp_old"]))))
        prod_vector_scalar_inplace(potential, scaling)
    return potential


def V_cycle(x, b, param, nlevel=0) -> None:  # multigrid.py:474-517
    lap_smoothing(x, b, param["Npre"])
    res_c = lap_restrict_residual(x, b)
    x_corr_c = lap_initialise_potential(res_c)
    if nlevel >= (param["ncoarse"] - 3):
        lap_smoothing(x_corr_c, res_c, param["Npre"])
    else:
        V_cycle(x_corr_c, res_c, param, nlevel + 1)
    add_prolongation(x, x_corr_c)
    lap_smoothing(x, b, param["Npost"])


def mg_linear(x, b, param, info: dict | None = None):  # multigrid.py:23-83
    theory = param["theory"].casefold()
    if param["compute_additional_field"] and theory == "fr":
        raise ValueError("Linear should not be used for scalaron field")
    mond_pot = (not param["compute_additional_field"]) and theory == "mond"
    if ("tolerance" not in param) or (param["nsteps"] % 3) == 0:
        tolerance = param["epsrel"] * lap_truncation_error(x)
        param["tolerance_mond" if mond_pot else "tolerance"] = tolerance
    tolerance = param["tolerance_mond"] if mond_pot else param["tolerance"]
    residual_err = 1e30
    cycles = 0
    while residual_err > tolerance:
        V_cycle(x, b, param)
        cycles += 1
        tmp = lap_residual_error(x, b)
        if tmp < tolerance or residual_err / tmp < 2:
            break
        residual_err = tmp
    if info is not None:
        info["cycles"] = cycles
    return x


def V_cycle_FAS(x, b, param, nlevel=0, rhs=None) -> None:  # multigrid.py:521-579 (f(R) branches)
    q, m = np.float32(param["fR_q"]), int(param["fR_n"])
    cubic_smoothing(x, b, q, param["Npre"], rhs, m)
    if rhs is None:
        res_c = minus_restriction(cubic_operator(x, b, q, m))
    else:
        res_c = restriction(cubic_residual_with_rhs(x, b, q, rhs, m))
    x_c = restriction(x)
    x_corr_c = x_c.copy()
    b_c = restriction(b)
    L_c = cubic_operator(x_c, b_c, q, m)
    linear_operator_vectors_inplace(res_c, np.float32(4), L_c, np.float32(1))
    if nlevel >= (param["ncoarse"] - 3):
        cubic_smoothing(x_corr_c, b_c, q, param["Npre"], res_c, m)
    else:
        V_cycle_FAS(x_corr_c, b_c, param, nlevel + 1, res_c)
    add_vector_scalar_inplace(x_corr_c, x_c, np.float32(-1))
    add_prolongation(x, x_corr_c)
    cubic_smoothing(x, b, q, param["Npost"], rhs, m)


def FAS(x, b, param, info: dict | None = None):  # multigrid.py:88-138
    q, m = np.float32(param["fR_q"]), int(param["fR_n"])
    if m not in (1, 2):
        raise NotImplementedError(f"Only f(R) with n = 1 and 2, currently {param['fR_n']=}")
    if ("tolerance_FAS" not in param) or (param["nsteps"] % 3) == 0:
        param["tolerance_FAS"] = param["epsrel"] * cubic_truncation_error(x, b, q, m)
    tolerance = param["tolerance_FAS"]
    residual_err = 1e30
    cycles = 0
    while residual_err > tolerance:
        V_cycle_FAS(x, b, param)
        cycles += 1
        tmp = cubic_residual_error(x, b, q, m)
        if tmp < tolerance or residual_err / tmp < 2:
            break
        residual_err = tmp
    if info is not None:
        info["cycles"] = cycles
    return x


def _fr_additional_field(additional_field, density, param, tables):  # solver.py:326-359
    c = 299792458.0
    Rbar = 3 * param["Om_m"] * param["aexp"] ** (-3) + 12 * param["Om_lambda"]
    Rbar0 = 3 * param["Om_m"] + 12 * param["Om_lambda"]
    fR_a = -param["aexp"] ** 2 * ((Rbar0 / Rbar) ** (param["fR_n"] + 1)) * 10.0 ** (-param["fR_logfR0"])
    c2 = (c * 1e-3 * param["unit_t"] / (param["unit_l"] * param["aexp"])) ** 2
    f1 = np.float32(param["aexp"] * param["Om_m"] / (c2 * 6)) / (-fR_a)
    f2 = np.float32(Rbar / 3 * param["aexp"] ** 4 - param["Om_m"] * param["aexp"]) / (6 * c2) / (-fR_a)
    dens_term = linear_operator(density, f1, f2)
    q = np.float32(-param["aexp"] ** 4 * Rbar / (18 * c2)) / (-fR_a)
    param["fR_q"] = q
    additional_field = initialise_potential(additional_field, dens_term, param, tables)
    return FAS(additional_field, dens_term, param)


def _mond_g0(param):  # solver.py:418-425
    return (param["mond_g0"] * 1e-3 * 1e-10 * param["unit_t"] ** 2 / param["unit_l"]
            * param["aexp"] ** (1 + param["mond_scale_factor_exponent"]))


def _mond_rhs(density, additional_field, param) -> None:  # solver.py:413-443
    fn = param["mond_function"].casefold()
    if fn not in _MOND_KIND:
        raise NotImplementedError(f"MOND_FUNCTION={fn!r}, should be 'simple', 'n', 'beta', 'gamma' or 'delta'")
    mond_rhs(additional_field, density, _mond_g0(param), fn, param["mond_alpha"])


def _mond_additional_field(additional_field, density, param, tables):  # solver.py:360-374
    rhs_poisson_full(density, additional_field, param)  # compute_additional_field is True -> linear branch
    solver = param["linear_newton_solver"].casefold()
    if solver == "multigrid":
        additional_field = initialise_potential(additional_field, density, param, tables)
        return mg_linear(additional_field, density, param)
    if solver == "fft_7pt":
        return fft(density, param)
    raise NotImplementedError(f"{param['linear_newton_solver']=}, should be 'multigrid' or 'fft_7pt'")


# ------------------------------------------------------------------ initial conditions: nLPT (row f3)


def _kvec(n):
    k = _kfreq(n)
    return k[:, None, None], k[None, :, None], np.arange(n // 2 + 1, dtype=np.float32)[None, None, :]


def gradient(x) -> np.ndarray:  # fourier.py:730-781: i 2 pi k_m x -> [n, n, n/2+1, 3]
    n = x.shape[0]
    twopi = np.float32(2 * np.pi)
    re, im = -(twopi * x.imag), twopi * x.real
    out = np.empty(x.shape + (3,), dtype=np.complex64)
    for m, km in enumerate(_kvec(n)):
        out[..., m] = (re * km) + 1j * (im * km)
    return out


def hessian(x, ij) -> np.ndarray:  # fourier.py:784-839
    kk = _kvec(x.shape[0])
    c = (-kk[ij[0]] * kk[ij[1]]) * np.float32(4 * np.pi**2)
    return ((c * x.real) + 1j * (c * x.imag)).astype(np.complex64)


def sum_of_hessian(x, ij1, ij2) -> np.ndarray:  # fourier.py:842-904
    kk = _kvec(x.shape[0])
    c = -(kk[ij1[0]] * kk[ij1[1]] + kk[ij2[0]] * kk[ij2[1]]) * np.float32(4 * np.pi**2)
    return ((c * x.real) + 1j * (c * x.imag)).astype(np.complex64)


def diff_of_hessian(x, ij1, ij2) -> np.ndarray:  # fourier.py:907-961
    kk = _kvec(x.shape[0])
    c = -(kk[ij1[0]] * kk[ij1[1]] - kk[ij2[0]] * kk[ij2[1]]) * np.float32(4 * np.pi**2)
    return ((c * x.real) + 1j * (c * x.imag)).astype(np.complex64)


def _hr(phi_k, ij):
    return ifft_3D_real(hessian(phi_k, ij))


def _corner_blocks(x, n_out) -> np.ndarray:
    """initial_conditions.py:1859-1928 (pad: n -> 3n/2, trim: 3n/2 -> n): zeros, then the four low/high corner blocks
    of the x and y axes and the low block of z, `middle` = half the SMALLER size (the Nyquist planes are dropped)."""
    n_in = x.shape[0]
    middle = min(n_in, n_out) // 2
    low = np.arange(middle)                                   # frequencies 0 .. middle - 1
    keep_in = np.concatenate([low, np.arange(n_in - middle + 1, n_in)])      # ... and -(middle - 1) .. -1
    keep_out = np.concatenate([low, np.arange(n_out - middle + 1, n_out)])   # the same frequencies in the other grid
    out = np.zeros((n_out, n_out, n_out // 2 + 1), dtype=x.dtype)
    out[np.ix_(keep_out, keep_out, low)] = x[np.ix_(keep_in, keep_in, low)]
    return out


def pad(x) -> np.ndarray:  # initial_conditions.py:1859-1894
    return _corner_blocks(x, 3 * x.shape[0] // 2)


def trim(x) -> np.ndarray:  # initial_conditions.py:1897-1928
    return _corner_blocks(x, 2 * x.shape[0] // 3)


def _dealiased(rhs_fn, factor, *phis_k):
    """Orszag's 3/2 rule as the reference applies it to every product of Hessians (initial_conditions.py:1008,
    1034-1038): pad the potentials, form the products on the 3n/2 grid, transform, trim, transform back, times
    `factor` (1.5^3; 1.5^6 for the cubic term, :1120)."""
    out = rhs_fn(*[pad(p) for p in phis_k])
    return (ifft_3D_real(trim(fft_3D_real(out))) * np.float32(factor)).astype(np.float32)


def compute_3a_rhs(phi_1_k) -> np.ndarray:  # initial_conditions.py:1042-1121 (de-aliasing: _dealiased)
    one, two = np.float32(1), np.float32(2)
    out = _hr(phi_1_k, (0, 0))
    out = out * (one * _hr(phi_1_k, (1, 1)))
    out = out * (one * _hr(phi_1_k, (2, 2)))
    out = out + two * _hr(phi_1_k, (0, 1)) * _hr(phi_1_k, (0, 2)) * _hr(phi_1_k, (1, 2))
    for off, diag in (((1, 2), (0, 0)), ((0, 2), (1, 1)), ((0, 1), (2, 2))):
        t1 = _hr(phi_1_k, off)
        out = out + (-one) * t1 * t1 * _hr(phi_1_k, diag)
    return out.astype(np.float32)


def compute_3b_rhs(phi_1_k, phi_2_k) -> np.ndarray:  # initial_conditions.py:1162-1246
    one, half = np.float32(1), np.float32(0.5)
    out = _hr(phi_1_k, (0, 0))
    out = out * (half * ifft_3D_real(sum_of_hessian(phi_2_k, (1, 1), (2, 2))))
    out = out + half * _hr(phi_1_k, (1, 1)) * ifft_3D_real(sum_of_hessian(phi_2_k, (0, 0), (2, 2)))
    out = out + half * _hr(phi_1_k, (2, 2)) * ifft_3D_real(sum_of_hessian(phi_2_k, (0, 0), (1, 1)))
    for ij in ((0, 1), (0, 2), (1, 2)):
        out = out + (-one) * _hr(phi_1_k, ij) * _hr(phi_2_k, ij)
    return out.astype(np.float32)


# 3c: the three components of the transverse part (initial_conditions.py:1290-1361, 1405-1476, 1520-1591): per
# component (first term's two Hessians of phi_1 / phi_2, the pair of the second term, and the (off-diagonal,
# difference of diagonals) pair of the last two)
_3C = {"Ax": (((0, 2), (0, 1)), ((0, 2), (0, 1)), (1, 2), ((1, 1), (2, 2))),
       "Ay": (((0, 1), (1, 2)), ((0, 1), (1, 2)), (0, 2), ((2, 2), (0, 0))),
       "Az": (((1, 2), (0, 2)), ((1, 2), (0, 2)), (0, 1), ((0, 0), (1, 1)))}


def compute_3c_rhs(phi_1_k, phi_2_k, which: str) -> np.ndarray:
    one = np.float32(1)
    (a1, a2), (b2, b1), off, (d1, d2) = _3C[which]
    out = _hr(phi_1_k, a1)
    out = out * (one * _hr(phi_2_k, a2))
    out = out + (-one) * _hr(phi_2_k, b2) * _hr(phi_1_k, b1)
    out = out + one * _hr(phi_1_k, off) * ifft_3D_real(diff_of_hessian(phi_2_k, d1, d2))
    out = out + (-one) * _hr(phi_2_k, off) * ifft_3D_real(diff_of_hessian(phi_1_k, d1, d2))
    return out.astype(np.float32)


def compute_2ndorder_rhs(phi_1_k) -> np.ndarray:  # initial_conditions.py:976-1039 (de-aliasing: _dealiased)
    phi_2 = ifft_3D_real(hessian(phi_1_k, (0, 0)))
    phi_2 = phi_2 * ifft_3D_real(sum_of_hessian(phi_1_k, (1, 1), (2, 2)))
    phi_2 = phi_2 + np.float32(1) * ifft_3D_real(hessian(phi_1_k, (1, 1))) * ifft_3D_real(hessian(phi_1_k, (2, 2)))
    for ij in ((0, 1), (0, 2), (1, 2)):
        t = ifft_3D_real(hessian(phi_1_k, ij))
        phi_2 = phi_2 + np.float32(-1) * t * t
    return phi_2.astype(np.float32)


def lpt_particles(density_k_half, order: str, position_ics: str, scalars, dealiased: bool = False):
    """The grid part of initial_conditions.generate (initial_conditions.py:108-158, 1688-1856) from the Fourier-space
    density: inverse Laplacian -> displacement -> particles; `scalars`: Hz [km/s/Mpc], d1_z0, d1, f1, d2, f2, unit_t."""
    pc = 3.0856775814913674e16
    n = density_k_half.shape[0]
    Hz = scalars["Hz"] * scalars["unit_t"] / (1e3 * pc)
    pot1 = np.ascontiguousarray(density_k_half.astype(np.complex64))
    with np.errstate(all="ignore"):
        inverse_laplacian(pot1)
    psi = ifft_3D_real_grad(gradient(pot1))
    dplus_1 = np.float32(scalars["d1"] / scalars["d1_z0"])
    fH_1 = np.float32(scalars["f1"] * Hz)
    h = np.float32(1.0 / n)
    q = np.arange(n, dtype=np.float32) * h + (np.float32(0.5 / n) if position_ics.casefold() == "center" else np.float32(0))
    lattice = np.stack(np.meshgrid(q, q, q, indexing="ij"), axis=-1)
    position = (lattice + dplus_1 * (-psi)).astype(np.float32)
    velocity = ((dplus_1 * fH_1) * (-psi)).astype(np.float32)
    if order.casefold() in ("2lpt", "3lpt"):
        if dealiased:
            rhs2, rhs3a = (lambda p1: _dealiased(compute_2ndorder_rhs, 1.5**3, p1)), (lambda p1: _dealiased(compute_3a_rhs, 1.5**6, p1))
            rhs3b = lambda p1, p2: _dealiased(compute_3b_rhs, 1.5**3, p1, p2)  # noqa: E731
            rhs3c = lambda p1, p2, which: _dealiased(lambda a, b: compute_3c_rhs(a, b, which), 1.5**3, p1, p2)  # noqa: E731
        else:
            rhs2, rhs3a, rhs3b, rhs3c = compute_2ndorder_rhs, compute_3a_rhs, compute_3b_rhs, compute_3c_rhs
        rho2_k = fft_3D_real(rhs2(pot1))
        with np.errstate(all="ignore"):
            inverse_laplacian(rho2_k)
        psi2 = ifft_3D_real_grad(gradient(rho2_k))
        dplus_2 = np.float32(scalars["d2"] / scalars["d1_z0"] ** 2)
        fH_2 = np.float32(scalars["f2"] * Hz)
        position = (position + dplus_2 * psi2).astype(np.float32)
        velocity = (velocity + (dplus_2 * fH_2) * psi2).astype(np.float32)
    if order.casefold() == "3lpt":  # initial_conditions.py:159-198: every term through grad(inverse Laplacian)
        def add(rhs, d, f):
            nonlocal position, velocity
            psi3 = ifft_3D_real_grad(gradient_inverse_laplacian(fft_3D_real(rhs)))
            dplus = -np.float32(d / scalars["d1_z0"] ** 3)
            fH = np.float32(f * Hz)
            position = (position + dplus * psi3).astype(np.float32)
            velocity = (velocity + (dplus * fH) * psi3).astype(np.float32)

        with np.errstate(all="ignore"):
            add(rhs3a(pot1), scalars["d3a"], scalars["f3a"])
            add(rhs3b(pot1, rho2_k), scalars["d3b"], scalars["f3b"])
            for which in ("Ax", "Ay", "Az"):
                add(rhs3c(pot1, rho2_k, which), scalars["d3c"], scalars["f3c"])
    position, velocity = np.ascontiguousarray(position.reshape(-1, 3)), np.ascontiguousarray(velocity.reshape(-1, 3))
    periodic_wrap(position)
    return position, velocity
