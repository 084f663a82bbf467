"""Initial conditions from a power spectrum with first / second order Lagrangian perturbation theory: the nLPT branch of
the reference's initial_conditions.generate (initial_conditions.py:108-158; row f3 of SURVEY.md section 8).

Pipeline (initial_conditions.py:402-445, 531-728, 976-1039, 1635-1856): Gaussian white noise in Fourier space drawn
with NumPy's generator exactly as the reference draws it (same seed -> same field), times sqrt(P(k)) interpolated on
the grid -> inverse Laplacian -> displacement = gradient -> particles displaced from the lattice; second order from
the Hessian products of the first-order potential.  The random numbers and the transfer table are host work (a few
NumPy calls); every grid operation -- Green's function, k-space gradient / Hessians, FFTs, products, the particle
displacement -- is a HIP kernel behind the C ABI (include/pmstep.h).  1LPT, 2LPT and 3LPT, with or without the
reference's `dealiased_ICS` (Orszag's 3/2 rule: `_ProductGrid`).
"""
from __future__ import annotations

import logging

import numpy as np

from . import _lib as L
from . import fourier, iostream, utils

_PC = 3.0856775814913674e16  # astropy.constants.pc (initial_conditions.py:118)


# ---- host side: random field and transfer table (same draws and rounding as the reference) -------------------


def _hermitian_fill(upper: np.ndarray, corners) -> np.ndarray:
    """The reference's loop (initial_conditions.py:626-640) as array operations: for i <= n/2 and all (j, k),
    D[i, j, k] = upper[i, j, k] and D[-i, -j, -k] = conj(upper[i, j, k]); on the two self-conjugate planes (i = 0,
    n/2) the loop writes every cell twice and the later write (row-major iteration order) stays.  Returns the half
    spectrum D[:, :, :n/2+1], the only part the pipeline reads."""
    m1, n = upper.shape[0], upper.shape[1]
    m = m1 - 1
    D = np.empty((n, n, n), dtype=np.complex64)
    D[:m1] = upper
    idx = (-np.arange(n)) % n
    if m > 1:
        D[n - np.arange(1, m)] = np.conj(upper[1:m][:, idx][:, :, idx])
    order = np.arange(n)[:, None] * n + np.arange(n)[None, :]
    order_partner = order[idx][:, idx]  # iteration number of the (j, k) whose conjugate write lands here
    for p in (0, m):
        partner = np.conj(upper[p][idx][:, idx])
        D[p] = np.where(order > order_partner, upper[p], partner)
    for (a, b, c), v in corners:
        D[a, b, c] = v
    return np.ascontiguousarray(D[:, :, :m1])


def white_noise_fourier(ncells_1d: int, rng) -> np.ndarray:
    """Rayleigh amplitudes, uniform phases (initial_conditions.py:585-660): half spectrum [n, n, n/2+1] complex64."""
    n, m = ncells_1d, ncells_1d // 2
    ra = rng.random((m + 1, n, n), dtype=np.float32)
    rp = rng.random((m + 1, n, n), dtype=np.float32)
    amplitude = np.sqrt(-np.log((np.float32(1) - ra).astype(np.float64)))
    phase = (np.float32(2 * np.pi) * rp).astype(np.float64)
    upper = (amplitude * np.cos(phase) + 1j * (amplitude * np.sin(phase))).astype(np.complex64)
    pts = [(0, 0, m), (0, m, 0), (0, m, m), (m, 0, 0), (m, 0, m), (m, m, 0), (m, m, m)]
    corners = [((0, 0, 0), 0)] + [(p, np.sqrt(-np.log(float(np.float32(1) - rng.random(dtype=np.float32))))) for p in pts]
    return _hermitian_fill(upper, corners)


def white_noise_fourier_fixed(ncells_1d: int, rng, is_paired: bool) -> np.ndarray:
    """Unit amplitudes, uniform phases (+ pi when paired) (initial_conditions.py:664-728)."""
    n, m = ncells_1d, ncells_1d // 2
    rp = rng.random((m + 1, n, n), dtype=np.float32)
    shift = np.float32(np.pi) if is_paired else np.float32(0)
    phase = (np.float32(2 * np.pi) * rp + shift).astype(np.float64)
    upper = (np.cos(phase) + 1j * np.sin(phase)).astype(np.complex64)
    pts = [(0, 0, m), (0, m, 0), (0, m, m), (m, 0, 0), (m, 0, m), (m, m, 0), (m, m, m)]
    return _hermitian_fill(upper, [((0, 0, 0), 0)] + [(p, 1.0) for p in pts])


def get_transfer_grid(param) -> np.ndarray:
    """sqrt(P(k)) on the half grid [n, n, n/2+1], float64 (initial_conditions.py:531-581)."""
    k, Pk = np.loadtxt(param["power_spectrum_file"]).T
    n = int(round(param["npart"] ** (1.0 / 3.0)))
    if n**3 != param["npart"]:
        raise ValueError(f"{param['npart']=}, should be a cube")
    kf = 2 * np.pi / param["boxlen"]
    sqrtPk = (np.sqrt(Pk / param["boxlen"] ** 3) * n**3).astype(np.float32)
    k1 = np.fft.fftfreq(n, 1 / n)
    kz = k1[: n // 2 + 1].copy()
    kz[-1] = abs(kz[-1])  # (fftfreq gives -n/2 there; only its square is used)
    k_grid = np.sqrt(kz[None, None, :] ** 2 + k1[:, None, None] ** 2 + k1[None, :, None] ** 2)
    return np.interp(k_grid, k / kf, sqrtPk)


def generate_density_fourier(param) -> np.ndarray:
    """White noise times the transfer function: half spectrum complex64 (initial_conditions.py:402-445)."""
    n = int(round(param["npart"] ** (1.0 / 3.0)))
    rng = np.random.default_rng() if param["seed"] < 0 else np.random.default_rng(param["seed"])
    if param["fixed_ICS"]:
        density_k = white_noise_fourier_fixed(n, rng, param["paired_ICS"])
    else:
        density_k = white_noise_fourier(n, rng)
    return (density_k.astype(np.complex128) * get_transfer_grid(param)).astype(np.complex64)


# ---- device side ------------------------------------------------------------------------------------------


def _displacement(potential_k):
    """psi = ifft(i 2 pi k phi) as [n, n, n, 3] (fourier.gradient + ifft_3D_real_grad)."""
    torch = L.require_gpu()
    n = potential_k.shape[0]
    planar = torch.empty((3, n, n, n // 2 + 1), dtype=torch.complex64, device=potential_k.device)
    L.context(n).call("pm_kspace_gradient", L.ptr(potential_k), L.ptr(planar), L.hptr(fourier._kfreq(n)))
    return fourier._ifft_grad_planar(planar)


class _ProductGrid:
    """Where the products of Hessians are formed: the potential's own grid, or -- dealiased_ICS, Orszag's 3/2 rule
    (initial_conditions.py:1008, 1034-1038 and the same lines of every source term) -- one of 3 n / 2 cells per side:
    lift() pads the potential (initial_conditions.pad, :1859), hessian_real() is fourier.hessian / sum_of_hessian /
    diff_of_hessian + ifft_3D_real there, finish() transforms the product, trims it (:1897), transforms back and
    multiplies by the reference's factor."""

    def __init__(self, n, param):
        self.torch = L.require_gpu()
        self.n, self.ctx = n, L.context(n)
        self.dealiased = bool(param.get("dealiased_ICS", False))
        self.m = 3 * n // 2 if self.dealiased else n
        self.suffix = "_padded" if self.dealiased else ""
        self.kfreq = fourier._kfreq(self.m)

    def _spectrum(self, m, like):
        return self.torch.empty((m, m, m // 2 + 1), dtype=self.torch.complex64, device=like.device)

    def lift(self, potential_k):
        if not self.dealiased:
            return potential_k
        out = self._spectrum(self.m, potential_k)
        self.ctx.call("pm_dealias_resize", L.ptr(potential_k), L.ptr(out), 1)
        return out

    def hessian_real(self, potential_k, ij, ij2=(-1, -1)):
        m = self.m
        tmp = self.torch.empty_like(potential_k)
        self.ctx.call("pm_kspace_hessian" + self.suffix, L.ptr(potential_k), L.ptr(tmp), L.hptr(self.kfreq), ij[0], ij[1],
                      ij2[0], ij2[1])
        out = self.torch.empty((m, m, m), dtype=self.torch.float32, device=potential_k.device)
        self.ctx.call("pm_fft_c2r" + self.suffix, L.ptr(tmp), L.ptr(out))
        return out

    def diff_real(self, potential_k, d1, d2):  # fourier.diff_of_hessian: second pair given as 4 + axis
        return self.hessian_real(potential_k, d1, (4 + d2[0], 4 + d2[1]))

    def mul(self, y, f, x):  # utils.prod_vector_vector[_scalar]_inplace: y *= f * x
        self.ctx.call("pm_addcmul", L.ptr(y), float(f), L.ptr(y), L.ptr(x), y.numel(), 0)

    def addmul(self, y, f, a, b):  # utils.add_vector_vector_inplace: y += f * a * b
        self.ctx.call("pm_addcmul", L.ptr(y), float(f), L.ptr(a), L.ptr(b), y.numel(), 1)

    def addmul3(self, y, f, a, b, c):  # utils.add_vector_vector_vector_inplace: y += f * a * b * c
        self.ctx.call("pm_addcmul3", L.ptr(y), float(f), L.ptr(a), L.ptr(b), L.ptr(c), y.numel())

    def finish(self, product, factor=1.5**3):
        if not self.dealiased:
            return product
        n = self.n
        spec = self._spectrum(self.m, product)
        self.ctx.call("pm_fft_r2c_padded", L.ptr(product), L.ptr(spec))
        del product
        small = self._spectrum(n, spec)
        self.ctx.call("pm_dealias_resize", L.ptr(spec), L.ptr(small), 0)
        del spec
        out = self.torch.empty((n, n, n), dtype=self.torch.float32, device=small.device)
        self.ctx.call("pm_fft_c2r", L.ptr(small), L.ptr(out))
        self.ctx.call("pm_scale", L.ptr(out), float(np.float32(factor)), out.numel())
        return out


def compute_2ndorder_rhs(potential_1_k, param):
    """phi,xx (phi,yy + phi,zz) + phi,yy phi,zz - phi,xy^2 - phi,xz^2 - phi,yz^2 (initial_conditions.py:976-1039)."""
    g = _ProductGrid(potential_1_k.shape[0], param)
    p1 = g.lift(potential_1_k)
    phi_2 = g.hessian_real(p1, (0, 0))
    g.mul(phi_2, 1.0, g.hessian_real(p1, (1, 1), (2, 2)))
    g.addmul(phi_2, 1.0, g.hessian_real(p1, (1, 1)), g.hessian_real(p1, (2, 2)))
    for ij in ((0, 1), (0, 2), (1, 2)):
        tmp = g.hessian_real(p1, ij)
        g.addmul(phi_2, -1.0, tmp, tmp)
    return g.finish(phi_2)


def compute_3a_rhs(potential_1_k, param):
    """phi,xx phi,yy phi,zz + 2 phi,xy phi,xz phi,yz - phi,yz^2 phi,xx - phi,xz^2 phi,yy - phi,xy^2 phi,zz
    (initial_conditions.py:1042-1121, Scoccimarro 1998 B.2); de-aliased, the cubic term carries 1.5^6 (:1120)."""
    g = _ProductGrid(potential_1_k.shape[0], param)
    p1 = g.lift(potential_1_k)
    out = g.hessian_real(p1, (0, 0))
    g.mul(out, 1.0, g.hessian_real(p1, (1, 1)))
    g.mul(out, 1.0, g.hessian_real(p1, (2, 2)))
    t1, t2, t3 = (g.hessian_real(p1, ij) for ij in ((0, 1), (0, 2), (1, 2)))
    g.addmul3(out, 2.0, t1, t2, t3)
    del t1, t2, t3
    for off, diag in (((1, 2), (0, 0)), ((0, 2), (1, 1)), ((0, 1), (2, 2))):
        t1, t2 = g.hessian_real(p1, off), g.hessian_real(p1, diag)
        g.addmul3(out, -1.0, t1, t1, t2)
    return g.finish(out, 1.5**6)


def compute_3b_rhs(potential_1_k, potential_2_k, param):
    """Half the mixed products of the diagonal Hessians of phi_1 and phi_2 minus the off-diagonal ones
    (initial_conditions.py:1162-1246)."""
    g = _ProductGrid(potential_1_k.shape[0], param)
    p1, p2 = g.lift(potential_1_k), g.lift(potential_2_k)
    out = g.hessian_real(p1, (0, 0))
    g.mul(out, 0.5, g.hessian_real(p2, (1, 1), (2, 2)))
    g.addmul(out, 0.5, g.hessian_real(p1, (1, 1)), g.hessian_real(p2, (0, 0), (2, 2)))
    g.addmul(out, 0.5, g.hessian_real(p1, (2, 2)), g.hessian_real(p2, (0, 0), (1, 1)))
    for ij in ((0, 1), (0, 2), (1, 2)):
        g.addmul(out, -1.0, g.hessian_real(p1, ij), g.hessian_real(p2, ij))
    return g.finish(out)


# the transverse part, component by component (initial_conditions.py:1290-1361, 1405-1476, 1520-1591): Hessian
# components of (phi_1, phi_2) in the first product, of (phi_2, phi_1) in the second, then the off-diagonal one that
# multiplies the difference of two diagonal ones
_3C = {"Ax": (((0, 2), (0, 1)), ((0, 2), (0, 1)), (1, 2), ((1, 1), (2, 2))),
       "Ay": (((0, 1), (1, 2)), ((0, 1), (1, 2)), (0, 2), ((2, 2), (0, 0))),
       "Az": (((1, 2), (0, 2)), ((1, 2), (0, 2)), (0, 1), ((0, 0), (1, 1)))}


def compute_3c_rhs(potential_1_k, potential_2_k, which: str, param):
    g = _ProductGrid(potential_1_k.shape[0], param)
    p1, p2 = g.lift(potential_1_k), g.lift(potential_2_k)
    (a1, a2), (b2, b1), off, (d1, d2) = _3C[which]
    out = g.hessian_real(p1, a1)
    g.mul(out, 1.0, g.hessian_real(p2, a2))
    g.addmul(out, -1.0, g.hessian_real(p2, b2), g.hessian_real(p1, b1))
    g.addmul(out, 1.0, g.hessian_real(p1, off), g.diff_real(p2, d1, d2))
    g.addmul(out, -1.0, g.hessian_real(p2, off), g.diff_real(p1, d1, d2))
    return g.finish(out)


def _displacement_from_density(density):
    """psi = ifft(grad(inverse Laplacian(fft(density)))) as [n, n, n, 3]: the form the third-order terms use
    (initial_conditions.py:1152-1157: fourier.gradient_inverse_laplacian + ifft_3D_real_grad)."""
    return fourier.ifft_3D_real_grad(fourier.gradient_inverse_laplacian(fourier.fft_3D_real(density)))


def generate(param, tables):
    """(position, velocity) [Np, 3] float32 device tensors from 1LPT, 2LPT or 3LPT (initial_conditions.py:108-205), wrapped
    into the box and written as snapshot 0 like finalise_initial_conditions (:216-279)."""
    torch = L.require_gpu()
    order = param["initial_conditions"].casefold()
    if order not in ("1lpt", "2lpt", "3lpt"):
        raise ValueError(f"{param['initial_conditions']=}, should be 1LPT, 2LPT or 3LPT")
    lna_start = np.log(1.0 / (1 + param["z_start"]))
    logging.warning(f"{param['z_start']=}")
    Hz = float(tables[2](lna_start)) * param["unit_t"] / (1e3 * _PC)  # km/s/Mpc -> box units
    density_k = torch.from_numpy(generate_density_fourier(param)).cuda()
    n = density_k.shape[0]
    ctx = L.context(n)
    fourier.inverse_laplacian(density_k)
    potential_1_k = density_k
    psi = _displacement(potential_1_k)
    logging.warning("Compute 1LPT contribution")
    d1_z0 = float(tables[3](0))
    dplus_1 = np.float32(float(tables[3](lna_start)) / d1_z0)
    fH_1 = np.float32(float(tables[4](lna_start)) * Hz)
    position, velocity = torch.empty_like(psi), torch.empty_like(psi)
    POSITION = param["position_ICS"].casefold()
    if POSITION not in ("center", "edge"):
        raise NotImplementedError(f"{POSITION=}, should be 'center' or 'edge'")
    ctx.call("pm_lpt_displace", L.ptr(psi), float(dplus_1), float(dplus_1 * fH_1), 0 if POSITION == "center" else 1,
             L.ptr(position), L.ptr(velocity))
    del psi
    if order in ("2lpt", "3lpt"):
        logging.warning("Compute 2LPT contribution")
        density_2 = compute_2ndorder_rhs(potential_1_k, param)
        density_2_k = fourier.fft_3D_real(density_2)
        del density_2
        fourier.inverse_laplacian(density_2_k)
        psi_2 = _displacement(density_2_k)
        dplus_2 = np.float32(float(tables[5](lna_start)) / d1_z0**2)
        fH_2 = np.float32(float(tables[6](lna_start)) * Hz)
        ctx.call("pm_lpt_displace", L.ptr(psi_2), float(dplus_2), float(dplus_2 * fH_2), 2, L.ptr(position), L.ptr(velocity))
        del psi_2
    if order == "3lpt":  # initial_conditions.py:159-198
        potential_2_k = density_2_k

        def add(label, density_3, i_d):
            logging.warning(f"Compute 3LPT {label} contribution")
            psi_3 = _displacement_from_density(density_3)
            dplus_3 = -np.float32(float(tables[i_d](lna_start)) / d1_z0**3)
            fH_3 = np.float32(float(tables[i_d + 1](lna_start)) * Hz)
            ctx.call("pm_lpt_displace", L.ptr(psi_3), float(dplus_3), float(dplus_3 * fH_3), 2, L.ptr(position), L.ptr(velocity))

        add("a)", compute_3a_rhs(potential_1_k, param), 7)
        add("b)", compute_3b_rhs(potential_1_k, potential_2_k, param), 9)
        for which in ("Ax", "Ay", "Az"):
            add(f"c) {which}", compute_3c_rhs(potential_1_k, potential_2_k, which, param), 11)
    position, velocity = position.view(-1, 3), velocity.view(-1, 3)
    utils.periodic_wrap(position)
    if "base" in param:  # finalise_initial_conditions: the initial snapshot
        snap = dict(param.items()) if not isinstance(param, dict) else dict(param)
        snap["i_snap"] = 0
        iostream.write_snapshot_particles(position, velocity, snap)
        if snap["output_snapshot_format"].casefold() == "parquet":
            # the reference's finalise_initial_conditions names the side file of the INITIAL snapshot without the
            # snapshot number (initial_conditions.py:266-270: output_00000/param_{extra}.txt): written under that
            # name too, for tools that look for the reference's layout
            iostream.write_param_file(f"{snap['base']}/output_00000/param_{snap['extra']}.txt", iostream._public(snap))
    return position, velocity
