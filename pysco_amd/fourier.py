"""FFT wrappers (rocFFT, persistent plans), k-space Green's functions and the P(k) estimator.

Mirror of the reference's fourier.py entry points on the hot path (fourier.py:22-100, 104-147,
251-294, 460-595).  Half-spectra are complex64 [n, n, n//2+1] exactly as np.fft.rfftn lays them out.
The per-axis tables (k_i^2, sinc(k_i/n), sin^2(pi k_i/n)) are O(n) host-side setup evaluated with
the reference's own fp32 expressions; the O(n^3) work is on the GPU.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

from . import _lib as L


def _kfreq(n: int) -> np.ndarray:
    i = np.arange(n)
    return np.where(i >= n // 2, i - n, i).astype(np.float32)  # fourier.py:523-527


def green_tables(n: int, kind: int):
    """(k2, s, c0) for pm_greens: fp32 per-axis tables + scalar prefactor."""
    if kind == L.GREEN_PLAIN:  # fourier.py:470-489 ('>' middle, unlike the two below)
        i = np.arange(n)
        k = np.where(i > n // 2, n - i, i).astype(np.float32)
        return k**2, np.ones(n, np.float32), float(np.float32(-0.25 / np.pi**2))
    k = _kfreq(n)
    if kind == L.GREEN_COMPENSATED:  # fourier.py:517-543
        h = np.float32(1.0 / n)
        return k**2, np.sinc(k * h).astype(np.float32), float(np.float32(-0.25 / np.pi**2))
    if kind == L.GREEN_7PT:  # fourier.py:574-594
        pi_h = np.float32(np.pi / n)
        h = np.float32(1.0 / n)
        c0 = float(-(np.float32(0.25) * h**2))
        return np.zeros(n, np.float32), (np.sin(pi_h * k) ** 2).astype(np.float32), c0
    raise ValueError(f"unknown Green's function kind {kind}")


def fft_3D_real(x, threads: int = 1):
    """Unnormalised r2c transform of a real [n,n,n] grid (`threads` is accepted and ignored)."""
    torch = L.require_gpu()
    xd = L.to_device(x, torch.float32)
    n = xd.shape[0]
    ctx = L.context(n)
    xk = torch.empty((n, n, n // 2 + 1), dtype=torch.complex64, device=xd.device)
    ctx.call("pm_fft_r2c", L.ptr(xd), L.ptr(xk))
    return L.like_input(xk, x)


def ifft_3D_real(x, threads: int = 1):
    """c2r transform carrying 1/n^3; the device input buffer may be overwritten."""
    torch = L.require_gpu()
    xk = L.to_device(x, torch.complex64)
    n = xk.shape[0]
    ctx = L.context(n)
    out = torch.empty((n, n, n), dtype=torch.float32, device=xk.device)
    work = xk.clone() if L.is_device_array(x) else xk  # rocFFT c2r destroys its input
    ctx.call("pm_fft_c2r", L.ptr(work), L.ptr(out))
    return L.like_input(out, x)


def _greens_inplace(x, kind: int, p: int) -> None:
    torch = L.require_gpu()
    if not L.is_device_array(x):
        xk = L.to_device(x, torch.complex64)
    else:
        xk = x
        if not xk.is_contiguous() or xk.dtype != torch.complex64:
            raise ValueError("device half-spectrum must be contiguous complex64")
    n = xk.shape[0]
    ctx = L.context(n)
    k2, s, c0 = green_tables(n, kind)
    ctx.call("pm_greens", kind, int(p), L.ptr(xk), L.hptr(k2), L.hptr(s), c0)
    if not L.is_device_array(x):
        x[...] = xk.cpu().numpy()


def inverse_laplacian(x) -> None:  # fourier.py:460
    _greens_inplace(x, L.GREEN_PLAIN, 0)


def inverse_laplacian_compensated(x, p: int) -> None:  # fourier.py:502
    _greens_inplace(x, L.GREEN_COMPENSATED, p)


def inverse_laplacian_7pt(x) -> None:  # fourier.py:555
    _greens_inplace(x, L.GREEN_7PT, 0)


def _grad_planar(xk, p: int):
    """Three planar half-spectra [3, n, n, n/2+1] of the k-space force (device tensors)."""
    torch = L.require_gpu()
    n = xk.shape[0]
    out = torch.empty((3, n, n, n // 2 + 1), dtype=torch.complex64, device=xk.device)
    k = _kfreq(n)
    sincs = np.sinc(k * np.float32(1.0 / n)).astype(np.float32)
    L.context(n).call("pm_grad_inverse_laplacian", int(p), L.ptr(xk), L.ptr(out), L.hptr(k), L.hptr(sincs))
    return out


def gradient_inverse_laplacian(x):
    """-i k / (2 pi k^2) x as [n, n, n/2+1, 3] complex64 (fourier.py:606)."""
    torch = L.require_gpu()
    out = _grad_planar(L.to_device(x, torch.complex64), 0).permute(1, 2, 3, 0).contiguous()
    return L.like_input(out, x)


def gradient_inverse_laplacian_compensated(x, p: int):
    """The same with the mass-assignment window deconvolved, W^(-2p) (fourier.py:660)."""
    torch = L.require_gpu()
    out = _grad_planar(L.to_device(x, torch.complex64), int(p)).permute(1, 2, 3, 0).contiguous()
    return L.like_input(out, x)


def _ifft_grad_planar(planar):
    """[3, n, n, n/2+1] complex64 (destroyed) -> force grid [n, n, n, 3] float32."""
    torch = L.require_gpu()
    n = planar.shape[1]
    ctx = L.context(n)
    real = torch.empty((3, n, n, n), dtype=torch.float32, device=planar.device)
    for m in range(3):
        ctx.call("pm_fft_c2r", L.ptr(planar[m]), L.ptr(real[m]))
    out = torch.empty((n, n, n, 3), dtype=torch.float32, device=planar.device)
    ctx.call("pm_interleave3", L.ptr(real), n**3, L.ptr(out))
    return out


def ifft_3D_real_grad(x, threads: int = 1):
    """c2r transform of a [n, n, n/2+1, 3] spectrum over its first three axes -> [n, n, n, 3] (fourier.py:343)."""
    torch = L.require_gpu()
    xk = L.to_device(x, torch.complex64)
    return L.like_input(_ifft_grad_planar(xk.permute(3, 0, 1, 2).contiguous()), x)


def gradient(x):
    """i 2 pi k x as [n, n, n/2+1, 3] complex64 (fourier.py:730)."""
    torch = L.require_gpu()
    xk = L.to_device(x, torch.complex64)
    n = xk.shape[0]
    out = torch.empty((3, n, n, n // 2 + 1), dtype=torch.complex64, device=xk.device)
    L.context(n).call("pm_kspace_gradient", L.ptr(xk), L.ptr(out), L.hptr(_kfreq(n)))
    return L.like_input(out.permute(1, 2, 3, 0).contiguous(), x)


def _hessian(x, a1, b1, a2, b2):
    torch = L.require_gpu()
    xk = L.to_device(x, torch.complex64)
    n = xk.shape[0]
    out = torch.empty_like(xk)
    L.context(n).call("pm_kspace_hessian", L.ptr(xk), L.ptr(out), L.hptr(_kfreq(n)), a1, b1, a2, b2)
    return L.like_input(out, x)


def hessian(x, ij):
    """-k_i k_j 4 pi^2 x (fourier.py:784)."""
    return _hessian(x, int(ij[0]), int(ij[1]), -1, -1)


def sum_of_hessian(x, ij1, ij2):
    """-(k_i1 k_j1 + k_i2 k_j2) 4 pi^2 x (fourier.py:842)."""
    return _hessian(x, int(ij1[0]), int(ij1[1]), int(ij2[0]), int(ij2[1]))


def diff_of_hessian(x, ij1, ij2):
    """-(k_i1 k_j1 - k_i2 k_j2) 4 pi^2 x (fourier.py:907)."""
    return _hessian(x, int(ij1[0]), int(ij1[1]), 4 + int(ij2[0]), 4 + int(ij2[1]))


def fourier_grid_to_Pk(density_k, p: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(k, P(k), Nmodes) in grid units; also zeroes density_k[0,0,0] like the reference."""
    torch = L.require_gpu()
    xk = density_k if L.is_device_array(density_k) else L.to_device(density_k, torch.complex64)
    n = xk.shape[0]
    ctx = L.context(n)
    sincs = np.sinc(_kfreq(n) * np.float32(1.0 / n)).astype(np.float32)
    ks, ps, nm = (np.zeros(n, np.float64) for _ in range(3))
    ctx.call("pm_pk", int(p), L.ptr(xk), L.hptr(sincs), L.hptr(ks), L.hptr(ps), L.hptr(nm))
    if not L.is_device_array(density_k):
        density_k[0, 0, 0] = 0
    kmax = int(2 * (n // 2) / 3)
    with np.errstate(invalid="ignore", divide="ignore"):
        k = (ks[1:kmax] / nm[1:kmax]).astype(np.float32)
        pk = (ps[1:kmax] / nm[1:kmax]).astype(np.float32)
    return k, pk, nm[1:kmax].astype(np.float32)
