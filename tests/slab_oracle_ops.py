"""TEST-ONLY compute provider for pysco_amd.slab: the same interface as slab.HipOps, implemented with
the CPU oracle on CPU torch tensors, so that the multi-rank orchestration (partition, ghost
exchange, FFT transposes, particle migration) can be exercised with gloo and no GPU."""
import numpy as np
import torch

from oracle import pysco_oracle as orc
from pysco_amd import _lib as L


class OracleOps:
    def __init__(self, n):
        self.n = n

    @staticmethod
    def _np(t):
        return t.numpy()

    def _planes(self, nx, x0):
        return [(x0 + l) % self.n for l in range(nx)]

    def kick_drift_wrap(self, pos, vel, acc, half_dt, dt, x_lo, x_hi):
        p, v, a = self._np(pos), self._np(vel), self._np(acc)
        orc.add_vector_scalar_inplace(v, a, -np.float32(half_dt))
        orc.add_vector_scalar_inplace(p, v, np.float32(dt))
        orc.periodic_wrap(p)
        x = p[:, 0]
        return torch.from_numpy(np.nonzero((x < np.float32(x_lo)) | (x >= np.float32(x_hi)))[0].astype(np.int64))

    def deposit(self, scheme, pos, nx, x0):
        fn = {L.TSC: orc.TSC, L.CIC: orc.CIC, L.NGP: orc.NGP}[scheme]
        full = fn(self._np(pos), self.n) if pos.shape[0] else np.zeros((self.n,) * 3, np.float32)
        return torch.from_numpy(np.ascontiguousarray(full[self._planes(nx, x0)]))

    def interp_kick(self, scheme, force, nx, x0, pos, vel, half_dt):
        full = np.zeros((self.n, self.n, self.n, 3), np.float32)
        full[self._planes(nx, x0)] = self._np(force)
        fn = {L.TSC: orc.invTSC_vec, L.CIC: orc.invCIC_vec, L.NGP: orc.invNGP_vec}[scheme]
        acc = fn(full, self._np(pos)) if pos.shape[0] else np.zeros((0, 3), np.float32)
        mv = 0.0
        if vel is not None:
            v = self._np(vel)
            orc.add_vector_scalar_inplace(v, acc, -np.float32(half_dt))
            mv = float(np.abs(v).max()) if v.size else 0.0
        ma = float(np.abs(acc).max()) if acc.size else 0.0
        return torch.from_numpy(acc), ma, mv

    def max_abs(self, x):
        return float(x.abs().max()) if x.numel() else 0.0

    def axpy(self, y, x, a):
        orc.add_vector_scalar_inplace(self._np(y), self._np(x), np.float32(a))

    def linear_operator_inplace(self, x, f1, f2):
        a = np.ascontiguousarray(self._np(x))
        orc.linear_operator_inplace(a, f1, f2)
        x.copy_(torch.from_numpy(a))

    def scale_inplace(self, x, a):
        x.mul_(float(a))

    def fft2d_r2c(self, x):
        return torch.from_numpy(np.fft.rfft2(self._np(x).astype(np.float64), axes=(1, 2)).astype(np.complex64))

    def fft2d_c2r(self, xk):
        n = self.n
        r = np.fft.irfft2(self._np(xk).astype(np.complex128), s=(n, n), axes=(1, 2)) / n  # 1/n^3 in total
        return torch.from_numpy(np.ascontiguousarray(r.astype(np.float32)))

    def fft1d_x(self, data, inverse):
        a = self._np(data).astype(np.complex128)
        r = np.fft.ifft(a, axis=0) * self.n if inverse else np.fft.fft(a, axis=0)  # both unnormalised
        data.copy_(torch.from_numpy(r.astype(np.complex64)))

    def greens(self, kind, p, data, j0, nj, scale=None):
        n = self.n
        full = np.zeros((n, n, n // 2 + 1), np.complex64)
        full[:, j0:j0 + nj] = self._np(data)
        if kind == L.GREEN_PLAIN:
            orc.inverse_laplacian(full)
        elif kind == L.GREEN_COMPENSATED:
            orc.inverse_laplacian_compensated(full, p)
        else:
            orc.inverse_laplacian_7pt(full)
        if scale is not None:
            full *= np.float32(scale)
        data.copy_(torch.from_numpy(np.ascontiguousarray(full[:, j0:j0 + nj])))

    # numpy equivalents of the LDS-FFT passes (unnormalised transforms, 1/n^3 inside the x pass)
    def has_fftp(self, nx, nky):
        return True

    def fftp_forward(self, rhs):
        a = np.fft.rfft(self._np(rhs).astype(np.float64), axis=2)      # [x, y, kz]
        a = np.fft.fft(a, axis=1)                                        # [x, ky, kz]
        return torch.from_numpy(np.ascontiguousarray(a.transpose(1, 2, 0)).astype(np.complex64))  # [ky, kz, x]

    def fftp_xgreen(self, seg, kind, p, nx, nky, ky0, scale=None):
        n, nh = self.n, self.n // 2 + 1
        G = n // nx
        a = self._np(seg).reshape(G, nky, nh, nx).transpose(1, 2, 0, 3).reshape(nky, nh, n)   # [ky_l, kz, x]
        ak = np.fft.fft(a.astype(np.complex128), axis=2)                                       # [ky_l, kz, kx]
        full = np.zeros((n, n, nh), np.complex64)
        full[:, ky0:ky0 + nky, :] = ak.transpose(2, 0, 1)
        if kind == L.GREEN_PLAIN:
            orc.inverse_laplacian(full)
        elif kind == L.GREEN_COMPENSATED:
            orc.inverse_laplacian_compensated(full, p)
        else:
            orc.inverse_laplacian_7pt(full)
        if scale is not None:
            full *= np.float32(scale)
        ak = full[:, ky0:ky0 + nky, :].transpose(1, 2, 0).astype(np.complex128) / float(n) ** 3
        a = np.fft.ifft(ak, axis=2) * n                                                         # unnormalised
        out = a.reshape(nky, nh, G, nx).transpose(2, 0, 1, 3)
        seg.copy_(torch.from_numpy(np.ascontiguousarray(out).astype(np.complex64)).reshape(seg.shape))

    def fftp_inverse(self, c, nx):
        n = self.n
        a = self._np(c).reshape(n, n // 2 + 1, nx).transpose(2, 0, 1).astype(np.complex128)     # [x, ky, kz]
        a = np.fft.ifft(a, axis=1) * n
        r = np.fft.irfft(a, n=n, axis=2) * n
        return torch.from_numpy(np.ascontiguousarray(r.astype(np.float32)))

    def gradient(self, order, phi_ext, x0_phi, nx_force, x0_force):
        n = self.n
        full = np.zeros((n, n, n), np.float32)
        full[self._planes(phi_ext.shape[0], x0_phi)] = self._np(phi_ext)
        F = orc.derivative(full, order)
        return torch.from_numpy(np.ascontiguousarray(F[self._planes(nx_force, x0_force)]))
