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

    def gradient_fr(self, order, fr_n, f, phi_ext, u_ext, x0_phi, nx_force, x0_force):
        n = self.n
        a, b = np.zeros((n, n, n), np.float32), np.ones((n, n, n), np.float32)
        planes = self._planes(phi_ext.shape[0], x0_phi)
        a[planes], b[planes] = self._np(phi_ext), self._np(u_ext)
        F = orc.derivative_fR(a, b, np.float32(f), int(fr_n), order)
        return torch.from_numpy(np.ascontiguousarray(F[self._planes(nx_force, x0_force)]))

    def level_kernels(self):
        import torch.distributed as dist

        return NumpyLevelKernels(dist.get_rank(), dist.get_world_size())


class NumpyLevelKernels:
    """TEST-ONLY twin of pysco_amd.slab_multigrid.HipLevelKernels on CPU torch tensors.  Every level operation
    embeds the rank's planes (with their ghosts) at their global positions in a full periodic grid of that level,
    runs the single-process oracle function on it and takes the owned planes back: the owned outputs only depend
    on cells the rank holds, so this is the slab operation.  One-colour half-sweeps come from the oracle's
    red-then-black sweep: the red cells of its result are the red half; for the black half the grid is shifted by
    one cell so that black sits where the oracle relaxes first.  Tolerance-level, not bit-level, equivalence."""

    device = "cpu"

    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def _embed(self, x, fill=0.0):
        a = x.numpy()
        n, nx = a.shape[1], a.shape[0] - 2
        full = np.full((n, n, n), fill, np.float32)
        planes = [(self.rank * nx - 1 + l) % n for l in range(nx + 2)]
        full[planes] = a
        return full, planes

    def _own(self, full, nx):
        lo = self.rank * nx
        return full[lo:lo + nx]

    @staticmethod
    def _parity(n):
        i, j, k = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
        return (i + j + k) & 1

    def _half_sweep(self, x, sweep, colour):
        """sweep(y_full) does an in-place red-then-black oracle sweep on a full grid."""
        nx = x.shape[0] - 2
        full, _ = self._embed(x)
        n = full.shape[0]
        odd = self._parity(n) == 1
        if colour == 1:
            y = full.copy()
            sweep(y, 0)
            full[odd] = y[odd]
        else:
            y = np.ascontiguousarray(np.roll(full, 1, axis=2))
            sweep(y, 1)
            y = np.roll(y, -1, axis=2)
            full[~odd] = y[~odd]
        x[1:-1].copy_(torch.from_numpy(np.ascontiguousarray(self._own(full, nx))))

    def gs(self, x, b, x_lo, colour):
        fb, _ = self._embed(b)

        def sweep(y, shift):
            orc.lap_gauss_seidel(y, np.ascontiguousarray(np.roll(fb, shift, axis=2)), np.float32(1.25))

        self._half_sweep(x, sweep, colour)

    def restrict_residual(self, x, b, out):
        fx, _ = self._embed(x)
        fb, _ = self._embed(b)
        out[1:-1].copy_(torch.from_numpy(np.ascontiguousarray(self._own(orc.lap_restrict_residual(fx, fb), out.shape[0] - 2))))

    def restriction(self, x, scale, out):
        fx, _ = self._embed(x)
        r = orc.restriction(fx) * np.float32(scale / 0.125)
        out[1:-1].copy_(torch.from_numpy(np.ascontiguousarray(self._own(r, out.shape[0] - 2))))

    def prolongation_add(self, y, x_c):
        fy, _ = self._embed(y)
        fc, _ = self._embed(x_c)
        orc.add_prolongation(fy, fc)
        y[1:-1].copy_(torch.from_numpy(np.ascontiguousarray(self._own(fy, y.shape[0] - 2))))

    def operator(self, x, out):
        fx, _ = self._embed(x)
        out[1:-1].copy_(torch.from_numpy(np.ascontiguousarray(self._own(orc.lap_operator(fx), x.shape[0] - 2))))

    def residual_sumsq(self, x, b):
        fx, _ = self._embed(x)
        fb, _ = self._embed(b)
        r = self._own(orc.lap_residual(fx, fb), x.shape[0] - 2).astype(np.float64)
        return float((r * r).sum())

    def diff_sumsq(self, a, b, f):
        d = (np.float32(f) * a.numpy() - b.numpy()).astype(np.float64)
        return float((d * d).sum())

    def linear_operator(self, out, x, f1, f2):
        out.copy_(torch.from_numpy(np.float32(f1) * x.numpy() + np.float32(f2)))

    def mond_rhs(self, pot, out, g0, kind, alpha):
        fp, _ = self._embed(pot)
        fo = np.empty_like(fp)
        name = {0: "simple", 1: "n", 2: "beta", 3: "gamma", 4: "delta"}[int(kind)]
        orc.mond_rhs(fp, fo, np.float32(g0), name, alpha)
        out[1:-1].copy_(torch.from_numpy(np.ascontiguousarray(self._own(fo, pot.shape[0] - 2))))

    # f(R): scalaron fields are ~1, so unset cells of the embedding are filled with 1 (no 0 ** negative power anywhere)
    def fr_operator(self, n_fr, x, b, q, rhs, out):
        fx, _ = self._embed(x, 1.0)
        fb, _ = self._embed(b, 1.0)
        if rhs is None:
            r = orc.cubic_operator(fx, fb, np.float32(q), n_fr)
        else:
            r = orc.cubic_residual_with_rhs(fx, fb, np.float32(q), self._embed(rhs)[0], n_fr)
        out[1:-1].copy_(torch.from_numpy(np.ascontiguousarray(self._own(r, x.shape[0] - 2))))

    def fr_gs(self, n_fr, x, b, q, rhs, x_lo, colour):
        fb, _ = self._embed(b, 1.0)
        fr = None if rhs is None else self._embed(rhs)[0]

        def sweep(y, shift):
            with np.errstate(all="ignore"):
                orc.cubic_gauss_seidel(y, np.ascontiguousarray(np.roll(fb, shift, axis=2)), np.float32(q), np.float32(1.25),
                                       None if fr is None else np.ascontiguousarray(np.roll(fr, shift, axis=2)), n_fr)

        nx = x.shape[0] - 2
        full, _ = self._embed(x, 1.0)
        n = full.shape[0]
        odd = self._parity(n) == 1
        if colour == 1:
            y = full.copy()
            sweep(y, 0)
            full[odd] = y[odd]
        else:
            y = np.ascontiguousarray(np.roll(full, 1, axis=2))
            sweep(y, 1)
            y = np.roll(y, -1, axis=2)
            full[~odd] = y[~odd]
        x[1:-1].copy_(torch.from_numpy(np.ascontiguousarray(self._own(full, nx))))

    def fr_residual_sumsq(self, n_fr, x, b, q):
        fx, _ = self._embed(x, 1.0)
        fb, _ = self._embed(b, 1.0)
        r = self._own(orc.cubic_operator(fx, fb, np.float32(q), n_fr), x.shape[0] - 2).astype(np.float64)
        return float((r * r).sum())

    def fr_initialise_potential(self, n_fr, b, q, out):
        fb, planes = self._embed(b, 1.0)
        with np.errstate(all="ignore"):
            r = orc.cubic_initialise_potential(fb, np.float32(q)) if n_fr == 1 else orc.quartic_initialise_potential(fb, np.float32(q))
        out.copy_(torch.from_numpy(np.ascontiguousarray(r[planes])))

    def coarse_linear(self, res_full, param, nlevel):
        res = res_full.numpy()
        x = orc.lap_initialise_potential(res)
        if nlevel >= (param["ncoarse"] - 3):
            orc.lap_smoothing(x, res, param["Npre"])
        else:
            orc.V_cycle(x, res, param, nlevel + 1)
        return torch.from_numpy(x)

    def coarse_fas(self, x_full, b_full, res_full, param, nlevel):
        x, b, res = x_full.numpy(), b_full.numpy(), np.ascontiguousarray(res_full.numpy())
        q, n_fr = np.float32(param["fR_q"]), int(param["fR_n"])
        xc = x.copy()
        res = (np.float32(4) * res + orc.cubic_operator(x, b, q, n_fr)).astype(np.float32)
        with np.errstate(all="ignore"):
            if nlevel >= (param["ncoarse"] - 3):
                orc.cubic_smoothing(xc, b, q, param["Npre"], res, n_fr)
            else:
                orc.V_cycle_FAS(xc, b, param, nlevel + 1, res)
        return torch.from_numpy(xc - x)
