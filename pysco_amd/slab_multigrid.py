"""Linear multigrid on x-slabs with halo exchange (multi-GPU; no reference counterpart: PySCo is
single-process).  Mirrors pysco_amd/multigrid.py:linear / V_cycle (reference multigrid.py:23-83, 474-517)
level by level; every level array is [nx_l + 2, n_l, n_l] -- the rank's planes of that level plus one
ghost plane per side, refreshed from the ring neighbours before each kernel that reads x-neighbours
(twice per Gauss-Seidel sweep: the black half needs the neighbours' fresh red values).  Arithmetic per
cell is the single-GPU kernels' (same code, pmstep.h "slab" entry points); only the norms are summed in
a different order (per rank, then all-reduce).

Below the level where a rank would own fewer than two planes (or the grid is smaller than 16^3) the
coarse problem is a few thousand cells: it is all-gathered and every rank runs the single-GPU V-cycle
on it (deterministic, identical on all ranks), then takes its planes of the correction.
"""
from __future__ import annotations

import ctypes as C
import logging

import numpy as np

from . import _lib as L
from . import laplacian, multigrid


class HipLevelKernels:
    """The level operations of the slab multigrid through the C ABI (pmstep.h slab entry points).  Level arrays
    are [nx + 2, n, n] torch CUDA tensors.  tests/slab_oracle_ops.py holds a CPU twin (test-only) with the same
    methods so that the orchestration below can run under gloo without a GPU."""

    device = "cuda"

    def __init__(self):
        self.torch = L.require_gpu()

    @staticmethod
    def _dims(x):
        return x.shape[1], x.shape[0] - 2

    def gs(self, x, b, x_lo, colour) -> None:
        n, nx = self._dims(x)
        L.context(n).call("pm_lap_gauss_seidel_slab", L.ptr(x), L.ptr(b), n, nx, int(x_lo), int(colour), 1.25)

    def restrict_residual(self, x, b, out) -> None:
        n, nx = self._dims(x)
        L.context(n).call("pm_lap_restrict_residual_slab", L.ptr(x), L.ptr(b), n, nx, L.ptr(out))

    def restriction(self, x, scale, out) -> None:
        n, nx = self._dims(x)
        L.context(n).call("pm_restriction_slab", L.ptr(x), n, nx, float(scale), L.ptr(out))

    def prolongation_add(self, y, x_c) -> None:
        nc, nxc = self._dims(x_c)
        L.context(2 * nc).call("pm_prolongation_slab", L.ptr(y), L.ptr(x_c), nc, nxc, 1)

    def operator(self, x, out) -> None:
        n, nx = self._dims(x)
        L.context(n).call("pm_lap_operator_slab", L.ptr(x), None, n, nx, 0, L.ptr(out))

    def residual_sumsq(self, x, b) -> float:
        n, nx = self._dims(x)
        s = C.c_double(0)
        L.context(n).call("pm_lap_residual_sumsq_slab", L.ptr(x), L.ptr(b), n, nx, C.byref(s))
        return s.value

    def diff_sumsq(self, a, b, f) -> float:
        s = C.c_double(0)
        L.context(a.shape[1]).call("pm_diff_sumsq", L.ptr(a), L.ptr(b), float(f), a.numel(), C.byref(s))
        return s.value

    def linear_operator(self, out, x, f1, f2) -> None:
        L.context(x.shape[1]).call("pm_linear_operator", L.ptr(out), L.ptr(x), float(f1), float(f2), x.numel())

    def mond_rhs(self, pot, out, g0, kind, alpha) -> None:
        n, nx = self._dims(pot)
        L.context(n).call("pm_mond_rhs_slab", L.ptr(pot), L.ptr(out), n, nx, float(np.float32(g0)), int(kind),
                          float(np.float32(alpha)))

    # f(R)
    def fr_operator(self, n_fr, x, b, q, rhs, out) -> None:
        n, nx = self._dims(x)
        L.context(n).call("pm_fr_operator_slab", n_fr, L.ptr(x), L.ptr(b), q, None if rhs is None else L.ptr(rhs), n, nx,
                          L.ptr(out))

    def fr_gs(self, n_fr, x, b, q, rhs, x_lo, colour) -> None:
        n, nx = self._dims(x)
        L.context(n).call("pm_fr_gauss_seidel_slab", n_fr, L.ptr(x), L.ptr(b), q, None if rhs is None else L.ptr(rhs), n,
                          nx, int(x_lo), int(colour), 1.25)

    def fr_residual_sumsq(self, n_fr, x, b, q) -> float:
        n, nx = self._dims(x)
        s = C.c_double(0)
        L.context(n).call("pm_fr_residual_sumsq_slab", n_fr, L.ptr(x), L.ptr(b), q, n, nx, C.byref(s))
        return s.value

    def fr_initialise_potential(self, n_fr, b, q, out) -> None:
        name = "pm_cubic_initialise_potential_flat" if n_fr == 1 else "pm_quartic_initialise_potential_flat"
        L.context(b.shape[1]).call(name, L.ptr(b), q, b.shape[1], b.numel(), L.ptr(out))

    # the small replicated coarse problems: the single-GPU cycle on full grids
    def coarse_linear(self, res_full, param, nlevel):
        x_full = laplacian.initialise_potential(res_full)
        if nlevel >= (param["ncoarse"] - 3):
            laplacian.smoothing(x_full, res_full, param["Npre"])
        else:
            multigrid.V_cycle(x_full, res_full, param, nlevel + 1)
        return x_full

    def coarse_fas(self, x_full, b_full, res_full, param, nlevel):
        """Rest of a FAS cycle below this level; returns the coarse correction x_corr - x."""
        from . import utils

        x_corr_full = x_full.clone()
        L_full = multigrid.operator(x_full, param, b_full)
        utils.linear_operator_vectors_inplace(res_full, multigrid.normalisation_residual(param), L_full, np.float32(1))
        del L_full
        if nlevel >= (param["ncoarse"] - 3):
            multigrid.smoothing(x_corr_full, b_full, param["Npre"], param, res_full)
        else:
            multigrid.V_cycle_FAS(x_corr_full, b_full, param, nlevel + 1, res_full)
        utils.add_vector_scalar_inplace(x_corr_full, x_full, np.float32(-1))
        return x_corr_full


class SlabMultigrid:
    def __init__(self, slab, kernels=None):
        import torch

        self.slab = slab
        self.k = kernels if kernels is not None else HipLevelKernels()
        self.torch = torch
        self.world, self.rank = slab.world, slab.rank

    # ---- level geometry / communication ---------------------------------------------------------
    def nx_of(self, n_l: int) -> int:
        return n_l // self.world

    def distributed(self, n_l: int) -> bool:
        """This level is kept on slabs (every rank owns an even number >= 2 of planes, rows kernel applies)."""
        nx = n_l // self.world
        return n_l >= 16 and n_l % self.world == 0 and nx >= 2 and nx % 2 == 0

    def new(self, n_l: int):
        return self.torch.empty((self.nx_of(n_l) + 2, n_l, n_l), dtype=self.torch.float32, device=self.k.device)

    def extend(self, owned):
        """[nx, n, n] owned planes -> level array with (unfilled) ghosts."""
        x = self.new(owned.shape[1])
        x[1:-1].copy_(owned)
        return x

    def exchange(self, x) -> None:
        """Fill the two ghost planes of a level array from the ring neighbours."""
        from .slab import exchange_planes

        nx = x.shape[0] - 2
        from_left, from_right = exchange_planes(x[1], x[nx])
        x[0].copy_(from_left)
        x[nx + 1].copy_(from_right)

    def all_sum(self, v: float) -> float:
        from .slab import _dist

        dist = _dist()
        t = self.torch.tensor([v], dtype=self.torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def gather_full(self, x):
        """Level array -> the full periodic [n, n, n] grid on every rank."""
        from .slab import _dist, _stage, _unstage

        dist = _dist()
        mine = _stage(x[1:-1].contiguous())
        parts = [self.torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine)
        return _unstage(self.torch.cat(parts, dim=0), x)

    # ---- level operations (pmstep.h slab entry points) ---------------------------------------------------
    def smoothing(self, x, b, nsweeps: int) -> None:
        x_lo = self.rank * (x.shape[0] - 2)
        for _ in range(int(nsweeps)):
            for colour in (1, 0):  # red, then black (laplacian.py:870-1022)
                self.exchange(x)
                self.k.gs(x, b, x_lo, colour)

    def restrict_residual(self, x, b):
        out = self.new(x.shape[1] // 2)
        self.exchange(x)
        self.k.restrict_residual(x, b, out)
        return out

    def add_prolongation(self, x, x_c) -> None:
        self.exchange(x_c)
        self.k.prolongation_add(x, x_c)

    def initialise_potential(self, b):
        """-h^2/6 * b (laplacian.py:765) on the whole level array."""
        n = b.shape[1]
        out = self.torch.empty_like(b)
        h = np.float32(1.0 / n)
        self.k.linear_operator(out, b, np.float32(-float(h * h) / 6.0), 0.0)
        return out

    def residual_error(self, x, b) -> np.float32:
        self.exchange(x)
        return np.float32(np.sqrt(self.all_sum(self.k.residual_sumsq(x, b))))

    def truncation_error(self, x) -> np.float32:
        """|| R(L x) - L(R x) || (laplacian.py:502) on slabs."""
        n = x.shape[1]
        self.exchange(x)
        Lx = self.torch.empty_like(x)
        self.k.operator(x, Lx)
        RLx, Rx = self.new(n // 2), self.new(n // 2)
        self.k.restriction(Lx, 0.125, RLx)
        self.k.restriction(x, 0.125, Rx)
        self.exchange(Rx)
        LRx = self.torch.empty_like(Rx)
        self.k.operator(Rx, LRx)
        return np.float32(np.sqrt(self.all_sum(self.k.diff_sumsq(RLx[1:-1], LRx[1:-1], 1.0))))  # owned planes: contiguous

    # ---- the cycle (multigrid.py:474-517) -----------------------------------------------------------------------
    def V_cycle(self, x, b, param, nlevel: int = 0) -> None:
        n = x.shape[1]
        self.smoothing(x, b, param["Npre"])
        res_c = self.restrict_residual(x, b)
        n_c = n // 2
        if self.distributed(n_c):
            x_corr_c = self.initialise_potential(res_c)
            if nlevel >= (param["ncoarse"] - 3):
                self.smoothing(x_corr_c, res_c, param["Npre"])
            else:
                self.V_cycle(x_corr_c, res_c, param, nlevel + 1)
        else:  # small coarse problem: every rank solves all of it
            x_full = self.k.coarse_linear(self.gather_full(res_c), param, nlevel)
            x_corr_c = self._take_planes(x_full, n_c, n_c // self.world if n_c % self.world == 0 else None)
            del x_full
        del res_c
        if x_corr_c.shape[0] - 2 != (x.shape[0] - 2) // 2:
            raise RuntimeError("inconsistent slab multigrid geometry")
        if self.distributed(n_c):
            self.add_prolongation(x, x_corr_c)
        else:  # ghosts already taken from the full grid
            self.k.prolongation_add(x, x_corr_c)
        del x_corr_c
        self.smoothing(x, b, param["Npost"])

    def _take_planes(self, full, n_c: int, nxc):
        """My planes (+ one periodic ghost each side) of a full coarse grid, as a level array."""
        if nxc is None or nxc < 1:
            raise ValueError(f"slab multigrid: {n_c} planes cannot be split over {self.world} ranks")
        lo = self.rank * nxc
        idx = self.torch.arange(lo - 1, lo + nxc + 1, device=full.device) % n_c
        return full.index_select(0, idx).contiguous()

    def linear(self, x, b, param):
        """multigrid.linear (multigrid.py:23) on level arrays x, b ([nx+2, n, n]); in place on x."""
        THEORY = param["theory"].casefold()
        mond_pot = (not param["compute_additional_field"]) and "mond" == THEORY
        if ("tolerance" not in param) or (param["nsteps"] % 3) == 0:
            logging.info("Compute Truncation error")
            tolerance = param["epsrel"] * self.truncation_error(x)
            if mond_pot:
                param["tolerance_mond"] = tolerance
            else:
                param["tolerance"] = tolerance
        tolerance = param["tolerance_mond"] if mond_pot else param["tolerance"]
        residual_err = 1e30
        while residual_err > tolerance:
            self.V_cycle(x, b, param)
            residual_error_tmp = self.residual_error(x, b)
            logging.info(f"{residual_error_tmp=} {tolerance=}")
            if residual_error_tmp < tolerance or residual_err / residual_error_tmp < 2:
                break
            residual_err = residual_error_tmp
        return x

    def mond_rhs(self, pot, out, g0, kind: int, alpha) -> None:
        self.exchange(pot)
        self.k.mond_rhs(pot, out, g0, kind, alpha)

    # ---- f(R): Full Approximation Scheme on slabs (multigrid.py:88-138, 521-579; cubic.py / quartic.py) ----------
    def _fr(self, param):
        n_fr = int(param["fR_n"])
        if n_fr not in (1, 2):
            raise NotImplementedError(f"Only f(R) with n = 1 and 2, currently {param['fR_n']=}")
        return n_fr, float(np.float32(param["fR_q"]))

    def fr_operator(self, x, b, param, rhs=None):
        """operator(x, b, q) or, with rhs, residual_with_rhs = rhs - operator (cubic.py:23, 90)."""
        n_fr, q = self._fr(param)
        self.exchange(x)
        out = self.torch.empty_like(x)
        self.k.fr_operator(n_fr, x, b, q, rhs, out)
        return out

    def fr_smoothing(self, x, b, nsweeps: int, param, rhs=None) -> None:
        n_fr, q = self._fr(param)
        x_lo = self.rank * (x.shape[0] - 2)
        for _ in range(int(nsweeps)):
            for colour in (1, 0):
                self.exchange(x)
                self.k.fr_gs(n_fr, x, b, q, rhs, x_lo, colour)

    def restriction(self, x, scale=0.125):
        out = self.new(x.shape[1] // 2)
        self.k.restriction(x, scale, out)
        return out

    def fr_residual_error(self, x, b, param) -> np.float32:
        n_fr, q = self._fr(param)
        self.exchange(x)
        return np.float32(np.sqrt(self.all_sum(self.k.fr_residual_sumsq(n_fr, x, b, q))))

    def fr_truncation_error(self, x, b, param) -> np.float32:
        """|| 4 R(L x) - L(R x, R b) || (cubic.py:1021)."""
        RLx = self.restriction(self.fr_operator(x, b, param))
        LRx = self.fr_operator(self.restriction(x), self.restriction(b), param)
        return np.float32(np.sqrt(self.all_sum(self.k.diff_sumsq(RLx[1:-1], LRx[1:-1], 4.0))))

    def V_cycle_FAS(self, x, b, param, nlevel: int = 0, rhs=None) -> None:
        n = x.shape[1]
        n_c = n // 2
        self.fr_smoothing(x, b, param["Npre"], param, rhs)
        if rhs is None:   # multigrid.py:282: minus_restriction(operator) or restriction(residual_with_rhs)
            res_c = self.restriction(self.fr_operator(x, b, param), -0.125)
        else:
            res_c = self.restriction(self.fr_operator(x, b, param, rhs))
        x_c = self.restriction(x)
        b_c = self.restriction(b)
        if self.distributed(n_c):
            x_corr_c = x_c.clone()
            L_c = self.fr_operator(x_c, b_c, param)
            res_c.mul_(float(multigrid.normalisation_residual(param))).add_(L_c)  # 4 res + L (multigrid.py:549)
            del L_c
            if nlevel >= (param["ncoarse"] - 3):
                self.fr_smoothing(x_corr_c, b_c, param["Npre"], param, res_c)
            else:
                self.V_cycle_FAS(x_corr_c, b_c, param, nlevel + 1, res_c)
            x_corr_c.sub_(x_c)
            self.add_prolongation(x, x_corr_c)
        else:  # small coarse problem: every rank runs the rest of the cycle on all of it
            corr_full = self.k.coarse_fas(self.gather_full(x_c), self.gather_full(b_c), self.gather_full(res_c), param,
                                          nlevel)
            x_corr_c = self._take_planes(corr_full, n_c, n_c // self.world)
            self.k.prolongation_add(x, x_corr_c)
            del corr_full
        del res_c, b_c, x_c, x_corr_c
        self.fr_smoothing(x, b, param["Npost"], param, rhs)

    def FAS(self, x, b, param):
        """multigrid.FAS (multigrid.py:88) on level arrays; in place on x."""
        if ("tolerance_FAS" not in param) or (param["nsteps"] % 3) == 0:
            logging.info("Compute FAS Truncation error")
            param["tolerance_FAS"] = param["epsrel"] * self.fr_truncation_error(x, b, param)
        tolerance = param["tolerance_FAS"]
        residual_err = 1e30
        while residual_err > tolerance:
            self.V_cycle_FAS(x, b, param)
            residual_error_tmp = self.fr_residual_error(x, b, param)
            logging.info(f"{residual_error_tmp=} {tolerance=}")
            if residual_error_tmp < tolerance or residual_err / residual_error_tmp < 2:
                break
            residual_err = residual_error_tmp
        return x

    def fr_initialise_potential(self, b, param):
        """cubic / quartic initialise_potential (cubic.py:217, quartic.py:214): per cell, whole level array."""
        n_fr, q = self._fr(param)
        out = self.torch.empty_like(b)
        self.k.fr_initialise_potential(n_fr, b, q, out)
        return out

