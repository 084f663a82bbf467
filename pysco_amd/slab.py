"""Slab-decomposed PM step across the GPUs of one node (one process per GPU, torch.distributed).

PySCo itself is single-process ("no MPI parallelization"), so this layer has no reference
counterpart; parity is defined against the single-GPU result.  Decomposition: rank r owns the
x-planes [r*n/G, (r+1)*n/G) of every grid and the particles inside them.  Exchanges per step
(RCCL over xGMI with backend "nccl"; "gloo" stages through the host and is used by the CPU tests):

  deposit   1 ghost plane each way, added into the neighbour's edge plane      (send/recv, ring)
  FFT       2 all-to-all transposes per solve ([x_local, y, kz] <-> [x, y_local, kz])
  gradient  H = reach+1 ghost planes of phi each way (3 for the order-5 stencil + TSC gather)
  particles all-to-all of the particles that left the slab after the drift
  scalars   max|a|, max|v| all-reduce(MAX) for dt

The Newtonian FFT path is fully decomposed.  Multigrid / f(R) / MOND / P(k) run with decomposed
particles but a replicated grid solve (all-gather of the density slabs; halo-exchange multigrid is next).
All compute goes through an `ops` provider: `HipOps` (libpmstep, the product path).  The CPU tests
inject their own provider built on the oracle -- nothing in this package falls back to the CPU.
"""
from __future__ import annotations

import ctypes as C
import logging
import os
from typing import List, Optional

import numpy as np

from . import _lib as L
from . import fourier, utils

_log = logging.getLogger(__name__)

_REACH = {2: 1, 3: 1, 5: 2, 7: 3}
_SCHEME = {"cic": L.CIC, "tsc": L.TSC}


def _dist():
    import torch.distributed as dist

    return dist


def active() -> bool:
    """True when running under torch.distributed with more than one rank."""
    try:
        dist = _dist()
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    except Exception:
        return False


# ------------------------------------------------------------------------------------------------
# compute provider (product): thin wrappers over the C ABI


class HipOps:
    def __init__(self, n: int):
        self.torch = L.require_gpu()
        self.n = n
        self.ctx = L.context(n)

    def _window(self, on: bool) -> None:
        """Restrict the tile-range kernels to the tiles of the rank's planes (set by Slab) around a particle call;
        it stays on while prepared tile lists are held (they belong to the window they were made for)."""
        w = getattr(self, "window", None)
        if w is None:
            return
        if on:
            self.ctx.call("pm_set_tile_window", w[0], w[1])
        elif not getattr(self, "_lists_held", False):
            self.ctx.call("pm_set_tile_window", 0, self.n)

    # particles
    def kick_drift_wrap(self, pos, vel, acc, half_dt, dt, x_lo, x_hi):
        """Fused kick + drift + wrap; returns the (sorted) indices of the particles that left [x_lo, x_hi)."""
        torch, n = self.torch, pos.shape[0]
        cap = max(4096, n // 16)
        if getattr(self, "_leave_buf", None) is None or self._leave_buf.numel() < cap:
            self._leave_buf = torch.empty(cap, dtype=torch.int64, device=pos.device)
        count = C.c_int(0)
        self.ctx.call("pm_kick_drift_wrap_slab", L.ptr(pos), L.ptr(vel), L.ptr(acc), float(half_dt), float(dt), n,
                      float(x_lo), float(x_hi), L.ptr(self._leave_buf), cap, C.byref(count))
        if count.value > cap:  # more leavers than the list holds (unsorted / first placement): rescan
            x = pos[:, 0]
            return torch.nonzero((x < x_lo) | (x >= x_hi), as_tuple=False).squeeze(1)
        return torch.sort(self._leave_buf[:count.value]).values

    def khalo_buffer(self, nx_window):
        """The halo buffer of the fused kick + deposit for a window of nx_window planes (pmstep.h:
        pm_khalo_floats_slab), kept between steps."""
        buf = getattr(self, "_khalo_buf", None)
        need = int(L.load_library().pm_khalo_floats_slab(self.n, nx_window))
        if need <= 0:
            return None
        if buf is None or buf.numel() != need:
            buf = self._khalo_buf = self.torch.zeros(need, dtype=self.torch.float32, device="cuda")
        return buf

    def kick_drift_wrap_deposit(self, scheme, pos, vel, acc, half_dt, dt, x_lo, x_hi, nx, x0, khalo=None):
        """Fused first kick + drift + wrap + deposit into [nx, n, n] planes starting at x0 (ghosts included);
        returns (density, sorted leaver indices).  `khalo`: the halo contributions of the window's tiles stay in that
        buffer (pm_kick_drift_wrap_deposit_slab_khalo) for fftp_z_forward to add."""
        torch, n, npart = self.torch, self.n, pos.shape[0]
        cap = max(4096, npart // 16)
        if getattr(self, "_leave_buf", None) is None or self._leave_buf.numel() < cap:
            self._leave_buf = torch.empty(cap, dtype=torch.int64, device=pos.device)
        rho = torch.empty((nx, n, n), dtype=torch.float32, device=pos.device)
        count = C.c_int(0)
        self.ctx.call("pm_set_slab", nx, x0)
        self._window(True)
        try:
            if khalo is not None:
                self.ctx.call("pm_kick_drift_wrap_deposit_slab_khalo", scheme, L.ptr(pos), L.ptr(vel), L.ptr(acc),
                              float(half_dt), float(dt), npart, L.ptr(rho), float(x_lo), float(x_hi),
                              L.ptr(self._leave_buf), cap, C.byref(count), L.ptr(khalo))
            else:
                self.ctx.call("pm_kick_drift_wrap_deposit_slab", scheme, L.ptr(pos), L.ptr(vel), L.ptr(acc), float(half_dt),
                              float(dt), npart, L.ptr(rho), float(x_lo), float(x_hi), L.ptr(self._leave_buf), cap,
                              C.byref(count))
        finally:
            self.ctx.call("pm_set_slab", n, 0)
            self._window(False)
        if count.value > cap:
            x = pos[:, 0]
            leave = torch.nonzero((x < x_lo) | (x >= x_hi), as_tuple=False).squeeze(1)
        else:
            leave = torch.sort(self._leave_buf[:count.value]).values
        return rho, leave

    def deposit(self, scheme, pos, nx, x0):
        n = self.n
        rho = self.torch.empty((nx, n, n), dtype=self.torch.float32, device=pos.device)
        self.ctx.call("pm_set_slab", nx, x0)
        self._window(True)
        try:
            self.ctx.call("pm_deposit", scheme, L.ptr(pos), pos.shape[0], L.ptr(rho), 1)
        finally:
            self.ctx.call("pm_set_slab", n, 0)
            self._window(False)
        return rho

    def interp_kick(self, scheme, force, nx, x0, pos, vel, half_dt):
        out = self.torch.empty_like(pos)
        self.ctx.call("pm_set_slab", nx, x0)
        self._window(True)
        complete = self.lists_complete()
        if not complete:
            self.ctx.call("pm_set_gather_mode", 2)  # chunk / anchor kernel: any particle order through LDS
        try:
            self.ctx.call("pm_interp_kick", scheme, L.ptr(force), L.ptr(pos), pos.shape[0], L.ptr(out),
                          L.ptr(vel) if vel is not None else None, float(half_dt))
        finally:
            self.ctx.call("pm_set_slab", self.n, 0)
            self._window(False)
            if not complete:
                self.ctx.call("pm_set_gather_mode", 1)
        a, v = C.c_float(0), C.c_float(0)
        self.ctx.call("pm_last_maxabs", C.byref(a), C.byref(v))
        return out, float(a.value), float(v.value)

    def gradient_interp_kick(self, order, scheme, phi_ext, x0_phi, pos, vel, half_dt):
        """Gradient + gather + kick from the potential slab (with ghosts) without a force array."""
        out = self.torch.empty_like(pos)
        self._window(True)
        try:
            self.ctx.call("pm_gradient_interp_kick", int(order), scheme, L.ptr(phi_ext), phi_ext.shape[0], int(x0_phi),
                          L.ptr(pos), pos.shape[0], L.ptr(out), L.ptr(vel) if vel is not None else None,
                          float(half_dt))
        finally:
            self._window(False)
        a, v, m = C.c_float(0), C.c_float(0), C.c_double(0)
        self.ctx.call("pm_last_maxabs", C.byref(a), C.byref(v))
        self.ctx.call("pm_last_acc_meanabs", C.byref(m))
        self.last_meanabs = float(m.value) or None
        return out, float(a.value), float(v.value)

    def set_destination_codes(self, codes):
        """Where the next fused kick + deposit records every particle's destination code (uint8 [Np]); None: off."""
        self.ctx.call("pm_set_destination_codes", L.ptr(codes) if codes is not None else None)

    def set_prediction(self, dt_next, acc_scale):
        self.ctx.call("pm_set_predicted_dt", float(np.float32(dt_next)))
        self.ctx.call("pm_set_predicted_acc_scale", float(np.float32(acc_scale)))

    def gradient_interp_kick_rebucket(self, order, scheme, phi_ext, x0_phi, pos, vel, half_dt, codes, pos_out, vel_out,
                                      acc_out, ids=None, ids_out=None, counted=False):
        """gradient_interp_kick writing position, kicked velocity and acceleration out of place in tile order
        (pmstep.h: pm_rebucket_prepare + pm_gradient_interp_kick_rebucket_slab); needs the lists of tile_lists(pos);
        counted: tile_lists(pos, codes) has counted the destinations already."""
        self._window(True)
        try:
            if not counted:
                self.ctx.call("pm_rebucket_prepare", L.ptr(pos), pos.shape[0], L.ptr(codes))
            self.ctx.call("pm_gradient_interp_kick_rebucket_slab", int(order), scheme, L.ptr(phi_ext), phi_ext.shape[0],
                          int(x0_phi), L.ptr(pos), L.ptr(vel), pos.shape[0], float(half_dt), L.ptr(pos_out), L.ptr(vel_out),
                          L.ptr(acc_out), L.ptr(ids), L.ptr(ids_out))
            self._lists_held = False  # (the gather dropped them: they described the old layout)
        finally:
            self._window(False)
        a, v, m = C.c_float(0), C.c_float(0), C.c_double(0)
        self.ctx.call("pm_last_maxabs", C.byref(a), C.byref(v))
        self.ctx.call("pm_last_acc_meanabs", C.byref(m))
        self.last_meanabs = float(m.value) or None
        return float(a.value), float(v.value)

    def sorted_stream_resized(self, pos):
        """The migration changed the particle count of the stream the last re-bucketing gather wrote: keep its tile
        offsets on file (pmstep.h: pm_sorted_stream_resized)."""
        self._window(True)
        try:
            self.ctx.call("pm_sorted_stream_resized", L.ptr(pos), pos.shape[0])
        finally:
            self._window(False)

    def tile_lists(self, pos, codes=None):
        """Tile ranges + straggler lists for these positions (pmstep.h: pm_tile_lists); None drops them.  `codes`:
        the particles' destination codes -- the same pass then also counts them for the re-bucketing gather
        (pm_tile_lists_codes)."""
        if pos is None:
            self.ctx.call("pm_tile_lists", None, 0)
            self._lists_held = False
            self._window(False)
        else:
            self._window(True)
            if codes is not None:
                self.ctx.call("pm_tile_lists_codes", L.ptr(pos), pos.shape[0], L.ptr(codes))
            else:
                self.ctx.call("pm_tile_lists", L.ptr(pos), pos.shape[0])
            self._lists_held = True

    def lists_complete(self) -> bool:
        from . import utils

        return utils.straggler_fraction(self.n)[1]

    def max_abs(self, x) -> float:
        if x.numel() == 0:
            return 0.0
        out = C.c_float(0)
        self.ctx.call("pm_max_abs", L.ptr(x), x.numel(), C.byref(out))
        return float(out.value)

    def axpy(self, y, x, a):
        self.ctx.call("pm_axpy", L.ptr(y), L.ptr(x), float(a), y.numel())

    # grids
    def linear_operator_inplace(self, x, f1, f2):
        self.ctx.call("pm_linear_operator", L.ptr(x), L.ptr(x), float(f1), float(f2), x.numel())

    def scale_inplace(self, x, a):
        self.ctx.call("pm_scale", L.ptr(x), float(a), x.numel())

    def fft2d_r2c(self, x):
        nb, n = x.shape[0], self.n
        out = self.torch.empty((nb, n, n // 2 + 1), dtype=self.torch.complex64, device=x.device)
        self.ctx.call("pm_fft2d_r2c", L.ptr(x), L.ptr(out), nb)
        return out

    def fft2d_c2r(self, xk):
        nb, n = xk.shape[0], self.n
        out = self.torch.empty((nb, n, n), dtype=self.torch.float32, device=xk.device)
        self.ctx.call("pm_fft2d_c2r", L.ptr(xk), L.ptr(out), nb)
        return out

    def fft1d_x(self, data, inverse: bool):
        ncol = data.numel() // self.n
        self.ctx.call("pm_fft1d_x", L.ptr(data), ncol, int(inverse))

    def greens(self, kind, p, data, j0, nj, scale=None):
        k2, s, c0 = fourier.green_tables(self.n, kind)
        if scale is not None:  # rhs scaling folded into the prefactor (see solver.pm)
            c0 = float(np.float32(c0) * np.float32(scale))
        self.ctx.call("pm_greens_slab", kind, int(p), L.ptr(data), L.hptr(k2), L.hptr(s), c0, j0, nj)

    # LDS-FFT passes of csrc/pm_fftpoisson.hip on a slab (no rocFFT, no stand-alone transposes)
    def has_fftp(self, nx, nky) -> bool:
        return 32 <= self.n <= 1024 and nx % 8 == 0 and nky % 8 == 0

    def fftp_forward(self, rhs):
        n, nx = self.n, rhs.shape[0]
        out = self.torch.empty((n, n // 2 + 1, nx), dtype=self.torch.complex64, device=rhs.device)
        self.ctx.call("pm_fftp_forward", L.ptr(rhs), nx, L.ptr(out))
        return out

    def fftp_xgreen(self, seg, kind, p, nx, nky, ky0, scale=None):
        k2, s, c0 = fourier.green_tables(self.n, kind)
        if scale is not None:
            c0 = float(np.float32(c0) * np.float32(scale))
        self.ctx.call("pm_fftp_xgreen", kind, int(p), L.ptr(seg), nx, nky, ky0, L.hptr(k2), L.hptr(s), c0)

    # stage-wise forms (pipelined against the all-to-alls)
    def fftp_z_forward(self, rhs, khalo=None):
        if khalo is not None:
            self.ctx.call("pm_fftp_z_forward_khalo", L.ptr(rhs), rhs.shape[0], L.ptr(khalo))
        else:
            self.ctx.call("pm_fftp_z_forward", L.ptr(rhs), rhs.shape[0])

    def fftp_y_forward(self, nx, kz0, nkz):
        out = self.torch.empty((self.n, nkz, nx), dtype=self.torch.complex64, device="cuda")
        self.ctx.call("pm_fftp_y_forward", nx, kz0, nkz, L.ptr(out))
        return out

    def fftp_xgreen_kz(self, seg, kind, p, nx, nky, ky0, kz0, nkz, scale=None):
        k2, s, c0 = fourier.green_tables(self.n, kind)
        if scale is not None:
            c0 = float(np.float32(c0) * np.float32(scale))
        self.ctx.call("pm_fftp_xgreen_kz", kind, int(p), L.ptr(seg), nx, nky, ky0, kz0, nkz, L.hptr(k2), L.hptr(s), c0)

    def fftp_y_inverse(self, c, nx, kz0, nkz):
        self.ctx.call("pm_fftp_y_inverse", L.ptr(c), nx, kz0, nkz)

    def fftp_z_inverse(self, nx):
        n = self.n
        out = self.torch.empty((nx, n, n), dtype=self.torch.float32, device="cuda")
        self.ctx.call("pm_fftp_z_inverse", nx, L.ptr(out))
        return out

    def fftp_inverse(self, c, nx):
        n = self.n
        out = self.torch.empty((nx, n, n), dtype=self.torch.float32, device=c.device)
        self.ctx.call("pm_fftp_inverse", L.ptr(c), nx, L.ptr(out))
        return out

    def gradient_fr(self, order, fr_n, f, phi_ext, u_ext, x0_phi, nx_force, x0_force):
        """f(R) force (mesh.derivative_fR) on nx_force planes from potential / scalaron slabs with ghosts."""
        n = self.n
        out = self.torch.empty((nx_force, n, n, 3), dtype=self.torch.float32, device=phi_ext.device)
        self.ctx.call("pm_gradient_fr_slab", int(order), int(fr_n), float(np.float32(f)), L.ptr(phi_ext), L.ptr(u_ext),
                      phi_ext.shape[0], int(x0_phi), L.ptr(out), nx_force, int(x0_force))
        return out

    def level_kernels(self):
        """Level operations of the slab multigrid (slab_multigrid.HipLevelKernels)."""
        from .slab_multigrid import HipLevelKernels

        return HipLevelKernels()

    def gradient(self, order, phi_ext, x0_phi, nx_force, x0_force):
        n = self.n
        out = self.torch.empty((nx_force, n, n, 3), dtype=self.torch.float32, device=phi_ext.device)
        self.ctx.call("pm_gradient_slab", int(order), L.ptr(phi_ext), phi_ext.shape[0], x0_phi, L.ptr(out), nx_force,
                      x0_force)
        return out


# ------------------------------------------------------------------------------------------------
# communication helpers (device tensors over nccl/RCCL; host staging when the backend is gloo)


def _stage(t):
    return t.cpu() if (t.is_cuda and _dist().get_backend() == "gloo") else t


def _unstage(t, like):
    return t.to(like.device) if t.device != like.device else t


# bytes this rank handed to the all-to-alls / the ring exchanges for OTHER ranks (bench.py reports them per step against
# the 7 x 153 GB/s of a GPU's xGMI links)
comm_stats = {"a2a_bytes": 0, "a2a_calls": 0, "ring_bytes": 0, "ring_calls": 0}


def _count_a2a(t) -> None:
    world = _dist().get_world_size()
    comm_stats["a2a_bytes"] += t.numel() * t.element_size() * (world - 1) // world
    comm_stats["a2a_calls"] += 1


def all_to_all(send, out_splits: Optional[List[int]] = None, in_splits: Optional[List[int]] = None):
    """all_to_all_single along dim 0; equal splits when both lists are None."""
    import torch

    dist = _dist()
    s = _stage(send.contiguous())
    _count_a2a(s)
    if out_splits is None:
        out = torch.empty_like(s)
    else:
        out = torch.empty((int(sum(out_splits)),) + tuple(s.shape[1:]), dtype=s.dtype, device=s.device)
    if s.is_complex():  # gloo has no complex support; reinterpret as float pairs
        dist.all_to_all_single(torch.view_as_real(out), torch.view_as_real(s), out_splits, in_splits)
    else:
        dist.all_to_all_single(out, s, out_splits, in_splits)
    return _unstage(out, send)


class _AllToAllHandle:
    """An all_to_all in flight (equal splits along dim 0): wait() gives the received tensor."""

    def __init__(self, send):
        import torch

        dist = _dist()
        self.like = send
        self.src = _stage(send.contiguous())  # kept alive until wait()
        _count_a2a(self.src)
        self.out = torch.empty_like(self.src)
        a, b = (torch.view_as_real(self.out), torch.view_as_real(self.src)) if self.src.is_complex() else (self.out, self.src)
        self.work = dist.all_to_all_single(a, b, async_op=True)

    def wait(self):
        self.work.wait()  # nccl: the current stream waits, not the host
        out, self.src = _unstage(self.out, self.like), None
        return out


def exchange_planes(to_left, to_right):
    """Send `to_left` to rank-1 and `to_right` to rank+1 (periodic); returns (from_left, from_right)."""
    import torch

    dist = _dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    left, right = (rank - 1) % world, (rank + 1) % world
    sl, sr = _stage(to_left.contiguous()), _stage(to_right.contiguous())
    comm_stats["ring_bytes"] += sl.numel() * sl.element_size() + sr.numel() * sr.element_size()
    comm_stats["ring_calls"] += 1
    from_right, from_left = torch.empty_like(sl), torch.empty_like(sr)
    # order matters when left == right (2 ranks): a rank's first send (to its left) is its peer's
    # "from my right", so receives are posted in the order [from_right, from_left]
    ops = [dist.P2POp(dist.isend, sl, left), dist.P2POp(dist.isend, sr, right),
           dist.P2POp(dist.irecv, from_right, right), dist.P2POp(dist.irecv, from_left, left)]
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    return _unstage(from_left, to_left), _unstage(from_right, to_left)


def all_reduce_max(values: List[float], device) -> List[float]:
    import torch

    dist = _dist()
    t = torch.tensor(values, dtype=torch.float32, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t.tolist()]


# ------------------------------------------------------------------------------------------------


_FUSE_SLAB_DEPOSIT = True  # False: kick_drift_wrap, migration, then a separate deposit
_PIPELINE_FFT = True    # False: the slab FFT exchanges the whole spectrum in one all-to-all each way
_SLAB_MULTIGRID = True  # False: multigrid / MOND under slabs all-gather the density and solve replicated
_SLAB_REBUCKET = True   # False: the gather writes in place and the particle stream of a rank is never re-sorted
_SLAB_KHALO = os.environ.get("PMSTEP_SLAB_KHALO", "1") != "0"  # halo buffer of the fused kick + deposit (Newton FFT steps)


class Slab:
    """Geometry of this rank's slab and the distributed PM step."""

    def __init__(self, n: int, ops=None):
        dist = _dist()
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.n = n
        if n % self.world or (n // self.world) < 8:
            raise ValueError(f"slab decomposition needs n/G >= 8 planes per rank (n={n}, G={self.world})")
        self.nx = n // self.world
        self.x_lo = self.rank * self.nx
        self.ops = ops if ops is not None else HipOps(n)
        if isinstance(self.ops, HipOps) and self.nx % 8 == 0:
            self.ops.window = (self.x_lo, self.nx)  # all particles of this rank live in these planes (after migrate)
        self.carry = ()  # extra per-particle arrays ([Np, ...]) that migrate with the particles (e.g. IDs)

    # ---- particles ------------------------------------------------------------------------------
    def owner_of(self, pos):
        """Rank that owns each particle (by its x coordinate in [0,1))."""
        return (pos[:, 0] * self.world).long().clamp_(0, self.world - 1)

    def migrate(self, *arrays):
        """Move particles to the rank that owns their x-plane.  arrays[0] is the position [Np,3];
        all arrays are permuted alike.  Stayers keep their order, arrivals are appended."""
        import torch

        pos = arrays[0]
        owner = self.owner_of(pos)
        leaving = owner != self.rank
        n_leave = int(leaving.sum())
        counts_out = torch.zeros(self.world, dtype=torch.int64, device=pos.device)
        if n_leave:
            idx = torch.nonzero(leaving, as_tuple=False).squeeze(1)
            dest = owner[idx]
            order = torch.argsort(dest, stable=True)
            idx = idx[order]
            counts_out = torch.bincount(dest, minlength=self.world)
        counts_in = all_to_all(counts_out)
        out_splits = [int(c) for c in counts_out.tolist()]
        in_splits = [int(c) for c in counts_in.tolist()]
        any_move = sum(in_splits) + sum(out_splits)
        result = []
        keep = ~leaving if n_leave else None
        for a in arrays:
            send = a[idx] if n_leave else a[:0]
            recv = all_to_all(send, in_splits, out_splits)
            if not any_move:
                result.append(a)
            else:
                stay = a[keep] if n_leave else a
                result.append(torch.cat([stay, recv], dim=0) if recv.shape[0] else stay.contiguous())
        return tuple(result)

    # ---- capacity-backed particle storage + O(leavers) migration ----------------------------------
    def _adopt(self, position, velocity):
        """Make (position, velocity, *carry) views of buffers with head-room so that arrivals can be
        appended without reallocating; no-op when the caller hands back last step's views."""
        import torch

        st = getattr(self, "_store", None)
        n = position.shape[0]
        if (st is not None and st["n"] == n and st["bufs"][0].data_ptr() == position.data_ptr()
                and st["bufs"][1].data_ptr() == velocity.data_ptr()):
            return
        cap = int(n * 1.03) + 4096
        codes = self._rebucket_on(position)
        placed = self._place_arrays(position, cap) if codes else None
        bufs = []
        for k, a in enumerate((position, velocity) + tuple(self.carry)):
            if placed is not None and k < 2:
                b = placed["live"][k]
            else:
                b = torch.empty((cap,) + tuple(a.shape[1:]), dtype=a.dtype, device=a.device)
            b[:n] = a
            bufs.append(b)
        if codes:  # destination codes of the incremental re-sort: one more per-particle array that migrates (13 = stays)
            bufs.append(torch.full((cap,), 13, dtype=torch.uint8, device=position.device))
        self._store = {"bufs": bufs, "n": n, "cap": cap, "codes": codes, "out": placed["out"] if placed else None,
                       "acc": placed["acc"] if placed else None}

    def _place_arrays(self, position, cap):
        """Every rank runs the placement search of the single-GPU loop (pysco_amd.main.rank_triples; DESIGN.md section
        5) on its own capacity-backed arrays: the streaming particle kernels run up to 15 % faster or slower with the
        physical memory an allocation happened to land in, and on G ranks the slowest draw sets the step.  Candidates
        of the store's capacity are allocated, the fused kick's own traffic is timed over every pair of them, the
        best triple becomes (position, velocity, acceleration) of the live set, the best disjoint one the set the
        re-bucketing gather writes; the rest is freed.  Large ranks only (a probe pass must be worth timing), and
        only where the memory is there; PMSTEP_SLAB_PLACEMENT=0 turns it off."""
        import torch

        from . import main as pm_main

        n = position.shape[0]
        if (os.environ.get("PMSTEP_SLAB_PLACEMENT", "1") == "0" or n < 2**22 or position.dtype != torch.float32
                or not position.is_cuda):
            return None
        free_bytes, _ = torch.cuda.mem_get_info(position.device)
        each = cap * 12
        n_cand = int(min(10, free_bytes // (each * max(1, self.world_on_device())) - 2))
        if n_cand < 6:
            return None
        cand = []
        try:
            for _ in range(n_cand):
                cand.append(torch.zeros((cap, 3), dtype=torch.float32, device=position.device))
        except torch.OutOfMemoryError:
            cand = cand[:-1]
        if len(cand) < 6:
            return None
        a, b, info = pm_main.rank_triples(cand, n & ~3)
        self.placement_probe_ms = info
        placed = {"live": [cand[a[0]], cand[a[1]]], "out": [cand[b[0]], cand[b[1]]], "acc": [cand[a[2]], cand[b[2]]]}
        del cand
        torch.cuda.empty_cache()
        return placed

    def world_on_device(self) -> int:
        """Ranks that share this rank's GPU (1 on a real node; all of them in a one-GPU rehearsal)."""
        import torch

        return max(1, -(-self.world // max(1, torch.cuda.device_count())))

    def _rebucket_on(self, position) -> bool:
        return bool(_SLAB_REBUCKET and hasattr(self.ops, "gradient_interp_kick_rebucket") and self.n >= 16
                    and getattr(position, "is_cuda", False) and 0 < position.shape[0] < 2**32)

    def _views(self):
        st = self._store
        v = [b[:st["n"]] for b in st["bufs"]]
        if st.get("codes"):
            self._codes, v = v[-1], v[:-1]
        else:
            self._codes = None
        self.carry = tuple(v[2:])
        return v[0], v[1]

    def _grow(self, need):
        import torch

        st = self._store
        cap = int(need * 1.05) + 4096
        new = []
        for b in st["bufs"]:
            nb = torch.empty((cap,) + tuple(b.shape[1:]), dtype=b.dtype, device=b.device)
            nb[:st["n"]] = b[:st["n"]]
            new.append(nb)
        st["bufs"], st["cap"], st["out"], st["acc"] = new, cap, None, None

    def _migrate_leavers(self, leave):
        """Send the particles at (sorted) indices `leave` to their owners and file the arrivals into the
        holes they leave; surplus holes are filled from the tail, surplus arrivals are appended.  Touches
        O(leavers + arrivals) elements -- never the whole arrays."""
        import torch

        st = self._store
        bufs, n = st["bufs"], st["n"]
        nl = int(leave.numel())
        dev = bufs[0].device
        counts_out = torch.zeros(self.world, dtype=torch.int64, device=dev)
        by_dest = leave
        if nl:
            dest = self.owner_of(bufs[0][leave])
            order = torch.argsort(dest, stable=True)
            by_dest = leave[order]
            counts_out = torch.bincount(dest, minlength=self.world)
        counts_in = all_to_all(counts_out)
        out_splits = [int(c) for c in counts_out.tolist()]
        in_splits = [int(c) for c in counts_in.tolist()]
        na = sum(in_splits)
        arrivals = [all_to_all(b[by_dest], in_splits, out_splits) for b in bufs]
        if nl == 0 and na == 0:
            return
        if na >= nl:
            extra = na - nl
            if n + extra > st["cap"]:
                self._grow(n + extra)
                bufs = st["bufs"]
            for b, a in zip(bufs, arrivals):
                if nl:
                    b[leave] = a[:nl]
                if extra:
                    b[n:n + extra] = a[nl:]
            st["n"] = n + extra
        else:
            n_new = n - (nl - na)
            rem = leave[na:]                                   # holes no arrival fills (ascending)
            tail = torch.arange(n_new, n, device=dev)
            movers = tail[~torch.isin(tail, rem)]              # live particles beyond the new end
            targets = rem[rem < n_new]                         # holes below it: same count
            for b, a in zip(bufs, arrivals):
                if na:
                    b[leave[:na]] = a
                if movers.numel():
                    b[targets] = b[movers]
            st["n"] = n_new

    # ---- the PM solve (solver.py:30-215, Newtonian FFT branch) -------------------------------------
    def pm(self, position, param, potential=None, additional_field=None, tables=(), kick=None):
        import torch

        n, ops = self.n, self.ops
        if n != 2 ** int(param["ncoarse"]):
            raise ValueError("slab was built for a different grid size")
        MASS_SCHEME = param["mass_scheme"].casefold()
        THEORY = param["theory"].casefold()
        if MASS_SCHEME == "cic":
            param["MAS_index"] = 2
        elif MASS_SCHEME == "tsc":
            param["MAS_index"] = 3
        else:
            raise NotImplementedError(f"{param['mass_scheme']=}, should be 'CIC' or 'TSC'")
        if THEORY not in ("newton", "parametrized", "fr", "mond"):
            raise NotImplementedError(f"{param['theory']=}, should be 'newton', 'fr', 'parametrized' or 'mond'")
        LINEAR_NEWTON_SOLVER = param["linear_newton_solver"].casefold()
        SAVE = param["save_power_spectrum"].casefold()
        if SAVE not in ("yes", "z_out", "no"):
            raise NotImplementedError(f"SAVE_POWER_SPECTRUM={SAVE!r}, should be 'yes', 'z_out' or 'no'")
        save_pk = SAVE == "yes" or (SAVE == "z_out" and bool(param["write_snapshot"]))
        scheme = _SCHEME[MASS_SCHEME]
        nx = self.nx
        decomposed = THEORY == "newton" and LINEAR_NEWTON_SOLVER in ("fft", "fft_7pt") and not save_pk
        # Newton + multigrid and QUMOND: halo-exchange multigrid / slab FFT (HIP provider, >= 4 planes per rank)
        grid_solvers = (not decomposed and not save_pk and _SLAB_MULTIGRID and hasattr(ops, "level_kernels")
                        and n >= 32 and nx >= 4 and nx % 2 == 0
                        and ((THEORY == "newton" and LINEAR_NEWTON_SOLVER == "multigrid")
                             or (THEORY == "mond" and LINEAR_NEWTON_SOLVER in ("multigrid", "fft_7pt"))
                             or (THEORY == "fr" and LINEAR_NEWTON_SOLVER in ("multigrid", "fft", "fft_7pt"))))
        if not decomposed and not grid_solvers:
            return self._pm_replicated_grid(position, param, potential, additional_field, tables, kick, scheme)
        param["save_pk"] = False
        param["parametrized_mu_z"] = np.float32(1)
        # (with a second kick to apply and the rank's own arrays: the gather will re-bucket, and the scan that makes the
        # lists counts the destinations in the same pass)
        codes = self._codes_for(position, kick)
        self._codes_counted = codes is not None
        if grid_solvers:
            lists = getattr(ops, "tile_lists", None)
            if lists is not None:
                lists(position, codes) if codes is not None else lists(position)
            try:
                return self._pm_decomposed_grid_solvers(position, param, potential, additional_field, tables, kick,
                                                        scheme)
            finally:
                if lists is not None:
                    lists(None)

        lists = getattr(ops, "tile_lists", None)
        if lists is not None:
            lists(position, codes) if codes is not None else lists(position)
        try:
            return self._pm_decomposed(position, param, scheme, kick)
        finally:
            if lists is not None:
                lists(None)

    def _codes_for(self, position, kick):
        """The destination codes of `position` if the gather of this pm() can re-bucket: the rank's own capacity-backed
        arrays with their codes, and a kick to apply."""
        st = getattr(self, "_store", None)
        if (kick is None or kick[0] is None or st is None or not st.get("codes") or getattr(self, "_codes", None) is None
                or position.shape[0] != st["n"] or st["n"] == 0 or position.data_ptr() != st["bufs"][0].data_ptr()
                or kick[0].data_ptr() != st["bufs"][1].data_ptr() or self._codes.shape[0] != st["n"]):
            return None
        return self._codes

    def _khalo_applies(self, param) -> bool:
        """Will the pm() of this step be the Newtonian slab FFT with the pipelined LDS passes (the one consumer of the
        halo buffer on a slab)?  The same tests as pm() / poisson_fft() make."""
        ops = self.ops
        if not (_SLAB_KHALO and _PIPELINE_FFT and hasattr(ops, "khalo_buffer") and hasattr(ops, "fftp_y_forward")):
            return False
        save = param["save_power_spectrum"].casefold()
        save_pk = save == "yes" or (save == "z_out" and bool(param["write_snapshot"]))
        return (param["theory"].casefold() == "newton" and param["linear_newton_solver"].casefold() in ("fft", "fft_7pt")
                and not save_pk and bool(getattr(ops, "has_fftp", None)) and ops.has_fftp(self.nx, self.n // self.world))

    def _deposit_folded(self, position, scheme, accept_khalo=False):
        """Density on the owned planes [nx, n, n] (a view into the deposit's [nx + 2W] array): deposit with W
        ghost planes each side (1, or what the fused kick + deposit of integrate() used), ghosts folded into
        the neighbours.  Not normalised."""
        nx, ops = self.nx, self.ops
        pre = getattr(self, "_predeposited", None)
        self._predeposited = None
        self._khalo_live = None
        if pre is not None:
            rho_ext, W, khalo = pre
            if khalo is not None and not accept_khalo:
                raise RuntimeError("the fused kick + deposit left a halo buffer for a solve that cannot consume it")
            self._khalo_live = khalo  # (what the window's tiles add to each other: consumed by fftp_z_forward)
        else:
            rho_ext, W = ops.deposit(scheme, position, nx + 2, self.x_lo - 1), 1
        from_left, from_right = exchange_planes(rho_ext[:W], rho_ext[nx + W:])
        rho_ext[W:2 * W] += from_left             # left neighbour's right ghosts are my first planes
        rho_ext[nx:nx + W] += from_right          # right neighbour's left ghosts are my last planes
        return rho_ext[W:nx + W]                  # contiguous view of the owned planes

    def _deposit_owned(self, position, param, scheme, accept_khalo=False):
        """The same, normalised to unit mean (solver.py:113)."""
        n, ops = self.n, self.ops
        rho = self._deposit_folded(position, scheme, accept_khalo)
        if n**3 != param["npart"]:
            ops.scale_inplace(rho, np.float32(n**3 / param["npart"]))
            if self._khalo_live is not None:
                ops.scale_inplace(self._khalo_live, np.float32(n**3 / param["npart"]))
        return rho

    def _gather(self, phi, position, param, scheme, kick):
        """Acceleration at the particles from the owned planes of the potential (+ second kick, maxima)."""
        import torch

        n, nx, ops = self.n, self.nx, self.ops
        # phi ghosts: gradient reach + 1 plane for the TSC/CIC gather of the force
        order = param["gradient_stencil_order"]
        if order not in _REACH:
            raise ValueError(f"gradient_order={order}, should be 2, 3, 5 or 7")
        H = _REACH[order] + 1
        from_left, from_right = exchange_planes(phi[:H], phi[nx - H:])
        phi_ext = torch.cat([from_left, phi, from_right], dim=0)
        vel, half_dt = (kick if kick is not None else (None, 0.0))
        fused = hasattr(ops, "gradient_interp_kick") and n >= 16
        if fused and hasattr(ops, "lists_complete") and not ops.lists_complete():
            fused = False  # more stragglers than the lists hold: force grid + order-independent gather
        st = getattr(self, "_store", None)
        if fused and self._codes_for(position, kick) is not None:
            # Incremental re-sort (on a slab): the gather writes position, kicked velocity and
            # acceleration out of place, every particle into the range of the tile the fused kick expected it in after
            # the next drift (codes recorded before the migration, which carried them along), so a rank's stream
            # stays tile-sorted without ever being sorted; the two sets of capacity-backed arrays trade places.
            npart, cap = st["n"], st["cap"]
            if st["out"] is None:
                st["out"] = [torch.empty_like(st["bufs"][0]), torch.empty_like(st["bufs"][1])]
            if st.get("acc"):  # (placed with the sets: the one that goes with the set being written)
                acc = st["acc"][1][:npart]
                st["acc"].reverse()
            else:
                acc = torch.empty((npart, 3), dtype=torch.float32, device=position.device)
            ids = ids_out = None
            if self.carry:
                ids = torch.arange(npart, dtype=torch.int32, device=position.device)
                ids_out = torch.empty_like(ids)
            ma, mv = ops.gradient_interp_kick_rebucket(order, scheme, phi_ext, self.x_lo - H, position, vel, half_dt,
                                                       self._codes, st["out"][0][:npart], st["out"][1][:npart], acc,
                                                       ids, ids_out, counted=getattr(self, "_codes_counted", False))
            old = [st["bufs"][0], st["bufs"][1]]
            st["bufs"][0], st["bufs"][1], st["out"] = st["out"][0], st["out"][1], old
            if self.carry:  # the caller's per-particle arrays follow the permutation
                perm = ids_out.long()
                for b in st["bufs"][2:2 + len(self.carry)]:
                    b[:npart] = b[:npart][perm]
            self._rebucketed = True
        elif fused:  # the gather differentiates its own LDS tile
            acc, ma, mv = ops.gradient_interp_kick(order, scheme, phi_ext, self.x_lo - H, position, vel, half_dt)
        else:
            force = ops.gradient(order, phi_ext, self.x_lo - H, nx + 2, self.x_lo - 1)
            acc, ma, mv = ops.interp_kick(scheme, force, nx + 2, self.x_lo - 1, position, vel, half_dt)
            del force
        del phi_ext
        self.local_max = (ma, mv)
        return acc

    def _pm_decomposed(self, position, param, scheme, kick):
        import torch

        rho = self._deposit_owned(position, param, scheme, accept_khalo=True)
        khalo, self._khalo_live = self._khalo_live, None
        param["compute_additional_field"] = False
        # solver.rhs_poisson's f1*(rho - 1) is folded into the Green's function (k = 0 is discarded)
        f1 = np.float32(1.5 * param["aexp"] * param["Om_m"] * param["parametrized_mu_z"])
        phi = self.poisson_fft(rho, param, f1, khalo=khalo)
        del rho
        acc = self._gather(phi, position, param, scheme, kick)
        empty = torch.empty(0, dtype=torch.float32, device=position.device)
        return acc, phi, empty

    def _pm_decomposed_grid_solvers(self, position, param, potential, additional_field, tables, kick, scheme):
        """Newton with the multigrid solver and QUMOND (multigrid or fft_7pt) on slabs: solver.py:113-215 with
        every grid operation decomposed -- halo-exchange multigrid (slab_multigrid.py), the slab FFT, the MOND
        source stencil on a slab.  Potential and extra field are returned (and expected back for the warm
        start, solver.py:273-280) as the rank's owned planes [nx, n, n]."""
        import torch

        from .slab_multigrid import SlabMultigrid

        n, nx, ops = self.n, self.nx, self.ops
        mg = getattr(self, "_mg", None) or SlabMultigrid(self, ops.level_kernels())
        self._mg = mg
        THEORY = param["theory"].casefold()
        SOLVER = param["linear_newton_solver"].casefold()
        dev = position.device

        def owned(t):  # [nx, n, n] planes of a field handed back by the caller, or None
            if t is None or t.numel() == 0:
                return None
            if tuple(t.shape) == (n, n, n):  # a full grid from an earlier replicated solve
                return t[self.x_lo:self.x_lo + nx]
            if tuple(t.shape) != (nx, n, n):
                raise ValueError(f"slab solve expects [nx, n, n] = {(nx, n, n)} fields, got {tuple(t.shape)}")
            return t

        def solve(x0, rhs_lvl, warm_scale):
            """One Poisson solve on a level array rhs_lvl ([nx+2, n, n]); returns the solution level array."""
            if SOLVER == "multigrid":
                if x0 is None:
                    x = mg.initialise_potential(rhs_lvl)
                else:
                    x = mg.extend(x0)
                    if warm_scale is not None:
                        ops.scale_inplace(x, warm_scale)
                return mg.linear(x, rhs_lvl, param)
            sol = self.poisson_fft(rhs_lvl[1:nx + 1], param)  # fft / fft_7pt
            return mg.extend(sol)

        rho = self._deposit_owned(position, param, scheme)
        dens = mg.extend(rho)
        del rho
        f1 = np.float32(1.5 * param["aexp"] * param["Om_m"] * param["parametrized_mu_z"])
        param["compute_additional_field"] = True
        af = None
        if THEORY == "mond":  # solver.py:360-374: Newtonian potential first
            if SOLVER not in ("multigrid", "fft_7pt"):
                raise NotImplementedError(f"{param['linear_newton_solver']=}, should be 'multigrid' or 'fft_7pt'")
            ops.linear_operator_inplace(dens, f1, -f1)
            af = solve(owned(additional_field), dens, None)
        if THEORY == "fr":  # solver.py:326-359: scalaron u by FAS on slabs
            from . import solver as _solver

            Rbar = 3 * param["Om_m"] * param["aexp"] ** (-3) + 12 * param["Om_lambda"]
            Rbar0 = 3 * param["Om_m"] + 12 * param["Om_lambda"]
            fR_a = -param["aexp"] ** 2 * ((Rbar0 / Rbar) ** (param["fR_n"] + 1)) * 10.0 ** (-param["fR_logfR0"])
            c2 = (_solver._C_LIGHT * 1e-3 * param["unit_t"] / (param["unit_l"] * param["aexp"])) ** 2
            g1 = np.float32(param["aexp"] * param["Om_m"] / (c2 * 6)) / (-fR_a)
            g2 = np.float32(Rbar / 3 * param["aexp"] ** 4 - param["Om_m"] * param["aexp"]) / (6 * c2) / (-fR_a)
            dens_term = torch.empty_like(dens)
            mg.k.linear_operator(dens_term, dens, g1, g2)
            param["fR_q"] = np.float32(-param["aexp"] ** 4 * Rbar / (18 * c2)) / (-fR_a)
            u0 = owned(additional_field)
            af = mg.fr_initialise_potential(dens_term, param) if u0 is None else mg.extend(u0)
            af = mg.FAS(af, dens_term, param)
            del dens_term
        param["compute_additional_field"] = False
        if THEORY == "mond":  # solver.py:381-443: density <- div[nu(|grad phi_N| / g0) grad phi_N]
            g0 = (param["mond_g0"] * 1e-3 * 1e-10 * param["unit_t"] ** 2 / param["unit_l"]
                  * param["aexp"] ** (1 + param["mond_scale_factor_exponent"]))
            kinds = {"simple": 0, "n": 1, "beta": 2, "gamma": 3, "delta": 4}
            fn = param["mond_function"].casefold()
            if fn not in kinds:
                raise NotImplementedError(f"MOND_FUNCTION={fn!r}, should be 'simple', 'n', 'beta', 'gamma' or 'delta'")
            rhs = dens  # reused as the output buffer
            mg.mond_rhs(af, rhs, g0, kinds[fn], param["mond_alpha"])
        else:  # newton / fr: f1 (rho - 1)
            rhs = dens
            ops.linear_operator_inplace(rhs, f1, -f1)
        pot0 = owned(potential)
        warm = None
        if pot0 is not None:  # solver.py:273-280
            warm = np.float32(param["aexp"] * utils.table_eval(tables[3], np.log(param["aexp"]))
                              / (param["aexp_old"] * utils.table_eval(tables[3], np.log(param["aexp_old"]))))
        phi_lvl = solve(pot0, rhs, warm)
        del rhs, dens
        phi = phi_lvl[1:nx + 1]
        if THEORY == "fr":  # force = grad(phi) + half_c2 grad(u^(n+1)) (solver.py:167-203): force grid + gather
            acc = self._gather_fr(phi, af[1:nx + 1], position, param, scheme, kick)
        else:
            acc = self._gather(phi, position, param, scheme, kick)
        extra = af[1:nx + 1] if af is not None else torch.empty(0, dtype=torch.float32, device=dev)
        return acc, phi, extra

    def _gather_fr(self, phi, u, position, param, scheme, kick):
        """f(R) force from the owned planes of potential and scalaron (mesh.derivative_fR) and the gather."""
        import torch

        from . import solver as _solver

        n, nx, ops = self.n, self.nx, self.ops
        order = param["gradient_stencil_order"]
        if order not in _REACH:
            raise ValueError(f"gradient_order={order}, should be 2, 3, 5 or 7")
        H = _REACH[order] + 1
        ext = []
        for f in (phi, u):
            from_left, from_right = exchange_planes(f[:H], f[nx - H:])
            ext.append(torch.cat([from_left, f, from_right], dim=0))
        force = ops.gradient_fr(order, int(param["fR_n"]), _solver.fr_half_c2(param), ext[0], ext[1], self.x_lo - H, nx + 2,
                                self.x_lo - 1)
        del ext
        vel, half_dt = (kick if kick is not None else (None, 0.0))
        acc, ma, mv = ops.interp_kick(scheme, force, nx + 2, self.x_lo - 1, position, vel, half_dt)
        self.local_max = (ma, mv)
        return acc

    def _pm_replicated_grid(self, position, param, potential, additional_field, tables, kick, scheme):
        """Multigrid / f(R) / MOND / P(k) under slabs, first cut: the particles (deposit, gather, kick) stay
        decomposed, but the density slabs are all-gathered and every rank runs the single-GPU grid solve
        on the full mesh (no halo-exchange multigrid yet).  Returns the FULL potential / extra field so
        that the next step can warm-start from them (solver.py:273-280)."""
        import torch

        from . import solver

        n, nx, ops = self.n, self.nx, self.ops
        dist = _dist()
        mine = _stage(self._deposit_folded(position, scheme).contiguous())
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine)
        density = _unstage(torch.cat(parts, dim=0), position)
        del mine, parts
        if potential is None:
            potential = torch.empty(0, dtype=torch.float32, device=position.device)
        if additional_field is None:
            additional_field = torch.empty(0, dtype=torch.float32, device=position.device)
        solver._WRITE_PK = self.rank == 0
        try:
            potential, additional_field, force = solver._solve_grid(density, param, potential, additional_field,
                                                                    tables)
        finally:
            solver._WRITE_PK = True
        del density
        planes = [(self.x_lo - 1 + l) % n for l in range(nx + 2)]
        force_loc = force[planes].contiguous()  # own planes + one ghost each side
        del force
        vel, half_dt = (kick if kick is not None else (None, 0.0))
        acc, ma, mv = ops.interp_kick(scheme, force_loc, nx + 2, self.x_lo - 1, position, vel, half_dt)
        self.local_max = (ma, mv)
        return acc, potential, additional_field

    def poisson_fft(self, rhs, param, scale=None, khalo=None):
        """solver.fft (solver.py:453-522) on slabs: 2-D r2c, transpose, 1-D c2c, Green's function in
        the transposed layout, and back."""
        n, nx, G, ops = self.n, self.nx, self.world, self.ops
        nh, ny = n // 2 + 1, n // self.world
        MAS_index = param["MAS_index"]
        solver = param["linear_newton_solver"].casefold()
        if solver == "fft":
            kind, p = (L.GREEN_PLAIN, 0) if MAS_index == 0 else (L.GREEN_COMPENSATED, int(MAS_index))
        else:
            kind, p = L.GREEN_7PT, 0
        if getattr(ops, "has_fftp", None) and ops.has_fftp(nx, ny) and hasattr(ops, "fftp_y_forward") and _PIPELINE_FFT:
            # The same in chunks of kz so that the all-to-alls (xGMI) run under the FFT passes (HBM) of the
            # neighbouring chunks: y forward of chunk c+1 while chunk c is in flight, x pass of chunk c while
            # c+1 arrives and c-1 returns, y inverse likewise.
            nchunk = 4 if nh >= 16 else 1
            base, rem = divmod(nh, nchunk)
            ranges, k0 = [], 0
            for i in range(nchunk):
                ranges.append((k0, base + (1 if i < rem else 0)))
                k0 += ranges[-1][1]
            ops.fftp_z_forward(rhs, khalo) if khalo is not None else ops.fftp_z_forward(rhs)
            khalo = None
            fwd = [_AllToAllHandle(ops.fftp_y_forward(nx, kz0, nkz)) for kz0, nkz in ranges]   # [n (ky), nkz, nx] each
            bwd = []
            for (kz0, nkz), h in zip(ranges, fwd):
                seg = h.wait()                                                 # [G (src), ny, nkz, nx]
                ops.fftp_xgreen_kz(seg, kind, p, nx, ny, self.rank * ny, kz0, nkz, scale)
                bwd.append(_AllToAllHandle(seg))
            del fwd
            for (kz0, nkz), h in zip(ranges, bwd):
                ops.fftp_y_inverse(h.wait(), nx, kz0, nkz)
            del bwd
            return ops.fftp_z_inverse(nx)
        if khalo is not None:
            raise RuntimeError("a halo buffer was handed to a slab FFT that cannot consume it")
        if getattr(ops, "has_fftp", None) and ops.has_fftp(nx, ny):
            # LDS-FFT passes: the forward stage already writes [ky][kz][x_local], so both all-to-alls
            # exchange contiguous blocks and the x pass reads its lines as G segments of nx elements
            c = ops.fftp_forward(rhs)                                          # [n (ky), nh, nx]
            seg = all_to_all(c)                                                # [G (src), ny, nh, nx]
            del c
            ops.fftp_xgreen(seg, kind, p, nx, ny, self.rank * ny, scale)
            back = all_to_all(seg)                                             # [n (ky), nh, nx] again
            del seg
            return ops.fftp_inverse(back, nx)
        spec = ops.fft2d_r2c(rhs)                                              # [nx, n, nh]
        send = spec.view(nx, G, ny, nh).permute(1, 0, 2, 3).contiguous()       # [G(dst), nx, ny, nh]
        del spec
        t = all_to_all(send).view(n, ny, nh)                                   # [x (src-major), ny, nh]
        del send
        ops.fft1d_x(t, inverse=False)
        ops.greens(kind, p, t, self.rank * ny, ny, scale)
        ops.fft1d_x(t, inverse=True)
        back = all_to_all(t.view(G, nx, ny, nh))                               # [G(src = y block), nx, ny, nh]
        del t
        spec = back.permute(1, 0, 2, 3).contiguous().view(nx, n, nh)
        del back
        return ops.fft2d_c2r(spec)

    # ---- one integrator step (integration.py:17-264) ---------------------------------------------
    def integrate(self, position, velocity, acceleration, potential, additional_field, tables, param,
                  t_snap_next=np.float32(0)):
        from . import utils

        ops = self.ops
        self._adopt(position, velocity)
        position, velocity = self._views()
        key = (acceleration.data_ptr(), acceleration.shape[0], velocity.data_ptr(), velocity.shape[0])
        if getattr(self, "_max_for", None) == key:
            ma, mv = self.local_max
        else:
            ma, mv = ops.max_abs(acceleration), ops.max_abs(velocity)
        ma, mv = all_reduce_max([ma, mv], position.device)
        dx = np.float32(0.5 ** param["ncoarse"])
        with np.errstate(divide="ignore"):
            dt1 = np.float32(param["Courant_factor"]) * np.sqrt(dx / np.float32(ma))      # integration.py:267
            dt2 = np.float32(param["Courant_factor"]) * dx / np.float32(mv)                # :298
        aexp_factor = 1.0 + 0.01 * param["max_aexp_stepping"]
        dt3 = np.float32(utils.table_eval(tables[1], np.log(aexp_factor * param["aexp"]))
                         - utils.table_eval(tables[1], np.log(param["aexp"])))
        dt = np.min([dt1, dt2, dt3])
        if (param["t"] + dt) > t_snap_next:
            dt = t_snap_next - param["t"]
            param["write_snapshot"] = True
        else:
            param["write_snapshot"] = False
        INTEGRATOR = param["integrator"].casefold()
        if INTEGRATOR not in ("leapfrog", "euler"):
            raise NotImplementedError("ERROR: Integrator must be 'leapfrog' or 'euler'")

        def advance_clock():
            param["t"] += dt
            param["aexp_old"] = param["aexp"]
            param["aexp"] = np.exp(utils.table_eval(tables[0], param["t"]))
            utils.set_units(param)

        x_lo, x_hi = self.rank / self.world, (self.rank + 1) / self.world  # exact in fp32 (G is a power of two)
        if INTEGRATOR == "leapfrog":
            half_dt = np.float32(0.5 * dt)
            # Fused first kick + drift + deposit (pmstep.h): the deposit is done before the migration, by the old
            # owner, into ghost planes wide enough for the farthest a particle can get in this step:
            # |dx| <= dt max|v| + dt^2/2 max|a| cells (+ 1.5 for the TSC cloud).
            W = 0
            MASS_SCHEME = param["mass_scheme"].casefold()
            if (_FUSE_SLAB_DEPOSIT and hasattr(ops, "kick_drift_wrap_deposit") and MASS_SCHEME in _SCHEME
                    and self.n >= 16 and 0 < position.shape[0] < 2**32 and np.isfinite(ma) and np.isfinite(mv)):
                reach = (float(dt) * float(mv) + 0.5 * float(dt) ** 2 * float(ma)) * self.n + 1.5
                if reach < 4.0:
                    W = int(np.ceil(reach))
            codes = getattr(self, "_codes", None)
            if W > 0:
                if codes is not None:
                    # where every particle is expected after the NEXT drift (integration.predict_next_step; the
                    # growth of the accelerations from this rank's own sample): recorded by the fused kick
                    from . import integration

                    integration.predict_next_step(dt, dt1, dt2, tables, param, t_snap_next, getattr(ops, "last_meanabs", None))
                    ops.set_prediction(param.get("_dt_pred") or dt, param.get("_acc_scale_pred", 1.0))
                    ops.set_destination_codes(codes)
                # halo buffer (single-GPU step: DESIGN.md section 4): what the window's tiles add to each other stays
                # out of the density slab until the FFT's z pass loads its lines -- when that is the solve that follows
                khalo = ops.khalo_buffer(self.nx) if self._khalo_applies(param) else None
                try:
                    if khalo is not None:
                        rho_ext, leave = ops.kick_drift_wrap_deposit(_SCHEME[MASS_SCHEME], position, velocity, acceleration,
                                                                     half_dt, np.float32(dt), x_lo, x_hi, self.nx + 2 * W,
                                                                     self.x_lo - W, khalo=khalo)
                    else:
                        rho_ext, leave = ops.kick_drift_wrap_deposit(_SCHEME[MASS_SCHEME], position, velocity, acceleration,
                                                                     half_dt, np.float32(dt), x_lo, x_hi, self.nx + 2 * W,
                                                                     self.x_lo - W)
                finally:
                    if codes is not None:
                        ops.set_destination_codes(None)
                param["_dt_last"] = float(dt)
                self._predeposited = (rho_ext, W, khalo)
            else:
                leave = ops.kick_drift_wrap(position, velocity, acceleration, half_dt, np.float32(dt), x_lo, x_hi)
                if codes is not None:
                    codes.fill_(13)  # nothing recorded: every particle is filed under the tile it is in
            advance_clock()
            self._migrate_leavers(leave)
            position, velocity = self._views()
            if codes is not None and hasattr(ops, "sorted_stream_resized") and position.shape[0] > 0:
                ops.sorted_stream_resized(position)
            self._rebucketed = False
            acceleration, potential, additional_field = self.pm(position, param, potential, additional_field, tables,
                                                                kick=(velocity, half_dt))
            if self._rebucketed:  # the gather wrote the other set of arrays
                position, velocity = self._views()
            if _log.isEnabledFor(logging.DEBUG):
                _log.debug("slab step: rank %d dt %.5g guess of the next %s acceleration scale %s leavers %d held %d "
                           "re-bucketed %s", self.rank, float(dt), param.get("_dt_pred"), param.get("_acc_scale_pred"),
                           int(leave.numel()), position.shape[0], self._rebucketed)
        else:
            # drift with the old velocity, then kick with the old acceleration (integration.py:178-183)
            leave = ops.kick_drift_wrap(position, velocity, acceleration, np.float32(0), np.float32(dt), x_lo, x_hi)
            ops.axpy(velocity, acceleration, -np.float32(dt))
            advance_clock()
            self._migrate_leavers(leave)
            position, velocity = self._views()
            acceleration, potential, additional_field = self.pm(position, param, potential, additional_field, tables)
            self.local_max = (self.local_max[0], ops.max_abs(velocity))
        self._max_for = (acceleration.data_ptr(), acceleration.shape[0], velocity.data_ptr(), velocity.shape[0])
        return position, velocity, acceleration, potential, additional_field


_slabs = {}


def get(n: int, ops=None) -> Slab:
    key = (n, id(ops))
    if key not in _slabs:
        _slabs[key] = Slab(n, ops)
    return _slabs[key]
