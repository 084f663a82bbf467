"""ctypes binding of libpmstep.so (include/pmstep.h) and the device-array plumbing.

There is deliberately NO CPU fallback: if the HIP library is missing or no GPU is visible, the
first compute call raises.  PyTorch is used only to own device memory and the stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PMSTEP_LIB", os.path.join(_HERE, "libpmstep.so"))  # override: diagnostic builds only

NGP, CIC, TSC = 1, 2, 3
GREEN_PLAIN, GREEN_COMPENSATED, GREEN_7PT = 0, 1, 2

_vp = C.c_void_p
_i64 = C.c_int64
_f = C.c_float
_d = C.c_double
_i = C.c_int

# name -> argtypes (restype is always int unless listed in _RESTYPES)
SIGNATURES: Dict[str, tuple] = {
    "pm_version": (),
    "pm_last_error": (),
    "pm_ctx_create": (C.POINTER(_vp), _i, _vp),
    "pm_ctx_destroy": (_vp,),
    "pm_ctx_set_stream": (_vp, _vp),
    "pm_ctx_sync": (_vp,),
    "pm_timer_start": (_vp,),
    "pm_timer_stop_ms": (_vp, C.POINTER(_f)),
    "pm_deposit": (_vp, _i, _vp, _i64, _vp, _i),
    "pm_interp": (_vp, _i, _i, _vp, _vp, _i64, _vp),
    "pm_gradient": (_vp, _i, _vp, _vp),
    "pm_gradient_fr": (_vp, _i, _i, _f, _vp, _vp, _vp),
    "pm_axpy": (_vp, _vp, _vp, _f, _i64),
    "pm_scale": (_vp, _vp, _f, _i64),
    "pm_probe_update": (_vp, _vp, _vp, _vp, _i64),
    "pm_linear_operator": (_vp, _vp, _vp, _f, _f, _i64),
    "pm_linear_operator_vectors": (_vp, _vp, _f, _vp, _f, _i64),
    "pm_max_abs": (_vp, _vp, _i64, C.POINTER(_f)),
    "pm_periodic_wrap": (_vp, _vp, _i64),
    "pm_morton_keys": (_vp, _vp, _i64, _vp),
    "pm_reorder_particles": (_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp),
    "pm_fft_r2c": (_vp, _vp, _vp),
    "pm_fft_c2r": (_vp, _vp, _vp),
    "pm_greens": (_vp, _i, _i, _vp, _vp, _vp, _f),
    "pm_grad_inverse_laplacian": (_vp, _i, _vp, _vp, _vp, _vp),
    "pm_interleave3": (_vp, _vp, _i64, _vp),
    "pm_kspace_gradient": (_vp, _vp, _vp, _vp),
    "pm_kspace_hessian": (_vp, _vp, _vp, _vp, _i, _i, _i, _i),
    "pm_addcmul": (_vp, _vp, _f, _vp, _vp, _i64, _i),
    "pm_addcmul3": (_vp, _vp, _f, _vp, _vp, _vp, _i64),
    "pm_dealias_resize": (_vp, _vp, _vp, _i),
    "pm_kspace_hessian_padded": (_vp, _vp, _vp, _vp, _i, _i, _i, _i),
    "pm_fft_r2c_padded": (_vp, _vp, _vp),
    "pm_fft_c2r_padded": (_vp, _vp, _vp),
    "pm_lpt_displace": (_vp, _vp, _f, _f, _i, _vp, _vp),
    "pm_pk": (_vp, _i, _vp, _vp, _vp, _vp, _vp),
    "pm_poisson_fft": (_vp, _i, _i, _vp, _vp, _vp, _vp, _f),
    "pm_poisson_fft_fused": (_vp, _i, _i, _vp, _vp, _vp, _vp, _f),
    "pm_poisson_fft_fused_khalo": (_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _f),
    "pm_fftp_forward": (_vp, _vp, _i, _vp),
    "pm_fftp_xgreen": (_vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _f),
    "pm_fftp_inverse": (_vp, _vp, _i, _vp),
    "pm_fftp_z_forward": (_vp, _vp, _i),
    "pm_fftp_y_forward": (_vp, _i, _i, _i, _vp),
    "pm_fftp_xgreen_kz": (_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _f),
    "pm_fftp_y_inverse": (_vp, _vp, _i, _i, _i),
    "pm_fftp_z_inverse": (_vp, _i, _vp),
    "pm_kick_drift_wrap": (_vp, _vp, _vp, _vp, _f, _f, _i64),
    "pm_interp_kick": (_vp, _i, _vp, _vp, _i64, _vp, _vp, _f),
    "pm_tile_lists": (_vp, _vp, _i64),
    "pm_tile_lists_info": (_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i)),
    "pm_kick_drift_wrap_deposit": (_vp, _i, _vp, _vp, _vp, _f, _f, _i64, _vp),
    "pm_kick_drift_wrap_deposit_khalo": (_vp, _i, _vp, _vp, _vp, _f, _f, _i64, _vp, _vp),
    "pm_khalo_apply": (_vp, _vp, _vp),
    "pm_khalo_floats": (_i,),
    "pm_khalo_floats_slab": (_i, _i),
    "pm_set_fr_march_gradient": (_vp, _i),
    "pm_kick_drift_wrap_deposit_slab_khalo": (_vp, _i, _vp, _vp, _vp, _f, _f, _i64, _vp, _f, _f, _vp, _i, C.POINTER(_i), _vp),
    "pm_fftp_z_forward_khalo": (_vp, _vp, _i, _vp),
    "pm_kick_drift_wrap_deposit_slab": (_vp, _i, _vp, _vp, _vp, _f, _f, _i64, _vp, _f, _f, _vp, _i, C.POINTER(_i)),
    "pm_gradient_interp_kick": (_vp, _i, _i, _vp, _i, _i, _vp, _i64, _vp, _vp, _f),
    "pm_gradient_interp_kick_rebucket": (_vp, _i, _i, _vp, _vp, _vp, _i64, _f, _vp, _vp, _vp, _vp, _vp),
    "pm_gradient_interp_kick_rebucket_slab": (_vp, _i, _i, _vp, _i, _i, _vp, _vp, _i64, _f, _vp, _vp, _vp, _vp, _vp),
    "pm_set_destination_codes": (_vp, _vp),
    "pm_rebucket_prepare": (_vp, _vp, _i64, _vp),
    "pm_tile_lists_codes": (_vp, _vp, _i64, _vp),
    "pm_sorted_stream_resized": (_vp, _vp, _i64),
    "pm_set_predicted_dt": (_vp, _f),
    "pm_set_predicted_acc_scale": (_vp, _f),
    "pm_last_acc_meanabs": (_vp, C.POINTER(C.c_double)),
    "pm_last_maxabs": (_vp, C.POINTER(_f), C.POINTER(_f)),
    "pm_set_gather_mode": (_vp, _i),
    "pm_restriction": (_vp, _vp, _i, _f, _vp),
    "pm_prolongation": (_vp, _vp, _vp, _i, _i),
    "pm_lap_operator": (_vp, _vp, _vp, _i, _i, _vp),
    "pm_lap_restrict_residual": (_vp, _vp, _vp, _i, _vp),
    "pm_lap_residual_error": (_vp, _vp, _vp, _i, C.POINTER(_f)),
    "pm_diff_norm": (_vp, _vp, _vp, _f, _i64, C.POINTER(_f)),
    "pm_lap_vcycle": (_vp, _vp, _vp, _i, _i, _i),
    "pm_set_vcycle_graphs": (_vp, _i),
    "pm_set_fused_coarse": (_vp, _i),
    "pm_set_fr_rows_sweep": (_vp, _i),
    "pm_set_mond_fast_nu": (_vp, _i),
    "pm_set_fr_fused_restrict": (_vp, _i),
    "pm_vcycle_graph_stats": (_vp, C.POINTER(C.c_uint), C.POINTER(C.c_uint)),
    "pm_lap_smoothing": (_vp, _vp, _vp, _i, _i),
    "pm_fr_vcycle": (_vp, _i, _vp, _vp, _f, _i, _i, _i),
    "pm_fr_smoothing": (_vp, _i, _vp, _vp, _f, _vp, _i, _i),
    "pm_set_fused_sweep": (_vp, _i),
    "pm_lap_gauss_seidel": (_vp, _vp, _vp, _i, _f),
    "pm_cubic_operator": (_vp, _vp, _vp, _f, _vp, _i, _vp),
    "pm_cubic_residual_error": (_vp, _vp, _vp, _f, _i, C.POINTER(_f)),
    "pm_cubic_initialise_potential": (_vp, _vp, _f, _i, _vp),
    "pm_cubic_gauss_seidel": (_vp, _vp, _vp, _f, _vp, _i, _f),
    "pm_quartic_operator": (_vp, _vp, _vp, _f, _vp, _i, _vp),
    "pm_quartic_residual_error": (_vp, _vp, _vp, _f, _i, C.POINTER(_f)),
    "pm_quartic_initialise_potential": (_vp, _vp, _f, _i, _vp),
    "pm_quartic_gauss_seidel": (_vp, _vp, _vp, _f, _vp, _i, _f),
    "pm_mond_rhs": (_vp, _vp, _vp, _i, _f, _i, _f),
    "pm_set_slab": (_vp, _i, _i),
    "pm_set_tile_window": (_vp, _i, _i),
    "pm_lap_gauss_seidel_slab": (_vp, _vp, _vp, _i, _i, _i, _i, _f),
    "pm_lap_restrict_residual_slab": (_vp, _vp, _vp, _i, _i, _vp),
    "pm_restriction_slab": (_vp, _vp, _i, _i, _f, _vp),
    "pm_prolongation_slab": (_vp, _vp, _vp, _i, _i, _i),
    "pm_lap_operator_slab": (_vp, _vp, _vp, _i, _i, _i, _vp),
    "pm_lap_residual_sumsq_slab": (_vp, _vp, _vp, _i, _i, C.POINTER(C.c_double)),
    "pm_diff_sumsq": (_vp, _vp, _vp, _f, _i64, C.POINTER(C.c_double)),
    "pm_mond_rhs_slab": (_vp, _vp, _vp, _i, _i, _f, _i, _f),
    "pm_fr_operator_slab": (_vp, _i, _vp, _vp, _f, _vp, _i, _i, _vp),
    "pm_fr_residual_sumsq_slab": (_vp, _i, _vp, _vp, _f, _i, _i, C.POINTER(C.c_double)),
    "pm_fr_gauss_seidel_slab": (_vp, _i, _vp, _vp, _f, _vp, _i, _i, _i, _i, _f),
    "pm_gradient_fr_slab": (_vp, _i, _i, _f, _vp, _vp, _i, _i, _vp, _i, _i),
    "pm_cubic_initialise_potential_flat": (_vp, _vp, _f, _i, _i64, _vp),
    "pm_quartic_initialise_potential_flat": (_vp, _vp, _f, _i, _i64, _vp),
    "pm_kick_drift_wrap_slab": (_vp, _vp, _vp, _vp, _f, _f, _i64, _f, _f, _vp, _i, C.POINTER(_i)),
    "pm_gradient_slab": (_vp, _i, _vp, _i, _i, _vp, _i, _i),
    "pm_fft2d_r2c": (_vp, _vp, _vp, _i),
    "pm_fft2d_c2r": (_vp, _vp, _vp, _i),
    "pm_fft1d_x": (_vp, _vp, _i64, _i),
    "pm_greens_slab": (_vp, _i, _i, _vp, _vp, _vp, _f, _i, _i),
    "pm_gradient_interp_kick_codes": (_vp, _i, _i, _vp, _vp, _vp, _i64, _f, _vp),
    "pm_rebucket_kick_ready": (_vp, _vp, _i64, C.POINTER(_i)),
    "pm_kick_drift_wrap_deposit_rebucket": (_vp, _i, _vp, _vp, _vp, _f, _f, _i64, _vp, _vp, _vp, _vp, _vp, _vp),
    "pm_lap_solve": (_vp, _vp, _vp, _i, _i, _i, _f, _i, C.POINTER(_i), C.POINTER(_f)),
    "pm_fr_solve": (_vp, _i, _vp, _vp, _f, _i, _i, _i, _f, _i, C.POINTER(_i), C.POINTER(_f)),
    # the fp64 device mode (pm_real64.hip)
    "pm_ctx_create_real": (C.POINTER(_vp), _i, _vp, _i),
    "pm_ctx_real_bytes": (_vp,),
    "pm_real64_kick_drift_wrap": (_vp, _vp, _vp, _vp, _d, _d, _i64),
    "pm_real64_scale_add": (_vp, _vp, _d, _d, _i64),
    "pm_real64_max_abs": (_vp, _vp, _i64, C.POINTER(_d)),
    "pm_real64_deposit": (_vp, _i, _vp, _i64, _vp),
    "pm_real64_kick_drift_wrap_deposit": (_vp, _i, _vp, _vp, _vp, _d, _d, _i64, _vp),
    "pm_real64_poisson_fft": (_vp, _i, _i, _vp, _vp, _vp, _vp, _d),
    "pm_real64_gradient": (_vp, _i, _vp, _vp),
    "pm_real64_interp_kick": (_vp, _i, _vp, _vp, _i64, _vp, _vp, _d, C.POINTER(_d), C.POINTER(_d)),
    "pm_real64_pk": (_vp, _i, _vp, _vp, _vp, _vp, _vp),
    "pm_real64_reorder_particles": (_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i),
    "pm_real64_stream_age": (_vp, _vp, _vp, _vp, _vp),
    "pm_real64_reserve_sort": (_vp, _i64),
}
_RESTYPES = {"pm_last_error": C.c_char_p, "pm_khalo_floats": C.c_int64, "pm_khalo_floats_slab": C.c_int64}

_lib = None


class PMStepError(RuntimeError):
    pass


def load_library():
    """dlopen libpmstep.so and attach signatures.  Works without a GPU (symbols only)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build the HIP extension first (python -m pysco_amd.build). "
                "pysco_amd has no CPU fallback."
            )
        # torch first: it ships its own HIP runtime, and libpmstep must bind to that one (loaded the other way
        # round the process ends up with two runtimes and ours sees no device)
        _torch()
        lib = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
            fn.argtypes = list(argtypes)
            fn.restype = _RESTYPES.get(name, C.c_int)
        _lib = lib
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load_library().pm_last_error()
        raise PMStepError(f"libpmstep error {rc}: {msg.decode() if msg else '?'}")


# ------------------------------------------------------------------------------------------------
# device plumbing


def _torch():
    import torch

    return torch


_gpu_seen = False


def require_gpu():
    global _gpu_seen
    torch = _torch()
    if not _gpu_seen:  # (asked once: torch.cuda.is_available() costs 2 us and a step asks sixty times)
        if not torch.cuda.is_available():
            raise PMStepError("pysco_amd needs an MI355X (no HIP device visible); there is no CPU fallback")
        _gpu_seen = True
    return torch


# entry points that run on a level of a multigrid hierarchy: position of their grid-size argument (after the context),
# so that the stage profiler files the calls per level ("pm_lap_gauss_seidel@512")
_PROFILE_SIZE_ARG = {"pm_lap_gauss_seidel": 2, "pm_lap_smoothing": 2, "pm_fr_smoothing": 5, "pm_fr_vcycle": 4, "pm_lap_vcycle": 2, "pm_lap_solve": 2, "pm_fr_solve": 4, "pm_lap_restrict_residual": 2, "pm_lap_operator": 2,
                     "pm_lap_residual_error": 2, "pm_prolongation": 2, "pm_restriction": 1, "pm_cubic_gauss_seidel": 4,
                     "pm_quartic_gauss_seidel": 4, "pm_cubic_operator": 4, "pm_quartic_operator": 4,
                     "pm_cubic_residual_error": 3, "pm_quartic_residual_error": 3, "pm_mond_rhs": 2}

_ctxs: Dict[tuple, "Context"] = {}
_profiler = None  # list of (name, start_event, stop_event) while stage profiling is enabled


def profile_start() -> None:
    """Start recording a (HIP event, HIP event) pair around every C-ABI call (bench.py roofline)."""
    global _profiler
    _profiler = []


def profile_mark() -> None:
    """End of one step: profile_stop(per_step=True) splits the record here."""
    if _profiler is not None:
        _profiler.append((None, None, None))


def profile_stop(per_step: bool = False):
    """Stop recording; returns {entry point: (calls, total milliseconds)} after synchronising -- with
    `per_step`, also the list of {entry point: milliseconds} of every stretch closed by profile_mark()."""
    global _profiler
    rec, _profiler = _profiler or [], None
    _torch().cuda.synchronize()
    out: Dict[str, Tuple[int, float]] = {}
    steps, cur = [], {}
    for name, e0, e1 in rec:
        if name is None:
            steps.append(cur)
            cur = {}
            continue
        ms = e0.elapsed_time(e1)
        c, t = out.get(name, (0, 0.0))
        out[name] = (c + 1, t + ms)
        cur[name] = cur.get(name, 0.0) + ms
    return (out, steps) if per_step else out


class Context:
    """pm_ctx for one (device, grid size, arithmetic type)."""

    def __init__(self, n: int, real_bytes: int = 4):
        torch = require_gpu()
        self.lib = load_library()
        self.n = int(n)
        self.real_bytes = int(real_bytes)
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.handle = _vp()
        stream = torch.cuda.current_stream().cuda_stream
        check(self.lib.pm_ctx_create_real(C.byref(self.handle), self.n, _vp(stream), self.real_bytes))
        self._stream = stream
        self._stream_obj = torch.cuda.current_stream()

    def call(self, name: str, *args) -> None:
        """Invoke one C-ABI entry point on this context; raises PMStepError on failure.  When the
        stage profiler is on, the call is bracketed by events on the stream it launches on."""
        fn = getattr(self.lib, name)
        prof = _profiler
        if prof is None:
            check(fn(self.handle, *args))
            return
        torch = _torch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # (recorded on the stream object cached with the context's stream: Event.record() without one looks the
        # current stream up again, 10 us a time -- 0.3 ms of a 13-call step spent measuring it)
        stream = self._stream_obj
        e0.record(stream)
        check(fn(self.handle, *args))
        e1.record(stream)
        size_arg = _PROFILE_SIZE_ARG.get(name)
        prof.append((name if size_arg is None else f"{name}@{args[size_arg]}", e0, e1))

    def use_current_stream(self):
        torch = _torch()
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)  # (the raw handle: 0.3 us; current_stream(): 8 us)
        stream = raw(self.device.index) if raw is not None else torch.cuda.current_stream().cuda_stream
        if stream != self._stream or self._stream_obj is None:
            if stream != self._stream:
                check(self.lib.pm_ctx_set_stream(self.handle, _vp(stream)))
            self._stream = stream
            self._stream_obj = torch.cuda.current_stream()

    def __del__(self):
        try:
            if self.handle:
                self.lib.pm_ctx_destroy(self.handle)
        except Exception:
            pass


def context(n: int, real_bytes: int = 4) -> Context:
    """The context of (current device, n) -- of the fp64 device mode with real_bytes = 8 (pm_real64_* entry points)."""
    torch = require_gpu()
    key = (torch.cuda.current_device(), int(n)) if real_bytes == 4 else (torch.cuda.current_device(), int(n), int(real_bytes))
    ctx = _ctxs.get(key)
    if ctx is None:
        ctx = _ctxs[key] = Context(n, real_bytes)
    ctx.use_current_stream()
    return ctx


def is_device_array(x) -> bool:
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


def to_device(x, dtype=None):
    """numpy array or torch tensor -> contiguous CUDA tensor (no copy if already one)."""
    torch = require_gpu()
    if is_device_array(x):
        if not x.is_cuda:
            x = x.cuda()
        if dtype is not None and x.dtype != dtype:
            x = x.to(dtype)
        return x.contiguous()
    a = np.ascontiguousarray(x)
    t = torch.from_numpy(a).cuda()
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t


def like_input(t, template):
    """Return `t` (a CUDA tensor) in the same kind as `template` (numpy in -> numpy out)."""
    if is_device_array(template):
        return t
    return t.cpu().numpy()


def ptr(t) -> _vp:
    return _vp(t.data_ptr()) if t is not None else _vp(None)


def hptr(a: np.ndarray) -> _vp:
    return a.ctypes.data_as(_vp)
