"""QUMOND source term div[nu(|grad phi_N|/g0) grad phi_N] for the five interpolating-function
families (reference mond.py:171-932).  `out` is overwritten in place."""
from __future__ import annotations

import numpy as np

from . import _lib as L

_KIND = {"simple": 0, "n": 1, "beta": 2, "gamma": 3, "delta": 4}


def _rhs(potential, out, g0, kind: str, alpha) -> None:
    torch = L.require_gpu()
    pd_ = L.to_device(potential, torch.float32)
    ctx = L.context(4)
    args = (pd_.shape[0], float(np.float32(g0)), _KIND[kind], float(np.float32(alpha)))
    if L.is_device_array(out):
        ctx.call("pm_mond_rhs", L.ptr(pd_), L.ptr(out), *args)
    else:
        od = torch.empty_like(pd_)
        ctx.call("pm_mond_rhs", L.ptr(pd_), L.ptr(od), *args)
        out[...] = od.cpu().numpy()


def rhs_simple(potential, out, g0) -> None:  # mond.py:171
    _rhs(potential, out, g0, "simple", 1.0)


def rhs_n(potential, out, g0, n) -> None:  # mond.py:322
    _rhs(potential, out, g0, "n", n)


def rhs_beta(potential, out, g0, beta) -> None:  # mond.py:476
    _rhs(potential, out, g0, "beta", beta)


def rhs_gamma(potential, out, g0, gamma) -> None:  # mond.py:630
    _rhs(potential, out, g0, "gamma", gamma)


def rhs_delta(potential, out, g0, delta) -> None:  # mond.py:784
    _rhs(potential, out, g0, "delta", delta)
