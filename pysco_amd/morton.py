"""Morton (Z-order) keys, 21 bits per axis (reference morton.py:42-137)."""
from __future__ import annotations

from . import _lib as L


def positions_to_keys(positions):
    torch = L.require_gpu()
    pos = L.to_device(positions, torch.float32)
    ctx = L.context(4)
    keys = torch.empty(pos.shape[0], dtype=torch.int64, device=pos.device)
    ctx.call("pm_morton_keys", L.ptr(pos), pos.shape[0], L.ptr(keys))
    return L.like_input(keys, positions)
