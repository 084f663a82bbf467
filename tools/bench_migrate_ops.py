#!/usr/bin/env python3
"""Cost of the torch index plumbing in Slab.migrate at 1024^3 / 8 particles per rank (development tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.bench_kernels import timeit

npart = 1024**3 // 8
pos = torch.rand((npart, 3), device="cuda")
vel = torch.rand((npart, 3), device="cuda")
world, rank = 8, 3
pos[:, 0] = (rank + pos[:, 0]) / world
pos[::997, 0] += 0.001  # ~0.1 % near the upper face, some cross


def detect():
    owner = (pos[:, 0] * world).long().clamp_(0, world - 1)
    leaving = owner != rank
    n = int(leaving.sum())
    idx = torch.nonzero(leaving, as_tuple=False).squeeze(1)
    return owner, leaving, idx, n


owner, leaving, idx, n = detect()
print("leavers", n, "of", npart)
print(f"detect (owner, mask, sum, nonzero)   {timeit(detect):8.3f} ms")
keep = ~leaving
print(f"compaction pos[keep], vel[keep]      {timeit(lambda: (pos[keep], vel[keep])):8.3f} ms")
print(f"gather leavers pos[idx], vel[idx]    {timeit(lambda: (pos[idx], vel[idx])):8.3f} ms")
