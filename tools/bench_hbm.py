#!/usr/bin/env python3
"""HBM ceilings of this GPU with plain torch kernels (development tool): copy, fill, read-only sum."""
import torch


def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for gib in (1, 4):
    n = gib * 2**28
    a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    b = torch.empty_like(a)
    ms = t(lambda: b.copy_(a)); print(f"{gib} GiB copy      {ms:7.3f} ms  {2 * 4 * n / ms / 1e6:7.0f} GB/s")
    ms = t(lambda: b.fill_(1.0)); print(f"{gib} GiB fill      {ms:7.3f} ms  {4 * n / ms / 1e6:7.0f} GB/s")
    ms = t(lambda: a.sum()); print(f"{gib} GiB sum       {ms:7.3f} ms  {4 * n / ms / 1e6:7.0f} GB/s")
    ms = t(lambda: torch.add(a, 1.0, out=b)); print(f"{gib} GiB add->out  {ms:7.3f} ms  {2 * 4 * n / ms / 1e6:7.0f} GB/s")
    ms = t(lambda: a.add_(1.0)); print(f"{gib} GiB add_ inpl {ms:7.3f} ms  {2 * 4 * n / ms / 1e6:7.0f} GB/s")
    del a, b
