#!/usr/bin/env python3
"""Per-step kernel durations from a rocprofv3 --kernel-trace CSV of bench.py: one column per step (a step starts at
its tile_bounds_kernel / tile_extra_kernel launch, or at kr_counts_kernel when the fused kick re-buckets), one row per kernel, milliseconds.
    python tools/trace_per_step.py <kernel_trace.csv> [every]"""
import collections
import csv
import re
import sys


def short(n):
    m = re.match(r"(?:void )?((?:rocprim::)?[A-Za-z_0-9:]+(?:<[^(]*?>)?)", n)
    return (m.group(1) if m else n)[:56]


rows = list(csv.DictReader(open(sys.argv[1])))
every = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
steps, cur = [], None
for r in rows:
    nm = r["Kernel_Name"]
    if ("tile_bounds_kernel" in nm or "kr_counts_kernel" in nm
            or ("tile_extra_kernel" in nm and (cur is None or "kdw_deposit_kernel<3>" in cur))):
        cur = collections.OrderedDict()
        steps.append(cur)
    if cur is None or nm.startswith("void at::") or "at::native" in nm:
        continue
    k = short(nm)
    cur[k] = cur.get(k, 0.0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
steps = [s for s in steps if any("kdw_deposit" in k for k in s)]
keys = []
for s in steps:
    for k in s:
        if k not in keys:
            keys.append(k)
sel = steps[::every]
print(f"{len(steps)} steps (warm-up included); columns: every {every} step(s); ms")
for k in keys:
    if max(s.get(k, 0) for s in steps) >= 0.005:
        print(f"{k:58s}" + " ".join(f"{s.get(k, 0):6.2f}" for s in sel))
print(f"{'sum':58s}" + " ".join(f"{sum(s.values()):6.2f}" for s in sel))
