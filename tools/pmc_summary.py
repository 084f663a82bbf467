#!/usr/bin/env python3
"""Reduce two rocprofv3 --pmc collections (FETCH_SIZE pass, WRITE_SIZE pass) of ONE bench.py command line to HBM-side
bytes per C-ABI stage call, averaged over the TIMED steps, with the gfx950 corrections of MI355X_MICROARCH.md (HBM):
FETCH_SIZE is in KiB and counts 64 B per 128-B request -> x2 x1024; WRITE_SIZE in KiB, exact.

    python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> \
           --ncells 1024 --steps 20 --warmup 5 --rebucket-fraction 0.0 --rms-cells 0.5

bench.py reports `roofline.traffic` from the output only when every one of these matches its own command line."""
import argparse
import collections
import csv
import json
import re

# kernels of each stage (substring match on the kernel name)
STAGES = {
    "pm_kick_drift_wrap_deposit_khalo": ["kdw_deposit_kernel", "deposit_list_kernel", "deposit_list_flat_kernel", "tile_extra_kernel",
                                         "tile_bounds_kernel", "list_compact_kernel", "list_bounds_kernel", "rocprim"],
    "pm_gradient_interp_kick_rebucket": ["interp_ranges_kernel<3, 5, true, true>", "rebucket_counts_kernel"],
    # round 3, "the kick re-buckets": the gather in place + decisions, the fused kick writing the new stream
    "pm_gradient_interp_kick_codes": ["interp_ranges_kernel<3, 5, true, false, true>", "kr_segcap_kernel"],
    "pm_kick_drift_wrap_deposit_rebucket": ["kdw_deposit_kernel<3, true>", "kr_counts_kernel", "deposit_list_kernel", "deposit_list_flat_kernel",
                                            "tile_extra_kernel", "list_compact_kernel", "list_bounds_kernel", "rocprim"],
    "pm_gradient_interp_kick": ["interp_ranges_kernel<3, 5, true, false>"],
    "pm_poisson_fft_fused_khalo": ["fftp_z_r2c", "fftp_y<", "fftp_x_green", "fftp_z_c2r"],
}


def short(name):
    m = re.match(r"(?:void )?((?:rocprim::)?[A-Za-z_0-9:]+(?:<[^(]*?>)?)", name)
    return (m.group(1) if m else name)[:70]


def timed_rows(path, warmup):
    """Rows of the timed steps: everything after the `warmup`-th gather-with-kick dispatch (one per step)."""
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    seen, start = 0, 0
    for n, r in enumerate(rows):
        if "interp_ranges_kernel<3, 5, true" in r["Kernel_Name"]:
            seen += 1
            if seen == warmup:
                start = n + 1
                break
    return rows[start:]


ap = argparse.ArgumentParser()
ap.add_argument("fetch")
ap.add_argument("write")
ap.add_argument("out")
ap.add_argument("--ncells", type=int, default=1024)
ap.add_argument("--steps", type=int, required=True)
ap.add_argument("--warmup", type=int, required=True)
ap.add_argument("--rebucket-fraction", type=float, default=0.0)
ap.add_argument("--rms-cells", type=float, default=0.5)
a = ap.parse_args()

per_kernel = collections.OrderedDict()
for which, path, factor in (("fetch", a.fetch, 2.0 * 1024.0), ("write", a.write, 1024.0)):
    for r in timed_rows(path, a.warmup):
        k = short(r["Kernel_Name"])
        d = per_kernel.setdefault(k, {"fetch": 0.0, "write": 0.0, "launches": 0})
        d[which] += factor * float(r["Counter_Value"])
        if which == "fetch":
            d["launches"] += 1
stages = {}
for stage, pats in STAGES.items():
    ks = {k: v for k, v in per_kernel.items() if any(p in k for p in pats)}
    if not ks:
        continue
    calls = a.steps if stage != "pm_gradient_interp_kick" else max(1, sum(v["launches"] for v in ks.values()))
    if stage in ("pm_gradient_interp_kick_rebucket", "pm_gradient_interp_kick_codes"):
        calls = max(1, sum(v["launches"] for k, v in ks.items() if "interp_ranges" in k))
    if stage == "pm_kick_drift_wrap_deposit_rebucket":
        calls = max(1, sum(v["launches"] for k, v in ks.items() if "kr_counts_kernel" in k))
    f, w = sum(v["fetch"] for v in ks.values()), sum(v["write"] for v in ks.values())
    stages[stage] = {"calls": calls, "fetch_bytes_per_call": f / calls, "write_bytes_per_call": w / calls,
                     "hbm_bytes_per_call": (f + w) / calls,
                     "kernels": {k: {"launches": v["launches"], "fetch_bytes": v["fetch"], "write_bytes": v["write"]}
                                 for k, v in ks.items()}}
json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on "
                     f"bench.py --gpus 1 --steps {a.steps} --warmup {a.warmup} --no-cpu-baseline; dispatches of the timed "
                     "steps only; FETCH_SIZE x2 (gfx950), KiB -> bytes",
           "ncells": a.ncells, "steps": a.steps, "warmup": a.warmup, "rebucket_fraction": a.rebucket_fraction,
           "rms_cells": a.rms_cells, "stages": stages}, open(a.out, "w"), indent=1)
for s, v in stages.items():
    print(f"{s:40s} calls {v['calls']:3d}  fetch {v['fetch_bytes_per_call'] / 1e9:7.2f} GB  write {v['write_bytes_per_call'] / 1e9:7.2f} GB per call")
