#!/usr/bin/env python3
"""Reduce two rocprofv3 --pmc collections (FETCH_SIZE pass, WRITE_SIZE pass) of `bench.py` to a small
per-kernel JSON: HBM-side bytes per launch with the gfx950 corrections of MI355X_MICROARCH.md (HBM):
FETCH_SIZE is in KiB and counts 64 B per 128-B request -> x2 x1024; WRITE_SIZE in KiB, exact.
    python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import collections
import csv
import json
import re
import sys


def last_per_kernel(path):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        d.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    # the largest of the later launches: a kernel launched twice per step (owners + a few surplus workgroups
    # of the fused deposit) must not be averaged with its small second launch
    return {k: max(v[len(v) // 2:]) for k, v in d.items()}


def short(name):
    m = re.match(r"(?:void )?([A-Za-z_0-9]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name


fetch, write, out = sys.argv[1:4]
F, W = last_per_kernel(fetch), last_per_kernel(write)
keep = ("kick_drift_wrap", "kdw_deposit", "deposit_list", "deposit_ranges", "interp_ranges", "gradient_march", "fftp_",
        "greens_kernel", "ew_kernel", "straggler_scan",
        "tile_bounds", "max_abs")
res = {}
for k in F:
    if any(s in k for s in keep):
        fb, wb = 2.0 * 1024.0 * F[k], 1024.0 * W.get(k, 0.0)
        res[short(k)] = {"fetch_bytes": fb, "write_bytes": wb, "hbm_bytes": fb + wb}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py --steps 2 --warmup 1, "
                     "1024^3, 1 GPU; FETCH_SIZE x2 (gfx950), KiB -> bytes", "kernels": res}, open(out, "w"), indent=1)
for k, v in res.items():
    print(f"{k:48s} fetch {v['fetch_bytes'] / 1e9:8.2f} GB  write {v['write_bytes'] / 1e9:8.2f} GB")
