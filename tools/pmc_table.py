#!/usr/bin/env python3
"""Per-kernel means of a rocprofv3 --pmc counter_collection.csv (development tool).
    python tools/pmc_table.py <counter_collection.csv> [kernel-name-substring ...]"""
import collections
import csv
import re
import sys

path, keys = sys.argv[1], sys.argv[2:] or ["interp", "kdw", "deposit", "fftp"]
d = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    k = re.match(r"(?:void )?(?:\(anonymous namespace\)::)?([A-Za-z_0-9:]+(?:<[^>]*>)?)", r["Kernel_Name"]).group(1)
    d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in d.items():
    if any(s in k for s in keys):
        print(f"{k}   {sum(dur[k]) / len(dur[k]):.3f} ms")
        for c, x in v.items():
            print(f"    {c:26s} {sum(x) / len(x):.5g}")
