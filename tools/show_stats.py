#!/usr/bin/env python3
"""Pretty-print a rocprofv3 *_kernel_stats.csv (development tool)."""
import csv
import glob
import sys

f = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/*/*/*_kernel_stats.csv"))[-1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for r in list(csv.DictReader(open(f)))[:top]:
    print(f"{r['Name'][:84]:84s} calls={r['Calls']:>4s} avg_ms={float(r['AverageNs']) / 1e6:8.3f} pct={r['Percentage']}")
