#!/usr/bin/env python3
"""Prints ms/step, stage averages, the straggler series and the dt-guess errors of a bench.py JSON line (development)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms_per_step", round(d["ms_per_step"], 3), {s["stage"][3:30]: round(s["avg_ms"], 2) for s in d["roofline"]["stages"][:3]})
if "straggler_series" in d:
    print("stragglers", d["straggler_series"])
    ds = d.get("dt_series")
    if ds:
        print("dt guess error", [round(ds[i][1] / ds[i + 1][0] - 1, 4) for i in range(len(ds) - 1)])
for r in d.get("per_step", []):
    print(r["step"], r["stragglers"], r["wall_ms"], r["ms"])
