#!/usr/bin/env python3
"""Does the speed of the streaming update depend on WHICH arrays are combined, beyond each array's own placement?
Allocates candidates of 1024^3 x 3 floats, times the in-place pass over each (pm_scale by 1) and pm_kick_drift_wrap
(dt = 0: values unchanged) over triples (position, velocity, acceleration) of them.  (development tool)"""
import itertools, os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pysco_amd import _lib as L

np_ = 1024**3
ctx = L.context(4)
cand = [torch.rand((np_, 3), device="cuda") for _ in range(12)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best

single = [timed(lambda t=t: ctx.call("pm_scale", L.ptr(t), 1.0, t.numel())) for t in cand]
print("single", [round(s, 2) for s in single], flush=True)
order = sorted(range(len(cand)), key=lambda i: single[i])
fast = order[:8]
random.seed(1)
triples = random.sample(list(itertools.permutations(fast, 3)), 40)
res = []
for a, b, c in triples:
    ms = timed(lambda: ctx.call("pm_kick_drift_wrap", L.ptr(cand[a]), L.ptr(cand[b]), L.ptr(cand[c]), 0.0, 0.0, np_))
    res.append((ms, a, b, c))
    print(f"pos {a:2d} vel {b:2d} acc {c:2d}: {ms:.2f} ms   singles {single[a]:.2f} {single[b]:.2f} {single[c]:.2f}", flush=True)
res.sort()
print("best", res[:3], "worst", res[-3:])
