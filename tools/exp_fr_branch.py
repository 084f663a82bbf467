import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
from pysco_amd import solver
n = 512
param, tables = bench.make_param(n, n**3, "fr")
pos, vel = bench.zeldovich_device(n, rms_cells=0.5)
from pysco_amd import utils
utils.periodic_wrap(pos)
acc, phi, u = solver.pm(pos, param, tables=tables)
q = float(param["fR_q"]); h2 = 1.0 / n**2
print("fR_q", q, "d1 at 512", 27 * h2 * q, "u stats", float(u.min()), float(u.mean()), float(u.max()))
p = -(u.double() ** 2).mean()
print("p ~", float(p), "disc sign", float((27*h2*q)**2 + 108 * p**3))
for lvl in range(1, 8):
    nn = n >> lvl; print(nn, "d1", 27 * q / nn**2, "disc", (27*q/nn**2)**2 + 108*float(p)**3)
