import ctypes as C, sys, time
sys.path.insert(0, ".")
import torch
from pysco_amd import _lib as L
n = 512
ctx = L.context(4)
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.rand((n, n, n), generator=g, device="cuda"); b = torch.rand((n, n, n), generator=g, device="cuda")
out = C.c_float(0)
vals = {}
for fused in (0, 1, 0, 1):
    ctx.call("pm_set_fused_sweep", fused)
    for _ in range(3): ctx.call("pm_lap_residual_error", L.ptr(x), L.ptr(b), n, C.byref(out))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): ctx.call("pm_lap_residual_error", L.ptr(x), L.ptr(b), n, C.byref(out))
    print("march" if fused else "rows ", "ms", round((time.perf_counter() - t0) / 50 * 1e3, 4), repr(out.value))
    vals[fused] = out.value
ctx.call("pm_set_fused_sweep", 1)
# exact check against torch fp64
xd, bd = x.double(), b.double()
lap = (torch.roll(xd,1,0)+torch.roll(xd,-1,0)+torch.roll(xd,1,1)+torch.roll(xd,-1,1)+torch.roll(xd,1,2)+torch.roll(xd,-1,2)-6*xd)*n*n
print("fp64 reference", float(torch.sqrt(((-lap+bd)**2).sum())), "rel diff march/rows", abs(vals[1]-vals[0])/vals[0])
