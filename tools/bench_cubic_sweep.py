import sys, os, time
sys.path.insert(0, "/root/repo")
import torch
from pysco_amd import _lib as L
n = 512
ctx = L.context(4)
g = torch.Generator(device="cuda"); g.manual_seed(1)
b = torch.rand((n, n, n), generator=g, device="cuda") * 0.5 + 0.2
x = torch.rand((n, n, n), generator=g, device="cuda") * 0.2 + 0.9
for q in (-0.1, -5.0):
    for _ in range(2):
        ctx.call("pm_cubic_gauss_seidel", L.ptr(x), L.ptr(b), q, None, n, 1.25)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        ctx.call("pm_cubic_gauss_seidel", L.ptr(x), L.ptr(b), q, None, n, 1.25)
    torch.cuda.synchronize()
    print(os.environ.get("PMSTEP_LIB", "base")[-16:], "q", q, "sweep ms", (time.perf_counter() - t0) / 10 * 1e3, float(x.mean()))
