"""GPU parity tests for the grid solvers: linear multigrid, f(R) cubic FAS, QUMOND, and solver.pm
for those theories -- HIP kernels (through the C ABI) vs the CPU oracle and the reference goldens."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import pysco_oracle as orc  # noqa: E402  (checker only)
from tests.helpers import param_from_json, synth_tables  # noqa: E402


@pytest.fixture(scope="module")
def amd():
    import torch

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from pysco_amd import cubic, laplacian, mesh, mond, multigrid, quartic, solver, utils

    class NS:
        pass

    ns = NS()
    ns.cubic, ns.laplacian, ns.mesh, ns.mond, ns.multigrid, ns.solver, ns.utils, ns.torch = (
        cubic, laplacian, mesh, mond, multigrid, solver, utils, torch)
    ns.quartic = quartic
    return ns


def _scale(x):
    return float(np.abs(x).max())


def test_transfer_and_laplacian_bit_exact(amd, golden):
    g = golden("multigrid_n16")
    x, b, xc = g["x"], g["b"], g["xc"]
    np.testing.assert_array_equal(amd.mesh.restriction(x), g["restriction"])
    np.testing.assert_array_equal(amd.mesh.minus_restriction(x), g["minus_restriction"])
    np.testing.assert_array_equal(amd.mesh.prolongation(xc), g["prolongation"])
    y = x.copy()
    amd.mesh.add_prolongation(y, xc)
    np.testing.assert_array_equal(y, g["add_prolongation"])
    np.testing.assert_array_equal(amd.laplacian.operator(x), g["operator"])
    np.testing.assert_array_equal(amd.laplacian.residual(x, b), g["residual"])
    np.testing.assert_array_equal(amd.laplacian.restrict_residual(x, b), g["restrict_residual"])
    np.testing.assert_array_equal(amd.laplacian.initialise_potential(b), g["initialise_potential"])
    y = x.copy()
    amd.laplacian.gauss_seidel(y, b, np.float32(1.25))
    np.testing.assert_array_equal(y, g["gauss_seidel_1"])
    y = x.copy()
    amd.laplacian.smoothing(y, b, 3)
    np.testing.assert_array_equal(y, g["smoothing_3"])
    np.testing.assert_allclose(amd.laplacian.residual_error(x, b), g["residual_error"], rtol=2e-6)
    np.testing.assert_allclose(amd.laplacian.truncation_error(x), g["truncation_error"], rtol=2e-6)
    for m in (4, 8):  # coarsest levels of the V-cycle
        y = g[f"small{m}_x"].copy()
        amd.laplacian.gauss_seidel(y, g[f"small{m}_b"], np.float32(1.25))
        np.testing.assert_array_equal(y, g[f"small{m}_gs"])
    np.testing.assert_array_equal(amd.laplacian.restrict_residual(g["small8_x"], g["small8_b"]),
                                  g["small8_restrict_residual"])
    y = g["small8_x"].copy()
    amd.mesh.add_prolongation(y, g["small4_x"])
    np.testing.assert_array_equal(y, g["small8_add_prolongation"])


def test_linear_multigrid_bit_exact(amd, golden):
    g = golden("multigrid_n16")
    param = param_from_json(g["lin_param_json"])
    param.pop("tolerance", None)
    phi = amd.torch.from_numpy(g["vc_phi0"].copy()).cuda()
    rhs = amd.torch.from_numpy(g["vc_rhs"]).cuda()
    amd.multigrid.V_cycle(phi, rhs, param)
    np.testing.assert_array_equal(phi.cpu().numpy(), g["vc_phi1"])
    phi = amd.torch.from_numpy(g["vc_phi0"].copy()).cuda()
    out = amd.multigrid.linear(phi, rhs, param)
    assert out is phi  # relaxed in place, same object returned (multigrid.py:77, 83)
    np.testing.assert_allclose(param["tolerance"], g["lin_tolerance"], rtol=1e-5)
    np.testing.assert_array_equal(phi.cpu().numpy(), g["lin_phi"])  # same cycle count, same bits
    with pytest.raises(ValueError):
        amd.multigrid.linear(phi, rhs, dict(param, theory="fr", compute_additional_field=True))


def test_multigrid_random_64_vs_oracle(amd):
    """64^3 random problem: every level of the V-cycle against the oracle, bit for bit."""
    rng = np.random.default_rng(5)
    n = 64
    b = rng.normal(size=(n, n, n)).astype(np.float32)
    b -= b.mean()
    param = {"theory": "newton", "compute_additional_field": False, "ncoarse": 6, "Npre": 2, "Npost": 1,
             "epsrel": 1e-2, "nsteps": 0}
    x_ref = orc.lap_initialise_potential(b)
    x = amd.torch.from_numpy(x_ref.copy()).cuda()
    orc.V_cycle(x_ref, b, dict(param))
    amd.multigrid.V_cycle(x, amd.torch.from_numpy(b).cuda(), dict(param))
    np.testing.assert_array_equal(x.cpu().numpy(), x_ref)


def test_cubic(amd, golden):
    g = golden("cubic_n16")
    u, b, rhs, q = g["u"], g["b"], g["rhs"], np.float32(g["q"])
    s = _scale(g["operator"])
    np.testing.assert_array_equal(amd.cubic.operator(u, b, q), orc.cubic_operator(u, b, q))
    np.testing.assert_array_equal(amd.cubic.residual_with_rhs(u, b, q, rhs), orc.cubic_residual_with_rhs(u, b, q, rhs))
    np.testing.assert_allclose(amd.cubic.operator(u, b, q), g["operator"], rtol=0, atol=3e-6 * s)
    np.testing.assert_allclose(amd.cubic.initialise_potential(b, q), g["initialise_potential"], rtol=2e-5)
    np.testing.assert_allclose(amd.cubic.initialise_potential(b, q), orc.cubic_initialise_potential(b, q), rtol=3e-7)
    for with_rhs in (False, True):
        y, yo = u.copy(), u.copy()
        if with_rhs:
            amd.cubic.gauss_seidel_with_rhs(y, b, q, rhs, np.float32(1.25))
            orc.cubic_gauss_seidel(yo, b, q, np.float32(1.25), rhs)
            ref = g["gauss_seidel_with_rhs"]
        else:
            amd.cubic.gauss_seidel(y, b, q, np.float32(1.25))
            orc.cubic_gauss_seidel(yo, b, q, np.float32(1.25))
            ref = g["gauss_seidel"]
        np.testing.assert_allclose(y, yo, rtol=0, atol=3e-7)  # device pow/acos/cos vs libm: <= 1-2 ulp of u ~ 1
        np.testing.assert_allclose(y, ref, rtol=0, atol=5e-6)
    np.testing.assert_allclose(amd.cubic.residual_error(u, b, q), g["residual_error"], rtol=1e-5)
    np.testing.assert_allclose(amd.cubic.truncation_error(u, b, q), g["truncation_error"], rtol=1e-4)


def test_cubic_root_branches(amd):
    """All three branches of solution_cubic_equation (incl. the NaN one) through a 1-sweep GS on a
    grid engineered so that cell (1,1,1) sees the requested (p, d1)."""
    g = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "cubic_n16.npz"))
    pd_, ref = g["root_pd"], g["root"]
    got = np.array([orc.solution_cubic_equation(p, d) for p, d in pd_], dtype=np.float32)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    # device: u = 0 everywhere -> p = h2*b, d1 = 27*h2*q - 27*rhs; relax with f=1 from x=0 -> x = root
    n = 16
    h2 = np.float32(1.0 / n**2)
    sel = np.arange(0, len(pd_), 7)[: n**3 // 2]
    b = np.zeros((n, n, n), np.float32)
    rhs = np.zeros((n, n, n), np.float32)
    cells = [(i, j, k) for i in range(n) for j in range(n) for k in range(n) if (i + j + k) % 2 == 1][: len(sel)]
    for (i, j, k), s in zip(cells, sel):
        b[i, j, k] = pd_[s, 0] / h2
        rhs[i, j, k] = -pd_[s, 1] / np.float32(27)
    x = np.zeros((n, n, n), np.float32)
    xo = x.copy()
    amd.cubic.gauss_seidel_with_rhs(x, b, np.float32(0), rhs, np.float32(1))
    orc.cubic_gauss_seidel(xo, b, np.float32(0), np.float32(1), rhs)
    red = np.array([x[c] for c in cells])
    red_o = np.array([xo[c] for c in cells])
    assert np.array_equal(np.isnan(red), np.isnan(red_o))
    ok = ~np.isnan(red_o)
    np.testing.assert_allclose(red[ok], red_o[ok], rtol=3e-6, atol=1e-6)
    assert ok.sum() > 100 and (~ok).sum() > 0


def test_quartic(amd, golden):
    g = golden("quartic_n16")
    u, b, rhs, q = g["u"], g["b"], g["rhs"], np.float32(g["q"])
    s = _scale(g["operator"])
    np.testing.assert_array_equal(amd.quartic.operator(u, b, q), orc.cubic_operator(u, b, q, 2))
    np.testing.assert_array_equal(amd.quartic.residual_with_rhs(u, b, q, rhs), orc.cubic_residual_with_rhs(u, b, q, rhs, 2))
    np.testing.assert_allclose(amd.quartic.operator(u, b, q), g["operator"], rtol=0, atol=5e-6 * s)
    np.testing.assert_allclose(amd.quartic.initialise_potential(b, q), g["initialise_potential"], rtol=2e-5)
    np.testing.assert_allclose(amd.quartic.initialise_potential(b, q), orc.quartic_initialise_potential(b, q), rtol=3e-7)
    for with_rhs in (False, True):
        y, yo = u.copy(), u.copy()
        if with_rhs:
            amd.quartic.gauss_seidel_with_rhs(y, b, q, rhs, np.float32(1.25))
            orc.cubic_gauss_seidel(yo, b, q, np.float32(1.25), rhs, 2)
            ref = g["gauss_seidel_with_rhs"]
        else:
            amd.quartic.gauss_seidel(y, b, q, np.float32(1.25))
            orc.cubic_gauss_seidel(yo, b, q, np.float32(1.25), None, 2)
            ref = g["gauss_seidel"]
        np.testing.assert_allclose(y, yo, rtol=0, atol=3e-7)
        np.testing.assert_allclose(y, ref, rtol=0, atol=5e-6)
    np.testing.assert_allclose(amd.quartic.residual_error(u, b, q), g["residual_error"], rtol=1e-5)
    np.testing.assert_allclose(amd.quartic.truncation_error(u, b, q), g["truncation_error"], rtol=1e-4)


def test_mond(amd, golden):
    g = golden("mond_n16")
    for name, fn, kw, alpha in (("simple", amd.mond.rhs_simple, {}, 1), ("n", amd.mond.rhs_n, {"n": 2}, 2),
                                ("beta", amd.mond.rhs_beta, {"beta": 1.3}, 1.3),
                                ("gamma", amd.mond.rhs_gamma, {"gamma": 1.3}, 1.3),
                                ("delta", amd.mond.rhs_delta, {"delta": 1.3}, 1.3)):
        out = np.empty_like(g["pot"])
        fn(g["pot"], out, np.float32(g["g0"]), **kw)
        ref = g["rhs_" + name]
        np.testing.assert_allclose(out, ref, rtol=0, atol=2e-5 * _scale(ref), err_msg=name)
        oo = np.empty_like(out)
        orc.mond_rhs(g["pot"], oo, np.float32(g["g0"]), name, alpha)
        np.testing.assert_allclose(out, oo, rtol=0, atol=4e-6 * _scale(ref), err_msg=name + " vs oracle")


PM_MG = ["newton_multigrid", "fr_multigrid", "fr2_multigrid", "mond_multigrid", "mond_fft7pt"]


@pytest.mark.parametrize("tag", PM_MG)
@pytest.mark.parametrize("resident", [False, True])
def test_pm_theories_vs_golden(amd, golden, tag, resident):
    g = golden("pm_" + tag)
    tables = synth_tables()
    param = param_from_json(g["param0_json"])
    up = (lambda a: amd.torch.from_numpy(np.ascontiguousarray(a)).cuda()) if resident else (lambda a: a)
    host = (lambda t: t.cpu().numpy()) if resident else (lambda t: t)
    acc, pot, extra = amd.solver.pm(up(g["pos"]), param, tables=tables)
    sa, sp = _scale(g["acc1"]), _scale(g["pot1"])
    np.testing.assert_allclose(host(pot), g["pot1"], rtol=0, atol=2e-4 * sp)
    np.testing.assert_allclose(host(acc), g["acc1"], rtol=0, atol=1e-3 * sa)
    if g["extra1"].size:
        np.testing.assert_allclose(host(extra), g["extra1"], rtol=0, atol=2e-4 * _scale(g["extra1"]))
    else:
        assert host(extra).size == 0
    ref1 = param_from_json(g["param1_json"])
    for key in ("tolerance", "tolerance_FAS", "tolerance_mond", "fR_q"):
        if key in ref1:
            np.testing.assert_allclose(float(param[key]), float(ref1[key]), rtol=1e-3, err_msg=key)
    # second call: warm start from the previous potential / extra field (solver.py:273-280)
    param["nsteps"], param["aexp_old"], param["aexp"] = 1, param["aexp"], 0.51
    amd.utils.set_units(param)
    acc, pot, extra = amd.solver.pm(up(g["pos2"]), param, pot, extra, tables)
    np.testing.assert_allclose(host(pot), g["pot2"], rtol=0, atol=5e-4 * sp)
    np.testing.assert_allclose(host(acc), g["acc2"], rtol=0, atol=2e-3 * sa)


def test_pm_theory_errors(amd, golden):
    g = golden("pm_mond_multigrid")
    base = param_from_json(g["param0_json"])
    p = dict(base, mond_function="zeta")
    with pytest.raises(NotImplementedError):
        amd.solver.pm(g["pos"], p, tables=synth_tables())
    p = dict(base, linear_newton_solver="fft")
    with pytest.raises(NotImplementedError):
        amd.solver.pm(g["pos"], p, tables=synth_tables())
    gf = golden("pm_fr_multigrid")
    p = dict(param_from_json(gf["param0_json"]), fR_n=3)
    with pytest.raises(NotImplementedError):
        amd.solver.pm(gf["pos"], p, tables=synth_tables())


def test_multigrid_solves_the_7pt_problem_at_scale(amd):
    """256^3, beyond the oracle: the multigrid solver and the FFT with the 7-point Green's function solve the same
    discrete Poisson problem (laplacian.operator is the 7-point stencil), so their potentials must agree to the
    multigrid's stopping tolerance, and applying the operator to either must give back the zero-mean source."""
    from tests.helpers import zeldovich_particles

    t = amd.torch
    n = 256
    pos_h, _ = zeldovich_particles(n, seed=8, rms_cells=1.0)
    pos = t.from_numpy(pos_h).cuda()
    base = {"theory": "newton", "mass_scheme": "TSC", "npart": n**3, "ncoarse": 8, "Om_m": 0.25733, "aexp": 0.5,
            "gradient_stencil_order": 5, "save_power_spectrum": "no", "nthreads": 1, "Npre": 2, "Npost": 1,
            "epsrel": 1e-3, "nsteps": 0}
    acc_f, phi_f, _ = amd.solver.pm(pos, dict(base, linear_newton_solver="fft_7pt"))
    acc_m, phi_m, _ = amd.solver.pm(pos, dict(base, linear_newton_solver="multigrid"))
    phi_m = phi_m - phi_m.mean()          # the multigrid solution carries an arbitrary constant
    phi_f = phi_f - phi_f.mean()
    scale = float(phi_f.abs().max())
    assert float((phi_m - phi_f).abs().max()) < 2e-3 * scale
    assert float((acc_m - acc_f).abs().max()) < 2e-2 * float(acc_f.abs().max())
    # L(phi_fft7) reproduces the source f1 (rho - 1) to fp32 round-off of the transform
    rho = amd.mesh.TSC(pos, n)
    f1 = np.float32(1.5 * base["aexp"] * base["Om_m"])
    src = f1 * (rho - 1.0)
    lap = amd.laplacian.operator(phi_f)
    assert float((lap - src).abs().max()) < 2e-3 * float(src.abs().max())



@pytest.mark.parametrize("n", [8, 16, 64, 128])
def test_native_vcycle_is_the_python_recursion(amd, n):
    """pm_lap_vcycle (the whole V-cycle issued from C) against multigrid.V_cycle's recursion written out in Python
    over the per-operator entry points: same kernels in the same order, so the same bits."""
    rng = np.random.default_rng(n)
    t = amd.torch
    b = t.from_numpy(rng.normal(size=(n, n, n)).astype(np.float32)).cuda()
    b -= b.mean()
    x0 = t.from_numpy(rng.normal(size=(n, n, n)).astype(np.float32)).cuda() * 1e-3
    param = {"Npre": 2, "Npost": 1, "ncoarse": int(np.log2(n))}
    out = {}
    for native in (True, False):
        amd.multigrid._NATIVE_VCYCLE = native
        try:
            x = x0.clone()
            amd.multigrid.V_cycle(x, b, param)
            amd.multigrid.V_cycle(x, b, param)
            out[native] = x.cpu().numpy()
        finally:
            amd.multigrid._NATIVE_VCYCLE = True
    np.testing.assert_array_equal(out[True], out[False])
    assert np.isfinite(out[True]).all()


@pytest.mark.parametrize("slab,frn", [(False, 1), (True, 1), (False, 2)])
def test_marching_fr_operator_is_the_cell_kernels(amd, slab, frn):
    """cubic / quartic operator, residual with right-hand side and residual norm (cubic.py:23-158, 844-900) at 512^3: the
    marching kernel against the cell-per-thread one (pm_set_fr_rows_sweep 0) -- grids bit for bit, the fp64 sum of the
    norm to rounding; on a cube and on a slab level array with ghost planes (40 planes: a short second segment)."""
    import ctypes as C

    from pysco_amd import _lib as L

    t = amd.torch
    n, nx = 512, 40
    g = t.Generator(device="cuda")
    g.manual_seed(7 + frn)
    shape = (nx + 2, n, n) if slab else (n, n, n)
    x = 1e-2 * (1 + 0.3 * t.rand(shape, generator=g, device="cuda"))
    b = t.rand(shape, generator=g, device="cuda") - 0.5
    rhs = t.rand(shape, generator=g, device="cuda")
    q = -0.0123
    ctx = L.context(4)
    got = {}
    for march in (1, 0):
        ctx.call("pm_set_fr_rows_sweep", march)
        try:
            outs = []
            for r in (None, rhs):
                o = t.full(shape, float("nan"), device="cuda")
                if slab:
                    ctx.call("pm_fr_operator_slab", frn, L.ptr(x), L.ptr(b), q, L.ptr(r) if r is not None else None, n, nx, L.ptr(o))
                    o = o[1:-1]
                else:
                    ctx.call("pm_cubic_operator" if frn == 1 else "pm_quartic_operator", L.ptr(x), L.ptr(b), q,
                             L.ptr(r) if r is not None else None, n, L.ptr(o))
                outs.append(o.clone())
            if slab:
                sq = C.c_double(0)
                ctx.call("pm_fr_residual_sumsq_slab", frn, L.ptr(x), L.ptr(b), q, n, nx, C.byref(sq))
                norm2 = sq.value
            else:
                e = C.c_float(0)
                ctx.call("pm_cubic_residual_error" if frn == 1 else "pm_quartic_residual_error", L.ptr(x), L.ptr(b), q, n, C.byref(e))
                norm2 = float(e.value) ** 2
            got[march] = (outs, norm2)
        finally:
            ctx.call("pm_set_fr_rows_sweep", 1)
    for a, c in zip(got[1][0], got[0][0]):
        assert bool(t.isfinite(a).all()) and t.equal(a, c)
    assert abs(got[1][1] - got[0][1]) <= (1e-12 if slab else 3e-7) * got[0][1]
    # the norm is the norm of the operator's output
    assert abs(got[1][1] - float((got[1][0][0].double() ** 2).sum())) <= 1e-6 * got[1][1]


@pytest.mark.parametrize("slab", [False, True])
def test_marching_laplacian_operator_is_the_cell_kernel(amd, slab):
    """laplacian.operator and residual (laplacian.py:15-122) at 512^3 through the marching kernel against the
    cell-per-thread one, bit for bit; cube and slab level array (40 planes, ghosts either side)."""
    from pysco_amd import _lib as L

    t = amd.torch
    n, nx = 512, 40
    g = t.Generator(device="cuda")
    g.manual_seed(11)
    shape = (nx + 2, n, n) if slab else (n, n, n)
    x = t.rand(shape, generator=g, device="cuda")
    b = t.rand(shape, generator=g, device="cuda")
    ctx = L.context(4)
    got = {}
    for march in (1, 0):
        ctx.call("pm_set_fused_sweep", march)
        try:
            outs = []
            for mode in (0, 1):
                o = t.full(shape, float("nan"), device="cuda")
                if slab:
                    ctx.call("pm_lap_operator_slab", L.ptr(x), L.ptr(b), n, nx, mode, L.ptr(o))
                    o = o[1:-1]
                else:
                    ctx.call("pm_lap_operator", L.ptr(x), L.ptr(b) if mode else None, n, mode, L.ptr(o))
                outs.append(o.clone())
            got[march] = outs
        finally:
            ctx.call("pm_set_fused_sweep", 1)
    for a, c in zip(got[1], got[0]):
        assert bool(t.isfinite(a).all()) and t.equal(a, c)


@pytest.mark.parametrize("slab", [False, True])
def test_marching_residual_norm_is_the_row_kernels(amd, slab):
    """laplacian.residual_error at 512^3 (the convergence test after every V-cycle): the marching kernel (a plane of x
    loaded once, neighbours through registers and LDS) against the row kernel it replaces there -- the same per-cell
    expression, fp64 sums in another order -- and against a float64 evaluation; on a cube and on a slab with ghost
    planes (pm_lap_residual_sumsq_slab: 40 planes, two segments of the march, the second one short)."""
    import ctypes as C

    from pysco_amd import _lib as L

    t = amd.torch
    n, nx = 512, 40
    g = t.Generator(device="cuda")
    g.manual_seed(5)
    shape = (nx + 2, n, n) if slab else (n, n, n)
    x = t.rand(shape, generator=g, device="cuda")
    b = t.rand(shape, generator=g, device="cuda")
    ctx = L.context(4)
    got = {}
    for march in (1, 0):
        ctx.call("pm_set_fused_sweep", march)
        try:
            if slab:
                s = C.c_double(0)
                ctx.call("pm_lap_residual_sumsq_slab", L.ptr(x), L.ptr(b), n, nx, C.byref(s))
                got[march] = s.value
            else:
                o = C.c_float(0)
                ctx.call("pm_lap_residual_error", L.ptr(x), L.ptr(b), n, C.byref(o))
                got[march] = float(o.value) ** 2
        finally:
            ctx.call("pm_set_fused_sweep", 1)
    xd, bd = x.double(), b.double()
    if slab:   # owned planes 1 .. nx, neighbours in x are the ghost planes
        c = xd[1:-1]
        lap = xd[:-2] + xd[2:] + t.roll(c, 1, 1) + t.roll(c, -1, 1) + t.roll(c, 1, 2) + t.roll(c, -1, 2) - 6 * c
        want = float(((-lap * n * n + bd[1:-1]) ** 2).sum())
    else:
        lap = sum(t.roll(xd, sh, ax) for ax in range(3) for sh in (1, -1)) - 6 * xd
        want = float(((-lap * n * n + bd) ** 2).sum())
    assert abs(got[1] - got[0]) <= (1e-12 if slab else 3e-7) * got[0]
    assert abs(got[1] - want) <= 2e-6 * want   # (fp32 residuals per cell against fp64 ones)


@pytest.mark.parametrize("n,sweeps", [(256, 1), (256, 2), (256, 3), (512, 2)])
def test_single_pass_sweep_is_the_two_half_sweeps(amd, n, sweeps):
    """The single-pass red + black sweep (pm_lap_smoothing, n >= 256: one workgroup marching a column through LDS
    planes, out of place) against the two half-sweep launches per sweep: the same bits, whole grid."""
    from pysco_amd import _lib as L

    t = amd.torch
    g = t.Generator(device="cuda")
    g.manual_seed(n + sweeps)
    b = t.randn((n, n, n), generator=g, device="cuda")
    b -= b.mean()
    x0 = t.randn((n, n, n), generator=g, device="cuda") * 1e-4
    out = {}
    ctx = L.context(4)
    for fused in (1, 0):
        ctx.call("pm_set_fused_sweep", fused)
        try:
            x = x0.clone()
            amd.laplacian.smoothing(x, b, sweeps)
            out[fused] = x
        finally:
            ctx.call("pm_set_fused_sweep", 1)
    assert bool(t.isfinite(out[1]).all())
    assert bool((out[1] == out[0]).all()), float((out[1] - out[0]).abs().max())
    # and a whole V-cycle through both
    if sweeps == 2:
        param = {"Npre": 2, "Npost": 1, "ncoarse": int(np.log2(n))}
        vc = {}
        for fused in (1, 0):
            ctx.call("pm_set_fused_sweep", fused)
            try:
                x = x0.clone()
                amd.multigrid.V_cycle(x, b, param)
                vc[fused] = x
            finally:
                ctx.call("pm_set_fused_sweep", 1)
        assert bool((vc[1] == vc[0]).all())


@pytest.mark.parametrize("n,fr_n,with_rhs,sweeps", [(256, 1, False, 1), (256, 1, False, 2), (256, 1, True, 3), (256, 2, True, 1),
                                                   (256, 2, False, 2), (512, 1, False, 2)])
def test_fr_single_pass_sweep_is_the_two_half_sweeps(amd, n, fr_n, with_rhs, sweeps):
    """The f(R) smoothers' single-pass red + black sweep (pm_fr_smoothing, n >= 256: the marching kernel of the linear
    sweep with the cubic / quartic cell update) against the two half-sweep launches per sweep: the same bits, whole
    grid; and a whole FAS V-cycle (pm_fr_vcycle: ping-pong sweeps, out-of-place prolongation, graph replay) through
    both."""
    from pysco_amd import _lib as L

    t = amd.torch
    g = t.Generator(device="cuda")
    g.manual_seed(3 * n + 7 * fr_n + sweeps)
    b = t.rand((n, n, n), generator=g, device="cuda") * 0.5 + 0.2
    x0 = t.rand((n, n, n), generator=g, device="cuda") * 0.2 + 0.9
    rhs = (t.rand((n, n, n), generator=g, device="cuda") - 0.5) * 1e-3 if with_rhs else None
    ctx = L.context(4)
    out, vc = {}, {}
    try:
        for fused in (1, 0):
            ctx.call("pm_set_fused_sweep", fused)
            x = x0.clone()
            ctx.call("pm_fr_smoothing", fr_n, L.ptr(x), L.ptr(b), -0.1, L.ptr(rhs), n, sweeps)
            out[fused] = x
            if sweeps == 2 and not with_rhs and fr_n == 1:  # (the quartic FAS cycle of this random data ends in the reference's NaNs)
                x = x0.clone()
                for _ in range(2):  # the second call replays the captured graph
                    ctx.call("pm_fr_vcycle", fr_n, L.ptr(x), L.ptr(b), -0.1, n, 2, 1)
                vc[fused] = x
    finally:
        ctx.call("pm_set_fused_sweep", 1)
    assert bool(t.isfinite(out[1]).all()) and float((out[1] - x0).abs().max()) > 0
    assert bool((out[1] == out[0]).all()), float((out[1] - out[0]).abs().max())
    if vc:
        assert bool(t.isfinite(vc[1]).all()) and float((vc[1] - x0).abs().max()) > 0
        assert bool((vc[1] == vc[0]).all()), float((vc[1] - vc[0]).abs().max())


@pytest.mark.parametrize("fr_n", [1, 2])
def test_fr_fused_restriction_is_operator_plus_restrictions(amd, fr_n):
    """pm_fr_vcycle at 512^3 with the restricted operator / residual, solution and density term taken in one marching
    pass (fr_restrict_march_kernel) against the operator kernel + three pm_restriction launches: the same bits after two
    cycles (NaNs of the quartic's random data in the same places)."""
    from pysco_amd import _lib as L

    t = amd.torch
    n = 512
    g = t.Generator(device="cuda")
    g.manual_seed(40 + fr_n)
    b = t.rand((n, n, n), generator=g, device="cuda") * 0.5 + 0.2
    x0 = t.rand((n, n, n), generator=g, device="cuda") * 0.2 + 0.9
    ctx = L.context(4)
    out = {}
    try:
        for fused in (1, 0):
            ctx.call("pm_set_fr_fused_restrict", fused)
            x = x0.clone()
            for _ in range(2):
                ctx.call("pm_fr_vcycle", fr_n, L.ptr(x), L.ptr(b), -0.1, n, 2, 1)
            out[fused] = x
    finally:
        ctx.call("pm_set_fr_fused_restrict", 1)
    same = (out[1] == out[0]) | (t.isnan(out[1]) & t.isnan(out[0]))
    assert bool(same.all()), int((~same).sum())
    if fr_n == 1:
        assert bool(t.isfinite(out[1]).all()) and float((out[1] - x0).abs().max()) > 0


def test_mond_and_fR_at_benchmark_size(amd):
    """BASELINE configs 4 and 5 at their full 512^3, where the oracle cannot follow: size-independent properties.
    QUMOND: the multigrid and the fft_7pt solver solve the same two discrete Poisson problems (Newtonian potential,
    then the MOND source), so accelerations agree to the multigrid's stopping tolerance, and the Newtonian potential
    satisfies its 7-point equation.  f(R): the FAS solve leaves a scalaron whose residual is below its tolerance, the
    field is finite and positive, and the force responds (differs from the Newtonian one by the fifth force)."""
    import bench

    t = amd.torch
    n = 512
    pos, vel = bench.zeldovich_device(n, rms_cells=1.0)
    amd.utils.periodic_wrap(pos)
    pos, vel = amd.utils.reorder_particles(pos, vel)
    del vel
    # ---- QUMOND
    p_mg, tables = bench.make_param(n, n**3, "mond")
    p_mg["epsrel"] = 1e-3
    p_ff = dict(p_mg, linear_newton_solver="fft_7pt")
    acc_m, phi_m, newt_m = amd.solver.pm(pos, p_mg, tables=tables)
    acc_f, phi_f, newt_f = amd.solver.pm(pos, p_ff, tables=tables)
    for a, b_ in ((newt_m, newt_f), (phi_m, phi_f)):
        a, b_ = a - a.mean(), b_ - b_.mean()
        assert float((a - b_).abs().max()) < 3e-3 * float(b_.abs().max())
    assert float((acc_m - acc_f).abs().max()) < 3e-2 * float(acc_f.abs().max())
    assert bool(t.isfinite(acc_m).all())
    rho = amd.mesh.TSC(pos, n)
    src = np.float32(1.5 * p_mg["aexp"] * p_mg["Om_m"]) * (rho - 1.0)
    lap = amd.laplacian.operator(newt_f)
    assert float((lap - src).abs().max()) < 3e-3 * float(src.abs().max())
    # (the stopping rule also ends the cycles when one no longer halves the residual, multigrid.py:78: the bound that
    # always holds is relative to the source)
    res = float(amd.laplacian.residual_error(newt_m, src))
    assert res <= max(1.05 * float(p_mg["tolerance"]), 1e-3 * float(src.double().pow(2).sum().sqrt())), (res, p_mg["tolerance"])
    del acc_f, phi_f, newt_f, lap, src, rho, phi_m, newt_m
    # ---- f(R)
    p_fr, tables = bench.make_param(n, n**3, "fr")
    acc_r, phi_r, u = amd.solver.pm(pos, p_fr, tables=tables)
    assert bool(t.isfinite(u).all()) and float(u.min()) > 0 and bool(t.isfinite(acc_r).all())
    assert p_fr["_vcycles_fas"] >= 1 and p_fr["tolerance_FAS"] > 0
    p_nw = dict(p_fr, theory="newton")
    acc_n, _, _ = amd.solver.pm(pos, p_nw, tables=tables)
    rel = float((acc_r - acc_n).abs().max()) / float(acc_n.abs().max())
    assert 1e-4 < rel < 0.5, rel  # the fifth force is there, and it is a correction


@pytest.mark.parametrize("n,fr_n", [(32, 1), (64, 1), (32, 2)])
def test_native_fas_vcycle_is_the_python_recursion(amd, n, fr_n):
    """pm_fr_vcycle (the whole FAS V-cycle issued from C) against multigrid.V_cycle_FAS's recursion written out in
    Python over the per-operator entry points (multigrid.py:521-612), on the same scalaron guess and density: bit for
    bit (no atomics anywhere in the cycle, so both orders of issue give the same floats)."""
    torch = amd.torch
    g = torch.Generator(device="cuda")
    g.manual_seed(100 * n + fr_n)
    b = torch.rand((n, n, n), generator=g, device="cuda") * 0.5 + 0.2
    x0 = torch.rand((n, n, n), generator=g, device="cuda") * 0.2 + 0.9
    param = {"Npre": 2, "Npost": 1, "ncoarse": int(np.log2(n)), "fR_n": fr_n, "fR_q": np.float32(-0.1),
             "theory": "fr", "compute_additional_field": True}
    out = {}
    for native in (True, False):
        amd.multigrid._NATIVE_VCYCLE = native
        try:
            x = x0.clone()
            for _ in range(2):
                amd.multigrid.V_cycle_FAS(x, b, param)
            out[native] = x.cpu().numpy()
        finally:
            amd.multigrid._NATIVE_VCYCLE = True
    assert np.isfinite(out[True]).all()
    assert np.abs(out[True] - x0.cpu().numpy()).max() > 0
    np.testing.assert_array_equal(out[True], out[False])


def test_vcycle_graph_replay_is_the_plain_cycle(amd):
    """pm_lap_vcycle replayed from its captured hipGraph (second and later calls on the same arrays) against plain
    launches: same bits, for two grid sizes sharing one context, and the replay counter moves."""
    import ctypes as C

    torch = amd.torch
    from pysco_amd import _lib as L

    ctx = L.context(4)
    for n in (32, 128):
        g = torch.Generator(device="cuda")
        g.manual_seed(n)
        b = torch.rand((n, n, n), generator=g, device="cuda") - 0.5
        x0 = torch.rand((n, n, n), generator=g, device="cuda") * 1e-3
        out = {}
        for mode in (0, 1):
            ctx.call("pm_set_vcycle_graphs", mode)
            x = x0.clone()
            rep0, cap0 = C.c_uint(0), C.c_uint(0)
            ctx.call("pm_vcycle_graph_stats", C.byref(rep0), C.byref(cap0))
            for _ in range(4):
                ctx.call("pm_lap_vcycle", L.ptr(x), L.ptr(b), n, 2, 1)
            rep, cap = C.c_uint(0), C.c_uint(0)
            ctx.call("pm_vcycle_graph_stats", C.byref(rep), C.byref(cap))
            out[mode] = (x.cpu().numpy(), rep.value - rep0.value, cap.value - cap0.value)
        ctx.call("pm_set_vcycle_graphs", 1)
        assert out[0][1:] == (0, 0)
        assert out[1][1:] == (3, 1), out[1][1:]  # captured once, replayed three times
        np.testing.assert_array_equal(out[0][0], out[1][0])
        assert np.isfinite(out[1][0]).all() and np.abs(out[1][0] - x0.cpu().numpy()).max() > 0


@pytest.mark.parametrize("n", [8, 16, 32, 128])
def test_fused_coarse_levels_are_the_launches(amd, n):
    """pm_lap_vcycle with every level from 16^3 down run by one workgroup in LDS (lap_vcycle_coarse_kernel) against
    the same cycle issued level by level: bit for bit, for top-level grids of 8^3 (one kernel is the whole cycle) to
    128^3, plain launches and graph replay alike."""
    torch = amd.torch
    from pysco_amd import _lib as L

    ctx = L.context(4)
    g = torch.Generator(device="cuda")
    g.manual_seed(7 * n)
    b = torch.rand((n, n, n), generator=g, device="cuda") - 0.5
    x0 = (torch.rand((n, n, n), generator=g, device="cuda") - 0.5) * 1e-3
    out = {}
    try:
        for fused in (0, 1):
            ctx.call("pm_set_fused_coarse", fused)
            for graphs in (0, 1):
                ctx.call("pm_set_vcycle_graphs", graphs)
                x = x0.clone()
                for _ in range(3):
                    ctx.call("pm_lap_vcycle", L.ptr(x), L.ptr(b), n, 2, 1)
                out[(fused, graphs)] = x.cpu().numpy()
    finally:
        ctx.call("pm_set_fused_coarse", 1)
        ctx.call("pm_set_vcycle_graphs", 1)
    ref = out[(0, 0)]
    assert np.isfinite(ref).all() and np.abs(ref - x0.cpu().numpy()).max() > 0
    for key, val in out.items():
        np.testing.assert_array_equal(val, ref, err_msg=str(key))


@pytest.mark.parametrize("n,fr_n,with_rhs", [(16, 1, False), (64, 1, True), (64, 2, False), (32, 2, True), (128, 1, False)])
def test_fr_row_sweep_is_the_cell_sweep(amd, n, fr_n, with_rhs):
    """cubic / quartic Gauss-Seidel sweeps with the row kernels (eight consecutive k-cells per thread, 16-byte
    accesses) against the cell-per-thread kernels: bit for bit, with and without a right-hand side."""
    torch = amd.torch
    from pysco_amd import _lib as L

    ctx = L.context(4)
    g = torch.Generator(device="cuda")
    g.manual_seed(n + 10 * fr_n)
    b = torch.rand((n, n, n), generator=g, device="cuda") * 0.5 + 0.2
    x0 = torch.rand((n, n, n), generator=g, device="cuda") * 0.2 + 0.9
    rhs = (torch.rand((n, n, n), generator=g, device="cuda") - 0.5) * 1e-3 if with_rhs else None
    name = "pm_cubic_gauss_seidel" if fr_n == 1 else "pm_quartic_gauss_seidel"
    out = {}
    try:
        for rows in (0, 1):
            ctx.call("pm_set_fr_rows_sweep", rows)
            x = x0.clone()
            for _ in range(3):
                ctx.call(name, L.ptr(x), L.ptr(b), -0.1, L.ptr(rhs), n, 1.25)
            out[rows] = x.cpu().numpy()
    finally:
        ctx.call("pm_set_fr_rows_sweep", 1)
    assert np.isfinite(out[0]).all() and np.abs(out[0] - x0.cpu().numpy()).max() > 0
    np.testing.assert_array_equal(out[0], out[1])


@pytest.mark.parametrize("kind,alpha", [(0, 1.0), (1, 2.0), (2, 1.3), (3, 1.3), (4, 1.3)])
def test_mond_rhs_rows_is_the_cell_kernel(amd, kind, alpha):
    """pm_mond_rhs with four k-cells per thread (nine float4 rows + ten scalars for four cells) against the
    cell-per-thread kernel: bit for bit, every interpolating-function family."""
    torch = amd.torch
    from pysco_amd import _lib as L

    n = 64
    ctx = L.context(4)
    g = torch.Generator(device="cuda")
    g.manual_seed(kind)
    pot = torch.randn((n, n, n), generator=g, device="cuda") * 1e-3
    out = {}
    try:
        for rows in (0, 1):
            ctx.call("pm_set_fr_rows_sweep", rows)
            o = torch.empty_like(pot)
            ctx.call("pm_mond_rhs", L.ptr(pot), L.ptr(o), n, 1e-2, kind, alpha)
            out[rows] = o.cpu().numpy()
    finally:
        ctx.call("pm_set_fr_rows_sweep", 1)
    assert np.isfinite(out[0]).all() and np.abs(out[0]).max() > 0
    np.testing.assert_array_equal(out[0], out[1])


@pytest.mark.parametrize("n", [64, 256])
def test_mond_fast_nu_is_the_exact_chain_to_rounding(amd, n):
    """pm_mond_rhs for the simple function with the hardware's rsq / sqrt (the default) against the exact chain (IEEE
    sqrt, divide, sqrt): the 1.5e-7 relative difference of nu, seen through the divergence's cancellation -- a few 1e-6
    of the largest |nu grad phi| / h term per cell -- on a smooth potential with the dynamic range of a real one."""
    torch = amd.torch
    from pysco_amd import _lib as L

    ctx = L.context(4)
    g = torch.Generator(device="cuda")
    g.manual_seed(n)
    k = torch.arange(n, device="cuda", dtype=torch.float32) * (2 * np.pi / n)
    pot = (torch.sin(k)[:, None, None] * torch.cos(2 * k)[None, :, None] + 0.3 * torch.sin(3 * k + 1)[None, None, :]
           + 0.05 * torch.randn((n, n, n), generator=g, device="cuda")).contiguous() * 1e-2
    out = {}
    try:
        for fast in (1, 0):
            ctx.call("pm_set_mond_fast_nu", fast)
            o = torch.empty_like(pot)
            ctx.call("pm_mond_rhs", L.ptr(pot), L.ptr(o), n, 1e-2, 0, 1.0)
            out[fast] = o.double()
    finally:
        ctx.call("pm_set_mond_fast_nu", 1)
    assert bool(torch.isfinite(out[0]).all()) and float(out[0].abs().max()) > 0
    # the terms that cancel in a cell are nu * grad / h ~ n * nu * |grad phi|: bound them by n^2 * max |delta phi| * nu_max
    dphi = float((pot - torch.roll(pot, 1, 0)).abs().max())
    scale = float(out[0].abs().max())
    err = float((out[1] - out[0]).abs().max())
    assert err <= 2e-6 * max(scale, n * n * dphi), (err, scale)
    assert float((out[1] - out[0]).abs().mean()) <= 2e-7 * float(out[0].abs().mean())


@pytest.mark.parametrize("n", [16, 64])
def test_native_solve_loops_are_the_python_loops(amd, golden, n):
    """pm_lap_solve / pm_fr_solve (the stopping loops of multigrid.linear / multigrid.FAS issued from C: V-cycle, residual
    norm, "under the tolerance or no longer halving") against the same loops written out in Python: the same number
    of V-cycles and the same bits; at 16^3 also the reference's converged solve and its tolerance (golden)."""
    torch = amd.torch
    rng = np.random.default_rng(3 + n)
    if n == 16:
        g = golden("multigrid_n16")
        base = param_from_json(g["lin_param_json"])
        base.pop("tolerance", None)
        x0, b = g["vc_phi0"].copy(), g["vc_rhs"]
    else:
        b = rng.normal(size=(n, n, n)).astype(np.float32)
        b -= b.mean()
        x0 = orc.lap_initialise_potential(b)
        base = {"theory": "newton", "compute_additional_field": False, "ncoarse": int(np.log2(n)), "Npre": 2, "Npost": 1,
                "epsrel": 1e-3, "nsteps": 0}
    out = {}
    for native in (True, False):
        amd.multigrid._NATIVE_SOLVE = native
        try:
            param = dict(base)
            x = torch.from_numpy(x0.copy()).cuda()
            amd.multigrid.linear(x, torch.from_numpy(b).cuda(), param)
            out[native] = (x.cpu().numpy(), param["_vcycles_linear"], param["tolerance"])
        finally:
            amd.multigrid._NATIVE_SOLVE = True
    assert out[True][1] == out[False][1] and out[True][1] >= 2          # the same number of V-cycles
    np.testing.assert_array_equal(out[True][0], out[False][0])
    if n == 16:
        np.testing.assert_array_equal(out[True][0], g["lin_phi"])        # ... and the reference's converged solve
    # FAS (the scalaron)
    gq = torch.Generator(device="cuda")
    gq.manual_seed(n)
    dens = torch.rand((n, n, n), generator=gq, device="cuda") * 0.5 + 0.2
    u0 = torch.rand((n, n, n), generator=gq, device="cuda") * 0.2 + 0.9
    fas = {}
    for native in (True, False):
        amd.multigrid._NATIVE_SOLVE = native
        try:
            param = {"Npre": 2, "Npost": 1, "ncoarse": int(np.log2(n)), "fR_n": 1, "fR_q": np.float32(-0.1), "theory": "fr",
                     "compute_additional_field": True, "epsrel": 1e-3, "nsteps": 0}
            u = u0.clone()
            amd.multigrid.FAS(u, dens, param)
            fas[native] = (u.cpu().numpy(), param["_vcycles_fas"])
        finally:
            amd.multigrid._NATIVE_SOLVE = True
    assert fas[True][1] == fas[False][1] and fas[True][1] >= 1
    np.testing.assert_array_equal(fas[True][0], fas[False][0])
