"""Pins the CPU oracle (oracle/) against golden vectors captured from the reference source
(oracle/gen_golden.py).  CPU-only; runs everywhere."""
import numpy as np
import pytest

from oracle import pysco_oracle as orc

DEPOSIT_SETS = ["uniform_n16", "lattice_n16", "edge_n16", "cluster_n16", "uniform_n32", "lattice_n32"]


@pytest.mark.parametrize("tag", DEPOSIT_SETS)
def test_deposit_bit_exact(golden, tag):
    g = golden("deposit_" + tag)
    pos, n = g["pos"], int(g["n"])
    # interpreted TSC == TSC_seq (same summation order when run serially)
    np.testing.assert_array_equal(g["tsc"], g["tsc_seq"])
    # CIC/NGP: bit-exact.  TSC: the interpreter evaluates `dx**2` through numpy's scalar powf, which
    # differs from dx*dx (what Numba/LLVM compiles, and what the oracle does) by 1 ulp in ~0.2 % of
    # cases -> a few-ulp tolerance on the weights, far below any index/weight error (>= 1e-2).
    np.testing.assert_allclose(orc.TSC(pos, n), g["tsc"], rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(orc.CIC(pos, n), g["cic"])
    np.testing.assert_array_equal(orc.NGP(pos, n), g["ngp"])
    assert abs(float(orc.TSC(pos, n).sum(dtype=np.float64)) - pos.shape[0]) < 1e-3 * pos.shape[0]


@pytest.mark.parametrize("tag", DEPOSIT_SETS)
def test_gather_bit_exact(golden, tag):
    g = golden("deposit_" + tag)
    pos, F, S = g["pos"], g["F"], g["S"]
    np.testing.assert_allclose(orc.invTSC_vec(F, pos), g["inv_tsc_vec"], rtol=0, atol=2e-6)  # `**2`, see above
    np.testing.assert_array_equal(orc.invCIC_vec(F, pos), g["inv_cic_vec"])
    np.testing.assert_array_equal(orc.invNGP_vec(F, pos), g["inv_ngp_vec"])
    np.testing.assert_allclose(orc.invTSC(S, pos), g["inv_tsc"], rtol=0, atol=2e-6)


def test_gradient_bit_exact(golden):
    g = golden("gradient_n16")
    a, b, f = g["a"], g["b"], np.float32(g["f"])
    for order in (2, 3, 5, 7):
        np.testing.assert_array_equal(orc.derivative(a, order), g[f"d{order}"])
        # b**2 / b**3 go through numpy's scalar powf in the interpreter (1-ulp differences)
        np.testing.assert_allclose(orc.derivative_fR(a, b, f, 1, order), g[f"d{order}_fr1"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(orc.derivative_fR(a, b, f, 2, order), g[f"d{order}_fr2"], rtol=0, atol=2e-5)
    with pytest.raises(ValueError):
        orc.derivative(a, 4)


@pytest.mark.parametrize("n", [16, 32])
def test_kspace(golden, n):
    g = golden(f"kspace_n{n}")
    xk = g["xk"]
    np.testing.assert_array_equal(orc.fft_3D_real(g["x"]), xk)
    np.testing.assert_array_equal(orc.ifft_3D_real(xk.copy()), g["ifft_xk"])
    for name, fn in (("plain", orc.inverse_laplacian),
                     ("comp2", lambda a: orc.inverse_laplacian_compensated(a, 2)),
                     ("comp3", lambda a: orc.inverse_laplacian_compensated(a, 3)),
                     ("sevenpt", orc.inverse_laplacian_7pt)):
        a = xk.copy()
        fn(a)
        ref = g["green_" + name]
        assert a[0, 0, 0] == 0 and ref[0, 0, 0] == 0
        # fp32 sinc/sin/pow differ by a few ulp between libm and numpy's SIMD loops
        np.testing.assert_allclose(a, ref, rtol=3e-6, atol=0)
    for p in (0, 2, 3):
        a = xk.copy()
        k, pk, nm = orc.fourier_grid_to_Pk(a, p)
        assert a[0, 0, 0] == 0
        np.testing.assert_array_equal(nm, g[f"pk{p}_nmodes"])
        np.testing.assert_allclose(k, g[f"pk{p}_k"], rtol=2e-6)
        np.testing.assert_allclose(pk, g[f"pk{p}_pk"], rtol=1e-5)


def test_utils(golden):
    g = golden("utils")
    x, y0 = g["x"], g["y0"]
    for tag in ("p1", "m1", "gen"):
        y = y0.copy()
        orc.add_vector_scalar_inplace(y, x, np.float32(g["axpy_" + tag + "_a"]))
        np.testing.assert_array_equal(y, g["axpy_" + tag])
    y = y0.copy()
    orc.prod_vector_scalar_inplace(y, np.float32(1.7))
    np.testing.assert_array_equal(y, g["scale"])
    np.testing.assert_array_equal(orc.linear_operator(x, np.float32(0.193), np.float32(-0.193)), g["linop"])
    y = y0.copy()
    orc.linear_operator_vectors_inplace(y, np.float32(0.3), x, np.float32(-1.2))
    np.testing.assert_array_equal(y, g["linop_vec"])
    assert orc.max_abs(x) == g["max_abs"]
    w = g["wrap_in"].copy()
    orc.periodic_wrap(w)
    np.testing.assert_array_equal(w, g["wrap_out"])
    assert (w >= 0).all() and (w < 1).all()
    np.testing.assert_array_equal(orc.positions_to_keys(g["morton_pos"]), g["morton_keys"])
    p2, v2 = orc.reorder_particles(g["morton_pos"], g["morton_vel"])
    # np.argsort's default quicksort is not stable: compare sorted keys and the (key,pos,vel) multiset
    k_ref = orc.positions_to_keys(g["reorder_pos"])
    assert (np.diff(k_ref) >= 0).all()
    np.testing.assert_array_equal(orc.positions_to_keys(p2), k_ref)
    if len(np.unique(k_ref)) == len(k_ref):
        np.testing.assert_array_equal(p2, g["reorder_pos"])
        np.testing.assert_array_equal(v2, g["reorder_vel"])


PM_NEWTON = ["fft_tsc_o5", "fft_cic_o3", "fft7pt_tsc_o2", "fft_tsc_o7", "fft_tsc_o5_np5000_n16"]


def _scale(x):
    return float(np.abs(x).max())


@pytest.mark.parametrize("tag", PM_NEWTON)
def test_pm_newton(golden, tag):
    from tests.helpers import param_from_json, synth_tables

    g = golden("pm_newton_" + tag)
    param = param_from_json(g["param_json"])
    acc, pot, extra = orc.pm(g["pos"], param, tables=synth_tables())
    assert extra.size == 0 and g["extra"].size == 0
    np.testing.assert_allclose(pot, g["pot"], rtol=0, atol=2e-6 * _scale(g["pot"]))
    np.testing.assert_allclose(acc, g["acc"], rtol=0, atol=1e-5 * _scale(g["acc"]))
    for key in ("MAS_index", "save_pk", "compute_additional_field"):
        assert param[key] == param_from_json(g["param_json"])[key]


def test_pm_pk(golden):
    from tests.helpers import param_from_json, synth_tables

    g = golden("pm_newton_pk")
    param = param_from_json(g["param_json"])
    out = []
    orc.pm(g["pos"], param, tables=synth_tables(), pk_out=out)
    k, pk, nm = out[0]
    ref = g["pk_table"]
    np.testing.assert_allclose(k, ref[:, 0], rtol=2e-6)
    np.testing.assert_allclose(pk, ref[:, 1], rtol=1e-4)  # BASELINE.md parity bar for P(k)
    np.testing.assert_array_equal(nm, ref[:, 2])


@pytest.mark.parametrize("integ,nsteps", [("leapfrog", 10), ("euler", 3)])
def test_integrate(golden, integ, nsteps):
    from tests.helpers import param_from_json, synth_tables

    g = golden("integrate_" + integ)
    tables = synth_tables()
    param = param_from_json(g["param0_json"])
    p, v = g["pos0"].copy(), g["vel0"].copy()
    acc, pot, extra = orc.pm(p, dict(param), tables=tables)
    np.testing.assert_allclose(acc, g["acc0"], rtol=0, atol=1e-5 * _scale(g["acc0"]))
    t_next = np.float32(g["t_snap_next"])
    for s in range(nsteps):
        p, v, acc, pot, extra = orc.integrate(p, v, acc, pot, extra, tables, param, t_next)
        assert abs(param["t"] - g["t"][s]) < 1e-6
        assert abs(param["aexp"] - g["aexp"][s]) < 1e-6
        assert bool(param["write_snapshot"]) == bool(g["write_snapshot"][s])
        if s in (0, nsteps - 1):
            d = np.abs(p - g[f"pos_{s+1}"])
            d = np.minimum(d, 1 - d)
            assert d.max() <= 1e-5  # positions after 10 steps max |dx| <= 1e-5 box units
            np.testing.assert_allclose(v, g[f"vel_{s+1}"], rtol=0, atol=1e-5 * _scale(g[f"vel_{s+1}"]))


def test_transfer_and_laplacian_bit_exact(golden):
    g = golden("multigrid_n16")
    x, b, xc = g["x"], g["b"], g["xc"]
    np.testing.assert_array_equal(orc.restriction(x), g["restriction"])
    np.testing.assert_array_equal(orc.minus_restriction(x), g["minus_restriction"])
    np.testing.assert_array_equal(orc.prolongation(xc), g["prolongation"])
    y = x.copy()
    orc.add_prolongation(y, xc)
    np.testing.assert_array_equal(y, g["add_prolongation"])
    np.testing.assert_array_equal(orc.lap_operator(x), g["operator"])
    np.testing.assert_array_equal(orc.lap_residual(x, b), g["residual"])
    np.testing.assert_array_equal(orc.lap_restrict_residual(x, b), g["restrict_residual"])
    np.testing.assert_array_equal(orc.lap_initialise_potential(b), g["initialise_potential"])
    y = x.copy()
    orc.lap_gauss_seidel(y, b, np.float32(1.25))
    np.testing.assert_array_equal(y, g["gauss_seidel_1"])
    y = x.copy()
    orc.lap_smoothing(y, b, 3)
    np.testing.assert_array_equal(y, g["smoothing_3"])
    # norms: the interpreter squares through powf (1-ulp artefacts), fp32 sequential accumulation
    np.testing.assert_allclose(orc.lap_residual_error(x, b), g["residual_error"], rtol=2e-6)
    np.testing.assert_allclose(orc.lap_truncation_error(x), g["truncation_error"], rtol=2e-6)
    for m in (4, 8):
        y = g[f"small{m}_x"].copy()
        orc.lap_gauss_seidel(y, g[f"small{m}_b"], np.float32(1.25))
        np.testing.assert_array_equal(y, g[f"small{m}_gs"])
    np.testing.assert_array_equal(orc.lap_restrict_residual(g["small8_x"], g["small8_b"]), g["small8_restrict_residual"])
    y = g["small8_x"].copy()
    orc.add_prolongation(y, g["small4_x"])
    np.testing.assert_array_equal(y, g["small8_add_prolongation"])


def test_linear_multigrid(golden):
    from tests.helpers import param_from_json

    g = golden("multigrid_n16")
    param = param_from_json(g["lin_param_json"])
    param.pop("tolerance", None)
    phi = g["vc_phi0"].copy()
    orc.V_cycle(phi, g["vc_rhs"], param)
    np.testing.assert_array_equal(phi, g["vc_phi1"])  # whole V-cycle bit-exact
    phi = g["vc_phi0"].copy()
    info = {}
    phi = orc.mg_linear(phi, g["vc_rhs"], param, info)
    assert info["cycles"] == int(g["lin_cycles"])
    np.testing.assert_allclose(param["tolerance"], g["lin_tolerance"], rtol=1e-5)
    np.testing.assert_array_equal(phi, g["lin_phi"])


def test_cubic(golden):
    g = golden("cubic_n16")
    u, b, rhs, q = g["u"], g["b"], g["rhs"], np.float32(g["q"])
    s = float(np.abs(g["operator"]).max())
    np.testing.assert_allclose(orc.cubic_operator(u, b, q), g["operator"], rtol=0, atol=3e-6 * s)
    np.testing.assert_allclose(orc.cubic_residual_with_rhs(u, b, q, rhs), g["residual_with_rhs"], rtol=0, atol=3e-6 * s)
    ip, ipg = orc.cubic_initialise_potential(b, q), g["initialise_potential"]
    assert np.array_equal(np.isnan(ip), np.isnan(ipg))
    np.testing.assert_allclose(ip[~np.isnan(ip)], ipg[~np.isnan(ipg)], rtol=2e-5)
    roots = np.array([orc.solution_cubic_equation(p, d) for p, d in g["root_pd"]], dtype=np.float32)
    ref = g["root"]
    assert np.array_equal(np.isnan(roots), np.isnan(ref))  # same NaN branch structure (cubic.py:192-197)
    ok = ~np.isnan(ref)
    np.testing.assert_allclose(roots[ok], ref[ok], rtol=2e-6, atol=1e-6)
    y = u.copy()
    orc.cubic_gauss_seidel(y, b, q, np.float32(1.25))
    np.testing.assert_allclose(y, g["gauss_seidel"], rtol=0, atol=5e-6)
    y = u.copy()
    orc.cubic_gauss_seidel(y, b, q, np.float32(1.25), rhs)
    np.testing.assert_allclose(y, g["gauss_seidel_with_rhs"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(orc.cubic_residual_error(u, b, q), g["residual_error"], rtol=1e-5)
    np.testing.assert_allclose(orc.cubic_truncation_error(u, b, q), g["truncation_error"], rtol=1e-4)


def test_mond(golden):
    g = golden("mond_n16")
    ys = g["ys"]
    for name, kind, alpha in (("nu_simple", "simple", 1), ("nu_n2", "n", 2), ("nu_beta", "beta", 1.3),
                              ("nu_gamma", "gamma", 1.3), ("nu_delta", "delta", 1.3)):
        got = np.array([orc.mond_nu(y, kind, alpha) for y in ys], dtype=np.float32)
        np.testing.assert_allclose(got, g[name], rtol=5e-6, err_msg=name)
    for name, alpha in (("simple", 1), ("n", 2), ("beta", 1.3), ("gamma", 1.3), ("delta", 1.3)):
        out = np.empty_like(g["pot"])
        orc.mond_rhs(g["pot"], out, np.float32(g["g0"]), name, alpha)
        ref = g["rhs_" + name]
        np.testing.assert_allclose(out, ref, rtol=0, atol=2e-5 * float(np.abs(ref).max()), err_msg=name)


PM_MG = ["newton_multigrid", "fr_multigrid", "mond_multigrid", "mond_fft7pt"]


@pytest.mark.parametrize("tag", PM_MG)
def test_pm_multigrid_theories(golden, tag):
    from oracle.pysco_oracle import set_units
    from tests.helpers import param_from_json, synth_tables

    g = golden("pm_" + tag)
    tables = synth_tables()
    param = param_from_json(g["param0_json"])
    acc, pot, extra = orc.pm(g["pos"], param, tables=tables)
    sa, sp = _scale(g["acc1"]), _scale(g["pot1"])
    # iterative solves stop at epsrel * truncation error: compare at the solver's own accuracy
    np.testing.assert_allclose(pot, g["pot1"], rtol=0, atol=2e-4 * sp)
    np.testing.assert_allclose(acc, g["acc1"], rtol=0, atol=1e-3 * sa)
    if g["extra1"].size:
        np.testing.assert_allclose(extra, g["extra1"], rtol=0, atol=2e-4 * _scale(g["extra1"]))
    else:
        assert extra.size == 0
    ref1 = param_from_json(g["param1_json"])
    for key in ("tolerance", "tolerance_FAS", "tolerance_mond", "fR_q"):
        if key in ref1:
            np.testing.assert_allclose(float(param[key]), float(ref1[key]), rtol=1e-3, err_msg=key)
    # second call: warm start (solver.py:273-280)
    ref2 = param_from_json(g["param2_json"])
    param["nsteps"], param["aexp_old"], param["aexp"] = 1, param["aexp"], 0.51
    set_units(param)
    acc, pot, extra = orc.pm(g["pos2"], param, pot, extra, tables)
    np.testing.assert_allclose(pot, g["pot2"], rtol=0, atol=5e-4 * sp)
    np.testing.assert_allclose(acc, g["acc2"], rtol=0, atol=2e-3 * sa)


def test_quartic(golden):
    g = golden("quartic_n16")
    u, b, rhs, q = g["u"], g["b"], g["rhs"], np.float32(g["q"])
    s = float(np.abs(g["operator"]).max())
    np.testing.assert_allclose(orc.cubic_operator(u, b, q, 2), g["operator"], rtol=0, atol=5e-6 * s)
    np.testing.assert_allclose(orc.cubic_residual_with_rhs(u, b, q, rhs, 2), g["residual_with_rhs"], rtol=0, atol=5e-6 * s)
    np.testing.assert_allclose(orc.quartic_initialise_potential(b, q), g["initialise_potential"], rtol=2e-5)
    roots = np.array([orc.solution_quartic_equation(p, qq) for p, qq in g["root_pq"]], dtype=np.float32)
    ref = g["root"]
    assert np.array_equal(np.isnan(roots), np.isnan(ref))  # same branch structure, NaNs included
    ok = ~np.isnan(ref)
    np.testing.assert_allclose(roots[ok], ref[ok], rtol=3e-6, atol=1e-6)
    y = u.copy()
    orc.cubic_gauss_seidel(y, b, q, np.float32(1.25), None, 2)
    np.testing.assert_allclose(y, g["gauss_seidel"], rtol=0, atol=5e-6)
    y = u.copy()
    orc.cubic_gauss_seidel(y, b, q, np.float32(1.25), rhs, 2)
    np.testing.assert_allclose(y, g["gauss_seidel_with_rhs"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(orc.cubic_residual_error(u, b, q, 2), g["residual_error"], rtol=1e-5)
    np.testing.assert_allclose(orc.cubic_truncation_error(u, b, q, 2), g["truncation_error"], rtol=1e-4)


def test_pm_fr_n2(golden):
    test_pm_multigrid_theories(golden, "fr2_multigrid")


@pytest.mark.parametrize("n", [16, 32])
def test_oracle_full_fft_pieces(golden, n):
    """gradient_inverse_laplacian(_compensated), ifft_3D_real_grad and fft_force (fourier.py:343-396, 606-719,
    solver.py:524-566) against the reference's outputs (the pieces of the `full_fft` solver)."""
    g = golden(f"kspace_grad_n{n}")
    if n == 16:
        xk = g["xk"]
        for name, fn in (("grad_plain", lambda a: orc.gradient_inverse_laplacian(a)),
                         ("grad_comp2", lambda a: orc.gradient_inverse_laplacian_compensated(a, 2)),
                         ("grad_comp3", lambda a: orc.gradient_inverse_laplacian_compensated(a, 3))):
            got = fn(xk.copy())
            assert got.shape == (n, n, n // 2 + 1, 3) and got.dtype == np.complex64
            s = np.abs(g[name]).max()
            assert np.abs(got - g[name]).max() <= 4e-6 * s, name
            assert np.all(got[0, 0, 0] == 0)
        np.testing.assert_allclose(orc.ifft_3D_real_grad(g["grad_comp3"].copy()), g["ifft_grad_comp3"], rtol=0,
                                   atol=2e-6 * np.abs(g["ifft_grad_comp3"]).max())
    for mas in (0, 3):
        param = {"MAS_index": mas, "save_pk": False}
        f = orc.fft_force(g["x"].copy(), param)
        ref = g[f"fft_force_mas{mas}"]
        assert f.shape == (n, n, n, 3) and f.dtype == np.float32
        np.testing.assert_allclose(f, ref, rtol=0, atol=4e-6 * np.abs(ref).max())


@pytest.mark.parametrize("tag", ["fft", "multigrid"])
def test_oracle_pm_parametrized(golden, tag):
    """solver.pm with theory = parametrized (solver.py:95-112)."""
    from tests.helpers import param_from_json, synth_tables

    g = golden("pm_parametrized_" + tag)
    param = param_from_json(g["param_json"])
    ref = param_from_json(g["param_json"])
    acc, pot, extra = orc.pm(g["pos"], param, tables=synth_tables())
    assert param["parametrized_mu_z"] == pytest.approx(ref["parametrized_mu_z"], rel=1e-6) and ref["parametrized_mu_z"] != 1
    tol = 5e-6 if tag == "fft" else 1e-3
    np.testing.assert_allclose(pot, g["pot"], rtol=0, atol=tol * np.abs(g["pot"]).max())
    np.testing.assert_allclose(acc, g["acc"], rtol=0, atol=4 * tol * np.abs(g["acc"]).max())


LPT_CASES = ["1lpt_center_n16", "2lpt_center_n16", "2lpt_edge_fixed_paired_n16", "2lpt_center_n32", "3lpt_center_n16",
             "3lpt_edge_n32", "2lpt_dealiased_n16", "3lpt_dealiased_n16", "2lpt_dealiased_n32"]


@pytest.mark.parametrize("tag", LPT_CASES)
def test_oracle_lpt_vs_reference(golden, tag):
    """nLPT initial conditions (initial_conditions.py:108-158 and the functions it calls): the oracle's grid pipeline
    from the reference's Fourier-space density to particles, against the reference's positions and velocities."""
    import ast

    g = golden("lpt_" + tag)
    case, sc = ast.literal_eval(str(g["param_json"])), ast.literal_eval(str(g["scalars_json"]))
    pos, vel = orc.lpt_particles(g["density_k_half"], case["initial_conditions"], case["position_ICS"], sc,
                                 dealiased=case.get("dealiased_ICS", False))
    d = np.abs(pos - g["position"])
    assert np.minimum(d, 1 - d).max() < 2e-6
    assert np.abs(vel - g["velocity"]).max() < 1e-5 * np.abs(g["velocity"]).max()


@pytest.mark.parametrize("tag", LPT_CASES)
def test_host_side_random_field_vs_reference(golden, tag, tmp_path):
    """The product's host-side part of the initial conditions (pysco_amd.lpt: NumPy draws, Hermitian fill, transfer
    table; no GPU involved) reproduces the reference's Fourier-space density: same seed, same field."""
    import ast

    from pysco_amd import lpt

    g = golden("lpt_" + tag)
    case = ast.literal_eval(str(g["param_json"]))
    f = tmp_path / "pk.dat"
    np.savetxt(f, np.c_[g["k_table"], g["pk_table"]])
    dk = lpt.generate_density_fourier(dict(case, power_spectrum_file=str(f), boxlen=100.0))
    assert dk.dtype == np.complex64 and dk.shape == g["density_k_half"].shape
    assert np.abs(dk - g["density_k_half"]).max() <= 3e-7 * np.abs(g["density_k_half"]).max()
