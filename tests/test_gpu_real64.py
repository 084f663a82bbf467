"""GPU tests of the fp64 device mode (pm_real64_* entry points, pysco_amd/real64.py) against the fp64 CPU path
(oracle/pysco_oracle_f64.py: the pinned C restatement compiled with REAL = double): BASELINE.md section 6 asks for
fp64-GPU vs fp64-CPU <= 1e-10 on positions and P(k).  The tolerance is written where it is used."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import pysco_oracle as o32  # noqa: E402  (checker only)
from oracle import pysco_oracle_f64 as o64  # noqa: E402
from tests.helpers import synth_tables, zeldovich_particles  # noqa: E402


def _param(n, npart, scheme="TSC", solver="fft", order=5):
    p = {"theory": "newton", "mass_scheme": scheme, "npart": npart, "ncoarse": int(np.log2(n)), "Om_m": 0.25733,
         "Om_lambda": 0.742589237, "H0": 72.0, "boxlen": 100.0, "aexp": 0.5, "aexp_old": 0.49,
         "linear_newton_solver": solver, "gradient_stencil_order": order, "save_power_spectrum": "no", "nthreads": 1,
         "Courant_factor": 1.0, "max_aexp_stepping": 10, "integrator": "leapfrog", "write_snapshot": False, "nsteps": 0}
    tables = synth_tables()
    p["t"] = float(tables[1](np.log(p["aexp"])))
    o32.set_units(p)
    return p, tables


def _particles(npart_1d, seed, rms, vel=0.3):
    pos, v = zeldovich_particles(npart_1d, seed=seed, rms_cells=rms, vel_factor=vel)
    rng = np.random.default_rng(seed)
    # genuine float64 positions: the float32 set plus sub-float32 detail
    pos = np.mod(pos.astype(np.float64) + 1e-9 * rng.standard_normal(pos.shape), 1.0)
    return pos, v.astype(np.float64) * (1 + 1e-9 * rng.standard_normal(v.shape))


@pytest.mark.parametrize("scheme", ["TSC", "CIC", "NGP"])
def test_real64_deposit_gradient_gather(scheme):
    import torch

    from pysco_amd import _lib as L

    n = 32
    pos, _ = _particles(40, seed=3, rms=0.9)
    edge = np.array([[0.0, 0.0, 0.0], [1 - 2.0**-53, 1 - 2.0**-53, 0.5], [0.5 / n, 1.5 / n, 1 - 0.5 / n],
                     [1.0 / n, 2.0 / n, 3.0 / n]])  # first / last cell, cell centres, cell faces
    pos = np.concatenate([pos, edge])
    ctx = L.context(n, 8)
    code = {"NGP": L.NGP, "CIC": L.CIC, "TSC": L.TSC}[scheme]
    pd = torch.from_numpy(pos).cuda()
    for shuffled in (False, True):  # sorted (tiles) and arbitrary order: the same sums
        p_h = pos if not shuffled else pos[np.random.default_rng(1).permutation(len(pos))]
        if not shuffled:
            from pysco_amd import utils

            order = []
            utils.reorder_particles(torch.from_numpy(p_h.astype(np.float32)).cuda(), _order_out=order)
            p_h = p_h[order[0].cpu().numpy()]
        t = torch.from_numpy(np.ascontiguousarray(p_h)).cuda()
        rho = torch.empty((n, n, n), dtype=torch.float64, device="cuda")
        ctx.call("pm_real64_deposit", code, L.ptr(t), t.shape[0], L.ptr(rho))
        ref = o64.deposit(scheme, p_h, n)
        assert abs(float(rho.sum()) - len(p_h)) < 1e-9            # mass
        assert np.abs(rho.cpu().numpy() - ref).max() <= 1e-12      # order of the double sums only
    phi = torch.from_numpy(np.random.default_rng(5).standard_normal((n, n, n))).cuda()
    for order in (2, 3, 5, 7):
        force = torch.empty((n, n, n, 3), dtype=torch.float64, device="cuda")
        ctx.call("pm_real64_gradient", order, L.ptr(phi), L.ptr(force))
        assert np.array_equal(force.cpu().numpy(), o64.derivative(phi.cpu().numpy(), order))  # same operations: same bits
    acc = torch.empty_like(pd)
    import ctypes as C

    ma = C.c_double(0)
    ctx.call("pm_real64_interp_kick", code, L.ptr(force), L.ptr(pd), pd.shape[0], L.ptr(acc), None, 0.0, C.byref(ma), None)
    ref = o64.gather(scheme, force.cpu().numpy(), pos)
    assert np.array_equal(acc.cpu().numpy(), ref)                  # the reference's summation order: same bits
    assert ma.value == np.abs(ref).max()


def test_real64_kick_drift_wrap_and_errors():
    import torch

    from pysco_amd import _lib as L
    from pysco_amd import integration, solver

    x = np.array([[-0.2, 1.3, 0.8], [-1e-9, 1.0, 0.0], [-3e-8, 0.25, 1 - 2.0**-53]])
    v = np.array([[0.1, -0.2, 0.3], [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]])
    a = np.array([[1.0, 2.0, -3.0], [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]])
    half_dt, dt = 0.05, 0.1
    xr, vr = x.copy(), v.copy()
    vr += (-half_dt) * a
    xr += dt * vr
    o64.periodic_wrap(xr)
    xd, vd, ad = (torch.from_numpy(t.copy()).cuda() for t in (x, v, a))
    L.context(16, 8).call("pm_real64_kick_drift_wrap", L.ptr(xd), L.ptr(vd), L.ptr(ad), half_dt, dt, 3)
    assert np.array_equal(xd.cpu().numpy(), xr) and np.array_equal(vd.cpu().numpy(), vr)
    # a float32 context refuses the fp64 entry points; unsupported theories and integrators say so
    with pytest.raises(L.PMStepError):
        L.context(16).call("pm_real64_kick_drift_wrap", L.ptr(xd), L.ptr(vd), L.ptr(ad), half_dt, dt, 3)
    par, tables = _param(16, 3)
    for key, val in (("theory", "fr"), ("linear_newton_solver", "multigrid")):
        q = dict(par)
        q[key] = val
        with pytest.raises(NotImplementedError):
            solver.pm(xd, q, tables=tables)
    q = dict(par)
    q["integrator"] = "euler"
    with pytest.raises(NotImplementedError):
        integration.integrate(xd, vd, ad, None, None, tables, q, 1.0)
    q = dict(par)
    q["mass_scheme"] = "NGP"
    with pytest.raises(NotImplementedError):
        solver.pm(xd, q, tables=tables)


@pytest.mark.parametrize("n,npart_1d,scheme,solver,order,resident", [(64, 64, "TSC", "fft", 5, True), (128, 128, "TSC", "fft", 5, True),
                                                                      (64, 48, "CIC", "fft_7pt", 3, False), (32, 64, "TSC", "fft", 7, True)])
def test_real64_steps_vs_fp64_cpu(n, npart_1d, scheme, solver, order, resident, tmp_path):
    """Five leapfrog steps in the fp64 device mode against the fp64 CPU path from the same float64 particles:
    positions <= 1e-10 box units, velocities and accelerations <= 1e-9 of their scale, P(k) <= 1e-10 per bin
    (what separates the two sides is the order of double sums and two FFT libraries)."""
    import torch

    from pysco_amd import integration, real64, solver as amd_solver

    pos, vel = _particles(npart_1d, seed=7, rms=0.7 * npart_1d / n)
    par_c, tables = _param(n, npart_1d**3, scheme, solver, order)
    par_g = dict(par_c)
    pc, vc = pos.copy(), vel.copy()
    ac, _ = o64.pm(pc, par_c)
    if resident:
        # a Morton-sorted device-resident stream (the order is a matter of speed: the CPU side keeps its own)
        from pysco_amd import utils

        order_out = []
        utils.reorder_particles(torch.from_numpy(pos.astype(np.float32)).cuda(), _order_out=order_out)
        perm = order_out[0].cpu().numpy()
        pg, vg = torch.from_numpy(pos[perm]).cuda(), torch.from_numpy(vel[perm]).cuda()
    else:
        perm = np.arange(len(pos))
        pg, vg = pos.copy(), vel.copy()
    ag, phig, exg = amd_solver.pm(pg, par_g, tables=tables)
    host = (lambda t: t.cpu().numpy()) if resident else (lambda t: t)
    assert np.abs(host(ag) - ac[perm]).max() <= 1e-11 * np.abs(ac).max()
    for _ in range(5):
        pc, vc, ac, _, _ = o64.integrate(pc, vc, ac, tables, par_c, 3e38)
        pg, vg, ag, phig, exg = integration.integrate(pg, vg, ag, phig, exg, tables, par_g, 3e38)
        assert abs(par_c["t"] - par_g["t"]) <= 1e-13 * abs(par_c["t"]) and abs(par_c["aexp"] - par_g["aexp"]) <= 1e-13
    d = np.abs(host(pg) - pc[perm])
    assert np.minimum(d, 1 - d).max() <= 1e-10
    assert np.abs(host(vg) - vc[perm]).max() <= 1e-9 * np.abs(vc).max()
    assert np.abs(host(ag) - ac[perm]).max() <= 1e-9 * np.abs(ac).max()
    # P(k) of the evolved particles: device estimator on the device's particles vs the CPU's on the CPU's
    dens = torch.empty((n, n, n), dtype=torch.float64, device="cuda")
    from pysco_amd import _lib as L

    code = {"CIC": L.CIC, "TSC": L.TSC}[scheme]
    pgd = pg if resident else torch.from_numpy(pg).cuda()
    L.context(n, 8).call("pm_real64_deposit", code, L.ptr(pgd), pgd.shape[0], L.ptr(dens))
    par_g["MAS_index"] = 3 if scheme == "TSC" else 2
    k, pk, nm = real64.power_spectrum(dens, par_g)
    k_c, pk_c, nm_c = o64.pk_of_density(o64.deposit(scheme, pc, n), par_g["MAS_index"])
    assert np.array_equal(nm, nm_c)
    assert np.abs(k - k_c * (2 * np.pi / par_g["boxlen"])).max() <= 1e-12 * k.max()
    norm = (par_g["boxlen"] / n**2) ** 3
    assert np.abs(pk - pk_c * norm).max() <= 1e-10 * np.abs(pk_c * norm).max()
    assert (np.abs(pk / (pk_c * norm) - 1) <= 1e-10).all()       # ... and per bin


def _morton21(pos):
    """morton.py:42-137 on float64 positions: 21 bits per axis, x the most significant of each triple."""
    q = np.floor(pos * 2097152.0).astype(np.int64)
    key = np.zeros(len(pos), dtype=np.int64)
    for b in range(21):
        for a in range(3):
            key |= ((q[:, a] >> b) & 1) << (3 * b + (2 - a))
    return key


def test_real64_reorder_is_a_stable_morton_sort_in_double():
    """utils.reorder_particles on float64 device arrays: sorted in double (nothing rounded to float32: two particles
    that differ only below float32 resolution keep their values and fall on the right side of each other), stable,
    velocity / acceleration carried along; with _key_bits only the top bits decide and ties keep the input order; and
    the stream age the gather reports goes to None after the sort (main.reorder_policy's measure)."""
    import torch

    from pysco_amd import real64, utils

    rng = np.random.default_rng(11)
    npart = 50_000
    pos = rng.random((npart, 3))
    pos[1] = pos[0] + np.array([2.0**-40, 0.0, 0.0])  # same float32, different float64 (and, at 21 bits, the same key)
    pos[7] = pos[3]
    vel, acc = rng.standard_normal((npart, 3)), rng.standard_normal((npart, 3))
    order = []
    p, v, a = utils.reorder_particles(torch.from_numpy(pos).cuda(), torch.from_numpy(vel).cuda(), torch.from_numpy(acc).cuda(),
                                      _order_out=order)
    assert p.dtype == torch.float64
    want = np.argsort(_morton21(pos), kind="stable")
    np.testing.assert_array_equal(order[0].cpu().numpy(), want)
    np.testing.assert_array_equal(p.cpu().numpy(), pos[want])
    np.testing.assert_array_equal(v.cpu().numpy(), vel[want])
    np.testing.assert_array_equal(a.cpu().numpy(), acc[want])
    for bits in (4, 9):
        order = []
        p2 = real64.reorder_particles(torch.from_numpy(pos).cuda(), _order_out=order, _key_bits=bits)
        want = np.argsort(_morton21(pos) >> (3 * (21 - bits)), kind="stable")
        np.testing.assert_array_equal(order[0].cpu().numpy(), want)
        np.testing.assert_array_equal(p2.cpu().numpy(), pos[want])
    # NumPy in, NumPy out
    ph, vh = utils.reorder_particles(pos, vel)
    assert isinstance(ph, np.ndarray) and ph.dtype == np.float64
    np.testing.assert_array_equal(ph, pos[np.argsort(_morton21(pos), kind="stable")])


def test_real64_stream_age_and_policy():
    """The gather's age measure (particles served one by one: too few of their chunk in their tile) is small on a sorted
    stream (with about two tile passes per chunk), large on a shuffled one, and main.reorder_policy re-sorts the
    shuffled one to tile resolution: same particles, sorted order, age reset."""
    import torch

    from pysco_amd import main as pm_main
    from pysco_amd import real64, solver

    n = 64
    pos, vel = _particles(64, seed=5, rms=0.7)
    param, tables = _param(n, len(pos))
    ps, vs = real64.reorder_particles(torch.from_numpy(pos).cuda(), torch.from_numpy(vel).cuda())
    acc, _, _ = solver.pm(ps, dict(param), tables=tables)
    # (pm without the fused kick does not read the maxima back: step once for the measure)
    from pysco_amd import integration

    p1 = dict(param)
    out = integration.integrate(ps.clone(), vs.clone(), acc.clone(), None, None, tables, p1, 1e30)
    age_sorted = real64.stream_age(n)
    assert age_sorted is not None and age_sorted[0] < 0.1 and age_sorted[1] <= 3.0, age_sorted
    same = pm_main.reorder_policy(out[0], out[1], out[2], dict(p1))  # the sorted stream is left alone
    assert same[0].data_ptr() == out[0].data_ptr()
    perm = torch.from_numpy(np.random.default_rng(2).permutation(len(pos))).cuda()
    pu, vu, au = ps[perm].contiguous(), vs[perm].contiguous(), acc[perm].contiguous()
    p2 = dict(param)
    pu, vu, au, _, _ = integration.integrate(pu, vu, au, None, None, tables, p2, 1e30)
    age_shuffled = real64.stream_age(n)
    assert age_shuffled[0] > 0.5, age_shuffled
    pr, vr, ar = pm_main.reorder_policy(pu, vu, au, p2)
    assert real64.stream_age(n) is None and pr.data_ptr() != pu.data_ptr()
    key = _morton21(pr.cpu().numpy()) >> (63 - (3 * 6 - 11))  # (the policy sorts to the 8 x 16 x 16-cell tiles)
    assert (np.diff(key) >= 0).all()
    np.testing.assert_array_equal(np.sort(pr.cpu().numpy(), axis=0), np.sort(pu.cpu().numpy(), axis=0))


def test_real64_fused_kick_deposit_is_the_two_passes():
    """pm_real64_kick_drift_wrap_deposit against pm_real64_kick_drift_wrap followed by pm_real64_deposit: the same
    position and velocity bit for bit, the same density up to the order of the double sums."""
    import torch

    from pysco_amd import _lib as L

    n = 64
    pos, vel = _particles(48, seed=8, rms=0.8)
    rng = np.random.default_rng(8)
    acc = rng.standard_normal(pos.shape) * 0.05
    ctx = L.context(n, 8)
    out = {}
    for fused in (False, True):
        p, v, a = (torch.from_numpy(x.copy()).cuda() for x in (pos, vel, acc))
        rho = torch.empty((n, n, n), dtype=torch.float64, device="cuda")
        if fused:
            ctx.call("pm_real64_kick_drift_wrap_deposit", L.TSC, L.ptr(p), L.ptr(v), L.ptr(a), 0.01, 0.02, p.shape[0], L.ptr(rho))
        else:
            ctx.call("pm_real64_kick_drift_wrap", L.ptr(p), L.ptr(v), L.ptr(a), 0.01, 0.02, p.shape[0])
            ctx.call("pm_real64_deposit", L.TSC, L.ptr(p), p.shape[0], L.ptr(rho))
        out[fused] = (p.cpu().numpy(), v.cpu().numpy(), rho.cpu().numpy())
    np.testing.assert_array_equal(out[True][0], out[False][0])
    np.testing.assert_array_equal(out[True][1], out[False][1])
    assert np.abs(out[True][2] - out[False][2]).max() <= 1e-12
    assert abs(out[True][2].sum() - len(pos)) < 1e-8 and np.abs(out[True][0] - pos).max() > 0


def test_real64_steps_with_the_resort_policy_match_in_place_steps():
    """Eight fp64 leapfrog steps with the order policy forced on every step (a tile re-sort of position and velocity
    right after the drift, fused kick + deposit on the others) against the same steps in place: the same particles by
    tag to 1e-12 (only the order of the deposit's double sums differs), and the policy did re-sort."""
    import torch

    from pysco_amd import integration, solver

    n = 64
    pos, vel = _particles(64, seed=21, rms=0.6)
    out = {}
    for policy in (False, True):
        param, tables = _param(n, len(pos))
        if policy:
            param["rebucket_straggler_fraction"] = 0.0
            param["resort_passes_per_chunk"] = 0.0  # (every step the gather has measured)
            param["particle_id"] = torch.arange(len(pos), dtype=torch.int32, device="cuda")
        p, v = torch.from_numpy(pos.copy()).cuda(), torch.from_numpy(vel.copy()).cuda()
        a, phi, extra = solver.pm(p, param, tables=tables)
        for _ in range(8):
            p, v, a, phi, extra = integration.integrate(p, v, a, phi, extra, tables, param, 1e30)
        if policy:
            assert param.get("_resorted_f64", 0) >= 5
            inv = np.argsort(param["particle_id"].cpu().numpy())
            out[policy] = (p.cpu().numpy()[inv], v.cpu().numpy()[inv], a.cpu().numpy()[inv])
        else:
            out[policy] = (p.cpu().numpy(), v.cpu().numpy(), a.cpu().numpy())
    d = np.abs(out[True][0] - out[False][0])
    assert np.minimum(d, 1 - d).max() <= 1e-12
    assert np.abs(out[True][1] - out[False][1]).max() <= 1e-11 * np.abs(out[False][1]).max()
    assert np.abs(out[True][2] - out[False][2]).max() <= 1e-9 * np.abs(out[False][2]).max()


@pytest.mark.parametrize("order", [2, 3, 5, 7])
def test_real64_marching_gradient_is_the_cpu_stencil(order):
    """pm_real64_gradient at 128^3 (the marching kernel: planes along x kept in registers) against the fp64 CPU
    stencils: bit-identical, every order."""
    import torch

    from pysco_amd import _lib as L

    n = 128
    rng = np.random.default_rng(order)
    phi = rng.standard_normal((n, n, n))
    pd = torch.from_numpy(phi).cuda()
    force = torch.empty((n, n, n, 3), dtype=torch.float64, device="cuda")
    L.context(n, 8).call("pm_real64_gradient", order, L.ptr(pd), L.ptr(force))
    np.testing.assert_array_equal(force.cpu().numpy(), o64.derivative(phi, order))
