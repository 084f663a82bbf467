#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by EXECUTING THE REFERENCE SOURCE.

Runs only in the build container (needs /root/reference, which never travels to the GPU box).
The reference's Numba kernels are executed as interpreted Python through the no-op decorators in
oracle/refshim/ (the reference's own DISABLE_JIT mode); FFTs go through its NumPy fallback
(fourier.py:129-130, 276-277).  Only the .npz outputs are committed -- inputs and expected
outputs, no reference text.

    python oracle/gen_golden.py [group ...]      # groups: see GROUPS at the bottom
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/pysco"
sys.path.insert(0, os.path.join(HERE, "refshim"))
sys.path.insert(1, REF)

import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
from scipy.interpolate import interp1d  # noqa: E402

import mesh, fourier, utils, solver, integration, multigrid, laplacian, cubic, quartic, mond, morton  # noqa: E402,E401
import iostream  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)
f32 = np.float32


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} KiB)")


# ---------------------------------------------------------------- shared synthetic inputs
# (tests/helpers.py holds the same generators so GPU-box tests can rebuild inputs from seeds; the
# fixtures nevertheless store the inputs so that no generator drift can unpin them.)


def particles_uniform(seed, npart):
    rng = np.random.default_rng(seed)
    return rng.random((npart, 3), dtype=np.float32)


def particles_lattice(seed, n, rms_cells):
    """Cell-centred lattice displaced by a smooth Gaussian field, wrapped to [0,1)."""
    rng = np.random.default_rng(seed)
    g = (np.arange(n, dtype=np.float64) + 0.5) / n
    x0 = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    psi = np.zeros_like(x0)
    for m in range(1, 4):
        for ax in range(3):
            ph = rng.random(3) * 2 * np.pi
            amp = rng.normal(size=3) / m
            psi[:, ax] += amp[0] * np.sin(2 * np.pi * m * x0[:, 0] + ph[0]) \
                + amp[1] * np.sin(2 * np.pi * m * x0[:, 1] + ph[1]) \
                + amp[2] * np.sin(2 * np.pi * m * x0[:, 2] + ph[2])
    psi *= rms_cells / n / np.sqrt((psi**2).sum(axis=1).mean())
    pos = (x0 + psi).astype(np.float32)
    utils.periodic_wrap(pos)
    return pos, psi.astype(np.float32)


def particles_edge(n):
    """Positions on cell faces/centres, at 0, just below 1, in the first and last cells."""
    h = 1.0 / n
    vals = [0.0, np.nextafter(f32(1), f32(0)), h, 0.5 * h, 1 - h, 1 - 0.5 * h, 0.5, 0.5 + 0.5 * h,
            np.nextafter(f32(h), f32(0)), np.nextafter(f32(h), f32(1)), 0.25 * h, 1 - 0.25 * h]
    vals = np.array(vals, dtype=np.float32)
    pts = np.stack(np.meshgrid(vals, vals, vals, indexing="ij"), axis=-1).reshape(-1, 3)
    return np.ascontiguousarray(pts, dtype=np.float32)


def base_param(ncoarse, npart, **kw):
    p = {
        "theory": "newton", "mass_scheme": "TSC", "npart": npart, "ncoarse": ncoarse,
        "Om_m": 0.25733, "Om_lambda": 0.742589237, "w0": -1.0, "wa": 0.0, "H0": 72.0,
        "boxlen": 100.0, "linear_newton_solver": "fft", "gradient_stencil_order": 5,
        "save_power_spectrum": "no", "aexp": 0.5, "aexp_old": 0.48, "nthreads": 1,
        "Npre": 2, "Npost": 1, "epsrel": 1e-2, "Courant_factor": 1.0, "max_aexp_stepping": 10,
        "integrator": "leapfrog", "t": -0.6, "write_snapshot": False, "nsteps": 0,
    }
    p.update(kw)
    s = pd.Series(p, dtype=object)
    utils.set_units(s)
    return s


def synth_tables():
    """Monotone stand-ins for tables[0]=lna(t), [1]=t(lna), [3]=D1(lna); EdS-like a ~ (1+1.5 t)^(2/3)."""
    t = np.linspace(-0.95, 0.0, 4000)
    a = (1.0 + t) ** (2.0 / 3.0)  # a(t=0) = 1
    lna = np.log(a)
    t0 = interp1d(t, lna, fill_value="extrapolate")
    t1 = interp1d(lna, t, fill_value="extrapolate")
    d1 = interp1d(lna, a, fill_value="extrapolate")
    return [t0, t1, None, d1] + [None] * 9


# ---------------------------------------------------------------- groups


def g_deposit():
    """G1/G2: deposit and inverse interpolation (mesh.py:2240-3088)."""
    for n in (16, 32):
        sets = {"uniform": particles_uniform(1, 3000), "lattice": particles_lattice(2, n, 0.7)[0]}
        if n == 16:
            sets["edge"] = particles_edge(n)
            sets["cluster"] = (0.5 + 0.02 * np.random.default_rng(3).normal(size=(2000, 3))).astype(np.float32)
        for tag, pos in sets.items():
            if n == 32 and tag == "lattice":
                pos = pos[::5].copy()
            rng = np.random.default_rng(10)
            F = rng.normal(size=(n, n, n, 3)).astype(np.float32)
            S = rng.normal(size=(n, n, n)).astype(np.float32)
            save(f"deposit_{tag}_n{n}", pos=pos, n=n,
                 ngp=mesh.NGP(pos, n), cic=mesh.CIC(pos, n), tsc=mesh.TSC(pos, n), tsc_seq=mesh.TSC_seq(pos, n),
                 F=F, S=S,
                 inv_ngp_vec=mesh.invNGP_vec(F, pos), inv_cic_vec=mesh.invCIC_vec(F, pos),
                 inv_tsc_vec=mesh.invTSC_vec(F, pos), inv_tsc=mesh.invTSC(S, pos),
                 inv_cic=mesh.invCIC(S, pos), inv_ngp=mesh.invNGP(S, pos))


def g_kspace():
    """G3/G4: Green's functions and the P(k) estimator (fourier.py:22-100, 460-595) + FFT wrappers."""
    for n in (16, 32):
        rng = np.random.default_rng(20 + n)
        x = rng.normal(size=(n, n, n)).astype(np.float32)
        xk = fourier.fft_3D_real(x, 1)
        out = {"x": x, "xk": xk, "ifft_xk": fourier.ifft_3D_real(xk.copy(), 1)}
        for name, fn in (("plain", lambda a: fourier.inverse_laplacian(a)),
                         ("comp2", lambda a: fourier.inverse_laplacian_compensated(a, 2)),
                         ("comp3", lambda a: fourier.inverse_laplacian_compensated(a, 3)),
                         ("sevenpt", lambda a: fourier.inverse_laplacian_7pt(a))):
            a = xk.copy()
            with np.errstate(all="ignore"):
                fn(a)
            out["green_" + name] = a
        for p in (0, 2, 3):
            a = xk.copy()
            k, pk, nm = fourier.fourier_grid_to_Pk(a, p)
            out[f"pk{p}_k"], out[f"pk{p}_pk"], out[f"pk{p}_nmodes"] = k, pk, nm
            out[f"pk{p}_dc"] = a[0, 0, 0]
        save(f"kspace_n{n}", **out)


def g_kspace_grad():
    """G3 (force form): gradient_inverse_laplacian(_compensated) (fourier.py:606-719), ifft_3D_real_grad (:343-396)
    and solver.fft_force (solver.py:524-566) -- the pieces of the `full_fft` solver, which pm() itself cannot
    reach in the reference (solver.py:181, 200 call the two-argument fft_force with three)."""
    for n in (16, 32):
        rng = np.random.default_rng(20 + n)  # the same field as kspace_n{n}
        x = rng.normal(size=(n, n, n)).astype(np.float32)
        xk = fourier.fft_3D_real(x, 1)
        out = {"x": x, "xk": xk}
        with np.errstate(all="ignore"):
            out["grad_plain"] = fourier.gradient_inverse_laplacian(xk.copy())
            out["grad_comp2"] = fourier.gradient_inverse_laplacian_compensated(xk.copy(), 2)
            out["grad_comp3"] = fourier.gradient_inverse_laplacian_compensated(xk.copy(), 3)
        out["ifft_grad_comp3"] = fourier.ifft_3D_real_grad(out["grad_comp3"].copy(), 1)
        if n == 32:  # keep the fixture small: the larger size pins the force itself
            for k in ("grad_plain", "grad_comp2", "grad_comp3", "xk"):
                del out[k]
        for mas in (0, 3):
            param = base_param(int(np.log2(n)), n**3, MAS_index=mas, save_pk=False, parametrized_mu_z=f32(1))
            with np.errstate(all="ignore"):
                out[f"fft_force_mas{mas}"] = solver.fft_force(x.copy(), param)
        save(f"kspace_grad_n{n}", **out)


def g_pm_parametrized():
    """f4: solver.pm with theory = parametrized (mu(z) of solver.py:95-112), FFT and multigrid solvers."""
    n, nc = 32, 5
    pos, _ = particles_lattice(50, n, 0.8)
    for tag, kw in (("fft", dict(theory="parametrized", parametrized_mu0=-0.1, Om_r=8.07417e-05)),
                    ("multigrid", dict(theory="parametrized", parametrized_mu0=0.2, Om_r=8.07417e-05, w0=-0.9, wa=0.1,
                                       linear_newton_solver="multigrid"))):
        param, acc, pot, extra = _pm_case(tag, n, nc, pos, **kw)
        save(f"pm_parametrized_{tag}", pos=pos, acc=acc, pot=pot, extra=extra, param_json=np.array(param_to_json(param)))
        print("   mu(z) =", param["parametrized_mu_z"])


def g_gradient():
    """G5: finite-difference gradients (mesh.py:639-850) and the f(R) fused variants (:860-2069)."""
    n = 16
    rng = np.random.default_rng(30)
    a = rng.normal(size=(n, n, n)).astype(np.float32)
    b = (1.0 + 0.1 * rng.normal(size=(n, n, n))).astype(np.float32)
    f = f32(0.37)
    out = {"a": a, "b": b, "f": f}
    for order in (2, 3, 5, 7):
        out[f"d{order}"] = mesh.derivative(a, order)
        out[f"d{order}_fr1"] = mesh.derivative_fR(a, b, f, 1, order)
        out[f"d{order}_fr2"] = mesh.derivative_fR(a, b, f, 2, order)
    save("gradient_n16", **out)


def g_utils():
    """G10/G11: element-wise helpers, periodic_wrap edge values, Morton keys and reorder."""
    rng = np.random.default_rng(40)
    x = rng.normal(size=1000).astype(np.float32)
    y0 = rng.normal(size=1000).astype(np.float32)
    out = {"x": x, "y0": y0}
    for tag, a in (("p1", f32(1)), ("m1", f32(-1)), ("gen", f32(-0.0371))):
        y = y0.copy()
        utils.add_vector_scalar_inplace(y, x, a)
        out["axpy_" + tag] = y
        out["axpy_" + tag + "_a"] = a
    y = y0.copy()
    utils.prod_vector_scalar_inplace(y, f32(1.7))
    out["scale"] = y
    out["linop"] = utils.linear_operator(x, f32(0.193), f32(-0.193))
    y = y0.copy()
    utils.linear_operator_vectors_inplace(y, f32(0.3), x, f32(-1.2))
    out["linop_vec"] = y
    out["max_abs"] = utils.max_abs(x)
    eps = -(0.5**25)
    w = np.array([-0.2, 1.3, 0.8, 0.0, 1.0, -1e-9, eps, eps * (1 + 2e-6), np.nextafter(f32(eps), f32(-1)),
                  -1e-7, -3e-8, -2.9e-8, np.nextafter(f32(1), f32(0)), np.nextafter(f32(1), f32(2)),
                  -0.999999, 1.999999, -5e-8, -6e-8], dtype=np.float32)
    out["wrap_in"] = w.copy()
    utils.periodic_wrap(w)
    out["wrap_out"] = w
    pos = particles_uniform(41, 2000)
    pos[:5] = [[0, 0, 0], [0.999999, 0.999999, 0.999999], [0.5, 0.25, 0.125], [1e-7, 0.3, 0.9], [0.3, 0.3, 0.3]]
    vel = rng.normal(size=pos.shape).astype(np.float32)
    out["morton_pos"] = pos
    out["morton_vel"] = vel
    out["morton_keys"] = morton.positions_to_keys(pos)
    p2, v2 = utils.reorder_particles(pos, vel)
    out["reorder_pos"], out["reorder_vel"] = p2, v2
    save("utils", **out)


def _pm_case(name, n, ncoarse, pos, **kw):
    param = base_param(ncoarse, pos.shape[0], **kw)
    tables = synth_tables()
    p_in = {k: v for k, v in param.items()}
    with np.errstate(all="ignore"):
        acc, pot, extra = solver.pm(pos, param, tables=tables)
    return param, acc, pot, extra


def g_pm_newton():
    """G12 (Newtonian rows): solver.pm end-to-end, solver.py:30-215."""
    n, nc = 32, 5
    pos, _ = particles_lattice(50, n, 0.8)
    for tag, kw in (("fft_tsc_o5", dict()),
                    ("fft_cic_o3", dict(mass_scheme="CIC", gradient_stencil_order=3)),
                    ("fft7pt_tsc_o2", dict(linear_newton_solver="fft_7pt", gradient_stencil_order=2)),
                    ("fft_tsc_o7", dict(gradient_stencil_order=7))):
        param, acc, pot, extra = _pm_case(tag, n, nc, pos, **kw)
        save(f"pm_newton_{tag}", pos=pos, acc=acc, pot=pot, extra=extra,
             param_json=np.array(param_to_json(param)))
    # npart != ncells**3 rescaling branch (solver.py:114-116)
    pos2 = particles_uniform(51, 5000)
    param, acc, pot, extra = _pm_case("sub", 16, 4, pos2)
    save("pm_newton_fft_tsc_o5_np5000_n16", pos=pos2, acc=acc, pot=pot, extra=extra,
         param_json=np.array(param_to_json(param)))


def param_to_json(param):
    import json

    d = {}
    for k, v in param.items():
        if isinstance(v, (np.floating, float)):
            d[k] = float(v)
        elif isinstance(v, (np.integer, int)) and not isinstance(v, bool):
            d[k] = int(v)
        elif isinstance(v, (bool, np.bool_)):
            d[k] = bool(v)
        else:
            d[k] = str(v)
    return json.dumps(d)


def g_pk_pm():
    """P(k) as solver.fft normalises and writes it (solver.py:497-507, iostream.py:268-304)."""
    import tempfile
    n, nc = 32, 5
    pos, _ = particles_lattice(50, n, 0.8)
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "power"))
        param = base_param(nc, pos.shape[0], save_power_spectrum="yes", base=td, extra="golden")
        acc, pot, extra = solver.pm(pos, param, tables=synth_tables())
        fn = os.path.join(td, "power", "pk_golden_00000.dat")
        pk = np.loadtxt(fn)
    save("pm_newton_pk", pos=pos, pk_table=pk, acc=acc, param_json=np.array(param_to_json(param)))


def g_integrate():
    """G13: 10 leapfrog steps (integration.py:17-264) at 32^3, Newton/FFT/TSC/order 5; and 3 euler steps."""
    n, nc = 32, 5
    pos, psi = particles_lattice(60, n, 0.5)
    vel = (0.02 * psi).astype(np.float32)
    tables = synth_tables()
    for integ, nsteps in (("leapfrog", 10), ("euler", 3)):
        p, v = pos.copy(), vel.copy()
        param = base_param(nc, p.shape[0], integrator=integ)
        param["t"] = float(tables[1](np.log(param["aexp"])))
        acc, pot, extra = solver.pm(p, param, tables=tables)
        acc0 = acc.copy()
        rec = {"pos0": pos, "vel0": vel, "acc0": acc0, "param0_json": np.array(param_to_json(param))}
        t_snap_next = f32(param["t"] + 0.05)
        ts, aexps, snaps = [], [], []
        for s in range(nsteps):
            p, v, acc, pot, extra = integration.integrate(p, v, acc, pot, extra, tables, param, t_snap_next)
            ts.append(param["t"])
            aexps.append(param["aexp"])
            snaps.append(bool(param["write_snapshot"]))
            if s in (0, nsteps - 1):
                rec[f"pos_{s+1}"], rec[f"vel_{s+1}"], rec[f"acc_{s+1}"] = p.copy(), v.copy(), acc.copy()
        rec["t"], rec["aexp"], rec["write_snapshot"] = np.array(ts), np.array(aexps), np.array(snaps)
        rec["t_snap_next"] = t_snap_next
        rec["pot_final"] = pot
        save(f"integrate_{integ}", **rec)


def g_multigrid():
    """G6/G7: transfer operators (mesh.py:14-551) and the linear multigrid pieces (laplacian.py, multigrid.py)."""
    n = 16
    rng = np.random.default_rng(70)
    x = rng.normal(size=(n, n, n)).astype(np.float32)
    b = rng.normal(size=(n, n, n)).astype(np.float32)
    xc = rng.normal(size=(n // 2,) * 3).astype(np.float32)
    out = {"x": x, "b": b, "xc": xc}
    out["restriction"] = mesh.restriction(x)
    out["minus_restriction"] = mesh.minus_restriction(x)
    out["prolongation"] = mesh.prolongation(xc)
    y = x.copy()
    mesh.add_prolongation(y, xc)
    out["add_prolongation"] = y
    out["operator"] = laplacian.operator(x)
    out["residual"] = laplacian.residual(x, b)
    out["restrict_residual"] = laplacian.restrict_residual(x, b)
    out["residual_error"] = laplacian.residual_error(x, b)
    out["truncation_error"] = laplacian.truncation_error(x)
    out["initialise_potential"] = laplacian.initialise_potential(b)
    y = x.copy()
    laplacian.gauss_seidel(y, b, f32(1.25))
    out["gauss_seidel_1"] = y
    y = x.copy()
    laplacian.smoothing(y, b, 3)
    out["smoothing_3"] = y
    # small grids: the coarsest levels of the V-cycle (4^3 and 8^3)
    for m in (4, 8):
        xs = rng.normal(size=(m, m, m)).astype(np.float32)
        bs = rng.normal(size=(m, m, m)).astype(np.float32)
        ys = xs.copy()
        laplacian.gauss_seidel(ys, bs, f32(1.25))
        out[f"small{m}_x"], out[f"small{m}_b"], out[f"small{m}_gs"] = xs, bs, ys
        if m == 8:
            out["small8_restrict_residual"] = laplacian.restrict_residual(xs, bs)
            yp = xs.copy()
            mesh.add_prolongation(yp, out["small4_x"])
            out["small8_add_prolongation"] = yp
    # one V-cycle and a converged solve on a smooth problem
    g = (np.arange(n) + 0.5) / n
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    rhs = (np.sin(2 * np.pi * X) * np.cos(4 * np.pi * Y) + 0.5 * np.sin(6 * np.pi * Z + 1.0)
           + 0.1 * rng.normal(size=X.shape)).astype(np.float32)
    rhs -= rhs.mean(dtype=np.float64).astype(np.float32)
    param = base_param(4, n**3, linear_newton_solver="multigrid")
    param["compute_additional_field"] = False
    phi = laplacian.initialise_potential(rhs)
    out["vc_rhs"], out["vc_phi0"] = rhs, phi.copy()
    multigrid.V_cycle(phi, rhs, param)
    out["vc_phi1"] = phi.copy()
    phi = out["vc_phi0"].copy()
    ncyc = [0]
    orig = multigrid.V_cycle

    def counting(*a, **k):
        if len(a) < 4 and "nlevel" not in k:
            ncyc[0] += 1
        return orig(*a, **k)

    multigrid.V_cycle = counting
    phi = multigrid.linear(phi, rhs, param)
    multigrid.V_cycle = orig
    out["lin_phi"], out["lin_cycles"], out["lin_tolerance"] = phi, ncyc[0], f32(param["tolerance"])
    out["lin_param_json"] = np.array(param_to_json(param))
    save("multigrid_n16", **out)


def fr_param(ncoarse, npart, **kw):
    return base_param(ncoarse, npart, theory="fr", fR_n=1, fR_logfR0=5, linear_newton_solver="multigrid",
                      boxlen=1000.0, **kw)


def g_cubic():
    """G8: f(R) n=1 operator pieces (cubic.py) and the FAS cycle (multigrid.py:88-138, 521-579)."""
    n = 16
    rng = np.random.default_rng(80)
    # the regime solver.py:334-348 produces for |f_R0| = 1e-5, boxlen 1000, a = 0.5: q h^2 ~ -5.5,
    # b = f1*rho + f2 ~ |q| (1 +- 0.4 delta); outside it cubic.initialise_potential cancels catastrophically
    u = (1.0 + 0.1 * rng.normal(size=(n, n, n))).astype(np.float32)
    b = (1415.0 + 170.0 * rng.normal(size=(n, n, n))).astype(np.float32)
    rhs = (0.05 * rng.normal(size=(n, n, n))).astype(np.float32)
    q = f32(-1415.0)
    out = {"u": u, "b": b, "rhs": rhs, "q": q}
    with np.errstate(all="ignore"):
        out["operator"] = cubic.operator(u, b, q)
        out["residual_with_rhs"] = cubic.residual_with_rhs(u, b, q, rhs)
        out["initialise_potential"] = cubic.initialise_potential(b, q)
        P, D = np.meshgrid(np.linspace(-3, 3, 41), np.linspace(-6, 6, 41), indexing="ij")
        pd_ = np.stack([P.ravel(), D.ravel()], axis=1).astype(np.float32)
        out["root_pd"] = pd_
        out["root"] = np.array([cubic.solution_cubic_equation(f32(p), f32(d)) for p, d in pd_], dtype=np.float32)
        y = u.copy()
        cubic.gauss_seidel(y, b, q, f32(1.25))
        out["gauss_seidel"] = y
        y = u.copy()
        cubic.gauss_seidel_with_rhs(y, b, q, rhs, f32(1.25))
        out["gauss_seidel_with_rhs"] = y
        out["residual_error"] = cubic.residual_error(u, b, q)
        out["truncation_error"] = cubic.truncation_error(u, b, q)
    save("cubic_n16", **out)


def g_quartic():
    """f(R) n=2 operator pieces (quartic.py), same regime as g_cubic."""
    n = 16
    rng = np.random.default_rng(81)
    u = (1.0 + 0.1 * rng.normal(size=(n, n, n))).astype(np.float32)
    b = (1415.0 + 170.0 * rng.normal(size=(n, n, n))).astype(np.float32)
    rhs = (0.05 * rng.normal(size=(n, n, n))).astype(np.float32)
    q = f32(-1415.0)
    out = {"u": u, "b": b, "rhs": rhs, "q": q}
    with np.errstate(all="ignore"):
        out["operator"] = quartic.operator(u, b, q)
        out["residual_with_rhs"] = quartic.residual_with_rhs(u, b, q, rhs)
        out["initialise_potential"] = quartic.initialise_potential(b, q)
        P, Q = np.meshgrid(np.linspace(-6, 6, 41), np.linspace(-8, 2, 41), indexing="ij")
        pq = np.stack([P.ravel(), Q.ravel()], axis=1).astype(np.float32)
        out["root_pq"] = pq
        out["root"] = np.array([quartic.solution_quartic_equation(f32(p_), f32(q_)) for p_, q_ in pq], dtype=np.float32)
        y = u.copy()
        quartic.gauss_seidel(y, b, q, f32(1.25))
        out["gauss_seidel"] = y
        y = u.copy()
        quartic.gauss_seidel_with_rhs(y, b, q, rhs, f32(1.25))
        out["gauss_seidel_with_rhs"] = y
        out["residual_error"] = quartic.residual_error(u, b, q)
        out["truncation_error"] = quartic.truncation_error(u, b, q)
    save("quartic_n16", **out)


def g_pm_fr2():
    """solver.pm for f(R) n = 2 (quartic.py), two consecutive calls."""
    n, nc = 32, 5
    pos, _ = particles_lattice(50, n, 0.8)
    param = base_param(nc, n**3, theory="fr", fR_n=2, fR_logfR0=5, linear_newton_solver="multigrid", boxlen=1000.0)
    tables = synth_tables()
    p0 = np.array(param_to_json(param))
    with np.errstate(all="ignore"):
        acc, pot, extra = solver.pm(pos, param, tables=tables)
        rec = {"pos": pos, "param0_json": p0, "acc1": acc.copy(), "pot1": pot.copy(), "extra1": extra.copy(),
               "param1_json": np.array(param_to_json(param))}
        param["nsteps"] = 1
        param["aexp_old"] = param["aexp"]
        param["aexp"] = 0.51
        utils.set_units(param)
        pos2 = pos.copy()
        pos2 += f32(0.2 / n) * np.sin(2 * np.pi * pos[:, ::-1]).astype(np.float32)
        utils.periodic_wrap(pos2)
        acc, pot, extra = solver.pm(pos2, param, pot, extra, tables)
    rec.update(pos2=pos2, acc2=acc, pot2=pot, extra2=extra, param2_json=np.array(param_to_json(param)))
    assert np.isfinite(acc).all()
    print("    fr n=2 finite; q =", param["fR_q"], "extra range", (extra.min(), extra.max()))
    save("pm_fr2_multigrid", **rec)


def g_mond():
    """G9: QUMOND interpolating functions and source-term stencils (mond.py)."""
    n = 16
    rng = np.random.default_rng(90)
    pot = rng.normal(size=(n, n, n)).astype(np.float32)
    ys = np.array([1e-3, 0.05, 0.3, 1.0, 2.5, 10.0, 300.0], dtype=np.float32)
    out = {"pot": pot, "ys": ys, "g0": f32(7.5)}
    out["nu_simple"] = np.array([mond.nu_simple(y) for y in ys], dtype=np.float32)
    out["nu_n2"] = np.array([mond.nu_n(y, 2) for y in ys], dtype=np.float32)
    out["nu_beta"] = np.array([mond.nu_beta(y, f32(1.3)) for y in ys], dtype=np.float32)
    out["nu_gamma"] = np.array([mond.nu_gamma(y, f32(1.3)) for y in ys], dtype=np.float32)
    out["nu_delta"] = np.array([mond.nu_delta(y, f32(1.3)) for y in ys], dtype=np.float32)
    for name, fn, kw in (("simple", mond.rhs_simple, {}), ("n", mond.rhs_n, {"n": 2}),
                         ("beta", mond.rhs_beta, {"beta": f32(1.3)}), ("gamma", mond.rhs_gamma, {"gamma": f32(1.3)}),
                         ("delta", mond.rhs_delta, {"delta": f32(1.3)})):
        o = np.empty_like(pot)
        fn(pot, o, out["g0"], **kw)
        out["rhs_" + name] = o
    save("mond_n16", **out)


def g_pm_mg():
    """G12 (multigrid / f(R) / MOND rows): solver.pm end-to-end, two consecutive calls each so that the
    warm-start branches (solver.py:273-280) are pinned too."""
    n, nc = 32, 5
    pos, _ = particles_lattice(50, n, 0.8)
    cases = (
        ("newton_multigrid", base_param(nc, n**3, linear_newton_solver="multigrid")),
        ("fr_multigrid", fr_param(nc, n**3)),
        ("mond_multigrid", base_param(nc, n**3, theory="mond", mond_function="simple", mond_g0=1.2, mond_alpha=1,
                                      mond_scale_factor_exponent=0, linear_newton_solver="multigrid")),
        ("mond_fft7pt", base_param(nc, n**3, theory="mond", mond_function="simple", mond_g0=1.2, mond_alpha=1,
                                   mond_scale_factor_exponent=0, linear_newton_solver="fft_7pt")),
    )
    tables = synth_tables()
    for tag, param in cases:
        p0 = np.array(param_to_json(param))
        with np.errstate(all="ignore"):
            acc, pot, extra = solver.pm(pos, param, tables=tables)
            rec = {"pos": pos, "param0_json": p0, "acc1": acc.copy(), "pot1": pot.copy(), "extra1": extra.copy(),
                   "param1_json": np.array(param_to_json(param))}
            # second call, warm start from (pot, extra), one step later
            param["nsteps"] = 1
            param["aexp_old"] = param["aexp"]
            param["aexp"] = 0.51
            utils.set_units(param)
            pos2 = pos.copy()
            pos2 += f32(0.2 / n) * np.sin(2 * np.pi * pos[:, ::-1]).astype(np.float32)
            utils.periodic_wrap(pos2)
            acc, pot, extra = solver.pm(pos2, param, pot, extra, tables)
        rec.update(pos2=pos2, acc2=acc, pot2=pot, extra2=extra, param2_json=np.array(param_to_json(param)))
        assert np.isfinite(acc).all(), tag
        print("   ", tag, "finite; |acc|max", np.abs(acc).max(), "extra range",
              (extra.min(), extra.max()) if extra.size else None)
        save("pm_" + tag, **rec)


GROUPS = {
    "deposit": g_deposit, "kspace": g_kspace, "gradient": g_gradient, "utils": g_utils,
    "pm_newton": g_pm_newton, "pk_pm": g_pk_pm, "integrate": g_integrate,
    "multigrid": g_multigrid, "cubic": g_cubic, "mond": g_mond, "pm_mg": g_pm_mg,
    "quartic": g_quartic, "pm_fr2": g_pm_fr2, "kspace_grad": g_kspace_grad, "pm_parametrized": g_pm_parametrized,
}

if __name__ == "__main__":
    which = sys.argv[1:] or list(GROUPS)
    for g in which:
        print(f"[{g}]")
        GROUPS[g]()
