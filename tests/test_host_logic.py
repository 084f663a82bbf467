"""CPU tests of host-side logic that needs no GPU: the particle-order policy, the next-step dt guess, the bench's
per-level byte accounting and the PMC summary reducer."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reorder_policy_on_host_arrays_keeps_the_reference_cadence(monkeypatch):
    """NumPy arrays: re-sort every n_reorder steps (main.py:146-150), nothing else."""
    from pysco_amd import main as amd_main
    from pysco_amd import utils

    calls = []
    monkeypatch.setattr(utils, "reorder_particles", lambda p, v, a, _order_out=None: (calls.append(1), (p, v, a))[1])
    pos = vel = acc = np.zeros((4, 3), np.float32)
    param = {"n_reorder": 3, "ncoarse": 4}
    for step in range(1, 10):
        param["nsteps"] = step
        amd_main.reorder_policy(pos, vel, acc, param)
    assert len(calls) == 3  # steps 3, 6, 9


def test_dt_guess_follows_the_binding_criterion(monkeypatch):
    """integrate()'s guess of the next dt (for the incremental re-sort): scale-factor criterion at the step's end time,
    CFL criteria with their own trends; it must anticipate the switch from one to the other."""
    import bench
    from pysco_amd import integration

    param, tables = bench.make_param(64, 64**3)
    param["rebucket_straggler_fraction"] = 0.0
    seen = {}
    monkeypatch.setattr(integration, "leapfrog", lambda *a, **k: seen.setdefault("dt", a[5]))
    # velocity criterion far from binding: the guess is the scale-factor criterion one step later
    monkeypatch.setattr(integration, "dt_CFL_maxacc", lambda acc, p: np.float32(10.0))
    monkeypatch.setattr(integration, "dt_CFL_maxvel", lambda vel, p: np.float32(10.0))
    integration.integrate(None, None, None, None, None, tables, param, np.float32(3e38))
    dt3 = float(integration.dt_weak_variation(tables[1], param))
    assert float(seen["dt"]) == pytest.approx(dt3, rel=1e-6)
    a_end = np.exp(tables[0](param["t"] + dt3))
    dt3_next = float(tables[1](np.log(1.1 * a_end)) - tables[1](np.log(a_end)))
    assert param["_dt_pred"] == pytest.approx(dt3_next, rel=1e-6) and param["_dt_pred"] > dt3
    # the velocity criterion just above this step's dt, max|v| (~ 1 / dt2) having grown 10 % over the last step: the
    # same increment again over a step of the same length (no acceleration trend known), and it binds next step
    monkeypatch.setattr(integration, "dt_CFL_maxvel", lambda vel, p: np.float32(1.05 * dt3))
    param["_dt_cfl_last"] = (10.0, 1.05 * dt3 * 1.1)
    integration.integrate(None, None, None, None, None, tables, param, np.float32(3e38))
    assert param["_dt_pred"] == pytest.approx(1.05 * dt3 / 1.2 * 1.1, rel=1e-4)
    # ... over a step half as long as the last one, half the increment
    param["_dt_cfl_last"], param["_dt_last"] = (10.0, 1.05 * dt3 * 1.1), 2.0 * float(seen["dt"])
    integration.integrate(None, None, None, None, None, tables, param, np.float32(3e38))
    assert param["_dt_pred"] == pytest.approx(1.05 * dt3 / 1.15 * 1.1, rel=1e-3)
    param.pop("_dt_last")
    # a snapshot right after this step clips the next one
    param["_dt_cfl_last"] = None
    t_snap = param["t"] + dt3 * 1.5
    integration.integrate(None, None, None, None, None, tables, param, np.float64(t_snap))
    assert param["_dt_pred"] == pytest.approx(0.5 * dt3, rel=1e-3)


def test_bench_level_accounting():
    import bench

    assert bench.alg_bytes_level("pm_kick_drift_wrap_deposit_khalo") is None
    assert bench.alg_bytes_level("pm_lap_gauss_seidel@512") == 12.0 * 512**3
    v = bench.alg_bytes_level("pm_lap_vcycle@256")
    assert v == pytest.approx((3 * 12 + 16 + 12) * 8 / 7 * 256**3)
    p, tables = bench.make_param(128, 128**3, "mond")
    assert p["theory"] == "mond" and p["linear_newton_solver"] == "multigrid" and "unit_t" in p
    assert set(bench.CONFIGS) == {"newton_fft", "newton_mg", "fr", "mond"}


def test_pmc_summary_reduces_the_timed_steps_only(tmp_path):
    rows = ["Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value"]
    did = 0

    def add(name, value):
        nonlocal did
        did += 1
        rows.append(f'{did},"{name}",X,{value}')

    add("void interp_ranges_kernel<3, 5, false, false>(float const*)", 5)  # the first pm()
    for step in range(3):  # 1 warm-up + 2 timed
        add("void kdw_deposit_kernel<3>(float*)", 100 * (step + 1))
        add("void fftp_z_r2c(float const*)", 10)
        add("void interp_ranges_kernel<3, 5, true, true>(float const*)", 1000 * (step + 1))
    f, w = tmp_path / "f.csv", tmp_path / "w.csv"
    f.write_text("\n".join(rows) + "\n")
    w.write_text("\n".join(rows) + "\n")
    out = tmp_path / "o.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), str(f), str(w), str(out), "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    d = json.load(open(out))
    assert d["steps"] == 2 and d["warmup"] == 1 and d["ncells"] == 1024
    dep = d["stages"]["pm_kick_drift_wrap_deposit_khalo"]
    assert dep["fetch_bytes_per_call"] == pytest.approx((200 + 300) / 2 * 2048)  # KiB x 2 (gfx950) -> bytes, per call
    assert dep["write_bytes_per_call"] == pytest.approx((200 + 300) / 2 * 1024)
    gat = d["stages"]["pm_gradient_interp_kick_rebucket"]
    assert gat["calls"] == 2 and gat["write_bytes_per_call"] == pytest.approx((2000 + 3000) / 2 * 1024)


def test_table_eval_is_interp1d_bit_for_bit():
    """utils.table_eval: a linear extrapolating interp1d evaluated with interp1d's own operations on scalars -- same
    float64 bits inside, at the knots, between them and outside the range; other interpolators are called as they are."""
    from scipy.interpolate import interp1d

    from pysco_amd import utils

    rng = np.random.default_rng(2)
    x = np.sort(rng.uniform(-3.0, 2.0, 400))
    y = np.cumsum(rng.uniform(0.1, 1.0, 400))
    f = interp1d(x, y, fill_value="extrapolate")
    probes = list(rng.uniform(-4.0, 3.0, 300)) + [x[0], x[-1], x[7], x[0] - 1.0, x[-1] + 2.0, np.float32(0.3), 0]
    for q in probes:
        got, ref = utils.table_eval(f, q), f(q)
        assert np.float64(got).tobytes() == np.float64(ref).tobytes(), (q, got, ref)
    cubic = interp1d(x, y, kind="cubic", fill_value="extrapolate")
    assert float(utils.table_eval(cubic, 0.1)) == float(cubic(0.1))
    bounded = interp1d(x, y)
    assert float(utils.table_eval(bounded, 0.1)) == float(bounded(0.1))
    np.testing.assert_array_equal(utils.table_eval(f, np.array([0.1, 0.2])), f(np.array([0.1, 0.2])))
