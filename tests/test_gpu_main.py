"""GPU tests of the run driver (pysco_amd.main.run: parameter container -> tables -> initial conditions -> step loop
-> snapshots), SURVEY.md section 8 rows f1 / f2: the example configuration end to end against the same loop written
with the CPU oracle, and a restart from a snapshot against the uninterrupted run."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import pysco_oracle as orc  # noqa: E402  (checker only)
from tests.helpers import zeldovich_particles  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _example_param(tmp_path, **over):
    from pysco_amd import iostream

    param = dict(iostream.read_param_file(os.path.join(ROOT, "examples", "param.ini")).items())
    param.update(base=str(tmp_path), verbose=0, save_power_spectrum="no")
    param.update(over)
    return param


def _write_initial_snapshot(param, pos, vel, aexp):
    """Initial conditions as snapshot 0 of the run (the restart path of initial_conditions.generate)."""
    from pysco_amd import iostream

    extra = f"{param['theory'].casefold()}_{param['linear_newton_solver']}_ncoarse{param['ncoarse']}"
    snap = {"base": param["base"], "extra": extra, "i_snap": 0, "aexp": aexp, "nsteps": 0,
            "output_snapshot_format": param["output_snapshot_format"]}
    iostream.write_snapshot_particles(pos, vel, snap)


@pytest.mark.parametrize("policy", ["in_place", "product"])
def test_example_run_end_to_end_vs_oracle_loop(tmp_path, policy, monkeypatch):
    """examples/param.ini (Newton, 128^3 mesh and particles, TSC, multigrid solver, order-5 gradient, leapfrog): ten
    steps from z = 49 through main.run on the device, snapshots at two redshifts, against the oracle's loop on the
    same tables.  (Initial conditions: a Zel'dovich-like particle set handed in as snapshot 0.)
    policy "in_place": the reference's semantics (arrays updated in place, no re-sort within the run); "product":
    what main.run does when nothing is said -- prediction + incremental re-sort in every step -- with the snapshots'
    particles matched to the oracle's through int32 tags recorded as each snapshot is written."""
    import copy

    from pysco_amd import cosmotable, iostream
    from pysco_amd import main as amd_main

    n = 128
    pos, vel = zeldovich_particles(n, seed=12, rms_cells=0.3, vel_factor=0.02)
    pos, vel = orc.reorder_particles(pos, vel)  # Morton order: the run then has no reason to permute the particles
    z_out = "[30, 18]"
    if policy == "in_place":
        param = _example_param(tmp_path, z_out=z_out, initial_conditions=0, output_snapshot_format="parquet",
                               rebucket_straggler_fraction=None, n_reorder=1000, reorder_straggler_fraction=1.0)
    else:
        param = _example_param(tmp_path, z_out=z_out, initial_conditions=0, output_snapshot_format="parquet")
    _write_initial_snapshot(param, pos, vel, 1.0 / 50)
    ref_param = copy.deepcopy(param)
    tags = []
    if policy == "product":
        param["particle_id"] = True
        write = iostream.write_snapshot_particles

        def write_and_note(position, velocity, par):
            tags.append(par["particle_id"].long().cpu().numpy().copy())
            return write(position, velocity, par)

        monkeypatch.setattr(iostream, "write_snapshot_particles", write_and_note)
    amd_main.run(param)
    assert param["nsteps"] >= 8 and param["i_snap"] == 3
    if policy == "product":
        assert len(tags) == 2 and not np.array_equal(tags[-1], np.arange(n**3))  # (the stream WAS permuted)

    # the same loop with the oracle (main.py:97-156), same tables
    par = ref_param
    par["extra"] = param["extra"]
    tables = cosmotable.generate(par)
    par["aexp"] = 1.0 / 50
    par["write_snapshot"] = False
    orc.set_units(par)
    par["nsteps"] = 0
    par["t"] = float(tables[1](np.log(par["aexp"])))
    p, v = pos.copy(), vel.copy()
    acc, phi, extra = orc.pm(p, par, tables=tables)
    aexp_out = np.sort(1.0 / (np.array([30, 18]) + 1.0))
    t_out = tables[1](np.log(aexp_out))
    par["i_snap"] = 1
    snaps = []
    while par["aexp"] < aexp_out[-1] and par["i_snap"] <= len(t_out):  # (as run_nbody: see its comment)
        par["nsteps"] += 1
        p, v, acc, phi, extra = orc.integrate(p, v, acc, phi, extra, tables, par, t_out[par["i_snap"] - 1])
        if par["write_snapshot"]:
            snaps.append((par["aexp"], p.copy(), v.copy()))
            par["i_snap"] += 1
    assert par["nsteps"] == param["nsteps"]
    assert abs(par["aexp"] - param["aexp"]) < 1e-6
    for i, (a_ref, p_ref, v_ref) in enumerate(snaps, start=1):
        d = f"{tmp_path}/output_{i:05d}"
        p_dev, v_dev = iostream.read_snapshot_particles_parquet(f"{d}/particles_{param['extra']}.parquet")
        side = iostream.read_param_file(f"{d}/param_{param['extra']}_{i:05d}.txt")
        assert abs(float(side["aexp"]) - a_ref) < 1e-6 and int(side["i_snap"]) == i
        if tags:  # back into the order of the initial conditions
            inv = np.argsort(tags[i - 1])
            p_dev, v_dev = p_dev[inv], v_dev[inv]
        dd = np.abs(p_dev - p_ref)
        assert np.minimum(dd, 1 - dd).max() < 1e-5  # positions after ten steps
        assert np.abs(v_dev - v_ref).max() < 2e-3 * np.abs(v_ref).max()


@pytest.mark.parametrize("fmt", ["hdf5", "parquet"])
def test_restart_reproduces_the_uninterrupted_run(tmp_path, fmt):
    """Run A: z = 49 -> 30 -> 22 in one go.  Run B: the same parameter file restarted from A's snapshot 1
    (initial_conditions = 1, initial_conditions.py:79-107).  Same final particles to 1e-5."""
    from pysco_amd import iostream
    from pysco_amd import main as amd_main

    n = 64
    pos, vel = zeldovich_particles(n, seed=2, rms_cells=0.4, vel_factor=0.02)
    pos, vel = orc.reorder_particles(pos, vel)
    common = dict(ncoarse=6, npart=n**3, linear_newton_solver="fft", output_snapshot_format=fmt,
                  rebucket_straggler_fraction=None, n_reorder=1000, reorder_straggler_fraction=1.0)
    a_dir = tmp_path / "a"
    pa = _example_param(a_dir, z_out="[30, 22]", initial_conditions=0, **common)
    _write_initial_snapshot(pa, pos, vel, 1.0 / 50)
    amd_main.run(pa)
    ext = "h5" if fmt == "hdf5" else "parquet"
    read = iostream.read_snapshot_particles_hdf5 if fmt == "hdf5" else iostream.read_snapshot_particles_parquet
    final_a = read(f"{a_dir}/output_00002/particles_{pa['extra']}.{ext}")

    # (a restart takes every parameter of the snapshot, `base` included: it continues in the directory of run A)
    os.rename(f"{a_dir}/output_00002", f"{a_dir}/output_00002_run_a")
    pb = _example_param(a_dir, z_out="[30, 22]", initial_conditions=1, **common)
    amd_main.run(pb)
    assert pb["nsteps"] == pa["nsteps"] and pb["i_snap"] == pa["i_snap"]
    final_b = read(f"{a_dir}/output_00002/particles_{pb['extra']}.{ext}")
    d = np.abs(final_a[0] - final_b[0])
    assert np.minimum(d, 1 - d).max() < 1e-5
    assert np.abs(final_a[1] - final_b[1]).max() < 1e-4 * np.abs(final_a[1]).max()


LPT_CASES = ["1lpt_center_n16", "2lpt_center_n16", "2lpt_edge_fixed_paired_n16", "2lpt_center_n32", "3lpt_center_n16",
             "3lpt_edge_n32", "2lpt_dealiased_n16", "3lpt_dealiased_n16", "2lpt_dealiased_n32"]


@pytest.mark.parametrize("tag", LPT_CASES)
def test_lpt_initial_conditions_vs_reference(golden, tag, tmp_path):
    """pysco_amd.lpt.generate (host-side random field + HIP kernels for every grid operation) against the reference's
    initial_conditions.generate on the same seed, spectrum and growth factors (tests/golden/lpt_*.npz)."""
    import ast

    from pysco_amd import lpt

    g = golden("lpt_" + tag)
    case, sc = ast.literal_eval(str(g["param_json"])), ast.literal_eval(str(g["scalars_json"]))
    f = tmp_path / "pk.dat"
    np.savetxt(f, np.c_[g["k_table"], g["pk_table"]])
    lna = np.log(sc["aexp"])
    const = lambda at_start, at_zero=None: (lambda x: at_start if abs(x - lna) < 1e-9 else at_zero)  # noqa: E731
    tables = [None, None, const(sc["Hz"]), const(sc["d1"], sc["d1_z0"]), const(sc["f1"]), const(sc["d2"]), const(sc["f2"])]
    tables += [const(sc[k]) for k in ("d3a", "f3a", "d3b", "f3b", "d3c", "f3c") if k in sc]
    param = dict(case, power_spectrum_file=str(f), boxlen=100.0, z_start=1.0 / sc["aexp"] - 1, unit_t=sc["unit_t"],
                 dealiased_ICS=case.get("dealiased_ICS", False))
    pos, vel = lpt.generate(param, tables)
    pos, vel = pos.cpu().numpy(), vel.cpu().numpy()
    d = np.abs(pos - g["position"])
    assert np.minimum(d, 1 - d).max() < 3e-6
    assert np.abs(vel - g["velocity"]).max() < 2e-5 * np.abs(g["velocity"]).max()


@pytest.mark.parametrize("up", [1, 0])
@pytest.mark.parametrize("n", [16, 64])
def test_dealias_pad_and_trim_bit_exact(n, up):
    """pm_dealias_resize against initial_conditions.pad / trim as the oracle restates them (pinned through the
    de-aliased LPT goldens): a copy of corner blocks into zeros, so bit for bit -- Nyquist planes dropped both ways."""
    import torch

    from pysco_amd import _lib as L

    m = 3 * n // 2
    n_in = n if up else m
    rng = np.random.default_rng(n + up)
    x = (rng.standard_normal((n_in, n_in, n_in // 2 + 1)) + 1j * rng.standard_normal((n_in, n_in, n_in // 2 + 1))).astype(np.complex64)
    want = orc.pad(x) if up else orc.trim(x)
    out = torch.full(want.shape, float("nan"), dtype=torch.complex64, device="cuda")
    L.context(n).call("pm_dealias_resize", L.ptr(torch.from_numpy(x).cuda()), L.ptr(out), up)
    np.testing.assert_array_equal(out.cpu().numpy(), want)
    assert not want[n // 2].any() and not want[:, n // 2].any() and not want[:, :, n // 2].any()


@pytest.mark.parametrize("order,dealiased", [("2LPT", False), ("3LPT", False), ("3LPT", True)])
def test_lpt_against_the_oracle_at_128(tmp_path, order, dealiased):
    """2LPT / 3LPT at 128^3 (beyond the goldens; de-aliased: products on a 192^3 grid, rocFFT lengths 2^6 x 3): the
    device pipeline against the oracle's, fed the same Fourier density."""
    import torch

    from pysco_amd import cosmotable, lpt, utils

    n = 128
    param = _example_param(tmp_path, initial_conditions=order, npart=n**3, dealiased_ICS=dealiased,
                           power_spectrum_file=os.path.join(ROOT, "examples", "pk_example.dat"))
    del param["base"]  # (no snapshot from this call)
    tables = cosmotable.generate(dict(param, base=str(tmp_path), extra="t"))
    param["aexp"] = 1.0 / (1 + param["z_start"])
    utils.set_units(param)
    pos, vel = lpt.generate(param, tables)
    lna = np.log(param["aexp"])
    sc = {"Hz": float(tables[2](lna)), "d1_z0": float(tables[3](0)), "d1": float(tables[3](lna)), "f1": float(tables[4](lna)),
          "d2": float(tables[5](lna)), "f2": float(tables[6](lna)), "unit_t": param["unit_t"]}
    sc.update({k: float(tables[7 + i](lna)) for i, k in enumerate(("d3a", "f3a", "d3b", "f3b", "d3c", "f3c"))})
    p_ref, v_ref = orc.lpt_particles(lpt.generate_density_fourier(param), order, "center", sc, dealiased=dealiased)
    d = np.abs(pos.cpu().numpy() - p_ref)
    assert np.minimum(d, 1 - d).max() < 3e-6
    assert np.abs(vel.cpu().numpy() - v_ref).max() < 2e-5 * np.abs(v_ref).max()
    lattice_rms = float(torch.as_tensor(np.sqrt((((pos.cpu().numpy().reshape(n, n, n, 3) * n) % 1 - 0.5) ** 2).sum(-1).mean())))
    assert 0.01 < lattice_rms < 1.0  # displaced, not scrambled


def test_run_from_a_power_spectrum_2lpt(tmp_path):
    """examples/param.ini as it stands (initial_conditions = 2LPT from examples/pk_example.dat, multigrid, 128^3):
    main.run end to end on the device, first snapshots written, finite state."""
    from pysco_amd import iostream
    from pysco_amd import main as amd_main

    param = _example_param(tmp_path, z_out="[40, 30]", output_snapshot_format="hdf5",
                           power_spectrum_file=os.path.join(ROOT, "examples", "pk_example.dat"))
    amd_main.run(param)
    assert param["nsteps"] >= 4 and param["i_snap"] == 3
    p0, v0 = iostream.read_snapshot_particles_hdf5(f"{tmp_path}/output_00000/particles_{param['extra']}.h5")
    p2, v2 = iostream.read_snapshot_particles_hdf5(f"{tmp_path}/output_00002/particles_{param['extra']}.h5")
    assert p0.shape == (128**3, 3) and np.isfinite(p2).all() and np.isfinite(v2).all()
    assert 0 <= p2.min() and p2.max() < 1
    attrs = iostream.read_snapshot_attributes_hdf5(f"{tmp_path}/output_00002/particles_{param['extra']}.h5")
    assert abs(float(attrs["aexp"]) - 1.0 / 31) < 1e-6 and int(attrs["nsteps"]) == param["nsteps"]
    # growth: the rms displacement from the lattice grows roughly like D1 between a = 1/50 and 1/31
    def rms(p):
        return np.sqrt(((((p * 128) % 1) - 0.5) ** 2).sum(-1).mean())
    assert rms(p2) > 1.2 * rms(p0)


def test_tune_placement_keeps_the_values_and_fills_the_pool():
    """main.tune_placement: the particle data end up in (possibly different) allocations chosen by timing the update's
    own access pattern over combinations of candidates (pm_probe_update leaves every value as it is); the three arrays
    of the gather's output set are in the pool, distinct from the live ones."""
    import torch

    from pysco_amd import _lib as L
    from pysco_amd import main as amd_main
    from pysco_amd import mesh

    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    pos = torch.rand((300_000, 3), generator=g, device="cuda")
    vel = torch.randn((300_000, 3), generator=g, device="cuda")
    acc = torch.randn((300_000, 3), generator=g, device="cuda")
    vel[7, 1] = -0.0
    ref = [t.clone() for t in (pos, vel, acc)]
    # the probe itself leaves the arrays bit for bit
    L.context(4).call("pm_probe_update", L.ptr(pos), L.ptr(vel), L.ptr(acc), pos.shape[0])
    for t, r in zip((pos, vel, acc), ref):
        assert torch.equal(t.view(torch.int32), r.view(torch.int32))
    try:
        out = amd_main.tune_placement(pos, vel, acc, max_candidates=5)
        assert len(out) == 3
        for t, r in zip(out, ref):
            assert torch.equal(t.view(torch.int32), r.view(torch.int32))
        ptrs = {t.data_ptr() for t in out}
        pool = list(mesh._rebucket_pool)
        assert len(pool) == 3 and len(ptrs | {t.data_ptr() for t in pool}) == 6
        assert isinstance(amd_main.last_placement_probe_ms, dict) and len(amd_main.last_placement_probe_ms["pairs"]) == 28
        two = amd_main.tune_placement(out[0], out[1], None, max_candidates=4)
        assert len(two) == 2 and torch.equal(two[0], ref[0]) and torch.equal(two[1], ref[1])
    finally:
        mesh._rebucket_pool.clear()
