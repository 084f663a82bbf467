"""Multi-rank slab decomposition on the CPU: 2 gloo ranks running pysco_amd.slab's orchestration
(ghost-plane exchange, FFT all-to-all transposes, particle migration, all-reduced dt) with the
test-only oracle compute provider, compared with the single-process oracle step."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import pysco_oracle as orc
from tests.helpers import synth_tables, zeldovich_particles

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def run_ranks(world, outdir, backend, n, steps, timeout=600, integrator="leapfrog", flow="sym", theory=None,
              sorted_start=False):
    port = _free_port()
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="2")
    extra = [theory] if theory else []
    if sorted_start:
        extra = [theory or "newton", "sorted"]
    procs = [subprocess.Popen([sys.executable, "-m", "tests.slab_worker", str(r), str(world), port, str(outdir),
                               backend, str(n), str(steps), integrator, flow] + extra, cwd=ROOT, env=env)
             for r in range(world)]
    codes = [p.wait(timeout=timeout) for p in procs]
    assert codes == [0] * world, codes
    return [np.load(os.path.join(outdir, f"rank{r}.npz")) for r in range(world)]


def reference_run(n, steps, integrator="leapfrog", flow="sym", theory=None):
    pos, vel = zeldovich_particles(n, seed=11, rms_cells=1.2, vel_factor=0.5)
    if flow == "asym":
        vel[:] = 0
        vel[(pos[:, 0] > 0.25) & (pos[:, 0] < 0.5), 0] = 1.0
    tables = synth_tables()
    param = {"theory": "newton", "mass_scheme": "TSC", "npart": n**3, "ncoarse": int(np.log2(n)), "Om_m": 0.25733,
             "Om_lambda": 0.742589237, "H0": 72.0, "boxlen": 100.0, "aexp": 0.5, "aexp_old": 0.49,
             "linear_newton_solver": "fft", "gradient_stencil_order": 5, "save_power_spectrum": "no", "nthreads": 1,
             "Courant_factor": 1.0, "max_aexp_stepping": 10, "integrator": "leapfrog", "write_snapshot": False,
             "nsteps": 0}
    param["integrator"] = integrator
    if theory:
        param.update({"Npre": 2, "Npost": 1, "epsrel": 1e-2, "linear_newton_solver": "multigrid"})
        if theory in ("mond", "mond_fft7pt"):
            param.update({"theory": "mond", "mond_function": "simple", "mond_g0": 1.2, "mond_alpha": 1,
                          "mond_scale_factor_exponent": 0})
            if theory == "mond_fft7pt":
                param["linear_newton_solver"] = "fft_7pt"
        elif theory == "fr":
            param.update({"theory": "fr", "fR_n": 1, "fR_logfR0": 5, "boxlen": 1000.0})
    param["t"] = float(tables[1](np.log(param["aexp"])))
    orc.set_units(param)
    p, v = pos.copy(), vel.copy()
    acc, phi, extra = orc.pm(p, param, tables=tables)
    acc0, phi0 = acc.copy(), phi.copy()
    for _ in range(steps):
        p, v, acc, phi, extra = orc.integrate(p, v, acc, phi, extra, tables, param, np.float32(1))
    return dict(pos0=pos, acc0=acc0, phi0=phi0, pos=p, vel=v, acc=acc, phi=phi, t=param["t"], aexp=param["aexp"])


def check_against_reference(ranks, ref, n, world):
    nx = n // world
    ids0 = np.concatenate([r["id0"] for r in ranks])
    assert sorted(ids0.tolist()) == list(range(n**3))  # every particle owned exactly once
    acc0 = np.concatenate([r["acc0"] for r in ranks])
    sa = np.abs(ref["acc0"]).max()
    np.testing.assert_allclose(acc0, ref["acc0"][ids0], rtol=0, atol=2e-5 * sa)
    phi0 = np.concatenate([r["phi0"] for r in ranks], axis=0)
    assert phi0.shape == (n, n, n) and ranks[0]["phi0"].shape[0] == nx
    np.testing.assert_allclose(phi0, ref["phi0"], rtol=0, atol=5e-6 * np.abs(ref["phi0"]).max())
    ids = np.concatenate([r["id"] for r in ranks])
    assert sorted(ids.tolist()) == list(range(n**3))  # migration lost / duplicated nothing
    moved = sum(int((np.floor(ref["pos0"][r["id"], 0] * world).astype(int) != k).sum()) for k, r in enumerate(ranks))
    assert moved > 0, "test should exercise particle migration"
    for k, r in enumerate(ranks):  # every particle sits on the rank that owns its x-plane
        assert (np.floor(r["pos"][:, 0] * world).astype(int).clip(0, world - 1) == k).all()
        assert abs(float(r["t"]) - ref["t"]) < 1e-6 and abs(float(r["aexp"]) - ref["aexp"]) < 1e-6
    pos = np.concatenate([r["pos"] for r in ranks])
    vel = np.concatenate([r["vel"] for r in ranks])
    d = np.abs(pos - ref["pos"][ids])
    d = np.minimum(d, 1 - d)
    assert d.max() < 1e-5, d.max()
    np.testing.assert_allclose(vel, ref["vel"][ids], rtol=0, atol=2e-5 * np.abs(ref["vel"]).max())
    acc = np.concatenate([r["acc"] for r in ranks])
    np.testing.assert_allclose(acc, ref["acc"][ids], rtol=0, atol=5e-5 * sa)
    return moved


def test_two_rank_slab_step_matches_single_process(tmp_path):
    n, steps, world = 32, 2, 2
    ranks = run_ranks(world, tmp_path, "oracle", n, steps)
    moved = check_against_reference(ranks, reference_run(n, steps), n, world)
    print(f"{moved} particles changed rank")


def test_two_rank_one_way_flow_fills_holes_and_appends(tmp_path):
    """Rank 0 only loses particles (holes refilled from its tail), rank 1 only gains (appended)."""
    n, steps, world = 32, 2, 2
    ranks = run_ranks(world, tmp_path, "oracle", n, steps, flow="asym")
    ref = reference_run(n, steps, flow="asym")
    moved = check_against_reference(ranks, ref, n, world)
    n0 = int((np.floor(ref["pos0"][:, 0] * world).astype(int) == 0).sum())
    assert len(ranks[0]["id"]) < n0 < n**3 - len(ranks[1]["id"]) + n0 and moved > 100


def test_two_rank_slab_euler(tmp_path):
    n, steps, world = 32, 2, 2
    ranks = run_ranks(world, tmp_path, "oracle", n, steps, integrator="euler")
    check_against_reference(ranks, reference_run(n, steps, "euler"), n, world)


def test_four_rank_slab_step(tmp_path):
    n, steps, world = 32, 1, 4
    ranks = run_ranks(world, tmp_path, "oracle", n, steps)
    check_against_reference(ranks, reference_run(n, steps), n, world)


def test_eight_rank_slab_step(tmp_path):
    """The shape of `bench.py --gpus 8` in the host logic: eight ranks (gloo), 8 planes each at 64^3 (the fewest a slab
    may own: one row of tiles) -- ring neighbours, eight-way all-to-all splits in the migration and in the FFT
    transposes, three ghost planes out of eight owned."""
    n, steps, world = 64, 1, 8
    ranks = run_ranks(world, tmp_path, "oracle", n, steps)
    check_against_reference(ranks, reference_run(n, steps), n, world)


@pytest.mark.parametrize("theory", ["newton_mg", "mond", "mond_fft7pt", "fr"])
def test_two_rank_grid_solvers_cpu(tmp_path, theory):
    """The halo-exchange multigrid / FAS / QUMOND orchestration of slab_multigrid.py on the CPU: 2 gloo ranks
    with the test-only level-kernel provider (tests/slab_oracle_ops.py: oracle functions on embedded grids)
    against the single-process oracle, at the solvers' own accuracy."""
    n, steps, world = 32, 1, 2
    ranks = run_ranks(world, tmp_path, "oracle", n, steps, theory=theory)
    ref = reference_run(n, steps, theory=theory)
    ids0 = np.concatenate([r["id0"] for r in ranks])
    acc0 = np.concatenate([r["acc0"] for r in ranks])
    np.testing.assert_allclose(acc0, ref["acc0"][ids0], rtol=0, atol=2e-3 * np.abs(ref["acc0"]).max())
    phi0 = np.concatenate([r["phi0"] for r in ranks], axis=0)
    assert phi0.shape == (n, n, n) and ranks[0]["phi0"].shape[0] == n // world   # decomposed, not replicated
    np.testing.assert_allclose(phi0, ref["phi0"], rtol=0, atol=5e-4 * np.abs(ref["phi0"]).max())
    ids = np.concatenate([r["id"] for r in ranks])
    assert sorted(ids.tolist()) == list(range(n**3))
    d = np.abs(np.concatenate([r["pos"] for r in ranks]) - ref["pos"][ids])
    assert np.minimum(d, 1 - d).max() < 1e-4


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_hip(tmp_path):
    """Rehearsal of the multi-GPU path on the 1-GPU box: 2 ranks share cuda:0, gloo stages the
    exchanges through the host, all compute is the HIP slab kernels."""
    n, steps, world = 64, 2, 2
    ranks = run_ranks(world, tmp_path, "hip", n, steps)
    check_against_reference(ranks, reference_run(n, steps), n, world)
    assert all(int(r["khalo"]) == 1 for r in ranks)  # (the Newtonian FFT step keeps its halo buffer on a slab)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_slab_step_without_the_halo_buffer(tmp_path, world, monkeypatch):
    """PMSTEP_SLAB_KHALO=0: the fused kick + deposit flushes every tile into the cleared slab with atomics (the form the
    multigrid / MOND / f(R) slab steps keep): same trajectories."""
    monkeypatch.setenv("PMSTEP_SLAB_KHALO", "0")
    n, steps = 64, 2
    ranks = run_ranks(world, tmp_path, "hip", n, steps)
    check_against_reference(ranks, reference_run(n, steps), n, world)
    assert all(int(r["khalo"]) == 0 for r in ranks)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_slab_ranks_keep_their_streams_tile_sorted(tmp_path, world):
    """The incremental re-sort on slabs (pm_set_destination_codes -> migration -> pm_rebucket_prepare ->
    pm_gradient_interp_kick_rebucket_slab): from a Morton-sorted start every step re-buckets, the particle IDs follow
    the permutation (the comparison with the single-process run is by ID), and the fraction of a rank's particles
    found outside the tile they are filed under stays at the level of the arrivals + mispredictions instead of growing
    with the age of the sort."""
    n, steps = 64, 5
    ranks = run_ranks(world, tmp_path, "hip", n, steps, sorted_start=True)
    check_against_reference(ranks, reference_run(n, steps), n, world)
    for r in ranks:
        assert int(r["rebucketed"]) == steps
        # (every step, not the last alone: the bisection over the keys that used to stand in for the table after a
        # migration filed a tenth of a rank's particles under the wrong tiles whenever a probe landed on an arrival)
        assert max(r["stragglers"]) < 0.03, r["stragglers"]


@pytest.mark.gpu
def test_four_ranks_on_one_gpu_hip(tmp_path):
    """Four ranks (16 planes each at 64^3) sharing cuda:0: four-segment x lines in the pipelined slab FFT, two-tile
    windows, ring neighbours that differ, particles crossing three boundaries."""
    n, steps, world = 64, 2, 4
    ranks = run_ranks(world, tmp_path, "hip", n, steps)
    check_against_reference(ranks, reference_run(n, steps), n, world)
    assert all(int(r["khalo"]) == 1 for r in ranks)


@pytest.mark.gpu
def test_four_ranks_mond_multigrid_hip(tmp_path):
    """QUMOND with halo-exchange multigrid on four ranks: three distributed levels (64, 32, 16), replicated below."""
    n, steps, world = 64, 1, 4
    ranks = run_ranks(world, tmp_path, "hip", n, steps, theory="mond")
    ref = reference_run(n, steps, theory="mond")
    ids0 = np.concatenate([r["id0"] for r in ranks])
    acc0 = np.concatenate([r["acc0"] for r in ranks])
    np.testing.assert_allclose(acc0, ref["acc0"][ids0], rtol=0, atol=2e-3 * np.abs(ref["acc0"]).max())
    phi0 = np.concatenate([r["phi0"] for r in ranks], axis=0)
    np.testing.assert_allclose(phi0, ref["phi0"], rtol=0, atol=5e-4 * np.abs(ref["phi0"]).max())
    ids = np.concatenate([r["id"] for r in ranks])
    assert sorted(ids.tolist()) == list(range(n**3))
    d = np.abs(np.concatenate([r["pos"] for r in ranks]) - ref["pos"][ids])
    assert np.minimum(d, 1 - d).max() < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("theory", ["newton_mg", "mond", "mond_fft7pt", "fr", "newton_mg_replicated", "mond_replicated",
                                    "fr_replicated"])
def test_two_ranks_grid_solvers_hip(tmp_path, theory):
    """Multigrid / MOND / f(R) under slabs (HIP kernels, 2 ranks sharing cuda:0 over gloo) against the
    single-process oracle, at the solvers' own accuracy.  Decomposed: halo-exchange V-cycles and FAS cycles
    (slab_multigrid.py), the MOND source stencil and the f(R) force on slabs, the slab FFT for fft_7pt; the
    `_replicated` variants all-gather the density and solve the grid on every rank."""
    n, steps, world = 32, 1, 2
    ranks = run_ranks(world, tmp_path, "hip", n, steps, theory=theory)
    base = theory.replace("_replicated", "")
    ref = reference_run(n, steps, theory=base)
    ids0 = np.concatenate([r["id0"] for r in ranks])
    acc0 = np.concatenate([r["acc0"] for r in ranks])
    sa = np.abs(ref["acc0"]).max()
    np.testing.assert_allclose(acc0, ref["acc0"][ids0], rtol=0, atol=2e-3 * sa)
    decomposed = theory in ("newton_mg", "mond", "mond_fft7pt", "fr")
    sp = np.abs(ref["phi0"]).max()
    if decomposed:  # every rank holds its planes of the potential
        phi0 = np.concatenate([r["phi0"] for r in ranks], axis=0)
        assert phi0.shape == (n, n, n)
        np.testing.assert_allclose(phi0, ref["phi0"], rtol=0, atol=5e-4 * sp)
    else:           # every rank holds the full potential (warm start of the next solve)
        for r in ranks:
            assert r["phi0"].shape == (n, n, n)
            np.testing.assert_allclose(r["phi0"], ref["phi0"], rtol=0, atol=5e-4 * sp)
    ids = np.concatenate([r["id"] for r in ranks])
    assert sorted(ids.tolist()) == list(range(n**3))
    pos = np.concatenate([r["pos"] for r in ranks])
    d = np.abs(pos - ref["pos"][ids])
    assert np.minimum(d, 1 - d).max() < 1e-4
