"""GPU tests of the incremental re-sort (pm_gradient_interp_kick_rebucket): the gather writing the particle arrays
out of place in tile order.  The values must be those of the in-place step (pinned against the oracle and the
goldens in test_gpu_newton.py); only the particle order may differ, and int32 tags follow it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import pysco_oracle as orc  # noqa: E402  (checker only)
from tests.helpers import zeldovich_particles  # noqa: E402


def _tile_keys(pos, n):
    """Tile of every particle in stream order: 8 x 16 x 16 cells, Morton order of the cells with x most significant
     == cell-level Morton code >> 11."""
    c = np.minimum((pos.astype(np.float32) * np.float32(n)).astype(np.int64), n - 1)

    def spread(v):
        out = np.zeros_like(v)
        for b in range(11):
            out |= ((v >> b) & 1) << (3 * b)
        return out

    return (spread(c[:, 0]) << 2 | spread(c[:, 1]) << 1 | spread(c[:, 2])) >> 11


def _run(n, pos_h, vel_h, steps, rebucket, sort_first=True, scheme="TSC", order=5, predict=False, config="newton_fft"):
    import torch

    import bench
    from pysco_amd import _lib as L
    from pysco_amd import integration, solver, utils

    L._ctxs.clear()
    integration._PREDICT = predict
    param, tables = bench.make_param(n, pos_h.shape[0], config)
    param["mass_scheme"], param["gradient_stencil_order"] = scheme, order
    if scheme == "NGP":
        param["allow_ngp"] = True  # (the reference's pm() refuses NGP; the fused kernels serve it on request)
    if rebucket is not None:
        param["rebucket_straggler_fraction"] = rebucket
    pos, vel = torch.from_numpy(pos_h).cuda(), torch.from_numpy(vel_h).cuda()
    ids = torch.arange(pos.shape[0], dtype=torch.int32, device="cuda")
    if sort_first:
        order_out = []
        pos, vel = utils.reorder_particles(pos, vel, _order_out=order_out)
        ids = ids[order_out[0]].contiguous()
    param["particle_id"] = ids
    acc, pot, extra = solver.pm(pos, param, tables=tables)
    fracs = []
    for _ in range(steps):
        pos, vel, acc, pot, extra = integration.integrate(pos, vel, acc, pot, extra, tables, param, np.float32(3e38))
        fracs.append(utils.straggler_fraction(n)[0])
    integration._PREDICT = True
    ids = param["particle_id"].cpu().numpy()
    assert np.array_equal(np.sort(ids), np.arange(pos.shape[0]))  # a permutation: nobody lost, nobody twice
    inv = np.argsort(ids)
    return pos.cpu().numpy(), vel.cpu().numpy(), acc.cpu().numpy(), inv, fracs, param


@pytest.mark.parametrize("n,npart_1d,scheme,order", [(64, 64, "TSC", 5), (64, 48, "CIC", 3), (32, 64, "TSC", 7),
                                                     (128, 96, "TSC", 2), (64, 64, "NGP", 5), (32, 48, "NGP", 2)])
def test_rebucket_steps_match_inplace_steps(n, npart_1d, scheme, order):
    """Four leapfrog steps re-bucketing in every one of them vs the same steps in place: same particles (by tag),
    same positions / velocities / accelerations; the stream comes out tile-sorted; (32, 64): 8 particles per
    cell, every tile range is split over several workgroups."""
    pos_h, vel_h = zeldovich_particles(npart_1d, seed=3, rms_cells=0.7 * npart_1d / n, vel_factor=0.3)
    ref = _run(n, pos_h, vel_h, 4, None, scheme=scheme, order=order)
    new = _run(n, pos_h, vel_h, 4, 0.0, scheme=scheme, order=order)
    assert ref[5]["t"] == pytest.approx(new[5]["t"], rel=1e-6)
    for k, tol in ((0, 2e-6), (1, None), (2, None)):
        a, b = ref[k][ref[3]], new[k][new[3]]
        if k == 0:
            d = np.abs(a - b)
            assert np.minimum(d, 1 - d).max() <= tol
        else:
            assert np.abs(a - b).max() <= 2e-4 * np.abs(a).max()
    keys = _tile_keys(new[0], n)
    assert np.all(np.diff(keys) >= 0), "re-bucketed stream is not in tile order"
    assert max(ref[4]) > 0  # (the in-place stream has stragglers)


def test_rebucket_sorts_an_unsorted_stream_and_keeps_oracle_parity():
    """An unsorted (lattice-order) stream: every particle is a straggler, the lists hold them all, and one
    re-bucketing step is a full tile sort.  Values against the CPU oracle's leapfrog step, particle by tag."""
    n = 32
    pos_h, vel_h = zeldovich_particles(n, seed=9, rms_cells=0.5, vel_factor=0.3)
    pos, vel, acc, inv, fracs, param = _run(n, pos_h, vel_h, 2, 0.0, sort_first=False)
    assert fracs[0] > 0.2 and fracs[1] < 0.5 * fracs[0]  # second step: only that step's crossers
    import bench

    par, tables = bench.make_param(n, n**3)
    p, v = pos_h.copy(), vel_h.copy()
    a, phi, extra = orc.pm(p, par, tables=tables)
    for _ in range(2):
        p, v, a, phi, extra = orc.integrate(p, v, a, phi, extra, tables, par, np.float32(3e38))
    d = np.abs(pos[inv] - p)
    assert np.minimum(d, 1 - d).max() < 1e-5
    assert np.abs(vel[inv] - v).max() <= 2e-5 * np.abs(v).max()
    assert np.abs(acc[inv] - a).max() <= 2e-4 * np.abs(a).max()
    assert np.all(np.diff(_tile_keys(pos, n)) >= 0)


def test_rebucket_threshold_policy_and_table_reuse():
    """With a threshold the steps alternate between in-place and re-bucketing ones; a re-bucketed stream enters the
    next fused kick with its tile offsets on file (no range search) and shows only that step's crossers."""
    n = 64
    pos_h, vel_h = zeldovich_particles(n, seed=4, rms_cells=0.6, vel_factor=0.3)
    ref = _run(n, pos_h, vel_h, 6, None)
    new = _run(n, pos_h, vel_h, 6, 0.005)
    assert sum(new[4]) < sum(ref[4])  # the stragglers no longer pile up
    d = np.abs(ref[0][ref[3]] - new[0][new[3]])
    assert np.minimum(d, 1 - d).max() <= 3e-6


@pytest.mark.parametrize("n,npart_1d", [(64, 64), (32, 64), (16, 32)])
def test_rebucket_with_predicted_destinations(n, npart_1d):
    """Filing every particle under the tile it is expected to be in after the NEXT drift (pm_set_predicted_dt):
    same values as the in-place steps, and far fewer stragglers per step than filing under the current tile.
    (16, 32): two tiles along x, where the -1 and +1 neighbours are the same tile."""
    pos_h, vel_h = zeldovich_particles(npart_1d, seed=6, rms_cells=0.7 * npart_1d / n, vel_factor=0.5)
    ref = _run(n, pos_h, vel_h, 6, None)
    exact = _run(n, pos_h, vel_h, 6, 0.0)
    pred = _run(n, pos_h, vel_h, 6, 0.0, predict=True)
    for k in range(3):
        a, b = ref[k][ref[3]], pred[k][pred[3]]
        if k == 0:
            d = np.abs(a - b)
            assert np.minimum(d, 1 - d).max() <= 3e-6
        else:
            assert np.abs(a - b).max() <= 2e-4 * np.abs(a).max()
    if n >= 32:
        assert sum(pred[4][2:]) < 0.5 * sum(exact[4][2:]), (pred[4], exact[4])


def test_rebucket_requires_prepared_lists():
    import torch

    from pysco_amd import _lib as L
    from pysco_amd import mesh

    n = 32
    pos = torch.rand((1000, 3), device="cuda")
    vel = torch.zeros_like(pos)
    phi = torch.zeros((n, n, n), device="cuda")
    L._ctxs.clear()
    with pytest.raises(L.PMStepError, match="no stay counts prepared"):
        mesh.derivative_interp_vec_kick_rebucket("tsc", phi, 5, pos, vel, 0.1)


def test_rebucket_on_the_force_grid_path_of_fR():
    """f(R): the force is not the gradient of one potential, the gather reads the [n, n, n, 3] force grid
    (order 0 of pm_gradient_interp_kick_rebucket).  Same trajectories as the in-place steps."""
    n = 32
    pos_h, vel_h = zeldovich_particles(n, seed=5, rms_cells=0.5, vel_factor=0.3)
    ref = _run(n, pos_h, vel_h, 3, None, config="fr")
    new = _run(n, pos_h, vel_h, 3, 0.0, predict=True, config="fr")
    assert np.isfinite(new[0]).all()
    d = np.abs(ref[0][ref[3]] - new[0][new[3]])
    assert np.minimum(d, 1 - d).max() <= 5e-6
    assert np.abs(ref[2][ref[3]] - new[2][new[3]]).max() <= 2e-3 * np.abs(ref[2]).max()


def test_acc_meanabs_sample_and_predicted_scale():
    """pm_last_acc_meanabs: the fused gather's sampled mean |a_x| (every sixteenth tile; what integrate() extrapolates
    from step to step) -- its step-to-step ratio against that of the whole array; pm_set_predicted_acc_scale changes
    which tile a particle is filed under, never a value.  The sample rides on the read-back of pm_last_maxabs."""
    import ctypes as C

    import torch

    import bench
    from pysco_amd import _lib as L
    from pysco_amd import integration, solver, utils

    n = 128
    pos_h, vel_h = zeldovich_particles(n, seed=11, rms_cells=0.8, vel_factor=0.5)
    out = {}
    for scale in (1.0, 1.3):
        L._ctxs.clear()
        param, tables = bench.make_param(n, pos_h.shape[0], "newton_fft")
        param["rebucket_straggler_fraction"] = 0.0
        pos, vel = utils.reorder_particles(torch.from_numpy(pos_h).cuda(), torch.from_numpy(vel_h).cuda())
        param["particle_id"] = torch.arange(pos.shape[0], dtype=torch.int32, device="cuda")
        acc, pot, extra = solver.pm(pos, param, tables=tables)
        orig = L.Context.call
        sampled, full = [], []
        try:
            for step in range(4):
                pos, vel, acc, pot, extra = integration.integrate(pos, vel, acc, pot, extra, tables, param, np.float32(3e38))
                sampled.append(utils.fused_acc_meanabs(acc))
                full.append(float(acc[:, 0].abs().double().mean()))
                if step == 0 and scale != 1.0:  # from the second step on: a caller's own scale instead of the library's

                    def call(self, name, *args, _orig=orig, _scale=scale):
                        if name == "pm_set_predicted_acc_scale":
                            args = (_scale,)
                        return _orig(self, name, *args)

                    L.Context.call = call
        finally:
            L.Context.call = orig
        assert all(m is not None and 0.3 < m / f < 3.0 for m, f in zip(sampled, full)), (sampled, full)
        for k in range(1, 4):  # the growth from step to step is what the sample is for
            assert abs((sampled[k] / sampled[k - 1]) / (full[k] / full[k - 1]) - 1.0) < 0.02, (sampled, full)
        got = C.c_double(0)
        L.context(n).call("pm_last_acc_meanabs", C.byref(got))
        assert got.value == sampled[-1]
        inv = np.argsort(param["particle_id"].cpu().numpy())
        out[scale] = (pos.cpu().numpy()[inv], vel.cpu().numpy()[inv], acc.cpu().numpy()[inv], param.get("_acc_scale_pred"))
    assert 0.5 <= out[1.0][3] <= 2.0 and out[1.0][3] != 1.0
    for k in range(3):
        d = np.abs(out[1.0][k] - out[1.3][k])
        if k == 0:
            d = np.minimum(d, 1 - d)
        assert d.max() <= 2e-4 * max(1e-30, np.abs(out[1.0][k]).max())


@pytest.mark.parametrize("fused_scan", [False, True])
def test_rebucket_from_codes_alone(fused_scan):
    """The slab form of the re-sort on one GPU: destination codes in a caller's array (here: some "stay", some "one
    tile up in y"), counted by pm_rebucket_prepare -- or by pm_tile_lists_codes, in the pass that makes the lists --
    for a stream that is NOT tile-sorted (a fifth of it shuffled: those are served from the lists), then
    pm_gradient_interp_kick_rebucket_slab.  Every particle must land in the range of the tile its code names, with
    the acceleration and kicked velocity pm_gradient_interp_kick gives it."""
    import torch

    from pysco_amd import _lib as L
    from pysco_amd import mesh, utils

    n, npart = 64, 200_000
    L._ctxs.clear()
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    pos = torch.rand((npart, 3), generator=g, device="cuda")
    vel = torch.rand((npart, 3), generator=g, device="cuda") - 0.5
    pos, vel = utils.reorder_particles(pos, vel)
    swap = torch.randperm(npart, generator=g, device="cuda")[: npart // 5]
    pos[swap], vel[swap] = pos[swap.flip(0)].clone(), vel[swap.flip(0)].clone()  # strangers in their ranges
    phi = torch.rand((n, n, n), generator=g, device="cuda")
    codes = torch.full((npart,), 13, dtype=torch.uint8, device="cuda")
    up = torch.rand(npart, generator=g, device="cuda") < 0.3
    codes[up] = 13 + 3  # dtj = +1
    codes[:7] = 200     # rubbish is read as 26, never out of bounds
    half_dt = 0.01
    ref_vel = vel.clone()
    ref_acc = mesh.derivative_interp_vec_kick("tsc", phi, 5, pos, ref_vel, half_dt) if hasattr(mesh, "derivative_interp_vec_kick") else None
    ctx = L.context(n)
    if fused_scan:
        ctx.call("pm_tile_lists_codes", L.ptr(pos), npart, L.ptr(codes))
    else:
        ctx.call("pm_tile_lists", L.ptr(pos), npart)
        ctx.call("pm_rebucket_prepare", L.ptr(pos), npart, L.ptr(codes))
    pos_out, vel_out, acc_out = torch.empty_like(pos), torch.empty_like(vel), torch.empty_like(pos)
    ids = torch.arange(npart, dtype=torch.int32, device="cuda")
    ids_out = torch.empty_like(ids)
    ctx.call("pm_gradient_interp_kick_rebucket_slab", 5, L.TSC, L.ptr(phi), n, 0, L.ptr(pos), L.ptr(vel), npart,
             float(np.float32(half_dt)), L.ptr(pos_out), L.ptr(vel_out), L.ptr(acc_out), L.ptr(ids), L.ptr(ids_out))
    torch.cuda.synchronize()
    perm = ids_out.long().cpu().numpy()
    assert np.array_equal(np.sort(perm), np.arange(npart))
    assert torch.equal(pos_out.cpu(), pos.cpu()[perm])
    # destination tiles in stream order must be non-decreasing, and equal to "own tile + code"
    c = np.minimum(codes.cpu().numpy().astype(int), 26)[perm]
    cell = np.minimum((pos_out.cpu().numpy() * np.float32(n)).astype(np.int64), n - 1)
    ti, tj, tk = cell[:, 0] >> 3, cell[:, 1] >> 4, cell[:, 2] >> 4
    T = (n >> 3, n >> 4, n >> 4)
    di, dj, dk = c // 9 - 1, (c // 3) % 3 - 1, c % 3 - 1

    corner = np.stack([((ti + di) % T[0]) * 8, ((tj + dj) % T[1]) * 16, ((tk + dk) % T[2]) * 16], axis=1)
    dest = _tile_keys(((corner + 0.5) / n).astype(np.float32), n)  # key of the tile every particle was sent to
    assert (np.diff(dest) >= 0).all()
    if ref_acc is not None:
        assert torch.equal(acc_out.cpu(), ref_acc.cpu()[perm])
        assert torch.equal(vel_out.cpu(), ref_vel.cpu()[perm])


@pytest.mark.parametrize("change", [+500, -300, 0, +20000])
def test_sorted_stream_resized_after_a_migration(change):
    """What a slab rank does between two gathers, on one GPU: the re-bucketing gather leaves a tile-sorted stream and
    its offsets on file; a "migration" then overwrites some particles in place (arrivals in the leavers' holes) and
    appends a surplus / drops the tail.  pm_sorted_stream_resized keeps the offsets (clamped; the surplus filed under
    the LAST TILE OF THE WINDOW, which is not the last tile of the mesh here): the next scan finds the strangers one by
    one -- not a large part of the stream, as the bisected envelope of the keys did whenever a probe landed on an
    arrival (change != 0 here; the same count keeps the offsets anyway) -- and the next gather serves every particle,
    appended ones included.  change = +20000 takes the count past the 1/16 headroom of the re-sort's buffer, which
    holds the offsets on file: the buffer is re-allocated and the table must move with it (it used to be freed under
    the pointer the scan had just taken)."""
    import ctypes as C

    import torch

    from pysco_amd import _lib as L
    from pysco_amd import mesh, utils

    n, npart, holes = 64, 120_000, 64
    x_lo, nx = 16, 16  # two rows of tiles out of eight
    L._ctxs.clear()
    g = torch.Generator(device="cuda")
    g.manual_seed(17 + change)

    def in_window(count):
        p = torch.rand((count, 3), generator=g, device="cuda")
        p[:, 0] = (x_lo + p[:, 0] * nx) / n
        return p.clamp_(0, 1 - 2.0**-24)

    pos, _ = utils.reorder_particles(in_window(npart), torch.zeros((npart, 3), device="cuda"))
    vel = torch.rand((npart, 3), generator=g, device="cuda") - 0.5
    phi = torch.rand((n, n, n), generator=g, device="cuda")
    ctx = L.context(n)
    cap = npart + max(change, 0) + 1000
    half_dt = float(np.float32(0.01))

    def gather(p, v, count, codes):
        po, vo, ao = (torch.empty((cap, 3), device="cuda") for _ in range(3))
        ids = torch.arange(count, dtype=torch.int32, device="cuda")
        ids_out = torch.empty_like(ids)
        ctx.call("pm_tile_lists_codes", L.ptr(p), count, L.ptr(codes))
        s, c, ok = L._i64(0), L._i64(0), L._i(0)
        ctx.call("pm_tile_lists_info", C.byref(s), C.byref(c), C.byref(ok))
        assert ok.value == 1 and c.value == count
        ctx.call("pm_gradient_interp_kick_rebucket_slab", 5, L.TSC, L.ptr(phi), n, 0, L.ptr(p), L.ptr(v), count, half_dt,
                 L.ptr(po), L.ptr(vo), L.ptr(ao), L.ptr(ids), L.ptr(ids_out))
        return po, vo, ao, ids_out.long(), int(s.value)

    ctx.call("pm_set_tile_window", x_lo, nx)
    try:
        codes = torch.full((cap,), 13, dtype=torch.uint8, device="cuda")
        p1, v1, _, _, s0 = gather(pos, vel, npart, codes)
        assert s0 == 0
        # the migration: arrivals in holes (one of them, in the last tile of the window, exactly where the bisection
        # over the keys of the new count probes first: without the offsets on file every search for a tile of the
        # upper half turns left there), then the surplus appended / the tail dropped
        n_new = npart + change
        where = torch.randperm(npart - 400, generator=g, device="cuda")[:holes] + 50
        where[where == n_new >> 1] += 1
        where[0] = n_new >> 1
        arrivals = in_window(holes)
        arrivals[0] = torch.tensor([(x_lo + nx - 0.5) / n, 1 - 0.5 / n, 1 - 0.5 / n], device="cuda")
        p1[where] = arrivals
        if change > 0:
            p1[npart:n_new] = in_window(change)
            v1[npart:n_new] = 0.25
        want_v = v1[:n_new].clone()
        want_a = mesh.derivative_interp_vec_kick("tsc", phi, 5, p1[:n_new].clone(), want_v, half_dt)
        ctx.call("pm_sorted_stream_resized", L.ptr(p1), n_new)
        p2, v2, a2, perm, s1 = gather(p1, v1, n_new, codes)
        assert s1 <= holes + max(change, 0), s1          # strangers one by one
        assert s1 >= holes // 2                          # ... and they are found
        assert torch.equal(torch.sort(perm).values, torch.arange(n_new, device="cuda"))
        assert torch.equal(p2[:n_new], p1[:n_new][perm])
        assert torch.equal(a2[:n_new], want_a[perm]) and torch.equal(v2[:n_new], want_v[perm])
    finally:
        ctx.call("pm_set_tile_window", 0, n)
        ctx.call("pm_tile_lists", None, 0)


def test_bench_size_rebucket_matches_inplace():
    """BASELINE config 3 size (1024^3 / 1024^3), where the oracle cannot go: three leapfrog steps with the incremental
    re-sort (prediction on: the product's loop) against the same steps in place, compared particle by particle through
    the int32 tags; plus: the tags are a permutation, the stream leaves every step tile-sorted (a handful of
    stragglers), no particle is lost."""
    import torch

    import bench
    from pysco_amd import _lib as L
    from pysco_amd import integration, mesh, solver, utils

    if torch.cuda.get_device_properties(0).total_memory < 150e9:
        pytest.skip("needs the 288 GB of an MI355X")
    n = 1024
    pos0, vel0 = bench.zeldovich_device(n)
    utils.periodic_wrap(pos0)
    pos0, vel0 = utils.reorder_particles(pos0, vel0)
    out = {}
    for rebucket in (None, 0.0):
        L._ctxs.clear()
        mesh._rebucket_pool.clear()
        torch.cuda.empty_cache()
        param, tables = bench.make_param(n, n**3)
        if rebucket is not None:
            param["rebucket_straggler_fraction"] = rebucket
            param["particle_id"] = torch.arange(n**3, dtype=torch.int32, device="cuda")
        p, v = pos0.clone(), vel0.clone()
        acc, phi, extra = solver.pm(p, param, tables=tables)
        fracs = []
        for _ in range(3):
            p, v, acc, phi, extra = integration.integrate(p, v, acc, phi, extra, tables, param, np.float32(3e38))
            fracs.append(utils.straggler_fraction(n)[0])
        out[rebucket] = (p, v, acc, param.get("particle_id"), fracs)
        del phi, extra
    mesh._rebucket_pool.clear()
    (p0, v0, a0, _, f0), (p1, v1, a1, ids, f1) = out[None], out[0.0]
    assert p1.data_ptr() != pos0.data_ptr() and ids is not None
    seen = torch.zeros(n**3, dtype=torch.bool, device="cuda")
    seen[ids.long()] = True
    assert bool(seen.all())  # a permutation: nobody lost, nobody twice
    del seen
    idx = ids.long()
    d = (p1 - p0[idx]).abs()
    d = torch.minimum(d, 1 - d)
    assert float(d.max()) < 1e-6  # (the potential differs by the order of the deposit's atomics)
    del d
    assert float((v1 - v0[idx]).abs().max()) < 2e-4 * float(v0.abs().max())
    assert float((a1 - a0[idx]).abs().max()) < 2e-4 * float(a0.abs().max())
    assert f1[-1] < 1e-3 and f1[-1] < 0.2 * f0[-1], (f0, f1)  # filed where they are: the in-place stream ages instead


def test_kick_rebucket_overflow_falls_back_to_the_gather():
    """A cluster sitting on a tile face and moving across it: more movers into one tile than its incoming list holds
    (256 + a quarter of its own + an eighth of its x-face neighbours).  The gather raises the overflow flag, the next
    fused kick runs in place, integration.leapfrog re-buckets with the gather itself for _KR_HOLD_STEPS steps -- and the
    particles are the in-place run's (by tag), nobody lost, the stream tile-sorted again at the end."""
    n = 64
    pos_h, vel_h = zeldovich_particles(32, seed=12, rms_cells=0.3, vel_factor=0.05)
    rng = np.random.default_rng(5)
    m = 60000
    blob = np.empty((m, 3), np.float32)
    blob[:, 0] = (24.0 - rng.random(m) * 0.15) / n        # just below the x face between tiles 2 and 3
    blob[:, 1] = (20.0 + rng.random(m) * 8.0) / n
    blob[:, 2] = (36.0 + rng.random(m) * 8.0) / n
    vb = np.zeros((m, 3), np.float32)
    vb[:, 0] = 40.0 * np.abs(vel_h).max()                   # the fastest particles of the run: they set dt and move ~C cells
    pos_h = np.concatenate([pos_h, blob]).astype(np.float32)
    vel_h = np.concatenate([vel_h, vb]).astype(np.float32)
    ref = _run(n, pos_h, vel_h, 5, None, predict=True)
    new = _run(n, pos_h, vel_h, 5, 0.0, predict=True)
    assert "_kr_hold" in new[5], "the lists were expected to overflow"
    for k in range(3):
        a, b = ref[k][ref[3]], new[k][new[3]]
        if k == 0:
            d = np.abs(a - b)
            assert np.minimum(d, 1 - d).max() <= 3e-6
        else:
            assert np.abs(a - b).max() <= 2e-4 * np.abs(a).max()
    assert new[4][-1] < 0.05, new[4]
