"""The N-body section of the reference's driver (main.py:116-156) on top of solver.pm / integration.integrate:
first force evaluation, step loop up to the last output redshift, re-ordering of the particle stream every
`n_reorder` steps, snapshot hand-off.  Background tables and initial conditions are inputs here (the reference
builds them with astropy / its IC generator, main.py:104-112; both are outside this library's scope).

The re-ordering is more than a locality nicety on the GPU: the tile-range kernels serve every particle from LDS
whatever the order is (per-tile straggler lists), but the lists -- and with them the time per step -- grow with the
age of the sort; `n_reorder` bounds it exactly as it bounds cache misses in the reference.
"""
from __future__ import annotations

import logging
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import _lib as L
from . import integration, real64, solver, utils

__version__ = "1.0.9"  # of the reference release whose main.run this mirrors


REBUCKET_FRACTION = 0.0    # default of param["rebucket_straggler_fraction"] in this loop (device arrays): every step
                           # that found a straggler re-buckets (measured at 1024^3: 47.7 ms per step sustained against
                           # 48.5 / 49.7 with 0.02 / 0.04, where in-place and re-bucketing steps alternate)
REORDER_FRACTION = 0.10    # default of param["reorder_straggler_fraction"]
REORDER_AGE_FRACTION = 0.25  # default of param["reorder_age_fraction"] (fp64 device mode: pm_real64_stream_age)
REORDER_PASSES_PER_CHUNK = 3.5  # default of param["reorder_passes_per_chunk"] (idem: a sorted stream makes ~2)
last_placement_probe_ms = None  # tune_placement's timings of its candidate arrays, ascending (diagnostics)


def reserve_step_buffers(position) -> None:
    """The re-bucketing gather writes position, velocity and acceleration into a second set of arrays, which then
    trade places with the first every step (pysco_amd.mesh._rebucket_pool).  Allocating the second set up front keeps
    the device allocations -- 0.8 s per 12.9 GB array at 1024^3 -- out of the step loop."""
    if L.is_device_array(position):
        from . import mesh

        torch = L.require_gpu()
        while sum(1 for t in mesh._rebucket_pool if t.shape == position.shape and t.device == position.device) < 3:
            mesh._rebucket_pool.append(torch.empty_like(position))


def rank_triples(arrays, npart):
    """The measurement behind tune_placement, on any list of >= 6 device arrays of at least npart x 3 floats: times the
    fused kick's own traffic (pm_probe_update: two arrays rewritten in place, a third read, every value left as it
    is) over every pair, the leaders again best-of-three, and returns ((i, j, k) of the best triple, (i, j, k) of the
    best triple disjoint from it, {timings}) -- (updated, updated, read) = (position, velocity, acceleration).
    The x-slab step places each rank's capacity-backed arrays with it (pysco_amd.slab.Slab._adopt)."""
    torch = L.require_gpu()
    ctx = L.context(4)
    n = len(arrays)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def probe(i, j, k):  # the update's traffic over (arrays[i], arrays[j]) in place + arrays[k] read; x * 1 leaves every value
        e0.record()
        ctx.call("pm_probe_update", L.ptr(arrays[i]), L.ptr(arrays[j]), L.ptr(arrays[k]), npart)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    probe(0, 1, 2)  # (warm-up)
    pairs = []
    for i in range(n):
        for j in range(i + 1, n):
            k = next(m for m in range(n) if m != i and m != j)
            pairs.append((probe(i, j, k), i, j))
    pairs.sort()
    # the leaders again (one timing each is noisy at the 1 % level), best of three
    top = sorted((min(ms, probe(i, j, next(m for m in range(n) if m != i and m != j)),
                      probe(i, j, next(m for m in range(n) if m != i and m != j))), i, j) for ms, i, j in pairs[:8])
    set_a = top[0]
    rest = [t for t in top[1:] if not {t[1], t[2]} & {set_a[1], set_a[2]}]
    set_b = rest[0] if rest else next(t for t in pairs if not {t[1], t[2]} & {set_a[1], set_a[2]})
    used = {set_a[1], set_a[2], set_b[1], set_b[2]}

    def third(pair):  # the array read beside this pair: the best of those still free
        free = [m for m in range(n) if m not in used]
        best = min((probe(pair[1], pair[2], m), m) for m in free)
        used.add(best[1])
        return best

    acc_a, acc_b = third(set_a), third(set_b)
    info = {"pairs": [round(t[0], 3) for t in pairs], "live_set": round(acc_a[0], 3), "output_set": round(acc_b[0], 3)}
    logging.info(f"placement probe [ms]: pairs {pairs[0][0]:.2f} .. {pairs[-1][0]:.2f}, chosen triples {acc_a[0]:.2f} {acc_b[0]:.2f}")
    return (set_a[1], set_a[2], acc_a[1]), (set_b[1], set_b[2], acc_b[1]), info


def tune_placement(position, velocity, acceleration=None, max_candidates: int = 10):
    """Measured on MI355X: how fast the fused kick + drift + deposit streams through the particle arrays -- position
    and velocity updated in place, acceleration read -- depends on the physical memory the allocations landed in, and
    on the COMBINATION of arrays more than on each one: an in-place pass over a single 12.9 GB array runs at 5.08 or
    4.5 TB/s depending on the allocation (tools/bench_array_single.py), and the three-array update takes 11.05 to
    11.85 ms over triples of arrays that are all "fast" on their own, while a "slow" one can be part of the fastest
    triples (tools/exp_triples.py); the kernel then runs 13.2 or 15.1 ms at 1024^3.  Nothing in the virtual layout
    controls it, so this allocates candidate arrays, times the update's own access pattern (pm_probe_update: values
    unchanged) over every PAIR (updated, updated) of candidates, takes the best pair for position and velocity of the
    live set, the best disjoint pair for the set the re-bucketing gather writes (the fused kick updates that one in
    place a step later), then the best third array for each; the rest is freed.  One-time cost at 1024^3: 0.3 s per
    candidate allocated + about 11 ms per combination timed (1 s).  Returns the (possibly new) position, velocity[,
    acceleration] tensors; values and order unchanged."""
    if not L.is_device_array(position):
        return (position, velocity) if acceleration is None else (position, velocity, acceleration)
    from . import mesh

    torch = L.require_gpu()
    live = [position, velocity] + ([acceleration] if acceleration is not None else [])
    free_bytes, _ = torch.cuda.mem_get_info(position.device)
    each = position.numel() * 4
    # (the candidates that are not kept are freed again before anything else is allocated: they may use what is free now)
    n_cand = int(max(0, min(max_candidates, free_bytes // each - 2)))
    mesh._rebucket_pool.clear()
    cand = []
    try:
        for _ in range(n_cand):
            cand.append(torch.zeros_like(position))
    except torch.OutOfMemoryError:
        pass
    arrays = live + cand
    npart = position.shape[0]
    global last_placement_probe_ms
    if npart % 4 or len(arrays) < 6:  # (the probe streams whole float4; no room for a second set: leave things as they are)
        last_placement_probe_ms = None
        reserve_step_buffers(position)
        return tuple(live)
    (ia, ja, ka), (ib, jb, kb), last_placement_probe_ms = rank_triples(arrays, npart)
    set_a, set_b, acc_a, acc_b = (0, ia, ja), (0, ib, jb), (0, ka), (0, kb)
    homes = [arrays[set_a[1]], arrays[set_a[2]], arrays[acc_a[1]]][:len(live)]
    spare = [arrays[set_b[1]], arrays[set_b[2]], arrays[acc_b[1]]]
    same = lambda a, b: a.data_ptr() == b.data_ptr()  # noqa: E731
    # move the data: a home that still holds another live array's data waits until that array has moved (a cycle is
    # broken through a scratch copy)
    pending = [(x, h) for x, h in zip(live, homes) if not same(x, h)]
    while pending:
        ready = [k for k, (_, h) in enumerate(pending) if not any(same(h, x2) for x2, _ in pending)]
        if ready:
            x, h = pending.pop(ready[0])
            h.copy_(x)
        else:
            x, h = pending[0]
            pending[0] = (x.clone(), h)
    mesh._rebucket_pool.extend(spare)
    out = homes
    del cand, arrays, spare
    torch.cuda.empty_cache()
    return tuple(out)


def reorder_policy(position, velocity, acceleration, param):
    """What the loop does about the particle order after a step (main.py:146-150 re-sorts every n_reorder steps;
    on the GPU the order is a performance matter only: every kernel serves every particle whatever it is).
    Device arrays: integration.integrate keeps the stream tile-sorted on its own (incremental re-sort inside the
    gather, param["rebucket_straggler_fraction"]); a full Morton sort happens only when the stream has aged past
    param["reorder_straggler_fraction"] anyway or the straggler lists overflowed (unsorted input), never on the
    fixed cadence.  NumPy arrays or no incremental re-sort: the reference's cadence, plus the aged test."""
    device = L.is_device_array(position)
    frac, complete = (None, True)
    if device and real64.is_float64(position):
        # the fp64 mode steps in place over chunks of the sorted stream; its gather reports the share of particles it
        # served one by one because too few of their chunk shared their tile, and past param["reorder_age_fraction"]
        # a sort to cell resolution pays
        age = real64.stream_age(2 ** int(param["ncoarse"]))
        if age is not None and (age[0] > float(param.get("reorder_age_fraction", REORDER_AGE_FRACTION))
                                or age[1] > float(param.get("reorder_passes_per_chunk", REORDER_PASSES_PER_CHUNK))):
            logging.info(f"Reordering particles (fp64 stream, {age=})")
            return _reorder(position, velocity, acceleration, param, key_bits=-1)
        return position, velocity, acceleration
    if device:
        frac, complete = utils.straggler_fraction(2 ** int(param["ncoarse"]))
    aged = (not complete) or (frac is not None and frac > float(param.get("reorder_straggler_fraction", REORDER_FRACTION)))
    incremental = device and param.get("rebucket_straggler_fraction") is not None
    cadence = (not incremental) and param.get("n_reorder") and (param["nsteps"] % param["n_reorder"]) == 0
    if cadence or aged:
        logging.info(f"Reordering particles ({frac=})")
        position, velocity, acceleration = _reorder(position, velocity, acceleration, param)
    return position, velocity, acceleration


def _reorder(position, velocity, acceleration, param, key_bits=0):
    """utils.reorder_particles, with the optional tags of param["particle_id"] following the permutation.
    key_bits (fp64 arrays): sort to that many bits per axis only (real64.reorder_particles)."""
    ids = param.get("particle_id") if L.is_device_array(position) else None
    order = [] if ids is not None else None
    if key_bits and real64.is_float64(position):
        position, velocity, acceleration = real64.reorder_particles(position, velocity, acceleration, _order_out=order,
                                                                    _key_bits=key_bits, _ncells=2 ** int(param["ncoarse"]))
    else:
        position, velocity, acceleration = utils.reorder_particles(position, velocity, acceleration, _order_out=order)
    if ids is not None:
        param["particle_id"] = ids[order[0]]
    return position, velocity, acceleration


def run_nbody(position, velocity, param, tables: List, z_out: Sequence[float],
              on_snapshot: Optional[Callable] = None, on_step: Optional[Callable] = None):
    """main.py:116-156.  `param` needs what solver.pm / integration.integrate need plus `n_reorder`; `aexp`, `t`
    and the units must be set (main.py:106-113).  `on_snapshot(position, velocity, param)` replaces
    iostream.write_snapshot_particles; `on_step(param)` is called after every step.
    Returns (position, velocity, acceleration, potential, additional_field)."""
    if "nsteps" not in param:
        param["nsteps"] = 0
    if L.is_device_array(position):
        param.setdefault("rebucket_straggler_fraction", REBUCKET_FRACTION)
        if param["rebucket_straggler_fraction"] is not None and not real64.is_float64(position):
            reserve_step_buffers(position)
    acceleration, potential, additional_field = solver.pm(position, param, tables=tables)
    if L.is_device_array(position) and real64.is_float64(position):  # (idem, on the fp64 gather's own measure)
        position, velocity, acceleration = reorder_policy(position, velocity, acceleration, param)
    elif L.is_device_array(position):  # an unsorted input stream (initial conditions in lattice order) is sorted once
        frac, complete = utils.straggler_fraction(2 ** int(param["ncoarse"]))
        if not complete or (frac is not None and frac > float(param.get("reorder_straggler_fraction", REORDER_FRACTION))):
            position, velocity, acceleration = _reorder(position, velocity, acceleration, param)
        if param.get("rebucket_straggler_fraction") is not None and position.shape[0] >= 2**24 and not real64.is_float64(position):
            # (worth its 0.1 s only for large runs: see tune_placement)
            position, velocity, acceleration = tune_placement(position, velocity, acceleration)
    aexp_out = 1.0 / (np.array(z_out, dtype=np.float64) + 1)
    aexp_out.sort()
    t_out = tables[1](np.log(aexp_out))
    logging.info(f"{aexp_out=}")
    if "i_snap" not in param:
        param["i_snap"] = 1
    else:
        param["i_snap"] += 1
    # (second condition: the last snapshot's time is hit exactly, but exp(lna(t(ln a)))) may come back one ulp
    # below a; the reference then indexes t_out out of bounds, main.py:135)
    while param["aexp"] < aexp_out[-1] and param["i_snap"] <= len(t_out):
        param["nsteps"] += 1
        position, velocity, acceleration, potential, additional_field = integration.integrate(
            position, velocity, acceleration, potential, additional_field, tables, param,
            t_out[param["i_snap"] - 1])
        position, velocity, acceleration = reorder_policy(position, velocity, acceleration, param)
        if param["write_snapshot"]:
            if on_snapshot is not None:
                on_snapshot(position, velocity, param)
            param["i_snap"] += 1
        if on_step is not None:
            on_step(param)
        logging.info(f"{param['nsteps']=} {param['aexp']=} z = {1.0 / param['aexp'] - 1}")
    return position, velocity, acceleration, potential, additional_field


def run(param) -> None:
    """Drop-in for the reference's main.run (main.py:30-156): logging level, output directories, background tables
    (pysco_amd.cosmotable, astropy-free), initial conditions (pysco_amd.initial_conditions: snapshot restart or LPT),
    then run_nbody with snapshots written by pysco_amd.iostream.  `param`: dict or pandas Series (the reference's
    container); it is updated in place like the reference's Series.  The particle state lives on the GPU for the
    whole run: no CPU fallback."""
    import ast
    import os

    from . import cosmotable, initial_conditions, iostream

    levels = {0: logging.ERROR, 1: logging.WARNING, 2: logging.INFO}
    if param["verbose"] not in levels:
        raise ValueError(f"{param['verbose']=}, should be 0, 1 or 2")
    logging.basicConfig(level=levels[param["verbose"]], format="%(message)s", force=True)
    if not (isinstance(param, dict) or type(param).__name__ == "Series"):
        raise ValueError(f"{type(param)=}, should be a dictionnary or a Pandas Series")
    target, param = param, dict(param.items())
    param["write_snapshot"] = False
    if not param["nthreads"] > 0:
        param["nthreads"] = os.cpu_count() or 1  # host threads are irrelevant here; the key keeps its meaning
    logging.warning("FFT module: fused LDS FFT / rocFFT (MI355X)")

    extra = param["theory"].casefold()
    if extra == "fr":
        extra += f"{param['fR_logfR0']}_n{param['fR_n']}"
    elif extra == "mond":
        mond_function = param["mond_function"].casefold()
        extra += f"_g0_{param['mond_g0']}_exponent_{param['mond_scale_factor_exponent']}_{mond_function}"
        if mond_function != "simple":
            extra += f"_{param['mond_alpha']}"
    elif extra == "parametrized":
        extra += f"_mu0_{param['parametrized_mu0']}"
    extra += f"_{param['linear_newton_solver']}_ncoarse{param['ncoarse']}"
    param["extra"] = extra
    z_out = ast.literal_eval(param["z_out"]) if isinstance(param["z_out"], str) else list(param["z_out"])
    os.makedirs(f"{param['base']}/power", exist_ok=True)
    for i in range(len(z_out) + 1):
        os.makedirs(f"{param['base']}/output_{i:05d}", exist_ok=True)

    logging.warning("----- Compute background cosmology -----")
    tables = cosmotable.generate(param)
    param["aexp"] = 1.0 / (1 + param["z_start"])  # (aexp and t are overwritten when a snapshot is read)
    utils.set_units(param)
    if "nsteps" not in param:
        param["nsteps"] = 0
    logging.warning("----- Initial conditions -----")
    position, velocity = initial_conditions.generate(param, tables)
    utils.set_units(param)
    param["t"] = float(tables[1](np.log(param["aexp"])))
    logging.warning(f"{param['aexp']=} {param['t']=}")
    logging.warning("----- Run N-body -----")
    torch = L.require_gpu()
    position = L.to_device(position, torch.float32)
    velocity = L.to_device(velocity, torch.float32)
    if param.get("particle_id") is True:  # diagnostics: int32 tags 0 .. npart-1 in the order of the initial conditions
        param["particle_id"] = torch.arange(position.shape[0], dtype=torch.int32, device=position.device)
    try:
        return_state = run_nbody(position, velocity, param, tables, z_out, on_snapshot=iostream.write_snapshot_particles)
    finally:
        for k, v in param.items():  # the caller's container sees what the reference's Series would hold
            if not k.startswith("_") and not type(v).__module__.startswith("torch"):
                target[k] = v
    return return_state


def main() -> None:
    """python -m pysco_amd.main -c param.ini   (main.py:159-176)"""
    import argparse

    from . import iostream

    parser = argparse.ArgumentParser(description="Run a PM N-body simulation on an MI355X (PySCo-compatible parameter file).")
    parser.add_argument("-c", "--config_file", help="Configuration file", required=True)
    args = parser.parse_args()
    print(f"pysco_amd (drop-in for PySCo {__version__})\nrun parameters {args.config_file}")
    run(iostream.read_param_file(args.config_file))
    print("Run Completed!")


if __name__ == "__main__":
    main()
