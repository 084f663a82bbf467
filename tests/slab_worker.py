"""Worker for tests/test_slab_gloo.py (and the 2-rank GPU rehearsal): one rank of a slab-decomposed
run.  usage: python -m tests.slab_worker <rank> <world> <port> <outdir> <backend: oracle|hip> <n> <steps> [integrator]"""
import os
import sys

import numpy as np


def main():
    rank, world, port, outdir, backend, n, steps = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4],
                                                      sys.argv[5], int(sys.argv[6]), int(sys.argv[7]))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    if os.environ.get("PMSTEP_SLAB_LOG"):
        import logging

        logging.basicConfig(level=logging.DEBUG)

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pysco_amd import slab
    from tests.helpers import synth_tables, zeldovich_particles

    if backend == "oracle":
        from tests.slab_oracle_ops import OracleOps

        ops, dev = OracleOps(n), "cpu"
    else:
        ops, dev = None, "cuda"
        torch.cuda.set_device(0)
    pos, vel = zeldovich_particles(n, seed=11, rms_cells=1.2, vel_factor=0.5)
    tables = synth_tables()
    param = {"theory": "newton", "mass_scheme": "TSC", "npart": n**3, "ncoarse": int(np.log2(n)), "Om_m": 0.25733,
             "Om_lambda": 0.742589237, "H0": 72.0, "boxlen": 100.0, "aexp": 0.5, "aexp_old": 0.49,
             "linear_newton_solver": "fft", "gradient_stencil_order": 5, "save_power_spectrum": "no", "nthreads": 1,
             "Courant_factor": 1.0, "max_aexp_stepping": 10, "integrator": "leapfrog", "write_snapshot": False,
             "nsteps": 0}
    if len(sys.argv) > 8:
        param["integrator"] = sys.argv[8]
    if len(sys.argv) > 10 and sys.argv[10] != "newton":  # "mond" / "fr" / "newton_mg" [+ "_replicated": all-gathered density, replicated grid solve]
        kind = sys.argv[10]
        if kind.endswith("_replicated"):
            kind = kind[: -len("_replicated")]
            slab._SLAB_MULTIGRID = False
        if kind == "mond_fft7pt":
            kind, solver_name = "mond", "fft_7pt"
        else:
            solver_name = "multigrid"
        param.update({"Npre": 2, "Npost": 1, "epsrel": 1e-2, "linear_newton_solver": solver_name})
        if kind == "mond":
            param.update({"theory": "mond", "mond_function": "simple", "mond_g0": 1.2, "mond_alpha": 1,
                          "mond_scale_factor_exponent": 0})
        elif kind == "fr":
            param.update({"theory": "fr", "fR_n": 1, "fR_logfR0": 5, "boxlen": 1000.0})
    if len(sys.argv) > 9 and sys.argv[9] == "asym":  # one-way flow across x = 0.5: rank 0 only loses, rank 1 only gains
        vel[:] = 0
        vel[(pos[:, 0] > 0.25) & (pos[:, 0] < 0.5), 0] = 1.0
    param["t"] = float(tables[1](np.log(param["aexp"])))
    from pysco_amd import utils as amd_utils

    amd_utils.set_units(param)
    s = slab.get(n, ops)
    mine = (np.floor(pos[:, 0] * world).astype(int).clip(0, world - 1) == rank)
    p = torch.from_numpy(pos[mine].copy()).to(dev)
    v = torch.from_numpy(vel[mine].copy()).to(dev)
    ids = torch.from_numpy(np.nonzero(mine)[0].astype(np.int64)).to(dev)
    if len(sys.argv) > 11 and sys.argv[11] == "sorted" and backend == "hip":  # a Morton-sorted stream, as bench.py starts from
        order = []
        p, v = amd_utils.reorder_particles(p, v, _order_out=order)
        ids = ids[order[0]].contiguous()
    s.carry = (ids,)  # particle IDs migrate too
    acc, phi, extra = s.pm(p, param, tables=tables)
    out = {"acc0": acc.cpu().numpy(), "phi0": phi.cpu().numpy(), "id0": s.carry[0].cpu().numpy()}
    rebucketed, fracs = 0, []
    for _ in range(steps):
        p, v, acc, phi, extra = s.integrate(p, v, acc, phi, extra, tables, param, np.float32(1))
        rebucketed += int(bool(getattr(s, "_rebucketed", False)))
        if backend == "hip":  # stragglers the gather of this step found in the rank's stream (after the migration)
            fracs.append(amd_utils.straggler_fraction(n)[0] or 0.0)
    out["rebucketed"], out["stragglers"] = np.int64(rebucketed), np.asarray(fracs, dtype=np.float64)
    out["id"] = s.carry[0].cpu().numpy()
    out["khalo"] = np.int64(getattr(s.ops, "_khalo_buf", None) is not None)  # the fused kick + deposit used a halo buffer
    out.update(pos=p.cpu().numpy(), vel=v.cpu().numpy(), acc=acc.cpu().numpy(), phi=phi.cpu().numpy(),
               t=param["t"], aexp=param["aexp"])
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
