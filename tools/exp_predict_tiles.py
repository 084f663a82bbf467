#!/usr/bin/env python3
"""How well can the tile a particle will be in after the NEXT drift be predicted one step ahead?  (development tool)
Fractions per step: particles that change tile in the step (what an exact tile sort sees as stragglers one step later),
and particles whose predicted tile is wrong, for predictions made at the end of the previous step's drift:
  v   : x + dt_prev * v_half                       (velocity only)
  va  : x + dt_prev * (v_half - dt_prev * a_old)   (plus the acceleration the kick used)
    python tools/exp_predict_tiles.py --ncells 512 --steps 30"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ncells", type=int, default=512)
    ap.add_argument("--steps", type=int, default=30)
    args = ap.parse_args()
    import torch

    from bench import make_param, zeldovich_device
    from pysco_amd import integration, solver, utils

    n = args.ncells
    pos, vel = zeldovich_device(n)
    utils.periodic_wrap(pos)
    pos, vel = utils.reorder_particles(pos, vel)
    param, tables = make_param(n, n**3)
    acc, phi, extra = solver.pm(pos, param, tables=tables)

    def tile(x):
        c = (x * n).to(torch.int32).clamp_(0, n - 1)
        return ((c[:, 0] >> 3) * (n >> 4) + (c[:, 1] >> 4)) * (n >> 4) + (c[:, 2] >> 4)

    pred_v = pred_va = pred_vb = pred_vs = None
    amax_hist = []

    def s_guess_next(h, dtp):
        # ratio of max|a| over the last step, rescaled to the length of the next one
        a1, a0_, dt_ = h[-1]
        return (a1 / a0_) ** (float(dtp) / dt_) if a0_ > 0 and dt_ > 0 else 1.0

    t_prev = param["t"]
    dt_last = None
    for s in range(args.steps):
        x0, v0, a0 = pos.clone(), vel.clone(), acc.clone()
        pos, vel, acc, phi, extra = integration.integrate(pos, vel, acc, phi, extra, tables, param, np.float32(3e38))
        dt = float(param["t"] - t_prev)
        t_prev = param["t"]
        tl = tile(pos)
        cross = float((tl != tile(x0)).float().mean())
        msg = f"step {s + 1:3d} dt {dt:.4e} crossers {cross:.4f}"
        if pred_v is not None:
            msg += (f"  wrong: v {float((pred_v != tl).float().mean()):.4f}  va {float((pred_va != tl).float().mean()):.4f}"
                    f"  va(new acc) {float((pred_vb != tl).float().mean()):.4f}")
        # predictions for the NEXT step, from what the fused kick + drift of this step knows: new x, v_half, old a, dt
        v_half = v0 - np.float32(0.5 * dt) * a0
        ratio = min(1.5, max(0.5, dt / dt_last)) if dt_last else 1.0
        dt_last = dt
        dtp = np.float32(dt * ratio)  # guess of the next dt
        d = (dtp * v_half).clamp_(-1.0 / n, 1.0 / n)
        pred_v = tile((pos + d) % 1.0)
        d = (dtp * (v_half - np.float32(0.5 * (dt + float(dtp))) * a0)).clamp_(-1.0 / n, 1.0 / n)
        pred_va = tile((pos + d) % 1.0)
        d = (dtp * (v_half - np.float32(0.5 * (dt + float(dtp))) * acc)).clamp_(-1.0 / n, 1.0 / n)
        pred_vb = tile((pos + d) % 1.0)
        # (vs) the old acceleration scaled by the trend of max|a|: a_new ~ s a_old with one s for all particles
        amax = float(acc.abs().max())
        s_opt = float((acc * a0).sum() / (a0 * a0).sum())
        resid = float(((acc - s_opt * a0) ** 2).sum().sqrt() / (acc ** 2).sum().sqrt())
        if amax_hist:
            r_prev = amax_hist[-1][0] / amax_hist[-1][1] if amax_hist[-1][1] > 0 else 1.0
            s_guess = r_prev ** (dt / amax_hist[-1][2]) if amax_hist[-1][2] > 0 else 1.0
        else:
            s_guess = 1.0
        msg += f"  [a_new ~ s a_old: best s {s_opt:.4f} (residual {resid:.3f}), guessed {s_guess:.4f}, max ratio {amax / float(a0.abs().max()):.4f}]"
        if pred_vs is not None:
            msg += f"  vs {float((pred_vs != tl).float().mean()):.4f}"
        amax_hist.append((amax, float(a0.abs().max()), dt))
        d = (dtp * (v_half - np.float32(0.5 * (dt + float(dtp)) * s_guess_next(amax_hist, dtp)) * a0)).clamp_(-1.0 / n, 1.0 / n)
        pred_vs = tile((pos + d) % 1.0)
        # the same with the NEW acceleration (what the gather could do if it knew the counts beforehand)
        print(msg, flush=True)
        del x0, v0, a0, v_half, d


if __name__ == "__main__":
    main()
