#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY (an experiment ON the oracle, kept under oracle/ because it imports it; nothing in the product path,
tools/ or bench.py uses it).

How fine does the line-search ladder of the on-device actor optimiser (rcg_actor_optimize, oracle twin
rcg_oracle.actor_optimize_single) have to be?  CPU experiment on the reference's own F8 states (cost SLSQP reaches):
ladders over the same range (4 box widths .. 2^-28) with 64 / 32 / 16 step lengths, 5 and 10 iterations.

    python oracle/experiments/ladder_experiment.py

Result (2026-10, recorded in DESIGN.md 6): the three ladders reach the same cost to five digits on all three systems
(median J / J_slsqp: 3wrobot 1.00013 / 1.00002, 3wrobotNI 1.00032 / 1.00012, 2tank 1.00000 after 5 / 10 iterations),
so the kernel searches 16 step lengths per env and runs four envs per wave pass.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import rcg_oracle as O  # noqa: E402
from tests.conftest import load_golden  # noqa: E402
from tests.helpers import oracle_cfg  # noqa: E402


def optimise(cfg, obs, xs, u_init, iters, nalpha, log2_ratio):
    lo, hi = cfg.ctrl_bnds[:, 0], cfg.ctrl_bnds[:, 1]
    w = hi - lo
    u = np.array(u_init, dtype=np.float64).reshape(cfg.n_actor, cfg.du)
    J = float(O.actor_cost(u, obs, xs, cfg))
    for _ in range(iters):
        _, g = O.actor_grad(u, obs, xs, cfg)
        d = g * w * w
        gn = float(np.max(np.abs(d) / w))
        if not (gn > 0) or not np.isfinite(gn):
            break
        alphas = (1.0 / gn) * np.exp2(2.0 - log2_ratio * np.arange(nalpha))
        cand = np.minimum(np.maximum(u[None] - alphas[:, None, None] * d[None], lo), hi)
        bj, bi = O.argmin_first(O.actor_cost(cand, obs, xs, cfg)[None])
        if not (bj[0] < J):
            break
        u, J = cand[int(bi[0])], float(bj[0])
    return J


if __name__ == "__main__":
    for name in ("3wrobot", "3wrobotNI", "2tank"):
        meta, z = load_golden(f"F8_slsqp_actor_{name}")
        x = z["state"]
        cfg = oracle_cfg(name, n_actor=meta["N"], gamma=meta["gamma"], pred_step_size=meta["pred_step_size"])
        u0 = O.action_sqn_init(cfg, [0.5] if name == "2tank" else None)
        for iters in (5, 10):
            for nalpha, r in ((64, 0.5), (32, 1.0), (16, 2.0)):
                J = np.array([optimise(cfg, x[b], x[b], u0, iters, nalpha, r) for b in range(x.shape[0])])
                q = J / z["J_opt"]
                print(f"{name:10s} iters {iters:2d}  {nalpha:2d} step lengths, ratio 2^-{r}:  J/J_slsqp median {np.median(q):.5f} "
                      f"max {np.max(q):.5f}")
