#!/usr/bin/env python3
"""Study (not product): how many iterations the critic fit's active-set walk takes from different FEASIBLE start points on the stacks
of a closed loop (3-wheel robot, RQL, random start states, optimiser ticks - the scenario of tools/fit_ticks_probe.py, where half
of the weights end up on a bound).  The minimiser is unique, so the start only decides the length of the walk:
  cold      clip(w_init)                                   (the oracle's default)
  warm      the previous tick's weights                    (what the kernels do)
  clip-k    k rounds of "solve on the current free set, clip, the clipped ones become the fixed set" from the warm point, then the walk
python oracle/experiments/fit_warm_start_study.py [ticks] [envs]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
from oracle import rcg_oracle as O
from tests.helpers import oracle_cfg


def clip_rounds(A, b, w0, lo, hi, w, k):
    m = A.shape[0]
    mu = max(O.FIT_MU_REL * float(np.sum(A * A)) / m, 1e-30)
    for _ in range(k):
        free = (w > lo) & (w < hi)
        AF = A[:, free]
        rhs = b - A[:, ~free] @ w[~free] - AF @ w0[free]
        lam = np.linalg.solve(AF @ AF.T + mu * np.eye(m), rhs)
        z = w.copy()
        z[free] = w0[free] + AF.T @ lam
        w = np.clip(z, lo, hi)
    return w


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    rng = np.random.default_rng(7)
    for cs in ("quad-lin", "quadratic"):
        cfg = oracle_cfg("3wrobot", n_actor=10, mode=O.MODE_IDS["RQL"], critic_struct=O.CRITIC_IDS[cs], buffer_size=10, n_critic=4)
        x0 = np.stack([rng.uniform(-10, 10, B), rng.uniform(-10, 10, B), rng.uniform(-8, 8, B), rng.uniform(-3, 3, B), rng.uniform(-3, 3, B)], axis=-1)
        env = O.new_batch(cfg, x0)
        orig = O.critic_fit_single
        prev = {}
        tally = {k: [] for k in ("cold", "warm", "clip-1", "clip-2", "clip-3")}
        call = [0]

        def rec(A, b, w0, lo, hi, stats=None, w_start=None):
            e = call[0] % B
            call[0] += 1
            wp = prev.get(e, np.clip(w0, lo, hi))
            st = []
            w = orig(A, b, w0, lo, hi, stats=st)
            tally["cold"].append(st[0])
            st = []
            w2 = orig(A, b, w0, lo, hi, stats=st, w_start=wp)
            tally["warm"].append(st[0])
            assert np.max(np.abs(w2 - w) / np.maximum(np.abs(w), 1.0)) < 1e-5
            for k in (1, 2, 3):
                st = []
                w3 = orig(A, b, w0, lo, hi, stats=st, w_start=clip_rounds(A, b, w0, lo, hi, wp.copy(), k))
                tally[f"clip-{k}"].append(st[0] + k)
                assert np.max(np.abs(w3 - w) / np.maximum(np.abs(w), 1.0)) < 1e-5
            prev[e] = w
            return w

        O.critic_fit_single = rec
        for t in range(T):
            O.control_tick_opt(cfg, env, 5)
        O.critic_fit_single = orig
        n = len(tally["cold"])
        late = slice(n // 2, n)
        at_bound = np.mean([(np.mean((prev[e] <= O.critic_bounds(cfg.critic_struct, cfg.dc)[0]) | (prev[e] >= O.critic_bounds(cfg.critic_struct, cfg.dc)[1]))) for e in prev])
        print(f"{cs:10s} dc {cfg.dc}: {n} fits, weights on a bound at the end {at_bound:.0%}; iterations (solves) in the second half - mean / max: " +
              ", ".join(f"{k} {np.mean(v[late]):.1f} / {np.max(v[late])}" for k, v in tally.items()), flush=True)


if __name__ == "__main__":
    main()
