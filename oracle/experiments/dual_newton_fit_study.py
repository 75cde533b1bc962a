#!/usr/bin/env python3
"""Study (not product, not oracle): the critic fit's problem

    min 1/2 |A w - b|^2 + mu/2 |w - w0|^2,  lo <= w <= hi        (oracle/rcg_oracle.py::critic_fit_single)

solved by a damped semismooth Newton method on its m-dimensional dual instead of the primal active-set walk that changes one
variable per iteration.  With nu = -lambda / mu:  w(nu) = clip(w0 + A^T nu),  F(nu) = A w(nu) - b + mu nu = 0 at the optimum,
generalised Jacobian A_F A_F^T + mu I over the coordinates that are not clipped - a full step is the walk's own m x m solve, but
every coordinate may change sides at once.  The dual function is strongly concave, so a step length that increases it exists
and the method cannot cycle.  Question: iterations and agreement with the walk on the stacks of the fixtures and on random
rank-deficient stacks.   python oracle/experiments/dual_newton_fit_study.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import rcg_oracle as O


def dual_fit(A, b, w0, lo, hi, nu0=None, max_it=60, stats=None):
    m, dc = A.shape
    mu = max(O.FIT_MU_REL * float(np.sum(A * A)) / m, 1e-30)

    def w_of(nu):
        return np.clip(w0 + A.T @ nu, lo, hi)

    def g(nu, w):  # dual function in nu (to be maximised), up to the factor mu: g = -mu/2 |nu|^2 ... written via the Lagrangian
        lam = -mu * nu
        c = A.T @ lam
        return -0.5 * lam @ lam - lam @ b + np.sum(0.5 * mu * (w - w0) ** 2 + c * w)

    nu = np.zeros(m) if nu0 is None else nu0.copy()
    w = w_of(nu)
    it = 0
    for it in range(1, max_it + 1):
        z = w0 + A.T @ nu
        free = (z > lo) & (z < hi)
        F = A @ w - b + mu * nu
        AF = A[:, free]
        J = AF @ AF.T + mu * np.eye(m)
        d = -np.linalg.solve(J, F)
        # damped step: largest t in {1, 1/2, ...} that increases g (concave: the full step is taken whenever the active set is right)
        g0 = g(nu, w)
        t = 1.0
        while True:
            nu_t = nu + t * d
            w_t = w_of(nu_t)
            if g(nu_t, w_t) >= g0 or t < 1e-12:
                break
            t *= 0.5
        moved = np.max(np.abs(w_t - w)) if dc else 0.0
        nu, w = nu_t, w_t
        z = w0 + A.T @ nu
        if np.array_equal((z > lo) & (z < hi), free) and t == 1.0:
            break
    if stats is not None:
        stats.append(it)
    return w


def compare(name, stacks, lo, hi, w0):
    it_w, it_d, worst, worst_obj = [], [], 0.0, 0.0
    for A, b in stacks:
        sw, sd = [], []
        ww = O.critic_fit_single(A, b, w0, lo, hi, stats=sw)
        wd = dual_fit(A, b, w0, lo, hi, stats=sd)
        mu = max(O.FIT_MU_REL * float(np.sum(A * A)) / A.shape[0], 1e-30)
        P = lambda v: 0.5 * np.sum((A @ v - b) ** 2) + 0.5 * mu * np.sum((v - w0) ** 2)
        it_w += sw; it_d += sd
        worst = max(worst, float(np.max(np.abs(ww - wd) / np.maximum(np.abs(ww), 1.0))))
        worst_obj = max(worst_obj, (P(wd) - P(ww)) / max(P(np.clip(w0, lo, hi)), 1e-300))
    print(f"{name:34s} stacks {len(stacks):4d}  walk iters mean {np.mean(it_w):6.1f} max {np.max(it_w):4d} | dual Newton mean {np.mean(it_d):5.1f} "
          f"max {np.max(it_d):3d} | worst |dw| rel {worst:.2e}, objective excess {worst_obj:+.2e}")


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    from tests.conftest import load_golden
    from tests.test_critic_traces import CASES, MODES, trace_cfg
    for name, cs in CASES:
        for mode in MODES:
            meta, z = load_golden(f"F7c_trace_{name}_{mode}_{cs}")
            cfg = trace_cfg(meta)
            lo, hi = O.critic_bounds(cfg.critic_struct, cfg.dc)
            stacks = []
            for i in range(len(z["tick_t"])):
                if z["tick_fitted"][i]:
                    A, b = O.critic_td_system(z["tick_w_prev"][i][None], z["tick_obs_buf"][i][None], z["tick_act_buf"][i][None], cfg)
                    stacks.append((A[0], b[0]))
            compare(f"F7c {name} {mode} {cs}", stacks, lo, hi, np.ones(cfg.dc))
    # random stacks with many weights: 3 rows, 35 / 28 / 17 unknowns, both kinds of boxes, rank-deficient on purpose
    for dc, box in ((35, (-1e3, 1e3)), (28, (0.0, 1e3)), (17, (-1e3, 1e3)), (35, (0.0, 1e3))):
        stacks = []
        for _ in range(200):
            m = 3
            A = rng.normal(size=(m, dc)) * rng.uniform(0.1, 30.0, size=(1, dc))
            if rng.uniform() < 0.3:
                A[2] = A[0] * rng.uniform(0.5, 2.0) + 1e-9 * rng.normal(size=dc)
            b = rng.normal(size=m) * rng.uniform(1, 1e4)
            stacks.append((A, b))
        compare(f"random m=3 dc={dc} box={box}", stacks, np.full(dc, box[0]), np.full(dc, box[1]), np.ones(dc))
