#!/usr/bin/env python3
"""Fixture F4b: the reference's ``CtrlOptPred._actor_cost`` (controllers.py:1273-1328) on the PRODUCTION kernel's shape.

TEST INFRASTRUCTURE ONLY; runs in the build container (imports /root/reference through oracle/gen_fixtures.py).
F4 has ``state_sys != obs`` and one sequence per call, so on the GPU it reaches the generic kernel only.  Here, per
case, 2 envs x 64 candidate sequences with ``state_sys == obs`` (the control tick without ref_lag) - what k_actor_dma
serves: f32, K a multiple of 64, rows of N*du <= 40 reals, MPC (gamma = 1: the per-component instance; gamma = 0.95:
the discounted one) and RQL (critic instances, all four structures).  States and sequences are float32-exact values
stored as float32 (the reference evaluates them in float64), so the f32 kernel reads bit-identical inputs.

    python oracle/gen_f4b_fixture.py        -> tests/golden/F4b_actor_cost_dma_<system>.npz
"""
import numpy as np

import gen_fixtures as G


def main():
    systems, simulator, controllers = G.import_reference()
    rng_main = np.random.default_rng(20261004)
    for name in G.PRESETS:
        p = G.PRESETS[name]
        sys_obj = G.make_sys(systems, name)
        out, meta = {}, dict(system=name, cases=[])
        n_env, K = 2, 64
        cases = [(N, "MPC", "quad-nomix", g) for N in (3, 5, 10, 16) for g in (1.0, 0.95)]
        cases += [(N, "RQL", cs, 0.95) for N in (5, 16) for cs in ("quad-lin", "quadratic", "quad-nomix", "quad-mix")]
        if p["du"] == 1:
            cases += [(32, "MPC", "quad-nomix", 1.0)]  # (round-2 fixture: the longest row at the time)
        # rows up to 160 bytes = 40 floats: the robots' Nactor = 20 (a generator of their own, so that the cases above
        # keep the draws the committed fixture was made with)
        Nmax = 40 // p["du"]
        n_old = len(cases)
        cases += [(Nmax, "MPC", "quad-nomix", 1.0), (Nmax, "MPC", "quad-nomix", 0.95), (Nmax - 1, "RQL", "quad-mix", 0.95),
                  (Nmax, "SQL", "quad-nomix", 1.0)]
        rng_long = np.random.default_rng([20261005, p["ds"], p["du"]])
        for ci, (N, mode, cs, gamma) in enumerate(cases):
            rng = rng_main if ci < n_old else rng_long
            x = G.rand_states(rng, name, n_env).astype(np.float32)
            aseq = G.rand_actions(rng, name, (n_env, K, N)).astype(np.float32)
            c = G.make_ctrl(controllers, sys_obj, name, mode=mode, Nactor=N, gamma=gamma, critic_struct=cs)
            w = rng.uniform(0, 2, (n_env, c.dim_critic)).astype(np.float32)
            J = np.zeros((n_env, K))
            for i in range(n_env):
                c.state_sys = x[i].astype(np.float64)
                c.w_critic = w[i].astype(np.float64)
                for k in range(K):
                    J[i, k] = c._actor_cost(aseq[i, k].astype(np.float64).reshape(-1), x[i].astype(np.float64))
            tag = f"N{N}_{mode}_{cs}_g{gamma}"
            meta["cases"].append(dict(tag=tag, N=N, mode=mode, critic_struct=cs, gamma=gamma,
                                      pred_step_size=p["dt"] * p["mult"]))
            out.update({f"{tag}__state": x, f"{tag}__action_sqn": aseq, f"{tag}__w": w, f"{tag}__J": J})
        G.save(f"F4b_actor_cost_dma_{name}", meta, **out)


if __name__ == "__main__":
    main()
