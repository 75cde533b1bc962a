#!/usr/bin/env python3
"""Fixture F13: the reference's robots at LARGE HEADING (SURVEY §7 hard part 3: alpha is unbounded; it reached -72 rad in
a 2 s constant-torque run).

TEST INFRASTRUCTURE ONLY; runs in the build container (imports /root/reference through oracle/gen_fixtures.py).
Every other fixture keeps |alpha| < 8 (gen_fixtures.rand_states).  Here the heading spans |alpha| in
{30, 72, 300, 1e3, 1e4}, both signs, and every input is a float32-exact value stored as float32 - the float32 kernels read
bit for bit what the reference evaluated in float64, so the comparison measures the kernels' arithmetic (the f32 trig
behind its range reduction, the f32 rollout) and not the rounding of the inputs.

Per robot (Sys3WRobot, Sys3WRobotNI; the tank has no heading):
  rhs      `_state_dyn` / `closed_loop_rhs` (systems.py:308-323, 370-382, 213-253) on 16 states per magnitude and sign
  cost     `_actor_cost` (controllers.py:1273-1328), state_sys == obs, 2 envs (+A, -A) x 64 sequences per magnitude:
           MPC gamma = 1 (the per-component instance), MPC gamma = 0.95, RQL quad-nomix - the production kernels' shape
           (K = 64 -> k_actor_dma; the test also feeds the first 8 / 16 sequences of every env -> k_actor_dma_packed)
  grid     `_actor_cost` of the build's generated 256-level grid WITH THE LEVELS THE FLOAT32 KERNELS PRODUCE (grid_f32
           below; stored as `grid__levels`) from states AT REST (v = omega = 0; NI: any state) under a held zero action: one env
           step leaves such a state exactly where it is, so a fused tick (k_ticks_pk: env step + decision in one
           launch) decides on exactly the state the reference evaluated
  traj     the reference's Simulator (scipy RK45, simulator.py:71-168) under a constant action from a start whose
           heading is already large: per step (t, y) - replayed by k_sim<float> step by step (each step a map from the
           reference's own previous state) and as a free run

    python oracle/gen_large_heading_fixture.py        -> tests/golden/F13_large_heading_<system>.npz
"""
import numpy as np

import gen_fixtures as G
import rcg_oracle as O

MAGS = (30.0, 72.0, 300.0, 1e3, 1e4)


def headings(rng, n, A):
    """n float32 headings around +-A (alternating sign), spread over a few revolutions so that every quadrant occurs."""
    a = A + rng.uniform(-7.0, 7.0, n)
    a[1::2] *= -1.0
    return a.astype(np.float32)


def states(rng, name, n, A, rest=False):
    x = G.rand_states(rng, name, n).astype(np.float32)
    x[:, 2] = headings(rng, n, A)
    if rest and name == "3wrobot":
        x[:, 3:] = 0.0
    return x


def grid_f32(bnds, K, N):
    """The generated level grid as the FLOAT32 kernels produce it (rcg_kernels.hpp gen_candidate: level i of an input =
    fma((float) i, (hi - lo) / (g - 1), lo) with a correctly rounded float32 quotient; candidate k -> (k // g, k % g)):
    the product and the sum are exact in float64, so one rounding to float32 reproduces the fma."""
    g = int(round(np.sqrt(K)))
    lo, hi = bnds[:, 0].astype(np.float32), bnds[:, 1].astype(np.float32)
    step = ((hi - lo) / np.float32(g - 1)).astype(np.float32)
    i, j = np.divmod(np.arange(K), g)
    first = np.stack([(i * step[0].astype(np.float64) + lo[0]).astype(np.float32),
                      (j * step[1].astype(np.float64) + lo[1]).astype(np.float32)], axis=-1).astype(np.float64)
    return np.broadcast_to(first[:, None, :], (K, N, 2)).copy()


def main():
    systems, simulator, controllers = G.import_reference()
    for name in ("3wrobot", "3wrobotNI"):
        p = G.PRESETS[name]
        sys_obj = G.make_sys(systems, name)
        rng = np.random.default_rng([20261006, p["ds"]])
        out, meta = {}, dict(system=name, mags=list(MAGS), cost_cases=[], grid_K=256, traj=[])
        # ------------------------------------------------------------------ rhs
        xs, us = [], []
        for A in MAGS:
            xs.append(states(rng, name, 32, A))
            us.append(G.rand_actions(rng, name, (32,), overshoot=1.6).astype(np.float32))
        x, u = np.concatenate(xs), np.concatenate(us)
        dyn = np.stack([sys_obj._state_dyn(0.0, x[i].astype(np.float64), u[i].astype(np.float64)) for i in range(len(x))])
        clrhs, clipped = [], []
        for i in range(len(x)):
            sys_obj.receive_action(u[i].astype(np.float64))
            clrhs.append(sys_obj.closed_loop_rhs(0.0, x[i].astype(np.float64)))
            clipped.append(sys_obj.action.copy())
        out.update(rhs__state=x, rhs__action=u, rhs__state_dyn=dyn, rhs__closed_loop_rhs=np.stack(clrhs),
                   rhs__clipped_action=np.stack(clipped))
        # ------------------------------------------------------------------ cost (the production kernels' shape)
        N, K = 10, 64
        for A in MAGS:
            for mode, cs, gamma in (("MPC", "quad-nomix", 1.0), ("MPC", "quad-nomix", 0.95), ("RQL", "quad-nomix", 0.95)):
                xe = states(rng, name, 2, A)
                aseq = G.rand_actions(rng, name, (2, K, N)).astype(np.float32)
                c = G.make_ctrl(controllers, sys_obj, name, mode=mode, Nactor=N, gamma=gamma, critic_struct=cs)
                w = rng.uniform(0, 2, (2, c.dim_critic)).astype(np.float32)
                J = np.zeros((2, K))
                for i in range(2):
                    c.state_sys = xe[i].astype(np.float64)
                    c.w_critic = w[i].astype(np.float64)
                    for k in range(K):
                        J[i, k] = c._actor_cost(aseq[i, k].astype(np.float64).reshape(-1), xe[i].astype(np.float64))
                tag = f"A{A:g}_{mode}_g{gamma}"
                meta["cost_cases"].append(dict(tag=tag, A=A, N=N, mode=mode, critic_struct=cs, gamma=gamma,
                                               pred_step_size=p["dt"] * p["mult"]))
                out.update({f"cost_{tag}__state": xe, f"cost_{tag}__action_sqn": aseq, f"cost_{tag}__w": w,
                            f"cost_{tag}__J": J})
        # ------------------------------------------------------------------ grid (the generated 256-level grid, at rest)
        ocfg = O.OracleCfg(sys_id={"3wrobot": O.SYS_3WROBOT, "3wrobotNI": O.SYS_3WROBOT_NI}[name], n_actor=N,
                           pars=p["pars"], ctrl_bnds=np.array(p["bnds"], dtype=float),
                           R1=np.diag(np.array(p["R1"], dtype=float)))
        grid = grid_f32(np.array(p["bnds"], dtype=float), 256, N)  # [256, N, du], the float32 kernel's own levels
        assert np.max(np.abs(grid - O.grid_candidates(ocfg, 256))) < 2e-5
        xr = np.concatenate([states(rng, name, 4, A, rest=True) for A in MAGS])
        c = G.make_ctrl(controllers, sys_obj, name, mode="MPC", Nactor=N, gamma=1.0)
        Jg = np.zeros((len(xr), 256))
        for i in range(len(xr)):
            c.state_sys = xr[i].astype(np.float64)
            for k in range(256):
                Jg[i, k] = c._actor_cost(grid[k].reshape(-1), xr[i].astype(np.float64))
        out.update(grid__state=xr, grid__J=Jg, grid__levels=grid[:, 0, :].astype(np.float32),
                   grid__pred_step_size=np.array(p["dt"] * p["mult"]))
        # ------------------------------------------------------------------ traj (reference Simulator, constant action)
        const_u = {"3wrobot": [120.0, -35.0], "3wrobotNI": [8.0, -1.5]}[name]
        for A in (72.0, 1e3, 1e4):
            x0 = np.asarray(p["x0"], dtype=np.float32)
            x0[2] = np.float32(-A)
            if name == "3wrobot":
                x0[3:] = (2.0, -3.0)  # already moving and turning
            x0 = x0.astype(np.float64)
            sim = simulator.Simulator(sys_type="diff_eqn", closed_loop_rhs=sys_obj.closed_loop_rhs, sys_out=sys_obj.out,
                                      state_init=x0.copy(), disturb_init=[], action_init=np.zeros(p["du"]), t0=0, t1=2.0,
                                      dt=p["dt"], max_step=p["dt"] / 2, first_step=1e-6, atol=1e-5, rtol=1e-3,
                                      is_disturb=0, is_dyn_ctrl=0)
            sys_obj.receive_action(np.array(const_u))
            ts, ys = [0.0], [x0.copy()]
            while sim.t < 0.5:
                sim.sim_step()
                t, state, obs, full = sim.get_sim_step_data()
                ts.append(float(t))
                ys.append(np.array(full, dtype=float))
            tag = f"A{A:g}"
            meta["traj"].append(dict(tag=tag, A=A, action=const_u, dt=p["dt"]))
            out.update({f"traj_{tag}__t": np.array(ts), f"traj_{tag}__y": np.stack(ys)})
        meta.update(pars=p["pars"], bnds=p["bnds"])
        G.save(f"F13_large_heading_{name}", meta, **out)


if __name__ == "__main__":
    main()
