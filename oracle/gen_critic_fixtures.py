#!/usr/bin/env python3
"""Round-4 golden vectors: the reference's closed loop in the CRITIC modes, from starts where the critic steers.

    python oracle/gen_critic_fixtures.py     # writes tests/golden/F7c_trace_*.npz, F8c_slsqp_actor_*.npz

Why a second generator.  ``F7_trace_2tank_RQL`` (oracle/gen_fixtures.py) starts at the preset's ``x0 = [2, -2]``, where
the input sits on its lower bound for the whole run: that trace equals the MPC trace to 5e-11 and says nothing about
the critic.  The traces written here start where RQL / SQL decide differently from MPC, and the generator REFUSES to
write a trace whose actions stay within 1e-2 of the MPC run from the same start (``assert_discriminating``), so a
saturated critic-mode trace can never be committed again.

F7c_trace_<system>_<mode>_<critic_struct>  the loop body of presets/main_3wrobot.py:419-446 on the imported reference
    rows      [n_steps, 1 + ds + du + 2]   t, state, action, stage_obj, accum_obj per simulation step (as F7)
    rows_mpc  the same loop in MPC mode from the same start (what the critic modes are told apart from)
    tick_*    one entry per control tick (controllers.py:1440-1444), everything the decision of that tick saw:
              tick_t, tick_obs [dy], tick_state_sys [ds] (the state BEFORE receive_sys_state: App. A-2 lag),
              tick_w [dc] (w_critic the actor used), tick_w_prev [dc] (w_critic_prev the critic fit used),
              tick_obs_buf / tick_act_buf [buffer_size, d] (after the push, as the fit saw them),
              tick_action_sqn [N * du], tick_J (SLSQP's result, recomputed with the reference's own call and
              asserted bit-identical to the action the reference returned), tick_J_init (J at action_sqn_init),
              tick_nfev, tick_Jc (the reference's _critic_cost at the fitted weights), tick_Jc_init.
    mpc_tick_*  the same per-tick record (t, obs, state_sys, action_sqn, J, J_init, nfev) of the MPC run: SLSQP's decisions
              on the states the reference's own MPC loop visits (a tighter anchor for the optimiser than the closed-loop
              bands of the MPC traces).
F8c_slsqp_actor_<system>_<mode>_<critic_struct>  a subsample of those ticks as an optimiser-quality fixture in F8's layout
    (state = state_sys, obs, w, J_opt, action_sqn_opt, J_init, nfev) - the bar for rcg_actor_optimize in RQL / SQL.

Data only: inputs and the outputs the reference computed for them.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.gen_fixtures import PRESETS, import_reference, make_ctrl, make_sys, save  # noqa: E402

# (system, start state, t1, Nactor, critic structure); None = the preset's x0
CASES = [
    ("3wrobotNI", None, 0.3, 3, "quad-nomix"),
    ("3wrobotNI", None, 0.3, 3, "quad-mix"),
    ("3wrobot", None, 0.2, 5, "quad-nomix"),
    ("2tank", [0.2, 0.3], 3.0, 10, "quad-nomix"),
    ("2tank", [0.2, 0.3], 3.0, 10, "quadratic"),  # BASELINE configs[2]'s critic structure
    ("2tank", [0.2, 0.3], 3.0, 10, "quad-lin"),
]
DISCRIMINATION = 1e-2  # min over the run of max |action - MPC action| a critic-mode trace must exceed


def run_loop(systems, simulator, controllers, name, mode, x0, t1, Nactor, critic_struct, capture):
    from scipy.optimize import Bounds, minimize

    p = PRESETS[name]
    sys_obj = make_sys(systems, name)
    x0 = np.asarray(p["x0"] if x0 is None else x0, dtype=float)
    ctrl = make_ctrl(controllers, sys_obj, name, mode=mode, Nactor=Nactor, state_sys=x0.copy(),
                     critic_struct=critic_struct)
    sim = simulator.Simulator(sys_type="diff_eqn", closed_loop_rhs=sys_obj.closed_loop_rhs, sys_out=sys_obj.out,
                              state_init=x0.copy(), disturb_init=[], action_init=np.zeros(p["du"]), t0=0, t1=t1,
                              dt=p["dt"], max_step=p["dt"] / 2, first_step=1e-6, atol=1e-5, rtol=1e-3,
                              is_disturb=0, is_dyn_ctrl=0)
    rows, ticks = [], []
    while True:  # presets/main_3wrobot.py:419-446
        sim.sim_step()
        t, state, obs, full = sim.get_sim_step_data()
        clock_before = ctrl.ctrl_clock
        state_sys_before = np.array(ctrl.state_sys, dtype=float)
        w_prev_before = np.array(ctrl.w_critic_prev, dtype=float)
        action = controllers.ctrl_selector(t, obs, np.zeros(p["du"]), None, ctrl, mode)
        if capture and ctrl.ctrl_clock != clock_before:  # a control tick happened at this step
            # the same call as controllers.py:1393-1398, on the reference's own _actor_cost, before state_sys moves
            init = np.reshape(ctrl.action_sqn_init, [Nactor * p["du"]])
            res = minimize(lambda a: ctrl._actor_cost(a, obs), init, method="SLSQP", tol=1e-7,
                           bounds=Bounds(ctrl.action_sqn_min, ctrl.action_sqn_max, keep_feasible=True),
                           options={"maxiter": 300, "disp": False})
            assert np.array_equal(res.x[: p["du"]], np.asarray(action)), "recomputed SLSQP differs from the reference's"
            tk = dict(t=float(t), obs=np.array(obs, dtype=float), state_sys=state_sys_before,
                      w=np.array(getattr(ctrl, "w_critic", ctrl.w_critic_init), dtype=float), w_prev=w_prev_before,
                      obs_buf=np.array(ctrl.observation_buffer, dtype=float),
                      act_buf=np.array(ctrl.action_buffer, dtype=float), action_sqn=np.array(res.x, dtype=float),
                      J=float(res.fun), J_init=float(ctrl._actor_cost(init, obs)), nfev=int(res.nfev))
            if mode != "MPC":
                # _critic_cost reads self.w_critic_prev, which the fit has already overwritten: put the old one back
                keep = ctrl.w_critic_prev
                ctrl.w_critic_prev = w_prev_before
                tk["Jc"] = float(ctrl._critic_cost(tk["w"]))
                tk["Jc_init"] = float(ctrl._critic_cost(ctrl.w_critic_init))
                ctrl.w_critic_prev = keep
            ticks.append(tk)
        sys_obj.receive_action(action)
        ctrl.receive_sys_state(sys_obj._state)
        ctrl.upd_accum_obj(obs, action)
        rows.append(np.concatenate([[t], np.array(full, dtype=float), np.array(action, dtype=float),
                                    [ctrl.stage_obj(obs, action), ctrl.accum_obj_val]]))
        if t >= t1:
            break
    return np.stack(rows), ticks


def assert_discriminating(rows, rows_mpc, ds, du, what):
    n = min(len(rows), len(rows_mpc))
    gap = float(np.max(np.abs(rows[:n, 1 + ds:1 + ds + du] - rows_mpc[:n, 1 + ds:1 + ds + du])))
    assert gap > DISCRIMINATION, (f"{what}: the critic never steers (max |action - MPC action| = {gap:.3g}); "
                                  "a saturated trace is not a fixture of the critic modes")
    return gap


def main():
    systems, simulator, controllers = import_reference()
    for name, x0, t1, Nactor, cs in CASES:
        p = PRESETS[name]
        ds, du = p["ds"], p["du"]
        rows_mpc, ticks_mpc = run_loop(systems, simulator, controllers, name, "MPC", x0, t1, Nactor, cs, capture=True)
        mpc_arrays = {f"mpc_tick_{k}": np.stack([np.asarray(tk[k]) for tk in ticks_mpc])
                      for k in ("t", "obs", "state_sys", "action_sqn", "J", "J_init", "nfev")}
        for mode in ("RQL", "SQL"):
            rows, ticks = run_loop(systems, simulator, controllers, name, mode, x0, t1, Nactor, cs, capture=True)
            gap = assert_discriminating(rows, rows_mpc, ds, du, f"{name} {mode}")
            w_all = np.stack([tk["w"] for tk in ticks])
            assert np.max(np.abs(w_all - 1.0)) > 1e-3, f"{name} {mode}: the critic weights never left w_init"
            meta = dict(system=name, mode=mode, t1=t1, Nactor=Nactor, dt=p["dt"], critic_struct=cs, gamma=1.0,
                        Ncritic=4, buffer_size=10, x0=[float(v) for v in (p["x0"] if x0 is None else x0)],
                        pred_step_size=p["dt"] * p["mult"], max_action_gap_to_mpc=gap,
                        accum_obj=float(rows[-1, -1]), accum_obj_mpc=float(rows_mpc[-1, -1]),
                        columns="t,state...,action...,stage_obj,accum_obj")
            tick_arrays = {f"tick_{k}": np.stack([np.asarray(tk[k]) for tk in ticks]) for k in ticks[0]}
            save(f"F7c_trace_{name}_{mode}_{cs}", meta, rows=rows, rows_mpc=rows_mpc, **tick_arrays, **mpc_arrays)
            # optimiser-quality subsample: every tick of the short runs, every 3rd of the long one, at most 32
            step = max(1, len(ticks) // 32)
            sel = ticks[::step][:32]
            save(f"F8c_slsqp_actor_{name}_{mode}_{cs}",
                 dict(system=name, mode=mode, N=Nactor, gamma=1.0, critic_struct=cs, pred_step_size=p["dt"] * p["mult"],
                      note="ticks of F7c_trace: SLSQP from action_sqn_init on the reference's _actor_cost"),
                 state=np.stack([tk["state_sys"] for tk in sel]), obs=np.stack([tk["obs"] for tk in sel]),
                 w=np.stack([tk["w"] for tk in sel]), J_opt=np.array([tk["J"] for tk in sel]),
                 action_sqn_opt=np.stack([tk["action_sqn"] for tk in sel]),
                 J_init=np.array([tk["J_init"] for tk in sel]), nfev=np.array([tk["nfev"] for tk in sel]))
            print(f"  {name} {mode} {cs}: {len(rows)} sim steps, {len(ticks)} ticks, max action gap to MPC {gap:.3g}, "
                  f"accum {rows[-1, -1]:.4f} (MPC {rows_mpc[-1, -1]:.4f})")


if __name__ == "__main__":
    main()
