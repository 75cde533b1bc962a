#!/usr/bin/env python3
"""Fixture F7_long: closed-loop traces of the REFERENCE's own loop over whole seconds, the SURVEY.md section 6 quality
datapoint as data (3wrobot, MPC, Nactor = 5, 3 s: accum_obj 389.0 under the reference's SLSQP).

TEST INFRASTRUCTURE ONLY; runs in the build container (imports /root/reference through oracle/gen_fixtures.py, as
F7 does).  Loop body = presets/main_3wrobot.py:419-446: sim_step -> ctrl_selector -> receive_action ->
receive_sys_state(my_sys._state) -> upd_accum_obj.  Stored per sim step: t, state, action, stage_obj, accum_obj - the
columns of F7; data only.

    python oracle/gen_f7_long_fixture.py     -> tests/golden/F7_long_<system>_<mode>.npz   (about 1 minute)
"""
import time

import numpy as np

import gen_fixtures as G


def main():
    systems, simulator, controllers = G.import_reference()
    for name, mode, t1, Nactor in (("3wrobot", "MPC", 3.0, 5), ("3wrobotNI", "MPC", 3.0, 3), ("2tank", "MPC", 20.0, 10)):
        p = G.PRESETS[name]
        sys_obj = G.make_sys(systems, name)
        x0 = np.asarray(p["x0"], dtype=float)
        ctrl = G.make_ctrl(controllers, sys_obj, name, mode=mode, Nactor=Nactor, state_sys=x0.copy())
        sim = simulator.Simulator(sys_type="diff_eqn", closed_loop_rhs=sys_obj.closed_loop_rhs, sys_out=sys_obj.out,
                                  state_init=x0.copy(), disturb_init=[], action_init=np.zeros(p["du"]), t0=0, t1=t1,
                                  dt=p["dt"], max_step=p["dt"] / 2, first_step=1e-6, atol=1e-5, rtol=1e-3,
                                  is_disturb=0, is_dyn_ctrl=0)
        rows = []
        w0 = time.perf_counter()
        while True:
            sim.sim_step()
            t, state, obs, full = sim.get_sim_step_data()
            action = controllers.ctrl_selector(t, obs, np.zeros(p["du"]), None, ctrl, mode)
            sys_obj.receive_action(action)
            ctrl.receive_sys_state(sys_obj._state)
            ctrl.upd_accum_obj(obs, action)
            rows.append(np.concatenate([[t], np.array(full, dtype=float), np.array(action, dtype=float),
                                        [ctrl.stage_obj(obs, action), ctrl.accum_obj_val]]))
            if t >= t1:
                break
        rows = np.stack(rows)
        print(f"{name} {mode}: {len(rows)} sim steps to t = {rows[-1, 0]:.4f}, accum_obj {rows[-1, -1]:.4f}, "
              f"{time.perf_counter() - w0:.1f} s")
        G.save(f"F7_long_{name}_{mode}", dict(system=name, mode=mode, t1=t1, Nactor=Nactor, dt=p["dt"],
                                             columns="t,state...,action...,stage_obj,accum_obj"), rows=rows)


if __name__ == "__main__":
    main()
