#!/usr/bin/env python3
"""The reference's headless preset loop (presets/main_3wrobot.py:419-446, the loop body verbatim: sim_step, get_sim_step_data,
ctrl_selector, receive_action, receive_sys_state, upd_accum_obj, the unpacking of state_full, stage_obj, accum_obj) through the drop-in
classes at B = 1 - the 3-wheel robot, MPC, Nactor = 5, simulation steps of dt / 2 (what the reference's solver takes: max_step = dt / 2),
opt_iters = 30: simulation steps per second, with and without the fused loop step, and where the host time goes (cProfile, by own
time).  GPU box only.   python tools/b1_profile.py"""
import cProfile, pstats, sys, time
sys.path.insert(0, '.')
import numpy as np
from rcognita_amd import controllers
from tests.test_hip_ref_traces import make_loop_objects


def loop(t1, fuse=True):
    my_sys, my_ctrl_benchm, my_simulator = make_loop_objects("3wrobot", "MPC", 5, t1)
    my_simulator.fuse = fuse
    action_manual, my_ctrl_nominal, ctrl_mode = np.zeros(2), None, "MPC"
    n = 0
    while True:
        my_simulator.sim_step()
        t, state, observation, state_full = my_simulator.get_sim_step_data()
        action = controllers.ctrl_selector(t, observation, action_manual, my_ctrl_nominal, my_ctrl_benchm, ctrl_mode)
        my_sys.receive_action(action)
        my_ctrl_benchm.receive_sys_state(my_sys._state)
        my_ctrl_benchm.upd_accum_obj(observation, action)
        xCoord = state_full[0]
        yCoord = state_full[1]
        alpha = state_full[2]
        v = state_full[3]
        omega = state_full[4]
        stage_obj = my_ctrl_benchm.stage_obj(observation, action)
        accum_obj = my_ctrl_benchm.accum_obj_val
        n += 1
        if t >= t1 - 1e-12:
            return n, (xCoord, yCoord, alpha, v, omega, stage_obj, accum_obj)


loop(0.1)
for fuse in (True, False):
    t0 = time.perf_counter(); n, last = loop(5.0, fuse); dt = time.perf_counter() - t0
    print(f"fused loop step {fuse}: {n / dt:.0f} sim steps/s ({dt / n * 1e6:.1f} us per step), {n} steps, accum_obj {last[-1]:.4f}")
pr = cProfile.Profile(); pr.enable(); loop(1.0); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
