#!/usr/bin/env python3
"""The reference's headless preset loop (presets/main_3wrobot.py:419-446: the same calls in the same order - sim_step, get_sim_step_data,
ctrl_selector, receive_action, receive_sys_state, upd_accum_obj, the unpacking of state_full, stage_obj, accum_obj) through the drop-in
classes at B = 1 - the 3-wheel robot, MPC, Nactor = 5, simulation steps of dt / 2 (what the reference's solver takes: max_step = dt / 2),
opt_iters = 30: simulation steps per second, with and without the fused loop step, and where the host time goes (cProfile, by own
time).  GPU box only.   python tools/b1_profile.py"""
import cProfile, pstats, sys, time
sys.path.insert(0, '.')
import numpy as np
from rcognita_amd import controllers
from tests.test_hip_ref_traces import make_loop_objects


def loop(t1, fuse=True, ahead=True):
    """The calls of the reference's loop body, in its order (sim_step -> get_sim_step_data -> ctrl_selector -> receive_action ->
    receive_sys_state -> upd_accum_obj -> the state components -> stage_obj -> accum_obj), written out here - not the reference's text."""
    plant, ctrl, sim = make_loop_objects("3wrobot", "MPC", 5, t1)
    sim.fuse = fuse
    ctrl.speculate = ahead
    held = np.zeros(2)
    n = 0
    while True:
        sim.sim_step()
        t, _, obs, full = sim.get_sim_step_data()
        u = controllers.ctrl_selector(t, obs, held, None, ctrl, "MPC")
        plant.receive_action(u)
        ctrl.receive_sys_state(plant._state)
        ctrl.upd_accum_obj(obs, u)
        px, py, heading, speed, turn = (full[i] for i in range(5))
        rho = ctrl.stage_obj(obs, u)
        total = ctrl.accum_obj_val
        n += 1
        if t >= t1 - 1e-12:
            return n, (px, py, heading, speed, turn, rho, total)


loop(0.1)
for fuse, ahead in ((True, True), (True, False), (False, False)):
    t0 = time.perf_counter(); n, last = loop(5.0, fuse, ahead); dt = time.perf_counter() - t0
    print(f"fused loop step {fuse}, next step started ahead {ahead}: {n / dt:.0f} sim steps/s ({dt / n * 1e6:.1f} us per step), {n} steps, accum_obj {last[-1]:.4f}")
pr = cProfile.Profile(); pr.enable(); loop(1.0); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
