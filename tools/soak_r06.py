#!/usr/bin/env python3
"""Round-6 soak (GPU box): the paths added this round, run long - counters exact, nothing non-finite that was not frozen and
counted, no hang in the pinned-memory polling of rcg_loop_step.   python tools/soak_r06.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (PyTorch's HIP runtime first)

from rcognita_amd import Engine, _native as N  # noqa: E402
from rcognita_amd.pool import preset_engine_config  # noqa: E402
from tests.test_hip_ref_traces import make_loop_objects  # noqa: E402
from rcognita_amd import controllers  # noqa: E402

rng = np.random.default_rng(6)


def line(what, **kw):
    print(what + ": " + ", ".join(f"{k} = {v}" for k, v in kw.items()), flush=True)


# 1. the drop-in loop at B = 1 with the fused step: 100 000 iterations (50 000 decisions), the host polls pinned memory every call
my_sys, my_ctrl, my_sim = make_loop_objects("3wrobotNI", "MPC", 3, 1e9, opt_iters=10)
t0 = time.perf_counter()
n = 100000
for k in range(n):
    my_sim.sim_step()
    t, state, obs, full = my_sim.get_sim_step_data()
    a = controllers.ctrl_selector(t, obs, np.zeros(2), None, my_ctrl, "MPC")
    my_sys.receive_action(a)
    my_ctrl.receive_sys_state(my_sys._state)
    my_ctrl.upd_accum_obj(obs, a)
dt = time.perf_counter() - t0
line("B = 1 fused loop (3wrobotNI, MPC, Nactor 3)", steps=n, fused_steps=my_ctrl.fused_steps, fused_decisions=my_ctrl.fused_decisions,
     started_ahead=my_ctrl.spec_hits, dropped=my_ctrl.spec_drops,
     steps_per_s=round(n / dt), final_state=np.round(np.asarray(full, dtype=float), 6).tolist(), accum=float(my_ctrl.accum_obj_val),
     finite=bool(np.all(np.isfinite(full))))
assert my_ctrl.fused_steps == n and my_ctrl.fused_decisions == n // 2 and np.all(np.isfinite(full))
assert my_ctrl.spec_hits >= n - 3 and my_ctrl.spec_drops <= 1

# 2. the headline shape in float64, 20 000 ticks of the streamed tick
B, K, Nh = 65536, 256, 10
eng = Engine(preset_engine_config("3wrobot", B, Nactor=Nh, dtype="f64"))
eng.set_state(np.stack([rng.uniform(-10, 10, B), rng.uniform(-10, 10, B), rng.uniform(-np.pi, np.pi, B), rng.uniform(-1, 1, B),
                        rng.uniform(-1, 1, B)], axis=-1))
lo, hi = np.array([-300.0, -100.0]), np.array([300.0, 100.0])
cand = eng.to_device((lo + (hi - lo) * rng.random((B, K, Nh, 2))))
t0 = time.perf_counter()
T = 20000
eng.control_tick(cand, K=K, T=T)
eng.synchronize()
dt = time.perf_counter() - t0
s, _ = eng.episode_stats(from_accum=True)
line("C2 f64 streamed", ticks=T, ms_per_tick=round(dt / T * 1e3, 4), n_failed=s["n_failed"],
     step_idx_exact=bool(np.all(eng.get_field(N.FIELD_STEP_IDX) == T)), max_abs_heading=float(np.max(np.abs(eng.get_state()[:, 2]))))
assert np.all(eng.get_field(N.FIELD_STEP_IDX) == T) and s["n_failed"] == 0
eng.close()
del cand

# 3. RQL with 11 TD rows (k_critic_fit_gen), 4096 tanks, 1 500 ticks of the generated grid
eng = Engine(preset_engine_config("2tank", 4096, Nactor=10, dtype="f64", mode="RQL", critic_struct="quadratic", Ncritic=12,
                                  buffer_size=20))
eng.set_state(np.stack([rng.uniform(0, 2, 4096), rng.uniform(-2, 2, 4096)], axis=-1))
T = 1500
eng.control_tick(None, K=64, T=T)
s, _ = eng.episode_stats(from_accum=True)
w = eng.get_field(N.FIELD_W_CRITIC)
line("2tank RQL, 11 TD rows", ticks=T, kernel=eng.last_launch(N.KERNEL_CRITIC)["variant"], n_failed=s["n_failed"],
     weights_in_box=bool(np.all((w >= 0) & (w <= 1e3))), step_idx_exact=bool(np.all(eng.get_field(N.FIELD_STEP_IDX) == T)))
assert np.all(eng.get_field(N.FIELD_STEP_IDX) == T) and np.all((w >= 0) & (w <= 1e3))
eng.close()

# 4. MPC with a full R1 (DMA_MPC_GENF), 16 384 robots, 5 000 ticks
A = rng.uniform(-1, 1, (7, 7))
B = 16384
eng = Engine(preset_engine_config("3wrobot", B, Nactor=10, dtype="f32", R1=A @ A.T + np.eye(7)))
eng.set_state(np.stack([rng.uniform(-10, 10, B), rng.uniform(-10, 10, B), rng.uniform(-np.pi, np.pi, B), rng.uniform(-1, 1, B),
                        rng.uniform(-1, 1, B)], axis=-1))
cand = eng.to_device((lo + (hi - lo) * rng.random((B, 64, 10, 2))).astype(np.float32))
T = 5000
for _ in range(T):
    eng.control_tick(cand, K=64)
s, _ = eng.episode_stats(from_accum=True)
line("3wrobot MPC full R1", ticks=T, kernel=eng.last_launch(N.KERNEL_ACTOR), n_failed=s["n_failed"],
     step_idx_exact=bool(np.all(eng.get_field(N.FIELD_STEP_IDX) == T)))
assert np.all(eng.get_field(N.FIELD_STEP_IDX) == T)
print("soak ok")
