#!/usr/bin/env python3
"""Host-side cost of one B = 1 loop iteration of the drop-in classes WITHOUT a device: the same loop as tools/b1_profile.py on a stub
engine whose native calls return at once (any machine, no GPU) - what the mirror classes' own Python costs per simulation step,
by function (cProfile, own time).   python tools/b1_python_only.py"""
import cProfile, pstats, sys, time
sys.path.insert(0, '.')
import numpy as np
from rcognita_amd import controllers, simulator, engine


class StubEngine:
    """The calls the fused loop makes, answered with constants of the right shape."""
    def __init__(self, cfg):
        self.cfg, self.B = cfg, cfg.batch
        self.ds, self.du = {0: (5, 2), 1: (3, 2), 2: (2, 1)}[cfg.sys_id]
        self.dy, self.dc, self.real = self.ds, 0, np.float64
        self._row = np.zeros((self.B, self.ds + self.du + 2))
    def set_state(self, x, also_init=True): pass
    def set_field(self, f, v): pass
    def set_optimizer(self, memory=-1, ftol=None): pass
    def get_state(self): return np.zeros((self.B, self.ds))
    def loop_step(self, action, step, n_substeps=1, decide=False, push=False, fit=False, iters=10):
        out = np.empty((self.B, self.ds + self.du + 2)); out[:] = 0.5
        ds, du = self.ds, self.du
        return out[:, :ds], out[:, ds:ds + du], out[:, ds + du], out[:, ds + du + 1], None
    def loop_step_begin(self, *a, **k): pass
    def loop_step_end(self, drop=False): return None if drop else self.loop_step(None, 0)
    def stage_obj(self, y, a): return np.zeros(self.B)
    def actor_optimize(self, iters=0, obs=None, state_sys=None): return np.full((self.B, self.du), 0.5), np.full((self.B, 5, self.du), 0.5), np.zeros(self.B), np.zeros(self.B, np.int32)
    def sim_step(self, n, step=None): pass
    def get_field(self, f): return np.zeros((self.B, 1))
    def close(self): pass


controllers.Engine = simulator.Engine = StubEngine
from tests.test_hip_ref_traces import make_loop_objects  # noqa: E402


def loop(t1, speculate=True):
    plant, ctrl, sim = make_loop_objects("3wrobot", "MPC", 5, t1)
    ctrl.speculate = speculate
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
            return n


loop(0.5)
for spec in (True, False, True, False):
    t0 = time.perf_counter(); n = loop(20.0, spec); dt = time.perf_counter() - t0
    print(f"stub engine, speculate={spec}: {dt / n * 1e6:.1f} us of Python per simulation step ({n} steps)")
pr = cProfile.Profile(); pr.enable(); loop(5.0); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(16)
