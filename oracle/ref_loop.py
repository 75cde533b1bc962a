"""The reference's closed loop, restated around the oracle's operators and SciPy.  TEST INFRASTRUCTURE ONLY.

This is the *reference algorithm* (BASELINE.json configs[0]; BASELINE.md "CPU-baseline plan" item 1): one env,
``scipy.integrate.RK45`` exactly as ``Simulator`` configures it (rcognita/simulator.py:150: ``max_step = dt/2``,
``first_step = 1e-6``, ``atol = 1e-5``, ``rtol = 1e-3``) and ``scipy.optimize.minimize(method='SLSQP')`` exactly as
``_actor_optimizer`` / ``_critic_optimizer`` call it (rcognita/controllers.py:1373-1398, 1255-1264), driven by the
loop body of presets/main_3wrobot.py:419-446.  The arithmetic inside the callbacks is the oracle's
(oracle/rcg_oracle.py); the quirks of the reference loop are reproduced deliberately because the golden traces
(tests/golden/F7_trace_*.npz) contain them:

* the controller's sampling test is a bare float comparison on RK45's irregular time grid (controllers.py:1440-1442);
* ``state_sys`` is handed over AFTER ``compute_action`` (presets/main_3wrobot.py:425-428), so every rollout starts
  from the state of the previous sim step while ``observation_sqn[0]`` is the current observation;
* ``upd_accum_obj`` runs every sim step with ``sampling_time`` as the weight (controllers.py:1093);
* the action array returned by the controller is the array the system clips in place (systems.py:241-243);
* before the first tick the applied action is ``action_min / 10`` (or ``action_init``), while RK45's constructor
  evaluated the right-hand side once with zeros (systems.py:134).

It pins the oracle at the level of the whole loop (tests/test_ref_loop.py) and gives bench.py the CPU number of the
reference *algorithm* next to the CPU number of the GPU algorithm.
"""
from __future__ import annotations

import warnings

import numpy as np
from scipy.integrate import RK45
from scipy.optimize import Bounds, minimize

from . import rcg_oracle as O


class RefLoop:
    def __init__(self, cfg: O.OracleCfg, state_init, t1, action_init=None, atol=1e-5, rtol=1e-3, actor_tol=1e-7,
                 critic_tol=1e-7, critic="slsqp"):
        # actor_tol / critic_tol: SLSQP's `tol` (the reference hard-codes 1e-7, controllers.py:1264, 1396); other values
        # exist for ONE purpose - measuring how far the reference's own closed loop moves when nothing but the optimiser's
        # stopping rule changes (tests/test_critic_traces.py: the band a different optimiser can be held to)
        # critic = "exact": the reference's loop with ONE ingredient exchanged - its critic's SLSQP call replaced by the exact
        # minimiser of the build-defined objective (O.critic_fit).  SLSQP leaves 4 .. 74 % of the robots' fits at their start
        # point w_init (fixture field tick_critic_status), so on some traces this exchange alone moves the reference's own loop
        # by more than the band; tests/test_hip_ref_traces.py holds the HIP loop to THIS loop as well (F7c_sensitivity.json)
        self.actor_tol, self.critic_tol, self.critic_kind = actor_tol, critic_tol, critic
        self.cfg = cfg
        ds, du = cfg.ds, cfg.du
        self.dt = cfg.sampling_time
        self.t1 = t1
        self.sys_action = np.zeros(du)  # System.action (systems.py:134)
        lo, hi = cfg.ctrl_bnds[:, 0], cfg.ctrl_bnds[:, 1]
        self.action_curr = lo / 10 if action_init is None else np.asarray(action_init, dtype=float)
        self.action_sqn_init = np.tile(self.action_curr, cfg.n_actor)  # controllers.py:973-978
        self.sqn_min, self.sqn_max = np.tile(lo, cfg.n_actor), np.tile(hi, cfg.n_actor)
        self.state_sys = np.asarray(state_init, dtype=float)
        self.sys_state = np.zeros(ds)  # System._state: state of the last RHS evaluation (systems.py:251)
        self.ctrl_clock = 0.0
        self.critic_clock = 0.0
        self.critic_period = cfg.sampling_time * max(cfg.critic_every_ticks, 1)
        self.accum = 0.0
        self.act_buf = np.zeros((cfg.buffer_size, du))
        self.obs_buf = np.zeros((cfg.buffer_size, ds))
        dc = cfg.dc
        self.w_prev = np.ones(dc)
        self.w_init = self.w_prev
        self.w = np.ones(dc)
        self.Wmin, self.Wmax = O.critic_bounds(cfg.critic_struct, dc)
        self.nfev_actor = 0
        self.solver = RK45(self._closed_loop_rhs, 0.0, np.asarray(state_init, dtype=float), t1, max_step=self.dt / 2,
                           first_step=1e-6, atol=atol, rtol=rtol)

    # System.closed_loop_rhs (systems.py:213-253)
    def _closed_loop_rhs(self, t, y):
        a = self.sys_action
        if np.any(self.cfg.ctrl_bnds):
            for k in range(self.cfg.du):  # in place, as the reference
                a[k] = np.clip(a[k], self.cfg.ctrl_bnds[k, 0], self.cfg.ctrl_bnds[k, 1])
        self.sys_state = y
        return O.state_dyn(self.cfg.sys_id, y, a, self.cfg.pars)

    # Single-env cost in the reference's OPERATION ORDER (controllers.py:1063-1084, 1192-1214, 1284-1326):
    # SLSQP differentiates it by 2-point finite differences with step sqrt(eps), which turns a last-bit
    # difference of J into a different search path, so for a step-by-step comparison with the captured traces
    # the callback has to round exactly like the reference does (chi @ R1 @ chi as two BLAS products,
    # gamma**k, w @ regressor).  tests/test_ref_loop.py checks it against O.actor_cost as well.
    def _rho(self, y, u):
        c = self.cfg
        chi = np.concatenate([y if c.target is None else y - c.target, u])
        if c.stage_obj_struct == O.STAGE_QUADRATIC:
            return chi @ c.R1 @ chi
        return chi**2 @ c.R2 @ chi**2 + chi @ c.R1 @ chi

    def _q(self, y, u, w):
        return w @ O.critic_features(y, u, self.cfg)

    def _actor_cost(self, sqn, obs):
        self.nfev_actor += 1
        c = self.cfg
        u = np.reshape(sqn, [c.n_actor, c.du])
        ys = np.zeros([c.n_actor, c.ds])
        ys[0, :] = obs
        state = self.state_sys
        for k in range(1, c.n_actor):
            state = state + c.pred_step_size * O.state_dyn(c.sys_id, state, u[k - 1, :], c.pars)
            ys[k, :] = state
        J = 0
        if c.mode == O.MODE_MPC:
            for k in range(c.n_actor):
                J += c.gamma**k * self._rho(ys[k, :], u[k, :])
        elif c.mode == O.MODE_RQL:
            for k in range(c.n_actor - 1):
                J += c.gamma**k * self._rho(ys[k, :], u[k, :])
            J += self._q(ys[-1, :], u[-1, :], self.w)
        else:
            for k in range(c.n_actor):
                J += self._q(ys[k, :], u[k, :], self.w)
        return J

    def _critic_cost(self, w):
        c = self.cfg
        Jc = 0
        for k in range(c.n_critic - 1, 0, -1):
            yp, yn, up, un = self.obs_buf[k - 1, :], self.obs_buf[k, :], self.act_buf[k - 1, :], self.act_buf[k, :]
            e = self._q(yp, up, w) - c.gamma * self._q(yn, un, self.w_prev) - self._rho(yp, up)
            Jc += 1 / 2 * e**2
        return Jc

    def _actor_optimizer(self, obs):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            try:
                res = minimize(lambda a: self._actor_cost(a, obs), self.action_sqn_init.copy(), method="SLSQP",
                               tol=self.actor_tol, bounds=Bounds(self.sqn_min, self.sqn_max, keep_feasible=True),
                               options={"maxiter": 300, "disp": False})
                sqn = res.x
            except ValueError:  # controllers.py:1400-1402
                sqn = self.action_sqn_init.copy()
        return sqn[: self.cfg.du]

    def _critic_optimizer(self):
        if self.critic_kind == "exact":
            return O.critic_fit(self.cfg, self.w_prev[None], self.obs_buf[None], self.act_buf[None])[0]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return minimize(self._critic_cost, self.w_init, method="SLSQP", tol=self.critic_tol,
                            bounds=Bounds(self.Wmin, self.Wmax, keep_feasible=True),
                            options={"maxiter": 200, "disp": False}).x

    # CtrlOptPred.compute_action (controllers.py:1429-1493)
    def _compute_action(self, t, obs):
        if t - self.ctrl_clock >= self.cfg.sampling_time:
            self.ctrl_clock = t
            if self.cfg.mode != O.MODE_MPC:
                in_period = t - self.critic_clock
                self.act_buf = np.vstack([self.act_buf[1:], self.action_curr])
                self.obs_buf = np.vstack([self.obs_buf[1:], obs])
                if in_period >= self.critic_period:
                    self.critic_clock = t
                    self.w = self._critic_optimizer()
                    self.w_prev = self.w
                else:
                    self.w = self.w_prev
            action = self._actor_optimizer(obs)
            self.action_curr = action
            return action
        return self.action_curr

    def step(self):
        """One iteration of the loop body; returns the row the golden traces store."""
        self.solver.step()
        t, y = self.solver.t, self.solver.y
        obs = y
        action = self._compute_action(t, obs)
        self.sys_action = action  # receive_action: the SAME array object (aliasing)
        self.state_sys = self.sys_state  # receive_sys_state(my_sys._state), after compute_action
        rho = float(self._rho(obs, action))
        self.accum += rho * self.cfg.sampling_time  # upd_accum_obj, every sim step
        return np.concatenate([[t], np.array(y, dtype=float), np.array(action, dtype=float), [rho, self.accum]])

    def run(self):
        rows = []
        while True:
            rows.append(self.step())
            if self.solver.t >= self.t1:
                break
        return np.stack(rows)


def _worker_main(argv=None):
    """``python -m oracle.ref_loop --seconds S --nactor N``: ONE env of the reference's algorithm on ONE core for S seconds
    (the 3wrobot preset at the given horizon); prints one JSON line.  bench.py starts one of these per usable host core
    for its ``cpu_baseline.reference_algorithm`` figure (one env per process, core count printed)."""
    import argparse
    import json
    import time

    p = argparse.ArgumentParser()
    p.add_argument("--seconds", type=float, default=8.0)
    p.add_argument("--nactor", type=int, default=10)
    a = p.parse_args(argv)
    cfg = O.OracleCfg(sys_id=O.SYS_3WROBOT, n_actor=a.nactor, pars=[10.0, 1.0],
                      ctrl_bnds=np.array([[-300.0, 300.0], [-100.0, 100.0]]), R1=np.diag([1.0, 10.0, 1.0, 0, 0, 0, 0]),
                      gamma=1.0, dt_sim=0.01, sampling_time=0.01, pred_step_size=0.02)
    loop = RefLoop(cfg, np.array([5.0, 5.0, -3 * np.pi / 4, 0.0, 0.0]), t1=1e9)
    t0 = time.perf_counter()
    ticks, last = 0, None
    while time.perf_counter() - t0 < a.seconds:
        row = loop.step()
        act = tuple(row[1 + cfg.ds:1 + cfg.ds + cfg.du])
        if last is not None and act != last:
            ticks += 1
        last = act
    dt = time.perf_counter() - t0
    print(json.dumps({"ticks": ticks, "seconds": dt, "nfev_actor": loop.nfev_actor}))


if __name__ == "__main__":
    _worker_main()
