"""Simulator with the reference's interface (rcognita/simulator.py), backed by librcg.

``Simulator.sim_step`` integrates ``closed_loop_rhs`` with the build's fixed-step classical RK4
(``rcg_sim_step``) instead of scipy's adaptive RK45: one call advances exactly ``dt`` (``n_substeps``
RK4 substeps of ``dt / n_substeps``), time is ``t0 + k*dt`` with an integer step counter, so tick and
episode indexing is exact.  ``max_step, first_step, atol, rtol`` are accepted for signature
compatibility and ignored (the reference itself ignores ``max_step``, rcognita/simulator.py:150).
Against the reference's RK45 loop under a constant action the trajectories agree to < 1e-5 relative
(tests/test_hip_parity.py::test_F6_hip_rk4_vs_reference_rk45).

``state_init`` may be ``[ds]`` (one env, as in the reference) or ``[B, ds]`` (a batch of envs).
"""
from __future__ import annotations

import numpy as np

from . import _native as N
from .engine import Engine, EngineConfig
from .systems import System


class Simulator:
    def __init__(self, sys_type, closed_loop_rhs, sys_out, state_init, disturb_init=[], action_init=[], t0=0, t1=1,
                 dt=1e-2, max_step=0.5e-2, first_step=1e-6, atol=1e-5, rtol=1e-3, is_disturb=0, is_dyn_ctrl=0,
                 n_substeps=1, dtype="f64", device=0):
        if sys_type != "diff_eqn":
            raise NotImplementedError("only sys_type='diff_eqn' is on the native path (SURVEY.md 8a row 2)")
        if is_dyn_ctrl:
            raise NotImplementedError("is_dyn_ctrl is out of scope (SURVEY.md 8a row 1)")
        sys_obj = getattr(closed_loop_rhs, "__self__", None)
        if not isinstance(sys_obj, System):
            raise TypeError(
                "closed_loop_rhs must be the bound method of a rcognita_amd System (e.g. my_sys.closed_loop_rhs): "
                "arbitrary Python callables cannot run on the native path and there is no CPU fallback")
        self.sys = sys_obj
        self.sys_type = sys_type
        self.closed_loop_rhs = closed_loop_rhs
        self.sys_out = sys_out
        self.dt = dt
        self.t0, self.t1 = t0, t1
        self.n_substeps = int(n_substeps)

        state_init = np.asarray(state_init, dtype=float)
        self._batched = state_init.ndim == 2
        self.dim_state = state_init.shape[-1]
        x0 = state_init.reshape(-1, self.dim_state)
        self.B = x0.shape[0]
        spec = sys_obj.native_spec()
        a0 = np.zeros(sys_obj.dim_input) if len(action_init) == 0 else np.asarray(action_init, dtype=float)
        self.is_disturb = bool(is_disturb)
        if self.is_disturb != bool(sys_obj.is_disturb):
            raise ValueError("Simulator(is_disturb=...) must agree with the System it integrates")
        dist = dict(spec["disturb"])
        if self.is_disturb:  # state_full_init = [state_init, disturb_init] (simulator.py:131-134)
            self._q0 = np.zeros(sys_obj.dim_disturb) if len(disturb_init) == 0 else np.asarray(disturb_init, dtype=float)
            dist.update(disturb_init=self._q0, env_id_base=0)
        self._eng = Engine(EngineConfig(sys_id=spec["sys_id"], batch=self.B, dtype=dtype, device=device,
                                        pars=spec["pars"], ctrl_bnds=spec["ctrl_bnds"],
                                        dt_sim=float(dt) / self.n_substeps, sampling_time=float(dt),
                                        action_init=a0.reshape(-1)[: sys_obj.dim_input], **dist))
        self._eng.set_state(x0)
        if self.is_disturb:
            state_init = np.concatenate([state_init, np.broadcast_to(self._q0, state_init.shape[:-1] + self._q0.shape)],
                                        axis=-1)
        self.state_full_init = state_init.copy()
        self.dtype = dtype
        self._eng_stale = False  # the fused loop step advanced the state on the controller's handle, not on this one
        self.fuse = True  # build-specific: let sim_step run the fused loop step when a CtrlOptPred of the same System allows it
        self.step_idx = 0  # int: sim steps done in the current episode
        self.episode_idx = 0
        self.t = t0
        self.state_full = state_init.copy()
        self.state = self.state_full[..., 0:self.dim_state]
        self.observation = self.sys_out(self.state)

    def _shape(self, a):
        return a if self._batched else a[0]

    def sim_step(self, t_next=None):
        """One simulation step of length ``dt`` (rcognita/simulator.py:156-168): the action held by the
        system object is clipped to the control bounds and applied over the whole step.

        ``t_next`` (build-specific, optional): advance to that time instead - one step of length ``t_next - t``
        (rcg_sim_step_h).  The reference's solver picks its own steps; a caller that has a recorded time grid of the
        reference can walk it (tests/test_hip_ref_traces.py)."""
        act = np.asarray(self.sys.action, dtype=float)
        act = act.reshape(self.B, -1) if act.size == self.B * self.sys.dim_input else np.broadcast_to(act, (self.B, self.sys.dim_input))
        ctrl = self.sys._fused_controller(self) if self.fuse else None
        if ctrl is not None:  # the whole loop iteration in one native call (rcg_loop_step); compute_action / stage_obj pick it up
            self.step_idx += 1
            t_new = self.t0 + self.step_idx * self.dt if t_next is None else float(t_next)
            st = ctrl._fused_step(self, act, t_new, float(self.dt) if t_next is None else float(t_next) - float(self.t))
            if st is not None:
                self.t = t_new
                self._eng_stale = True
                self.state_full = self._shape(st)
                self.state = self.state_full[..., 0:self.dim_state]
                self.observation = self.sys_out(self.state)
                # what System.closed_loop_rhs leaves behind: the clipped action (systems.py:241-243); the controller's own last
                # decision - the optimiser's iterates live inside the box - needs no clip
                if self.sys.action is not getattr(ctrl, "_inb_action", None) and self.sys.ctrl_bnds.any():
                    b = self.sys.ctrl_bnds
                    self.sys.action = np.clip(np.asarray(self.sys.action, dtype=float), b[:, 0], b[:, 1])
                self.sys._state = self.state
                return
            self.step_idx -= 1  # (this step could not be fused after all: the separate calls below)
        if self._eng_stale:  # the controller's handle holds the current state: bring this one up to date
            self._eng.set_state(np.asarray(self.state_full, dtype=float).reshape(self.B, -1)[:, :self.dim_state], also_init=False)
            self._eng_stale = False
        ref = getattr(self.sys, "_ctrl_ref", None)
        if ref is not None and ref() is not None:
            ref()._fused_dirty = True  # the state moves outside the controller's handle
        self._eng.set_field(N.FIELD_ACTION, act)
        self.step_idx += 1
        if t_next is None:
            self._eng.sim_step(self.n_substeps)
            self.t = self.t0 + self.step_idx * self.dt
        else:
            self._eng.sim_step(self.n_substeps, step=float(t_next) - float(self.t))
            self.t = float(t_next)
        st = self._eng.get_state().astype(float)
        if self.is_disturb:
            st = np.concatenate([st, self._eng.get_field(N.FIELD_DISTURB).astype(float)], axis=-1)
        self.state_full = self._shape(st)
        self.state = self.state_full[..., 0:self.dim_state]  # simulator.py:166
        self.observation = self.sys_out(self.state)
        # what System.closed_loop_rhs leaves behind in the reference: the clipped action and the
        # state of the last RHS evaluation (rcognita/systems.py:241-251)
        if self.sys.ctrl_bnds.any():
            b = self.sys.ctrl_bnds
            self.sys.action = np.clip(np.asarray(self.sys.action, dtype=float), b[:, 0], b[:, 1])
        self.sys._state = self.state

    def get_sim_step_data(self):
        """rcognita/simulator.py:187-195."""
        return self.t, self.state, self.observation, self.state_full

    def status(self):
        """Per-env status words (bit 0: a non-finite state was produced; the env is frozen)."""
        return self._eng.get_field(N.FIELD_STATUS)

    def reset(self):
        """Episode reset (rcognita/simulator.py:197-204, whose ``.observation`` assignment never resets
        the solver state - SURVEY.md 3.4): here the state really returns to ``state_init``."""
        self._eng.episode_reset()
        self._eng_stale = False
        ref = getattr(self.sys, "_ctrl_ref", None)
        if ref is not None and ref() is not None:
            ref()._fused_dirty = True
        self.step_idx = 0
        self.episode_idx += 1
        self.t = self.t0
        self.state_full = self.state_full_init.copy()
        self.state = self.state_full[..., 0:self.dim_state]
        self.observation = self.sys_out(self.state)
