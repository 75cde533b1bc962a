"""Environment classes with the reference's interface (rcognita/systems.py), backed by librcg.

Same class names, constructor signature and method names as the reference, so a preset written for
``rcognita.systems`` runs against ``rcognita_amd.systems`` unchanged.  Differences:

* every method also accepts a leading batch axis (``state [B, ds]``, ``action [B, du]``);
* the arithmetic runs in the HIP kernels (``rcg_rhs``); there is no NumPy implementation here and no
  fallback - without the library or a GPU the calls raise;
* ``is_disturb`` / ``is_dyn_ctrl`` are rejected: no preset sets them and the reference's dynamic-
  controller branch is itself broken.
"""
from __future__ import annotations

import numpy as np

from . import _native as N
from .engine import Engine, EngineConfig


class System:
    """Interface class of dynamical systems a.k.a. environments (rcognita/systems.py:17-253)."""

    _sys_id = None  # set by the concrete systems
    name = "system"

    def __init__(self, sys_type, dim_state, dim_input, dim_output, dim_disturb, pars=[], ctrl_bnds=[], is_dyn_ctrl=0,
                 is_disturb=0, pars_disturb=[], dtype="f64", device=0):
        if self._sys_id is None:
            raise NotImplementedError(
                "only the built-in systems (Sys3WRobot, Sys3WRobotNI, Sys2Tank) run on the native path; "
                "rcognita_amd has no Python fallback for user-defined dynamics")
        if sys_type != "diff_eqn":
            raise NotImplementedError("only sys_type='diff_eqn' is on the native path (SURVEY.md 8a row 2)")
        if is_disturb or is_dyn_ctrl:
            raise NotImplementedError("is_disturb / is_dyn_ctrl are out of scope (SURVEY.md 8a rows 1, 8)")
        ds, du, npar = N.SYS_DIMS[self._sys_id]
        if (dim_state, dim_input, dim_output) != (ds, du, ds):
            raise ValueError(f"{type(self).__name__} has dims (state, input, output) = ({ds}, {du}, {ds})")
        self.sys_type = sys_type
        self.dim_state, self.dim_input, self.dim_output, self.dim_disturb = dim_state, dim_input, dim_output, dim_disturb
        self.pars = pars
        self.ctrl_bnds = np.zeros((du, 2)) if len(ctrl_bnds) == 0 else np.asarray(ctrl_bnds, dtype=float)
        self.is_dyn_ctrl, self.is_disturb, self.pars_disturb = is_dyn_ctrl, is_disturb, pars_disturb
        self._state = np.zeros(dim_state)
        self.action = np.zeros(dim_input)
        self._dim_full_state = dim_state
        self.dtype, self.device = dtype, device
        self._ops = None  # lazily created operator engine (batch 1: rcg_rhs takes any number of points)

    # ---- native plumbing ---------------------------------------------------------------------
    def _engine(self) -> Engine:
        if self._ops is None:
            self._ops = Engine(EngineConfig(sys_id=self._sys_id, batch=1, dtype=self.dtype, device=self.device,
                                            pars=list(self.pars), ctrl_bnds=self.ctrl_bnds))
        return self._ops

    def native_spec(self):
        """What a Simulator / CtrlOptPred needs to build its own handle for this system."""
        return dict(sys_id=self._sys_id, pars=list(self.pars), ctrl_bnds=self.ctrl_bnds)

    def _call_rhs(self, state, action, clip):
        state = np.asarray(state, dtype=float)
        action = np.asarray(action, dtype=float)
        lead = np.broadcast_shapes(state.shape[:-1], action.shape[:-1])
        s2 = np.broadcast_to(state, lead + (self.dim_state,)).reshape(-1, self.dim_state)
        a2 = np.broadcast_to(action, lead + (self.dim_input,)).reshape(-1, self.dim_input)
        d, ca = self._engine().rhs(s2, a2, clip=clip)
        return d.astype(float).reshape(lead + (self.dim_state,)), ca.astype(float).reshape(lead + (self.dim_input,))

    # ---- reference interface -----------------------------------------------------------------
    def _state_dyn(self, t, state, action, disturb=[]):
        """Right-hand side of the open-loop system (unclipped action), rcognita/systems.py:147-154."""
        return self._call_rhs(state, action, clip=False)[0]

    def _disturb_dyn(self, t, disturb):
        raise NotImplementedError("disturbance model is out of scope (SURVEY.md 8a row 8)")

    def _ctrl_dyn(self, t, action, observation):
        return np.zeros(self.dim_input)

    def out(self, state, action=[]):
        """System output = state for all built-in systems (rcognita/systems.py:185-198)."""
        return state

    def receive_action(self, action):
        """rcognita/systems.py:200-211."""
        self.action = action

    def closed_loop_rhs(self, t, state_full):
        """rcognita/systems.py:213-253: clip the stored action to ``ctrl_bnds`` (the reference clips the
        stored array in place; here the clipped value replaces ``self.action``), evaluate the dynamics,
        record ``_state``."""
        state = np.asarray(state_full, dtype=float)[..., 0:self.dim_state]
        d, ca = self._call_rhs(state, self.action, clip=bool(self.ctrl_bnds.any()))
        self.action = ca
        self._state = state
        return d


class Sys3WRobot(System):
    """Three-wheel robot with dynamical actuators (ENDI), rcognita/systems.py:255-351.
    state = (x, y, alpha, v, omega), action = (F, M), pars = (m, I)."""

    _sys_id = N.SYS_3WROBOT
    name = "3wrobot"


class Sys3WRobotNI(System):
    """Three-wheel robot with static actuators (non-holonomic integrator), rcognita/systems.py:353-399.
    state = (x, y, alpha), action = (v, omega)."""

    _sys_id = N.SYS_3WROBOT_NI
    name = "3wrobotNI"


class Sys2Tank(System):
    """Two-tank system with nonlinearity, rcognita/systems.py:401-428.
    state = (h1, h2), action = (u), pars = (tau1, tau2, K1, K2, K3)."""

    _sys_id = N.SYS_2TANK
    name = "2tank"
