"""Environment classes with the reference's interface (rcognita/systems.py), backed by librcg.

Same class names, constructor signature and method names as the reference, so a preset written for
``rcognita.systems`` runs against ``rcognita_amd.systems`` unchanged.  Differences:

* every method also accepts a leading batch axis (``state [B, ds]``, ``action [B, du]``);
* the arithmetic runs in the HIP kernels (``rcg_rhs``); there is no NumPy implementation here and no
  fallback - without the library or a GPU the calls raise;
* ``is_disturb=1`` (full state ``[state, disturb]``, rcognita/systems.py:140-145, 325-345) is on the native path
  (``rcg_rhs_full``, ``RCG_FLAG_DISTURB``).  The reference draws ``randn()`` from the unseeded global RNG inside every
  right-hand side; here the noise comes from the counter-based generator of rcg_disturb.hpp (key ``seed``): a
  ``System`` object called directly draws one vector per call, a ``Simulator`` one per RK4 substep and env;
* ``is_dyn_ctrl`` is rejected: no preset sets it and the reference's dynamic-controller branch is itself broken
  (SURVEY.md 8a row 1).
"""
from __future__ import annotations

import numpy as np

from . import _native as N
from .engine import Engine, EngineConfig


class System:
    """Interface class of dynamical systems a.k.a. environments (rcognita/systems.py:17-253)."""

    _sys_id = None  # set by the concrete systems
    _ctrl_ref = None  # weak reference to the CtrlOptPred built around this system (_register_controller)
    name = "system"

    def __init__(self, sys_type, dim_state, dim_input, dim_output, dim_disturb, pars=[], ctrl_bnds=[], is_dyn_ctrl=0,
                 is_disturb=0, pars_disturb=[], dtype="f64", device=0, seed=0):
        if self._sys_id is None:
            raise NotImplementedError(
                "only the built-in systems (Sys3WRobot, Sys3WRobotNI, Sys2Tank) run on the native path; "
                "rcognita_amd has no Python fallback for user-defined dynamics")
        if sys_type != "diff_eqn":
            raise NotImplementedError("only sys_type='diff_eqn' is on the native path (SURVEY.md 8a row 2)")
        if is_dyn_ctrl:
            raise NotImplementedError("is_dyn_ctrl is out of scope (SURVEY.md 8a row 1)")
        ds, du, npar = N.SYS_DIMS[self._sys_id]
        if is_disturb and dim_disturb != N.DIM_DISTURB[self._sys_id]:
            raise ValueError(f"{type(self).__name__} has dim_disturb = {N.DIM_DISTURB[self._sys_id]}")
        if (dim_state, dim_input, dim_output) != (ds, du, ds):
            raise ValueError(f"{type(self).__name__} has dims (state, input, output) = ({ds}, {du}, {ds})")
        self.sys_type = sys_type
        self.dim_state, self.dim_input, self.dim_output, self.dim_disturb = dim_state, dim_input, dim_output, dim_disturb
        self.pars = pars
        self.ctrl_bnds = np.zeros((du, 2)) if len(ctrl_bnds) == 0 else np.asarray(ctrl_bnds, dtype=float)
        self.is_dyn_ctrl, self.is_disturb, self.pars_disturb = is_dyn_ctrl, is_disturb, pars_disturb
        self._state = np.zeros(dim_state)
        self.action = np.zeros(dim_input)
        self._dim_full_state = dim_state + (dim_disturb if is_disturb else 0)  # systems.py:136-145
        if is_disturb and self._sys_id != N.SYS_2TANK:  # systems.py:303-306, 365-368
            self.sigma_disturb, self.mu_disturb, self.tau_disturb = pars_disturb[0], pars_disturb[1], pars_disturb[2]
        self.dtype, self.device, self.seed = dtype, device, int(seed)
        self._ops = None  # lazily created operator engine (batch 1: rcg_rhs takes any number of points)
        self._noise_calls = 0  # draws made by direct calls of _disturb_dyn / closed_loop_rhs on this object

    # ---- native plumbing ---------------------------------------------------------------------
    def _disturb_cfg(self):
        if not self.is_disturb:
            return {}
        pd = self.pars_disturb if len(self.pars_disturb) == 3 else [np.zeros(self.dim_disturb)] * 3
        return dict(is_disturb=True, pars_disturb=[np.asarray(v, dtype=float) for v in pd], seed=self.seed)

    def _engine(self) -> Engine:
        if self._ops is None:
            self._ops = Engine(EngineConfig(sys_id=self._sys_id, batch=1, dtype=self.dtype, device=self.device,
                                            pars=list(self.pars), ctrl_bnds=self.ctrl_bnds, **self._disturb_cfg()))
        return self._ops

    # ---- the fused loop step (rcg_loop_step): the System is what a Simulator and a CtrlOptPred have in common - the simulator is
    # built around `my_sys.closed_loop_rhs`, the controller around `my_sys._state_dyn` (presets/main_3wrobot.py:218-320) - so it is
    # where the two find each other
    def _register_controller(self, ctrl):
        import weakref

        self._ctrl_ref = weakref.ref(ctrl)

    def _fused_controller(self, sim):
        ref = getattr(self, "_ctrl_ref", None)
        ctrl = ref() if ref is not None else None
        return ctrl if (ctrl is not None and ctrl._can_fuse(sim)) else None

    def native_spec(self):
        """What a Simulator / CtrlOptPred needs to build its own handle for this system."""
        return dict(sys_id=self._sys_id, pars=list(self.pars), ctrl_bnds=self.ctrl_bnds, disturb=self._disturb_cfg())

    def _draw(self):
        """One noise vector ``[dim_disturb]`` for a direct call (the reference: ``randn()`` per component and call,
        systems.py:343): draw number ``_noise_calls`` of env 0, episode 0 of this object's generator."""
        eng = self._engine()
        eng.set_field(N.FIELD_SUBSTEP_IDX, np.array([self._noise_calls], dtype=np.int32))
        self._noise_calls += 1
        return eng.disturb_noise()[1][0, : self.dim_disturb].astype(float)

    def _call_rhs_full(self, state, disturb, action, xi, clip):
        state, disturb, action = (np.asarray(v, dtype=float) for v in (state, disturb, action))
        lead = np.broadcast_shapes(state.shape[:-1], disturb.shape[:-1], action.shape[:-1])
        bc = lambda a, d: np.broadcast_to(a, lead + (d,)).reshape(-1, d)
        s2, q2, a2 = bc(state, self.dim_state), bc(disturb, self.dim_disturb), bc(action, self.dim_input)
        x2 = np.broadcast_to(np.asarray(xi, dtype=float), q2.shape)
        d, dq, ca = self._engine().rhs_full(s2, q2, a2, x2, clip=clip)
        return (d.astype(float).reshape(lead + (self.dim_state,)), dq.astype(float).reshape(lead + (self.dim_disturb,)),
                ca.astype(float).reshape(lead + (self.dim_input,)))

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
        """Right-hand side of the open-loop system (unclipped action), rcognita/systems.py:147-154; with
        ``is_disturb`` and a disturbance given it enters as in systems.py:317-319, 373-376."""
        if self.is_disturb and len(disturb):
            return self._call_rhs_full(state, disturb, action, np.zeros(self.dim_disturb), clip=False)[0]
        return self._call_rhs(state, action, clip=False)[0]

    def _disturb_dyn(self, t, disturb, xi=None):
        """rcognita/systems.py:325-345, 384-394 (2tank: zeros, :421-424).  ``xi`` = the value of ``randn()`` per
        component; default: the next draw of this object's generator."""
        if not self.is_disturb:
            raise ValueError("_disturb_dyn needs a system created with is_disturb=1")
        xi = self._draw() if xi is None else xi
        q = np.asarray(disturb, dtype=float)
        zs = np.zeros(q.shape[:-1] + (self.dim_state,))
        return self._call_rhs_full(zs, q, np.zeros(self.dim_input), xi, clip=False)[1]

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
        state_full = np.asarray(state_full, dtype=float)
        state = state_full[..., 0:self.dim_state]
        if self.is_disturb:  # full state [state, disturb]; one noise draw per call
            disturb = state_full[..., self.dim_state:self.dim_state + self.dim_disturb]
            d, dq, ca = self._call_rhs_full(state, disturb, self.action, self._draw(), clip=bool(self.ctrl_bnds.any()))
            self.action = ca
            self._state = state
            return np.concatenate([d, dq], axis=-1)
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
