"""Predictive optimal controller with the reference's interface (rcognita/controllers.py:679-1493),
backed by librcg.

``CtrlOptPred`` keeps the reference's constructor signature, attributes and method names
(``compute_action, receive_sys_state, stage_obj, upd_accum_obj, reset, _actor_cost, _critic,
_critic_cost, _actor_optimizer, _critic_optimizer``) and adds a leading batch axis everywhere:
``observation [dy]`` or ``[B, dy]``.  All arithmetic of those methods runs in the HIP kernels; there is
no NumPy/SciPy path in this module.

What differs from the reference, by construction (SURVEY.md hard part 1):

* ``_actor_optimizer``: the reference calls SciPy SLSQP (finite-difference gradients, one env).  Here the decision is
  made on the device, for every mode (MPC / RQL / SQL), stage-cost and critic structure:
  ``actor_opt='auto'|'gradient'`` (default) - the on-device optimiser ``rcg_actor_optimize``: ``opt_iters`` iterations of
  adjoint gradient + limited-memory quasi-Newton direction + 16-way projected line search from the reference's start
  sequence; on the reference's own decisions (fixtures F8 / F8c) it ends within 0.5 % of SLSQP's cost; an env stops early once
  an accepted step lowers J by no more than ``opt_ftol`` = 1e-7, the accuracy the reference hands SLSQP (``tol=1e-7``,
  controllers.py:1396; ``opt_ftol=0``: always ``opt_iters`` steps unless a line search finds nothing better);
  ``actor_opt='sampling'`` - the device-side candidate search ``rcg_actor_search``: ``rounds`` rounds of ``n_candidates``
  sequences generated (Philox), evaluated and refined around each round's winner in one launch; no candidate is ever
  built on the host.  ``candidates=`` evaluates an explicit set instead (``rcg_actor_argmin``).
* ``_critic_optimizer``: bounded least squares on the TD stack solved natively (rcg_critic_update),
  see rcognita_amd/csrc/rcg_critic_fit.hpp.
* the sampling clock uses a tolerance instead of a bare float comparison, so that on the fixed-step
  simulation grid ``t = t0 + k*dt`` every sampling period triggers exactly one control tick.
"""
from __future__ import annotations

import numpy as np

from . import _native as N
from .engine import Engine, EngineConfig
from .systems import System

_F64 = np.dtype(np.float64)
_FZ_NO_TICK = {"tick": False}  # what a fused step that was no sample leaves for compute_action: nothing to take


def ctrl_selector(t, observation, action_manual, ctrl_nominal, ctrl_benchmarking, mode):
    """Main interface for various controllers (rcognita/controllers.py:40-63)."""
    if mode == "manual":
        action = action_manual
    elif mode == "nominal":
        action = ctrl_nominal.compute_action(t, observation)
    else:  # Controller for benchmarking
        action = ctrl_benchmarking.compute_action(t, observation)
    return action


class CtrlOptPred:
    def __init__(self, dim_input, dim_output, mode="MPC", ctrl_bnds=[], action_init=[], t0=0, sampling_time=0.1,
                 Nactor=1, pred_step_size=0.1, sys_rhs=[], sys_out=[], state_sys=[], prob_noise_pow=1,
                 is_est_model=0, model_est_stage=1, model_est_period=0.1, buffer_size=20, model_order=3,
                 model_est_checks=0, gamma=1, Ncritic=4, critic_period=0.1, critic_struct="quad-nomix",
                 stage_obj_struct="quadratic", stage_obj_pars=[], observation_target=[],
                 # ---- build-specific, keyword-only in spirit ----
                 candidates=None, actor_opt="auto", opt_iters=30, n_candidates=256, rounds=6, seed=0, dtype="f64",
                 device=0, clock_tol=1e-9, opt_ftol=1e-7):
        if is_est_model:
            raise NotImplementedError("is_est_model=1 needs the absent `sippy` package and is out of scope "
                                      "(SURVEY.md 2, component 3)")
        if mode not in ("MPC", "RQL", "SQL"):
            raise ValueError(f"unknown mode {mode!r}")
        sys_obj = getattr(sys_rhs, "__self__", None)
        if not isinstance(sys_obj, System):
            raise TypeError(
                "sys_rhs must be the bound `_state_dyn` of a rcognita_amd System: arbitrary Python models cannot "
                "run on the native path and there is no CPU fallback")
        self.sys = sys_obj
        self.dim_input, self.dim_output = dim_input, dim_output
        self.mode = mode
        self.ctrl_clock = t0
        self.sampling_time = sampling_time
        # relative slack of the two sampling tests of compute_action; 0 = the reference's bare float comparison
        # (controllers.py:1440, 1466), which a caller replaying the reference's own time stamps needs
        self.clock_tol = float(clock_tol)
        self.Nactor = Nactor
        self.pred_step_size = pred_step_size
        ctrl_bnds = np.asarray(ctrl_bnds, dtype=float)
        self.action_min = np.array(ctrl_bnds[:, 0])
        self.action_max = np.array(ctrl_bnds[:, 1])
        self.action_sqn_min = np.tile(self.action_min, Nactor)  # rep_mat(., 1, Nactor), controllers.py:970
        self.action_sqn_max = np.tile(self.action_max, Nactor)
        if len(action_init) == 0:  # controllers.py:973-978
            self.action_curr = self.action_min / 10
        else:
            self.action_curr = np.asarray(action_init, dtype=float)
        self.action_sqn_init = np.tile(self.action_curr, Nactor)
        self.sys_rhs, self.sys_out = sys_rhs, sys_out

        state_sys = np.asarray(state_sys, dtype=float)
        self._batched = state_sys.ndim == 2
        self.B = state_sys.shape[0] if self._batched else 1
        self.state_sys = state_sys
        self.action_curr = np.broadcast_to(self.action_curr, (self.B, dim_input)).copy() if self._batched else self.action_curr
        self.action_buffer = np.zeros([buffer_size, dim_input])
        self.observation_buffer = np.zeros([buffer_size, dim_output])

        self.is_est_model = 0
        self.prob_noise_pow = prob_noise_pow
        self.buffer_size = buffer_size
        self.critic_clock = t0
        self.gamma = gamma
        self.Ncritic = int(np.min([Ncritic, buffer_size - 1]))  # controllers.py:1015
        self.critic_period = critic_period
        self.critic_struct = critic_struct
        self.stage_obj_struct = stage_obj_struct
        self.stage_obj_pars = stage_obj_pars
        self.observation_target = observation_target
        self.accum_obj_val = np.zeros(self.B) if self._batched else 0

        spec = sys_obj.native_spec()
        R1 = np.asarray(stage_obj_pars[0], dtype=float)
        R2 = np.asarray(stage_obj_pars[1], dtype=float) if len(stage_obj_pars) > 1 else None
        tgt = None if len(observation_target) == 0 else np.asarray(observation_target, dtype=float)
        self._spec = None  # the next loop iteration, started ahead (see _speculate)
        self._stage_last = None  # (bytes of observation [B, dy], bytes of action [B, du], stage cost [B]) of the last evaluation
        self._eng_raw = Engine(EngineConfig(
            sys_id=spec["sys_id"], batch=self.B, dtype=dtype, device=device, Nactor=Nactor, mode=mode,
            stage_obj_struct=stage_obj_struct, critic_struct=critic_struct, Ncritic=Ncritic, buffer_size=buffer_size,
            dt_sim=sampling_time, sampling_time=sampling_time, pred_step_size=pred_step_size, gamma=gamma,
            pars=spec["pars"], ctrl_bnds=ctrl_bnds, R1=R1, R2=R2, observation_target=tgt,
            action_init=None if len(action_init) == 0 else action_init, seed=int(seed)))
        self.dim_critic = self._eng.dc
        lo, hi = (-1e3, 1e3) if critic_struct in ("quad-lin", "quad-mix") else (0.0, 1e3)
        self.Wmin, self.Wmax = lo * np.ones(self.dim_critic), hi * np.ones(self.dim_critic)
        self.w_critic_prev = np.ones(self.dim_critic)
        self.w_critic_init = self.w_critic_prev
        self.w_critic = np.ones(self.dim_critic)

        # actor search settings
        self.candidates = None if candidates is None else np.asarray(candidates, dtype=float)
        self.n_candidates, self.rounds = int(n_candidates), int(rounds)
        self.opt_iters = int(opt_iters)
        self.opt_ftol = float(opt_ftol)
        self._eng.set_optimizer(-1, ftol=self.opt_ftol)
        if actor_opt not in ("auto", "gradient", "sampling"):
            raise ValueError(f"actor_opt must be 'auto', 'gradient' or 'sampling', got {actor_opt!r}")
        self._use_gradient = actor_opt in ("auto", "gradient")
        self._search_draw = 0  # decisions made so far: the draw counter of the device-side search
        self._prev_opt_val = None  # last optimal sequence [B, N, du]
        self._prev_opt_on_device = False
        self.last_J = None
        self.last_idx = None
        # the fused loop step (rcg_loop_step, Simulator.sim_step): what it computed ahead for compute_action / stage_obj
        self._dtype = dtype
        self._fz = None
        self._fused_dirty = True  # the handle's STATE / buffers / weights must be uploaded before the next fused step
        self.fused_steps = 0      # statistics: loop iterations served by one native call ...
        self.fused_decisions = 0  # ... and decisions of compute_action taken from them
        self.speculate = True     # build-specific: start the next simulation step at the end of compute_action (see _speculate)
        self.spec_hits = self.spec_drops = 0  # ... iterations started ahead that Simulator.sim_step then asked for / did not
        self._spec_ready = False  # the last simulation step ran on this controller's handle: its STATE is the simulator's
        self._spec_dropped = False
        self._last_sim = None
        sys_obj._register_controller(self)

    @property
    def _eng(self):
        """The controller's handle.  Whoever touches it outside the fused loop first gets rid of an iteration that was started
        ahead: it is waited for and dropped, and the handle is marked for a fresh upload of the host objects' state."""
        if self._spec is not None:
            self._spec_drop()
        return self._eng_raw

    # the optimal sequence of the last decision; after a fused step it stays on the device until somebody asks
    @property
    def _prev_opt(self):
        if self._prev_opt_on_device:
            self._prev_opt_val = self._eng.get_field(N.FIELD_ACTION_SQN).astype(float)
            self._prev_opt_on_device = False
        return self._prev_opt_val

    @_prev_opt.setter
    def _prev_opt(self, v):
        self._prev_opt_val, self._prev_opt_on_device = v, False

    # ------------------------------------------------------------------ fused loop step
    def _can_fuse(self, sim):
        """One native call per loop iteration (rcg_loop_step) when: the decision is the on-device optimiser, no disturbance
        model, same batch and element type as the simulator, rows that fit the handle's pinned buffer, a non-empty TD stack."""
        key = (self.candidates is None, self._use_gradient)
        hit = getattr(self, "_can_fuse_cache", None)
        if hit is not None and hit[0]() is sim and hit[1] == key:  # (a weak reference: an id() may be reused by a later object)
            return hit[2]
        row = self.dim_output + self.dim_input + 2 + (self.dim_critic if self.mode != "MPC" else 0)
        ok = (sim.sys is self.sys and sim.B == self.B and sim.dtype == self._dtype and not sim.is_disturb
              and self.candidates is None and self._use_gradient and (self.mode == "MPC" or self.Ncritic - 1 >= 1)
              and self.B * (row + self.dim_input) * 8 <= 16384)
        import weakref

        self._can_fuse_cache = (weakref.ref(sim), key, ok)
        return ok

    def _tick_flags(self, t):
        """What compute_action(t, .) will do, WITHOUT doing it: (sample, critic refit) - the float clock tests of
        controllers.py:1440, 1466."""
        tick = (t - self.ctrl_clock) >= self.sampling_time * (1 - self.clock_tol)
        fit = tick and self.mode != "MPC" and (t - self.critic_clock) >= self.critic_period * (1 - self.clock_tol)
        return bool(tick), bool(fit)

    def _fused_sync(self, state):
        """Bring the handle to the host objects' state: the simulator's state, the critic buffers and weights."""
        self._eng.set_state(np.asarray(state, dtype=float).reshape(self.B, -1), also_init=False)
        if self._spec_dropped:  # a dropped step may have frozen an env on the handle (non-finite state) that the host never saw
            self._eng.set_field(N.FIELD_STATUS, np.zeros(self.B, dtype=np.uint32))
            self._spec_dropped = False
        if self.mode != "MPC":
            self._sync_critic_state()
            self._eng.set_field(N.FIELD_W_CRITIC, self._b(self.w_critic, self.dim_critic))
        self._fused_dirty = False

    def _spec_drop(self):
        spec, self._spec = self._spec, None
        if spec is not None:
            self._eng_raw.loop_step_end(drop=True)  # STATE / ACTION / buffers / weights on the handle are the dropped step's:
            self._fused_dirty = True                #   the next fused step uploads the host objects' anew
            self._spec_dropped = True
            self.spec_drops += 1

    def _speculate(self, action):
        """Called at the end of compute_action.  In the reference's loop (presets/main_3wrobot.py:419-446) the action compute_action
        returns is the one the system holds over the NEXT simulation step (`my_sys.receive_action(action)` follows at once), while
        the loop body still has its bookkeeping of the current step to do (receive_action, receive_sys_state, upd_accum_obj, the
        logger): the next iteration - the step, and what compute_action / stage_obj will ask for at its end - is enqueued now
        (rcg_loop_step_begin) and collected by Simulator.sim_step (rcg_loop_step_end) if, by then, it is still the step that is asked
        for: same simulator, same held action, same time and step length, same clocks.  Anything else - another action handed to
        the system, a reset, any other use of the controller's handle - drops it (the handle is then uploaded anew): a run is the
        same numbers with and without."""
        self._spec_ready = False
        sim = self._last_sim() if self._last_sim is not None else None
        if (sim is None or not self.speculate or not sim.fuse or self._spec is not None or self._fused_dirty or self._fz is not None
                or not sim._eng_stale):
            return
        act = np.asarray(action, dtype=float)
        if act.size != self.B * self.dim_input:  # (one action for every env: the ordinary path broadcasts it)
            return
        act = act.reshape(self.B, self.dim_input)
        t_new = sim.t0 + (sim.step_idx + 1) * sim.dt
        tick, fit = self._tick_flags(t_new)
        push = tick and self.mode != "MPC"
        if push and not self._same(self._b(self.action_curr, self.dim_input), act):
            return
        step = float(sim.dt)
        self._eng_raw.loop_step_begin(act, step, sim.n_substeps, decide=tick, push=push, fit=fit, iters=self.opt_iters)
        self._spec = (sim, act.tobytes(), t_new, step, tick, fit, push, sim.n_substeps, self.opt_iters)

    def _fused_step(self, sim, act, t_new, step):
        """Called by Simulator.sim_step: hold `act` over one step, and compute ahead what the loop body will ask for at
        t_new - the decision (if t_new is a sample), the critic push / refit, the stage cost.  Returns the new state [B, ds],
        or None when this iteration cannot be fused (the critic push would need the controller's own action_curr where it
        differs from the action the system holds: only before the first decision)."""
        if self._fz is not None and self._fz["tick"]:  # a decision computed ahead that nobody took: the handle's buffers moved on
            self._fused_dirty = True
        self._fz = None
        tick, fit = self._tick_flags(t_new)
        push = tick and self.mode != "MPC"
        res = None
        actb = act.tobytes()
        spec = self._spec
        if spec is not None:  # this iteration may already be running (_speculate): take it if it is exactly the one asked for
            if (spec[0] is sim and spec[2] == t_new and spec[3] == step and spec[4:] == (tick, fit, push, sim.n_substeps, self.opt_iters) and not self._fused_dirty
                    and spec[1] == actb
                    and not (push and not self._same(self._b(self.action_curr, self.dim_input), act))):
                self._spec = None
                res = self._eng_raw.loop_step_end()
                self.spec_hits += 1
            else:
                self._spec_drop()
        if res is None:
            if push and not self._same(self._b(self.action_curr, self.dim_input), act):
                self._fused_dirty = True
                return None
            if self._fused_dirty:
                self._fused_sync(np.asarray(sim.state_full, dtype=float).reshape(self.B, -1)[:, :self.dim_output])
            res = self._eng_raw.loop_step(act, step, sim.n_substeps, decide=tick, push=push, fit=fit, iters=self.opt_iters)
        st, a, stage, bj, w = res
        stb = st.tobytes()
        if tick:  # what compute_action must be asked with for the decision to be the one it would make: the new state as the
            # observation, the state before the step (the simulator still holds it) as state_sys
            xsb = np.asarray(sim.state_full, dtype=float).reshape(self.B, -1)[:, :self.dim_output].tobytes()
            self._fz = dict(t=t_new, obs=stb, xs=xsb, tick=True, fit=fit, action=a, J=bj, w=w)
            self._stage_last = (stb, a.tobytes(), stage)
        else:
            self._fz = _FZ_NO_TICK
            self._stage_last = (stb, actb, stage)
        self.fused_steps += 1
        if self._last_sim is None or self._last_sim() is not sim:
            import weakref

            self._last_sim = weakref.ref(sim)
        self._spec_ready = True
        return st

    # ------------------------------------------------------------------ helpers
    def _b(self, a, d):
        """host array ``[d]`` or ``[B, d]`` -> ``[B, d]``"""
        a = np.asarray(a, dtype=float)
        if a.size == self.B * d:  # already one row per env: a view, no broadcast machinery (the B = 1 loop calls this 5 x per step)
            return a.reshape(self.B, d)
        return np.broadcast_to(a.reshape(-1, d), (self.B, d))

    def _unb(self, a):
        return a if self._batched else a[0]

    @staticmethod
    def _same(a, b):
        """Bitwise sameness of two small float arrays of equal shape (what 'the inputs the fused step computed ahead for' means)."""
        return a is b or (a.shape == b.shape and a.tobytes() == b.tobytes())

    # ------------------------------------------------------------------ reference interface
    def reset(self, t0):
        """rcognita/controllers.py:1046-1054: only the clock and the current action are reset."""
        if self._spec is not None:
            self._spec_drop()
        self.ctrl_clock = t0
        self.action_curr = self._unb(np.broadcast_to(self.action_min / 10, (self.B, self.dim_input)).copy())

    def receive_sys_state(self, state):
        """rcognita/controllers.py:1056-1061."""
        self.state_sys = state

    def stage_obj(self, observation, action):
        """rcognita/controllers.py:1063-1084 (rcg_stage_obj).  The reference's loop evaluates it twice per step on the
        same arguments (upd_accum_obj, then the logger: presets/main_3wrobot.py:429-441): the last result is kept."""
        last = self._stage_last
        # the loop's two calls per step, on the very arrays the fused step (or the first call) saw: float64 arrays whose bytes are
        # the cached ones - one row per env either way, so equal bytes are equal arguments
        if (last is not None and type(observation) is np.ndarray and type(action) is np.ndarray and observation.dtype == _F64
                and action.dtype == _F64 and observation.tobytes() == last[0] and action.tobytes() == last[1]):
            out = last[2]
        else:
            y, a = self._b(observation, self.dim_output), self._b(action, self.dim_input)
            yb, ab = np.ascontiguousarray(y).tobytes(), np.ascontiguousarray(a).tobytes()
            if last is not None and yb == last[0] and ab == last[1]:
                out = last[2]
            else:
                out = self._eng.stage_obj(y, a).astype(float)
                self._stage_last = (yb, ab, out)
        return out.copy() if self._batched else float(out[0])

    def upd_accum_obj(self, observation, action):
        """rcognita/controllers.py:1086-1093."""
        self.accum_obj_val = self.accum_obj_val + self.stage_obj(observation, action) * self.sampling_time

    def _critic(self, observation, action, w_critic):
        """rcognita/controllers.py:1192-1214 (rcg_critic)."""
        out = self._eng.critic(self._b(observation, self.dim_output), self._b(action, self.dim_input),
                               self._b(w_critic, self.dim_critic)).astype(float)
        return out if self._batched else float(out[0])

    def _sync_critic_state(self):
        ob = np.broadcast_to(self.observation_buffer, (self.B,) + self.observation_buffer.shape[-2:])
        ab = np.broadcast_to(self.action_buffer, (self.B,) + self.action_buffer.shape[-2:])
        self._eng.set_field(N.FIELD_OBS_BUF, ob)
        self._eng.set_field(N.FIELD_ACT_BUF, ab)
        self._eng.set_field(N.FIELD_W_PREV, self._b(self.w_critic_prev, self.dim_critic))

    def _critic_cost(self, w_critic):
        """rcognita/controllers.py:1216-1245 (rcg_critic_cost) on the current buffers."""
        self._sync_critic_state()
        out = self._eng.critic_cost(self._b(w_critic, self.dim_critic)).astype(float)
        return out if self._batched else float(out[0])

    def _critic_optimizer(self):
        """Replacement of rcognita/controllers.py:1248-1271: native bounded least squares on the TD stack
        of the CURRENT buffers (no push)."""
        # rcg_critic_update pushes (ACTION, STATE) before it fits: upload the rows pre-shifted so that the push restores
        # them (the values travel through the handle's element type exactly as set_field + get_field would round them)
        ob = np.broadcast_to(self.observation_buffer, (self.B,) + self.observation_buffer.shape[-2:]).astype(self._eng.real)
        ab = np.broadcast_to(self.action_buffer, (self.B,) + self.action_buffer.shape[-2:]).astype(self._eng.real)
        self._eng.set_field(N.FIELD_OBS_BUF, np.concatenate([ob[:, :1] * 0, ob[:, :-1]], axis=1))
        self._eng.set_field(N.FIELD_ACT_BUF, np.concatenate([ab[:, :1] * 0, ab[:, :-1]], axis=1))
        self._eng.set_field(N.FIELD_W_PREV, self._b(self.w_critic_prev, self.dim_critic))
        self._eng.set_field(N.FIELD_STATE, ob[:, -1])
        self._eng.set_field(N.FIELD_ACTION, ab[:, -1])
        self._eng.critic_update(do_fit=True)
        w = self._eng.get_field(N.FIELD_W_CRITIC).astype(float)
        return w if self._batched else w[0]

    def _actor_cost(self, action_sqn, observation):
        """rcognita/controllers.py:1273-1328 (rcg_actor_cost).  ``action_sqn`` is the reference's flat
        ``[N*du]`` vector, or ``[K, N*du]`` / ``[B, K, N*du]`` for many candidates at once."""
        a = np.asarray(action_sqn, dtype=float)
        single = a.ndim == 1
        if a.ndim == 1:
            a = a[None, None]
        elif a.ndim == 2:
            a = a[None]
        a = np.broadcast_to(a, (self.B,) + a.shape[1:]).reshape(self.B, -1, self.Nactor, self.dim_input)
        w = self._b(self.w_critic, self.dim_critic) if self.mode != "MPC" else None
        J = self._eng.actor_cost(a, obs=self._b(observation, self.dim_output),
                                 state_sys=self._b(self.state_sys, self.dim_output), w=w).astype(float)
        if single:
            return J[:, 0] if self._batched else float(J[0, 0])
        return J if self._batched else J[0]

    # ------------------------------------------------------------------ actor
    def _actor_optimizer(self, observation):
        """Replacement of rcognita/controllers.py:1330-1427.  Returns the first action of the best sequence.  Every
        variant decides on the device; like the reference, every call starts from ``action_sqn_init``."""
        obs = self._b(observation, self.dim_output)
        xs = self._b(self.state_sys, self.dim_output)
        if self.mode != "MPC":
            self._eng.set_field(N.FIELD_W_CRITIC, self._b(self.w_critic, self.dim_critic))
        if self.candidates is not None:
            cand = self.candidates
            cand = np.broadcast_to(cand if cand.ndim == 4 else cand[None], (self.B,) + cand.shape[-3:])
            act, bj, bi = self._eng.actor_argmin(cand, obs=obs, state_sys=xs)
            self._prev_opt = cand[np.arange(self.B), bi]
        elif self._use_gradient:  # on-device optimiser (rcg_actor_optimize)
            act, useq, bj, bi = self._eng.actor_optimize(iters=self.opt_iters, obs=obs, state_sys=xs)
            self._prev_opt = useq.astype(float)
        else:  # device-side candidate search (rcg_actor_search); the draw is keyed by (seed, env, decision number)
            self._eng.set_field(N.FIELD_STEP_IDX, np.full(self.B, self._search_draw, dtype=np.int32))
            self._search_draw += 1
            act, useq, bj, bi = self._eng.actor_search(K=self.n_candidates, rounds=self.rounds, obs=obs, state_sys=xs)
            self._prev_opt = useq.astype(float)
        self.last_J, self.last_idx = bj.astype(float), bi
        act = act.astype(float)
        return act if self._batched else act[0]

    def compute_action(self, t, observation):
        """Main method (rcognita/controllers.py:1429-1493).  (Build-specific: when the simulation step before this call ran as a
        fused loop step, the NEXT one is started here with the returned action held - `_speculate`.)"""
        action = self._compute_action(t, observation)
        if self._spec_ready:
            self._speculate(action)
        return action

    def _compute_action(self, t, observation):
        time_in_sample = t - self.ctrl_clock
        fz, self._fz = self._fz, None
        if time_in_sample >= self.sampling_time * (1 - self.clock_tol):  # new sample
            if (fz is not None and fz["tick"] and fz["t"] == t
                    and np.ascontiguousarray(self._b(observation, self.dim_output)).tobytes() == fz["obs"]
                    and np.ascontiguousarray(self._b(self.state_sys, self.dim_output)).tobytes() == fz["xs"]):
                return self._take_fused(t, observation, fz)
            self._fused_dirty = True  # the separate calls below use the handle's fields as scratch (and fz, if any, was for other inputs)
            self.ctrl_clock = t
            if self.mode in ("RQL", "SQL"):
                time_in_critic_period = t - self.critic_clock
                # push_vec of both buffers (controllers.py:1463-1464); one shared buffer per controller in
                # the un-batched case, [B, buffer_size, d] when batched
                a, y = np.asarray(self.action_curr, dtype=float), np.asarray(observation, dtype=float)
                if self._batched:
                    if self.action_buffer.ndim == 2:
                        self.action_buffer = np.broadcast_to(self.action_buffer, (self.B,) + self.action_buffer.shape).copy()
                        self.observation_buffer = np.broadcast_to(self.observation_buffer,
                                                                  (self.B,) + self.observation_buffer.shape).copy()
                    self.action_buffer = np.concatenate([self.action_buffer[:, 1:], a[:, None]], axis=1)
                    self.observation_buffer = np.concatenate([self.observation_buffer[:, 1:], y[:, None]], axis=1)
                else:
                    self.action_buffer = np.vstack([self.action_buffer[1:], a])
                    self.observation_buffer = np.vstack([self.observation_buffer[1:], y])
                if time_in_critic_period >= self.critic_period * (1 - self.clock_tol):
                    self.critic_clock = t
                    self.w_critic = self._critic_optimizer()
                    self.w_critic_prev = self.w_critic
                else:
                    self.w_critic = self.w_critic_prev
            action = self._actor_optimizer(observation)
            self.action_curr = action
            return action
        return self.action_curr


    def _take_fused(self, t, observation, fz):
        """compute_action's bookkeeping (controllers.py:1440-1493) around a decision rcg_loop_step already made from exactly
        these inputs: clocks, the host copies of the critic buffers and weights, the action."""
        self.ctrl_clock = t
        if self.mode in ("RQL", "SQL"):
            a, y = np.asarray(self.action_curr, dtype=float), np.asarray(observation, dtype=float)
            if self._batched:
                if self.action_buffer.ndim == 2:
                    self.action_buffer = np.broadcast_to(self.action_buffer, (self.B,) + self.action_buffer.shape).copy()
                    self.observation_buffer = np.broadcast_to(self.observation_buffer,
                                                              (self.B,) + self.observation_buffer.shape).copy()
                self.action_buffer = np.concatenate([self.action_buffer[:, 1:], a[:, None]], axis=1)
                self.observation_buffer = np.concatenate([self.observation_buffer[:, 1:], y[:, None]], axis=1)
            else:
                self.action_buffer = np.vstack([self.action_buffer[1:], a])
                self.observation_buffer = np.vstack([self.observation_buffer[1:], y])
            if fz["fit"]:
                self.critic_clock = t
                self.w_critic = fz["w"].copy() if self._batched else fz["w"][0].copy()
                self.w_critic_prev = self.w_critic
            else:
                self.w_critic = self.w_critic_prev
        self._prev_opt_on_device = True
        self._inb_action = None
        self.fused_decisions += 1
        self.last_J, self.last_idx = fz["J"], None
        action = fz["action"] if self._batched else fz["action"][0]
        self.action_curr = action
        # the optimiser's iterates live inside the box, so Simulator.sim_step need not clip this object (float64 handles only: a
        # float32 bound may lie a rounding outside the float64 one the host clips with)
        self._inb_action = action if self._dtype == "f64" else None
        return action


class _CtrlNominal:
    """Shared plumbing of the two nominal controllers (rcognita/controllers.py:1495-1956): sampled like the reference
    (``compute_action`` holds ``action_curr`` between samples), arithmetic in ``rcg_nominal_action``.  ``observation``
    may carry a leading batch axis; the device handle is created on first use for that batch size."""
    _sys_id = None
    _dy = 0

    def _init(self, ctrl_gain, ctrl_bnds, t0, sampling_time, pars, dtype, device):
        self.ctrl_gain = ctrl_gain
        self.ctrl_bnds = np.asarray(ctrl_bnds, dtype=float) if len(ctrl_bnds) else np.zeros((2, 2))
        self.ctrl_clock = t0
        self.sampling_time = sampling_time
        self.action_curr = np.zeros(2)
        self._pars, self._dtype, self._device = pars, dtype, device
        self._eng = None

    def _engine(self, n):
        if self._eng is None or self._eng.B != n:
            if self._eng is not None:
                self._eng.close()
            self._eng = Engine(EngineConfig(sys_id=self._sys_id, batch=n, dtype=self._dtype, device=self._device,
                                            pars=self._pars, ctrl_bnds=self.ctrl_bnds if self.ctrl_bnds.any() else None))
        return self._eng

    def _run(self, observation, clip, want_lyap=False):
        y = np.asarray(observation, dtype=float)
        batched = y.ndim == 2
        y2 = y.reshape(-1, self._dy)
        out = self._engine(y2.shape[0]).nominal_action(y2, self.ctrl_gain, ctrl_pars=self._pars or None,
                                                       clip=clip and bool(self.ctrl_bnds.any()), want_lyap=want_lyap)
        a, L = out if want_lyap else (out, None)
        a = a.astype(float)
        if want_lyap:
            L = L.astype(float)
            return L if batched else float(L[0])
        return a if batched else a[0]

    def reset(self, t0):
        """Resets controller for use in multi-episode simulation (controllers.py:1538-1544, 1771-1777)."""
        self.ctrl_clock = t0
        self.action_curr = np.zeros(2)

    def compute_action(self, t, observation):
        """Sampled, clipped action (controllers.py:1693-1731, 1906-1935)."""
        time_in_sample = t - self.ctrl_clock
        if time_in_sample >= self.sampling_time * (1 - 1e-9):  # new sample (same clock tolerance as CtrlOptPred)
            self.ctrl_clock = t
            self.action_curr = self._run(observation, clip=True)
        return self.action_curr

    def compute_action_vanila(self, observation):
        """Same without the internal clock and without the clip (controllers.py:1733-1748, 1937-1947)."""
        self.action_curr = self._run(observation, clip=False)
        return self.action_curr

    def compute_LF(self, observation):
        """Lyapunov function value (controllers.py:1750-1755, 1949-1955)."""
        return self._run(observation, clip=False, want_lyap=True)


class CtrlNominal3WRobot(_CtrlNominal):
    """Nominal controller of the 3-wheel robot with dynamic actuators, nonsmooth backstepping
    (rcognita/controllers.py:1495-1755).  theta* (SciPy trust-constr in the reference) comes from the build-defined
    local search (downhill walk from theta = 0 on a 64-point grid + golden section) of rcg_nominal.hpp."""
    _sys_id, _dy = N.SYS_3WROBOT, 5

    def __init__(self, m, I, ctrl_gain=10, ctrl_bnds=[], t0=0, sampling_time=0.1, dtype="f64", device=0):
        self.m, self.I = m, I
        self._init(ctrl_gain, ctrl_bnds, t0, sampling_time, [float(m), float(I)], dtype, device)


class CtrlNominal3WRobotNI(_CtrlNominal):
    """Nominal parking controller of the 3-wheel robot with static actuators, disassembled subgradients
    (rcognita/controllers.py:1757-1956)."""
    _sys_id, _dy = N.SYS_3WROBOT_NI, 3

    def __init__(self, ctrl_gain=10, ctrl_bnds=[], t0=0, sampling_time=0.1, dtype="f64", device=0):
        self._init(ctrl_gain, ctrl_bnds, t0, sampling_time, [], dtype, device)
