"""CPU oracle (numpy, float64) for the rcognita hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package ``rcognita_amd`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do,
and there only as the checker.

This is a restatement of the reference algorithm, not a copy: every function works on arrays with
arbitrary leading batch dimensions (``[..., ds]``) instead of the reference's one-env-at-a-time
1-D arrays.  Each function cites the reference lines it follows (paths relative to
``/root/reference``).

Parity status: the reference has no tests and no golden vectors, so this oracle is
pinned against outputs of the reference itself, generated in the build container by
``oracle/gen_fixtures.py`` and committed under ``tests/golden/`` (checked by
``tests/test_oracle_golden.py``).

Third-party arithmetic on the reference path that is NOT under /root/reference:
``scipy.integrate.RK45`` (setup.py pins ``scipy >= 1.5.0``; installed 1.15.3) and
``scipy.optimize.minimize(method='SLSQP')``.  The fixed-step RK4 and candidate-argmin used by the
GPU path are the build's own definitions; they are restated here so
the HIP kernels have a bit-level-comparable CPU twin, and are compared against reference RK45
trajectories (constant action) in the golden tests.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

# ----------------------------------------------------------------------------------------------
# System table (rcognita/systems.py:255-428)
# ----------------------------------------------------------------------------------------------
SYS_3WROBOT = 0
SYS_3WROBOT_NI = 1
SYS_2TANK = 2

SYS_NAMES = {SYS_3WROBOT: "3wrobot", SYS_3WROBOT_NI: "3wrobotNI", SYS_2TANK: "2tank"}
SYS_DIMS = {  # sys_id -> (dim_state, dim_input, n_pars); dim_output == dim_state for all three
    SYS_3WROBOT: (5, 2, 2),
    SYS_3WROBOT_NI: (3, 2, 0),
    SYS_2TANK: (2, 1, 5),
}

MODE_MPC, MODE_RQL, MODE_SQL = 0, 1, 2
MODE_IDS = {"MPC": MODE_MPC, "RQL": MODE_RQL, "SQL": MODE_SQL}

STAGE_QUADRATIC, STAGE_BIQUADRATIC = 0, 1
STAGE_IDS = {"quadratic": STAGE_QUADRATIC, "biquadratic": STAGE_BIQUADRATIC}

CRITIC_QUAD_LIN, CRITIC_QUADRATIC, CRITIC_QUAD_NOMIX, CRITIC_QUAD_MIX = 0, 1, 2, 3
CRITIC_IDS = {
    "quad-lin": CRITIC_QUAD_LIN,
    "quadratic": CRITIC_QUADRATIC,
    "quad-nomix": CRITIC_QUAD_NOMIX,
    "quad-mix": CRITIC_QUAD_MIX,
}


def dim_critic(critic_struct: int, dim_output: int, dim_input: int) -> int:
    """Number of critic weights (rcognita/controllers.py:1024-1039)."""
    n = dim_output + dim_input
    if critic_struct == CRITIC_QUAD_LIN:
        return n * (n + 1) // 2 + n
    if critic_struct == CRITIC_QUADRATIC:
        return n * (n + 1) // 2
    if critic_struct == CRITIC_QUAD_NOMIX:
        return n
    if critic_struct == CRITIC_QUAD_MIX:
        return dim_output + dim_output * dim_input + dim_input
    raise ValueError(critic_struct)


def critic_bounds(critic_struct: int, dc: int):
    """(Wmin, Wmax) (rcognita/controllers.py:1026-1039)."""
    if critic_struct in (CRITIC_QUAD_LIN, CRITIC_QUAD_MIX):
        return -1e3 * np.ones(dc), 1e3 * np.ones(dc)
    return np.zeros(dc), 1e3 * np.ones(dc)


@dataclass
class OracleCfg:
    """Everything the path needs besides per-env data.  Mirrors ``rcg_cfg`` in include/rcg.h."""

    sys_id: int
    n_actor: int = 5
    mode: int = MODE_MPC
    gamma: float = 1.0
    pred_step_size: float = 0.02
    dt_sim: float = 0.01
    substeps_per_tick: int = 1
    sampling_time: float = 0.01
    pars: Sequence[float] = ()
    ctrl_bnds: Optional[np.ndarray] = None  # [du, 2]
    R1: Optional[np.ndarray] = None  # [n, n]
    R2: Optional[np.ndarray] = None  # [n, n]
    stage_obj_struct: int = STAGE_QUADRATIC
    target: Optional[np.ndarray] = None  # None <=> reference's ``observation_target == []``
    critic_struct: int = CRITIC_QUAD_NOMIX
    n_critic: int = 4
    buffer_size: int = 10
    ref_lag: bool = False  # rollout starts from the previous sim step's state
    accum_every_substep: bool = False  # reference quirk, Appendix A-5
    critic_every_ticks: int = 1  # critic_period / sampling_time (controllers.py:1466)

    def __post_init__(self):
        ds, du, npar = SYS_DIMS[self.sys_id]
        self.pars = np.asarray(self.pars, dtype=np.float64)
        assert self.pars.shape[-1] == npar if npar else True
        if self.ctrl_bnds is not None:
            self.ctrl_bnds = np.asarray(self.ctrl_bnds, dtype=np.float64).reshape(du, 2)
        n = ds + du
        if self.R1 is None:
            self.R1 = np.eye(n)
        self.R1 = np.asarray(self.R1, dtype=np.float64).reshape(n, n)
        if self.R2 is not None:
            self.R2 = np.asarray(self.R2, dtype=np.float64).reshape(n, n)
        if self.target is not None:
            self.target = np.asarray(self.target, dtype=np.float64).reshape(ds)
        # Ncritic = min(Ncritic, buffer_size - 1)  (rcognita/controllers.py:1015)
        self.n_critic = int(min(self.n_critic, self.buffer_size - 1))

    @property
    def ds(self):
        return SYS_DIMS[self.sys_id][0]

    @property
    def du(self):
        return SYS_DIMS[self.sys_id][1]

    @property
    def dc(self):
        return dim_critic(self.critic_struct, self.ds, self.du)


# ----------------------------------------------------------------------------------------------
# Environment: right-hand sides
# ----------------------------------------------------------------------------------------------
def state_dyn(sys_id: int, state, action, pars):
    """``Sys*._state_dyn`` without disturbance.

    3wrobot   rcognita/systems.py:308-323   pars = (m, I)
    3wrobotNI rcognita/systems.py:370-382   no pars
    2tank     rcognita/systems.py:412-419   pars = (tau1, tau2, K1, K2, K3)

    ``state [..., ds]``, ``action [..., du]``, ``pars [..., np]`` (broadcastable) -> ``[..., ds]``.
    The action is used as given: no clipping here (the rollout in ``_actor_cost`` calls this
    directly, rcognita/controllers.py:1294).
    """
    state = np.asarray(state, dtype=np.float64)
    action = np.asarray(action, dtype=np.float64)
    pars = np.asarray(pars, dtype=np.float64)
    shape = np.broadcast_shapes(state.shape[:-1], action.shape[:-1], pars.shape[:-1])
    d = np.zeros(shape + (state.shape[-1],))
    if sys_id == SYS_3WROBOT:
        m, inertia = pars[..., 0], pars[..., 1]
        d[..., 0] = state[..., 3] * np.cos(state[..., 2])
        d[..., 1] = state[..., 3] * np.sin(state[..., 2])
        d[..., 2] = state[..., 4]
        d[..., 3] = 1 / m * action[..., 0]
        d[..., 4] = 1 / inertia * action[..., 1]
    elif sys_id == SYS_3WROBOT_NI:
        d[..., 0] = action[..., 0] * np.cos(state[..., 2])
        d[..., 1] = action[..., 0] * np.sin(state[..., 2])
        d[..., 2] = action[..., 1]
    elif sys_id == SYS_2TANK:
        tau1, tau2, K1, K2, K3 = (pars[..., i] for i in range(5))
        d[..., 0] = 1 / tau1 * (-state[..., 0] + K1 * action[..., 0])
        d[..., 1] = 1 / tau2 * (-state[..., 1] + K2 * state[..., 0] + K3 * state[..., 1] ** 2)
    else:
        raise ValueError(sys_id)
    return d


def clip_action(action, ctrl_bnds):
    """Box clip applied by ``closed_loop_rhs`` (rcognita/systems.py:241-243).

    The reference clips iff ``ctrl_bnds.any()``; an all-zero bounds array means "unconstrained".
    Returned as a value (the reference mutates the stored array in place, SURVEY §8b Ownership).
    """
    action = np.asarray(action, dtype=np.float64)
    if ctrl_bnds is None or not np.any(ctrl_bnds):
        return action
    return np.clip(action, ctrl_bnds[:, 0], ctrl_bnds[:, 1])


def closed_loop_rhs(sys_id: int, state, action, pars, ctrl_bnds):
    """``System.closed_loop_rhs`` for ``is_disturb = is_dyn_ctrl = 0`` (rcognita/systems.py:213-253).

    Returns ``(rhs, clipped_action)``.
    """
    a = clip_action(action, ctrl_bnds)
    return state_dyn(sys_id, state, a, pars), a


def rk4_step(sys_id: int, state, action, pars, ctrl_bnds, h: float):
    """One classical RK4 step of the closed loop under a zero-order-held action.

    Build-defined replacement of scipy RK45 inside ``Simulator.sim_step``
    (rcognita/simulator.py:156-168).  The clip is evaluated in every stage,
    as the reference's RHS does; with a held action it is the same value each time.
    """
    k1, _ = closed_loop_rhs(sys_id, state, action, pars, ctrl_bnds)
    k2, _ = closed_loop_rhs(sys_id, state + (0.5 * h) * k1, action, pars, ctrl_bnds)
    k3, _ = closed_loop_rhs(sys_id, state + (0.5 * h) * k2, action, pars, ctrl_bnds)
    k4, _ = closed_loop_rhs(sys_id, state + h * k3, action, pars, ctrl_bnds)
    return state + (h / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)


# ----------------------------------------------------------------------------------------------
# Controller: stage objective, critic, actor cost
# ----------------------------------------------------------------------------------------------
def _chi(obs, act, target):
    obs = np.asarray(obs, dtype=np.float64)
    act = np.asarray(act, dtype=np.float64)
    shape = np.broadcast_shapes(obs.shape[:-1], act.shape[:-1])
    obs = np.broadcast_to(obs, shape + obs.shape[-1:])
    act = np.broadcast_to(act, shape + act.shape[-1:])
    if target is None:
        return np.concatenate([obs, act], axis=-1)
    return np.concatenate([obs - target, act], axis=-1)


def stage_obj(obs, act, cfg: OracleCfg):
    """``CtrlOptPred.stage_obj`` (rcognita/controllers.py:1063-1084)."""
    chi = _chi(obs, act, cfg.target)
    quad = np.einsum("...i,ij,...j->...", chi, cfg.R1, chi)
    if cfg.stage_obj_struct == STAGE_QUADRATIC:
        return quad
    chi2 = chi**2
    return np.einsum("...i,ij,...j->...", chi2, cfg.R2, chi2) + quad


def critic_features(obs, act, cfg: OracleCfg):
    """Regressor of ``CtrlOptPred._critic`` (rcognita/controllers.py:1200-1212).

    ``uptria2vec`` is the row-major upper triangle incl. diagonal (rcognita/utilities.py:81-96).
    ``quad-mix`` uses the raw observation; the target is ignored there (controllers.py:1212).
    """
    chi = _chi(obs, act, cfg.target)
    n = chi.shape[-1]
    cs = cfg.critic_struct
    if cs in (CRITIC_QUAD_LIN, CRITIC_QUADRATIC):
        iu, ju = np.triu_indices(n)
        quad = chi[..., iu] * chi[..., ju]
        if cs == CRITIC_QUADRATIC:
            return quad
        return np.concatenate([quad, chi], axis=-1)
    if cs == CRITIC_QUAD_NOMIX:
        return chi * chi
    if cs == CRITIC_QUAD_MIX:
        obs = np.asarray(obs, dtype=np.float64)
        act = np.asarray(act, dtype=np.float64)
        shape = np.broadcast_shapes(obs.shape[:-1], act.shape[:-1])
        obs = np.broadcast_to(obs, shape + obs.shape[-1:])
        act = np.broadcast_to(act, shape + act.shape[-1:])
        kron = (obs[..., :, None] * act[..., None, :]).reshape(shape + (-1,))
        return np.concatenate([obs**2, kron, act**2], axis=-1)
    raise ValueError(cs)


def critic(obs, act, w, cfg: OracleCfg):
    """``CtrlOptPred._critic`` = ``w @ regressor`` (rcognita/controllers.py:1192-1214)."""
    return np.sum(np.asarray(w, dtype=np.float64) * critic_features(obs, act, cfg), axis=-1)


def actor_cost(action_sqn, obs, state_sys, cfg: OracleCfg, pars=None, w_critic=None):
    """``CtrlOptPred._actor_cost`` for ``is_est_model = 0`` (rcognita/controllers.py:1273-1328).

    ``action_sqn [..., N*du]`` or ``[..., N, du]``; ``obs [..., dy]``; ``state_sys [..., ds]``.
    Explicit-Euler rollout with ``pred_step_size`` from ``state_sys`` using the unclipped
    ``_state_dyn`` (controllers.py:1290-1296), ``observation_sqn[0] = obs``; then
    MPC  sum_{k<N} gamma^k rho(y_k,u_k)                       (controllers.py:1304-1306)
    RQL  sum_{k<N-1} gamma^k rho + Q_w(y_{N-1}, u_{N-1})        (controllers.py:1307-1310)
    SQL  sum_{k<N} Q_w(y_k,u_k), undiscounted                   (controllers.py:1311-1326)
    """
    N, du, ds = cfg.n_actor, cfg.du, cfg.ds
    pars = cfg.pars if pars is None else pars
    u = np.asarray(action_sqn, dtype=np.float64)
    if not (u.ndim >= 2 and u.shape[-2:] == (N, du)):
        # flat, step-major [N*du] as the reference passes it (controllers.py:1284)
        u = u.reshape(u.shape[:-1] + (N, du))
    obs = np.asarray(obs, dtype=np.float64)
    state = np.asarray(state_sys, dtype=np.float64)
    y = obs  # y_0
    J = 0.0
    g = 1.0
    for k in range(N):
        if k > 0:
            state = state + cfg.pred_step_size * state_dyn(cfg.sys_id, state, u[..., k - 1, :], pars)
            y = state  # sys_out is the identity for all three systems (systems.py:347-351,396-399,426-428)
        if cfg.mode == MODE_MPC:
            J = J + g * stage_obj(y, u[..., k, :], cfg)
        elif cfg.mode == MODE_RQL:
            if k < N - 1:
                J = J + g * stage_obj(y, u[..., k, :], cfg)
            else:
                J = J + critic(y, u[..., k, :], w_critic, cfg)
        elif cfg.mode == MODE_SQL:
            J = J + critic(y, u[..., k, :], w_critic, cfg)
        else:
            raise ValueError(cfg.mode)
        g = g * cfg.gamma
    return J


def critic_cost(w, w_prev, obs_buf, act_buf, cfg: OracleCfg):
    """``CtrlOptPred._critic_cost`` (rcognita/controllers.py:1216-1245).

    ``obs_buf [..., buffer_size, dy]``, ``act_buf [..., buffer_size, du]`` with the newest row
    LAST (``push_vec``, rcognita/utilities.py:78-79).  The reference indexes rows
    ``0 .. Ncritic-1`` - the OLDEST rows (controllers.py:1231-1234).
    """
    Jc = 0.0
    for k in range(cfg.n_critic - 1, 0, -1):
        y_prev, y_next = obs_buf[..., k - 1, :], obs_buf[..., k, :]
        u_prev, u_next = act_buf[..., k - 1, :], act_buf[..., k, :]
        c_prev = critic(y_prev, u_prev, w, cfg)
        c_next = critic(y_next, u_next, w_prev, cfg)
        e = c_prev - cfg.gamma * c_next - stage_obj(y_prev, u_prev, cfg)
        Jc = Jc + 0.5 * e**2
    return Jc


def push_vec(buf, vec):
    """FIFO push: drop row 0, append at the bottom (rcognita/utilities.py:78-79). Batched."""
    return np.concatenate([buf[..., 1:, :], np.asarray(vec)[..., None, :]], axis=-2)


def critic_td_system(w_prev, obs_buf, act_buf, cfg: OracleCfg):
    """The TD stack of ``_critic_cost`` written as ``Jc(w) = 1/2 |A w - b|^2``.

    Row r (r = 0 .. Ncritic-2) is the reference's term k = r + 1:
    ``A[r] = phi(y_{k-1}, u_{k-1})``, ``b[r] = gamma * w_prev . phi(y_k, u_k) + rho(y_{k-1}, u_{k-1})``.
    """
    rows = []
    rhs = []
    for k in range(1, cfg.n_critic):
        rows.append(critic_features(obs_buf[..., k - 1, :], act_buf[..., k - 1, :], cfg))
        rhs.append(
            cfg.gamma * critic(obs_buf[..., k, :], act_buf[..., k, :], w_prev, cfg)
            + stage_obj(obs_buf[..., k - 1, :], act_buf[..., k - 1, :], cfg)
        )
    return np.stack(rows, axis=-2), np.stack(rhs, axis=-1)


# ----------------------------------------------------------------------------------------------
# Build-defined pieces with a CPU twin: candidate argmin, control tick
# ----------------------------------------------------------------------------------------------
def argmin_first(J):
    """Deterministic argmin over the last axis: lower J wins, ties go to the
    lower candidate index, NaN counts as +inf.  Returns ``(best_J, best_idx int32)``."""
    Jc = np.where(np.isnan(J), np.inf, J)
    idx = np.argmin(Jc, axis=-1).astype(np.int32)
    return np.take_along_axis(Jc, idx[..., None].astype(np.int64), axis=-1)[..., 0], idx


def grid_candidates(cfg: OracleCfg, K: int, action_prev=None):
    """Build-defined generated candidate set: constant-over-horizon sequences.

    du = 2: ``g x g`` level grid with ``g = isqrt(K)``, candidate ``k -> (i, j) = (k // g, k % g)``;
    du = 1: ``K`` levels.  Level ``i`` of an input is ``lo + i * (hi - lo) / (g - 1)``.
    Returns ``[K, N, du]``.
    """
    du, N = cfg.du, cfg.n_actor
    lo, hi = cfg.ctrl_bnds[:, 0], cfg.ctrl_bnds[:, 1]
    if du == 1:
        g = K
        lev = lo[0] + np.arange(g) * ((hi[0] - lo[0]) / max(g - 1, 1))
        first = lev[:, None]
    else:
        g = int(np.floor(np.sqrt(K) + 1e-9))
        assert g * g == K, "du=2 grid needs a square K"
        i, j = np.divmod(np.arange(K), g)
        first = np.stack(
            [lo[0] + i * ((hi[0] - lo[0]) / max(g - 1, 1)), lo[1] + j * ((hi[1] - lo[1]) / max(g - 1, 1))],
            axis=-1,
        )
    return np.broadcast_to(first[:, None, :], (K, N, du)).copy()


@dataclass
class EnvBatch:
    """Per-env data of a batch of closed loops (host twin of the device handle's SoA tensors)."""

    state: np.ndarray  # [B, ds]
    action: np.ndarray  # [B, du]   action currently applied (ZOH)
    accum: np.ndarray  # [B]
    step_idx: np.ndarray  # [B] int32  control ticks done in the current episode
    episode_idx: np.ndarray  # [B] int32
    pars: np.ndarray  # [B, np] or [np]
    state_prev: Optional[np.ndarray] = None  # state before the last substep (ref_lag)
    best_J: Optional[np.ndarray] = None
    best_idx: Optional[np.ndarray] = None
    w_critic: Optional[np.ndarray] = None  # [B, dc]
    w_prev: Optional[np.ndarray] = None
    obs_buf: Optional[np.ndarray] = None  # [B, buffer_size, dy]
    act_buf: Optional[np.ndarray] = None  # [B, buffer_size, du]
    tick_count: int = 0  # control ticks issued (drives the critic period)


def new_batch(cfg: OracleCfg, state0, action0=None, pars=None) -> EnvBatch:
    state0 = np.array(state0, dtype=np.float64).reshape(-1, cfg.ds)
    B = state0.shape[0]
    if action0 is None:
        # action_curr = action_min / 10 when action_init == [] (rcognita/controllers.py:973-975)
        action0 = np.broadcast_to(cfg.ctrl_bnds[:, 0] / 10.0, (B, cfg.du))
    dc = cfg.dc
    return EnvBatch(
        state=state0,
        action=np.array(np.broadcast_to(action0, (B, cfg.du)), dtype=np.float64),
        accum=np.zeros(B),
        step_idx=np.zeros(B, dtype=np.int32),
        episode_idx=np.zeros(B, dtype=np.int32),
        pars=np.asarray(cfg.pars if pars is None else pars, dtype=np.float64),
        state_prev=state0.copy(),
        w_critic=np.ones((B, dc)),
        w_prev=np.ones((B, dc)),
        obs_buf=np.zeros((B, cfg.buffer_size, cfg.ds)),
        act_buf=np.zeros((B, cfg.buffer_size, cfg.du)),
    )


def episode_reset(cfg: OracleCfg, env: EnvBatch, state_init, action_init=None):
    """Episode boundary, twin of rcg_episode_reset (``Simulator.reset`` simulator.py:197-204 as intended +
    ``CtrlOptPred.reset`` controllers.py:1046-1054): returns := accum, accum := 0, state := state_init, action :=
    action_init (``action_min / 10`` by default), step_idx := 0, episode_idx += 1; the critic weights and both buffers
    are RETAINED (the reference's reset touches only the clock and ``action_curr``).  Build-defined: the controller's
    tick counter - and with it the critic period - restarts with the episode.  Returns the episode's returns."""
    returns = env.accum.copy()
    B = env.state.shape[0]
    env.accum = np.zeros(B)
    env.state = np.array(state_init, dtype=np.float64).reshape(B, cfg.ds)
    env.state_prev = env.state.copy()
    a0 = cfg.ctrl_bnds[:, 0] / 10.0 if action_init is None else np.asarray(action_init, dtype=np.float64)
    env.action = np.array(np.broadcast_to(a0, (B, cfg.du)), dtype=np.float64)
    env.step_idx = np.zeros(B, dtype=np.int32)
    env.episode_idx = env.episode_idx + np.int32(1)
    env.tick_count = 0
    return returns


def sim_substeps(cfg: OracleCfg, env: EnvBatch, n_substeps: int):
    """``n_substeps`` RK4 steps of size ``dt_sim`` under the held action (row 10)."""
    for _ in range(n_substeps):
        env.state_prev = env.state
        env.state = rk4_step(cfg.sys_id, env.state, env.action, env.pars, cfg.ctrl_bnds, cfg.dt_sim)
        if cfg.accum_every_substep:
            env.accum = env.accum + stage_obj(env.state, env.action, cfg) * cfg.sampling_time


def control_tick(cfg: OracleCfg, env: EnvBatch, cand, force_idx=None):
    """One env.control-step (unit U2 of SURVEY §8d) for every env of the batch.

    ``force_idx [B]`` (int, entries < 0 = not forced): take this candidate instead of the argmin for those envs -
    used by the parity checker to follow a float32 run through an exact near-tie (``oracle/parity.py``).

    Order follows the reference loop (presets/main_3wrobot.py:419-429):
      1. sim_step        : ``substeps_per_tick`` RK4 substeps with the held, clipped action
      2. compute_action  : evaluate ``_actor_cost`` for the K candidates ``cand [B, K, N, du]`` (or
                           ``[K, N, du]`` shared) from the new observation, take the argmin, the new
                           action is the first ``du`` entries of the winner (controllers.py:1427)
      3. receive_action  : the action is held until the next tick
      4. upd_accum_obj   : ``accum += rho(obs, action) * sampling_time`` (controllers.py:1086-1093)
      5. ``step_idx += 1`` (int32)
    """
    sim_substeps(cfg, env, cfg.substeps_per_tick)
    if cfg.mode != MODE_MPC:
        # buffer push + critic refit every `critic_every_ticks` ticks (controllers.py:1458-1477)
        # critic_clock starts at t0 and tick j happens at t0 + (j + 1) dt, so `t - critic_clock >= critic_period`
        # (controllers.py:1466) first holds on tick every - 1, then every `every` ticks
        every = max(int(cfg.critic_every_ticks), 1)
        critic_update(cfg, env, do_fit=((env.tick_count + 1) % every) == 0)
    env.tick_count += 1
    obs = env.state
    state_sys = env.state_prev if cfg.ref_lag else env.state
    cand = np.asarray(cand, dtype=np.float64)
    if cand.ndim == 3:
        cand = np.broadcast_to(cand[None], (obs.shape[0],) + cand.shape)
    J = actor_cost(
        cand,
        obs[:, None, :],
        state_sys[:, None, :],
        cfg,
        pars=env.pars[:, None, :] if env.pars.ndim == 2 else env.pars,
        w_critic=None if env.w_critic is None else env.w_critic[:, None, :],
    )
    best_J, best_idx = argmin_first(J)
    if force_idx is not None:
        f = np.asarray(force_idx)
        best_idx = np.where(f >= 0, f, best_idx).astype(np.int32)
        best_J = np.take_along_axis(np.where(np.isnan(J), np.inf, J), best_idx[:, None].astype(np.int64), axis=1)[:, 0]
    env.best_J, env.best_idx = best_J, best_idx
    env.action = np.take_along_axis(cand[:, :, 0, :], best_idx[:, None, None].astype(np.int64), axis=1)[:, 0, :]
    if not cfg.accum_every_substep:
        env.accum = env.accum + stage_obj(obs, env.action, cfg) * cfg.sampling_time
    env.step_idx = env.step_idx + np.int32(1)
    return J


# ----------------------------------------------------------------------------------------------
# Build-defined actor optimiser (replacement of the SLSQP call in
# CtrlOptPred._actor_optimizer, controllers.py:1330-1427): adjoint gradient + projected line search
# ----------------------------------------------------------------------------------------------
OPT_NALPHA = 16  # step lengths tried per env and iteration = one row of 16 lanes (four envs share a wave per pass)


def state_jac_T(sys_id, x, u, pars, lam):
    """(A^T lam, B^T lam) with A = d f/d x, B = d f/d u of ``_state_dyn`` at (x, u); single env."""
    if sys_id == SYS_3WROBOT:
        s, c = np.sin(x[2]), np.cos(x[2])
        ax = np.array([0.0, 0.0, x[3] * (lam[1] * c - lam[0] * s), lam[0] * c + lam[1] * s, lam[2]])
        bu = np.array([lam[3] * (1 / pars[0]), lam[4] * (1 / pars[1])])
    elif sys_id == SYS_3WROBOT_NI:
        s, c = np.sin(x[2]), np.cos(x[2])
        ax = np.array([0.0, 0.0, u[0] * (lam[1] * c - lam[0] * s)])
        bu = np.array([lam[0] * c + lam[1] * s, lam[2]])
    else:
        tau1, tau2, K1, K2, K3 = pars
        ax = np.array([-lam[0] * (1 / tau1) + lam[1] * (1 / tau2) * K2, lam[1] * (1 / tau2) * (-1 + 2 * K3 * x[1])])
        bu = np.array([lam[0] * (1 / tau1) * K1])
    return ax, bu


def stage_obj_grad(chi, cfg: OracleCfg):
    """d rho / d chi of ``stage_obj`` (controllers.py:1076-1082) for ONE point ``chi [n]``:
    quadratic ``chi R1 chi`` -> ``(R1 + R1^T) chi``; biquadratic adds ``2 chi * ((R2 + R2^T) chi^2)``."""
    g = (cfg.R1 + cfg.R1.T) @ chi
    if cfg.stage_obj_struct == STAGE_BIQUADRATIC:
        g = g + 2.0 * chi * ((cfg.R2 + cfg.R2.T) @ (chi * chi))
    return g


def critic_grad(y, u, w, cfg: OracleCfg):
    """(d Q / d y, d Q / d u) of ``_critic`` = ``w . phi`` (controllers.py:1192-1214) for ONE point, every structure.
    chi = [y - target, u], so d chi / d y = I; quad-mix works on the RAW observation (controllers.py:1212)."""
    ds, du = cfg.ds, cfg.du
    n = ds + du
    cs = cfg.critic_struct
    if cs == CRITIC_QUAD_MIX:
        W = np.asarray(w[ds:ds + ds * du]).reshape(ds, du)
        gy = 2.0 * w[:ds] * y + W @ u
        gu = W.T @ y + 2.0 * w[ds + ds * du:] * u
        return gy, gu
    chi = np.concatenate([y if cfg.target is None else y - cfg.target, u])
    if cs == CRITIC_QUAD_NOMIX:
        g = 2.0 * w * chi
    else:
        iu, ju = np.triu_indices(n)
        M = np.zeros((n, n))
        M[iu, ju] = w[:len(iu)]  # uptria2vec order (utilities.py:81-96)
        g = (M + M.T) @ chi      # the diagonal appears in both -> 2 w_pp chi_p
        if cs == CRITIC_QUAD_LIN:
            g = g + w[len(iu):]
    return g[:ds], g[ds:]


def actor_grad(u, obs, state_sys, cfg: OracleCfg, pars=None, w_critic=None):
    """Gradient of ``_actor_cost`` (every mode, every stage / critic structure) w.r.t. the whole action sequence
    ``u [N, du]`` by one forward Euler rollout and one reverse (adjoint) sweep.  Returns ``(J, g [N, du])``.

    J = sum_k c_k(y_k, u_k) with y_0 = obs (not a function of u), y_k = x_k, x_{k+1} = x_k + h f(x_k, u_k):
      MPC c_k = gamma^k rho;  RQL c_k = gamma^k rho (k < N-1), c_{N-1} = Q_w;  SQL c_k = Q_w  (controllers.py:1304-1326).
    lam_k = d J / d x_k = d c_k / d y + (I + h A_k^T) lam_{k+1};  g_k = d c_k / d u + h B_k^T lam_{k+1}."""
    N, du, ds, h = cfg.n_actor, cfg.du, cfg.ds, cfg.pred_step_size
    pars = cfg.pars if pars is None else pars
    tgt = np.zeros(ds) if cfg.target is None else cfg.target
    X = [np.asarray(state_sys, dtype=np.float64)]
    for k in range(1, N):
        X.append(X[-1] + h * state_dyn(cfg.sys_id, X[-1], u[k - 1], pars))
    Y = [np.asarray(obs, dtype=np.float64)] + X[1:]
    gks, gk = [], 1.0
    for k in range(N):
        gks.append(gk)
        gk *= cfg.gamma

    def is_critic_step(k):
        return cfg.mode == MODE_SQL or (cfg.mode == MODE_RQL and k == N - 1)

    def step_grad(k):  # (d c_k / d y, d c_k / d u)
        if is_critic_step(k):
            return critic_grad(Y[k], u[k], np.asarray(w_critic, dtype=np.float64), cfg)
        gc = gks[k] * stage_obj_grad(np.concatenate([Y[k] - tgt, u[k]]), cfg)
        return gc[:ds], gc[ds:]

    J = float(actor_cost(u, obs, state_sys, cfg, pars=pars, w_critic=w_critic))
    g = np.zeros((N, du))
    lam = np.zeros(ds)  # d J / d x_{k+1}
    for k in range(N - 1, -1, -1):
        gy, gu = step_grad(k)
        g[k] = gu
        if k < N - 1:
            ax, bu = state_jac_T(cfg.sys_id, X[k], u[k], pars, lam)
            g[k] = g[k] + h * bu
            lam_k = lam + h * ax
        else:
            lam_k = np.zeros(ds)
        if k >= 1:  # y_0 is the observation, not a function of the actions
            lam_k = lam_k + gy
        lam = lam_k
    return J, g


OPT_MEMORY = None  # None: the library's default (opt_memory_default); an int: that many curvature pairs (0: steepest descent)


def opt_memory_default(cfg: "OracleCfg") -> int:
    """Curvature pairs rcg_actor_optimize keeps by default (rcg_handle.hpp::opt_memory_of): 4 for the critic modes and the
    non-diagonal stage costs, 0 for MPC with a diagonal quadratic stage cost."""
    diag = np.count_nonzero(cfg.R1 - np.diag(np.diag(cfg.R1))) == 0 and (
        cfg.stage_obj_struct == STAGE_QUADRATIC or np.count_nonzero(cfg.R2 - np.diag(np.diag(cfg.R2))) == 0)
    generic = not (cfg.mode == MODE_MPC and cfg.stage_obj_struct == STAGE_QUADRATIC and diag)
    return 4 if generic else 0



def _quad_sum(R, term, free):
    """sum_i term(i) over the (free) coordinates, associated as four lanes of a quad do it: p_q = the terms with i = q mod 4 in
    index order, total = (p0 + p1) + (p2 + p3)."""
    p = [0.0, 0.0, 0.0, 0.0]
    for i in range(R):
        if free is None or free[i]:
            p[i & 3] += term(i)
    return (p[0] + p[1]) + (p[2] + p[3])


def actor_optimize_single(cfg: OracleCfg, obs, state_sys, u_init, iters, pars=None, w_critic=None, memory=OPT_MEMORY, ftol=0.0):
    """Projected limited-memory quasi-Newton descent with a 16-way line search, every mode and cost structure.

    Per iteration, with g = grad J(u) (adjoint sweep) and the box [lo, hi] of width w:
      * free set: coordinate i is held iff it sits on a bound and -g_i points out of the box;
      * the pair (s, y) = (u - u_prev, g - g_prev) of the last ACCEPTED step joins a ring of ``memory`` pairs;
      * direction d = H g on the free coordinates by the L-BFGS two-loop recursion over the stored pairs restricted
        to the free set (a pair with s.y <= 1e-12 |s| |y| there is skipped), initial metric H0 = scale * diag(w^2) with
        scale = s.y / (y . w^2 y) of the newest pair (1 if that is not positive).  Without pairs, or if d is not a
        descent direction (d . g <= 0: the memory is dropped), d = w^2 g on the free set - box-scaled steepest descent;
      * 16 trial points clip(u - alpha_l d): quasi-Newton alpha_l = 2^(2 - l) (the unit step is l = 2), steepest
        descent alpha_l = 4^(1 - l) / max_i |d_i / w_i| (four box widths down to 2^-28);  lower J wins, then lower l;
      * the best trial replaces u if it lowers J; otherwise a quasi-Newton iteration drops its memory and the next one
        retries with steepest descent, a steepest-descent iteration ends the search;
      * ``ftol`` (rcg_set_optimizer_tol; the reference hands SLSQP ``tol=1e-7``, controllers.py:1396): an accepted step that
        lowered J by no more than ``ftol`` is the last one (0: no such test).
    Deterministic, no finite differences.  Every sum over coordinates is associated as k_actor_opt (rcg_actor_opt.hpp) forms
    it with four lanes per env (_quad_sum).
    Returns ``(u [N, du], J, accepted steps)``.  Round 3's optimiser was the ``memory = 0`` case (with the steepest-descent
    step scaled over all coordinates): it stalls 3-14 % above SLSQP on the critic-mode fixtures F8c, whose terminal
    action has 1e4 times the curvature of the others; four pairs close that to < 0.5 % in 20 iterations."""
    lo, hi = cfg.ctrl_bnds[:, 0], cfg.ctrl_bnds[:, 1]
    N, du = cfg.n_actor, cfg.du
    R = N * du
    wbox = np.tile(hi - lo, N)
    lo_f, hi_f = np.tile(lo, N), np.tile(hi, N)
    h0 = wbox * wbox
    M = opt_memory_default(cfg) if memory is None else int(memory)
    u = np.array(u_init, dtype=np.float64).reshape(N, du)
    J = float(actor_cost(u, obs, state_sys, cfg, pars=pars, w_critic=w_critic))
    S = np.zeros((max(M, 1), R))
    Y = np.zeros((max(M, 1), R))
    head, n_pairs, pending = 0, 0, False
    used = 0
    for _ in range(int(iters)):
        _, g2 = actor_grad(u, obs, state_sys, cfg, pars=pars, w_critic=w_critic)
        g = g2.reshape(R)
        uf = u.reshape(R)
        if pending:  # finish the pair of the last accepted step: Y[head] holds the gradient at its start point
            for i in range(R):
                Y[head, i] = g[i] - Y[head, i]
            head = (head + 1) % M
            n_pairs = min(n_pairs + 1, M)
            pending = False
        free = np.zeros(R, dtype=bool)
        for i in range(R):
            free[i] = not ((uf[i] <= lo_f[i] and g[i] > 0.0) or (uf[i] >= hi_f[i] and g[i] < 0.0))
        d = np.zeros(R)
        quasi = n_pairs > 0
        if quasi:
            q = np.where(free, g, 0.0)
            a_t, sy_t, ok_t = np.zeros(M), np.zeros(M), np.zeros(M, dtype=bool)
            scale = 1.0
            for t in range(n_pairs):  # newest -> oldest
                j = (head - 1 - t) % M
                # sums over the coordinates as k_actor_opt's four lanes per env form them (round 5): lane i mod 4 adds its
                # coordinates in index order, the quad adds (p0 + p1) + (p2 + p3)
                sy = _quad_sum(R, lambda i: S[j, i] * Y[j, i], free)
                ss = _quad_sum(R, lambda i: S[j, i] * S[j, i], free)
                yy = _quad_sum(R, lambda i: Y[j, i] * Y[j, i], free)
                sq = _quad_sum(R, lambda i: S[j, i] * q[i], free)
                yhy = _quad_sum(R, lambda i: (Y[j, i] * h0[i]) * Y[j, i], free)
                ok = sy > 0.0 and sy * sy > 1e-24 * (ss * yy)
                if t == 0 and ok and yhy > 0.0:
                    scale = sy / yhy
                ok_t[t], sy_t[t] = ok, sy
                if ok:
                    a_t[t] = sq / sy
                    for i in range(R):
                        if free[i]:
                            q[i] = q[i] - a_t[t] * Y[j, i]
            r = np.zeros(R)
            for i in range(R):
                r[i] = (scale * h0[i]) * q[i]
            for t in range(n_pairs - 1, -1, -1):  # oldest -> newest
                if not ok_t[t]:
                    continue
                j = (head - 1 - t) % M
                yr = _quad_sum(R, lambda i: Y[j, i] * r[i], free)
                c = a_t[t] - yr / sy_t[t]
                for i in range(R):
                    if free[i]:
                        r[i] = r[i] + S[j, i] * c
            dg = _quad_sum(R, lambda i: r[i] * g[i], None)
            if dg > 0.0 and np.isfinite(dg):
                d = r
            else:
                quasi = False
                n_pairs = 0
        if not quasi:
            for i in range(R):
                d[i] = g[i] * h0[i] if free[i] else 0.0
        gn = 0.0
        for i in range(R):
            m_ = abs(d[i]) / wbox[i]
            gn = m_ if m_ > gn else gn
        if not (gn > 0.0) or not np.isfinite(gn):
            break
        if quasi:
            alphas = np.exp2(2.0 - np.arange(OPT_NALPHA))
        else:
            alphas = (1.0 / gn) * np.exp2(2.0 - 2.0 * np.arange(OPT_NALPHA))
        cand = np.minimum(np.maximum(uf[None] - alphas[:, None] * d[None], lo_f), hi_f).reshape(OPT_NALPHA, N, du)
        Js = actor_cost(cand, obs, state_sys, cfg, pars=pars, w_critic=w_critic)
        bj, bi = argmin_first(Js[None])
        if not (bj[0] < J):
            if quasi:  # drop the memory, retry from the same point with steepest descent
                n_pairs = 0
                continue
            break
        u_new = cand[int(bi[0])]
        if M > 0:
            S[head] = u_new.reshape(R) - uf
            Y[head] = g
            pending = True
        gain = J - float(bj[0])
        u, J = u_new, float(bj[0])
        used += 1
        if gain <= ftol:
            break
    return u, J, used


def actor_optimize(cfg: OracleCfg, obs, state_sys, u_init, iters, pars=None, w_critic=None, memory=OPT_MEMORY, ftol=0.0):
    """Batched wrapper: ``obs/state_sys [B, ds]``, ``u_init [B, N, du]`` or ``[N, du]`` ->
    ``(u [B, N, du], J [B], iterations [B] int32)``."""
    obs = np.asarray(obs, dtype=np.float64).reshape(-1, cfg.ds)
    xs = np.asarray(state_sys, dtype=np.float64).reshape(-1, cfg.ds)
    B = obs.shape[0]
    u0 = np.broadcast_to(np.asarray(u_init, dtype=np.float64).reshape(-1, cfg.n_actor, cfg.du)
                         if np.ndim(u_init) == 3 else np.asarray(u_init, dtype=np.float64)[None], (B, cfg.n_actor, cfg.du))
    U, Js, its = [], [], []
    for b in range(B):
        p = None if pars is None else np.asarray(pars)[b]
        wb = None if w_critic is None else np.asarray(w_critic, dtype=np.float64).reshape(-1, cfg.dc)[b if np.ndim(w_critic) == 2 else 0]
        u, J, n = actor_optimize_single(cfg, obs[b], xs[b], u0[b], iters, pars=p, w_critic=wb, memory=memory, ftol=ftol)
        U.append(u)
        Js.append(J)
        its.append(n)
    return np.stack(U), np.array(Js), np.array(its, dtype=np.int32)


def action_sqn_init(cfg: OracleCfg, action_init=None):
    """``action_sqn_init`` of the reference: ``action_min / 10`` (or ``action_init``) tiled over the horizon
    (controllers.py:973-978)."""
    a = cfg.ctrl_bnds[:, 0] / 10.0 if action_init is None else np.asarray(action_init, dtype=np.float64)
    return np.tile(a, (cfg.n_actor, 1))


# ----------------------------------------------------------------------------------------------
# Build-defined critic fit (replacement of CtrlOptPred._critic_optimizer, controllers.py:1248-1271)
# ----------------------------------------------------------------------------------------------
FIT_MU_REL = 1e-8     # Tikhonov weight relative to trace(A A^T)/m
FIT_KKT_TOL = 1e-10   # a bound variable is released only if its multiplier has the wrong sign beyond this (relative)


def fit_max_iters(dc: int) -> int:
    """Iteration cap of the active-set loop (one m x m Cholesky solve per iteration)."""
    return 3 * dc + 10


def _chol_solve(H, rhs, floor=0.0):
    """Solve the SPD system ``H x = rhs`` by an unpivoted root-free Cholesky factorisation ``H = L D L^T`` (unit lower
    ``L``), written out in the same operation order as the HIP kernel (rcg_critic_fit.hpp) so both take the same steps:
    one reciprocal per pivot, no square root and no other division (the kernel is bound by float64 instruction issue and
    a divide or square root costs ten to twenty of them; round 1 used L L^T with 3 m of each per solve).  Pivots are
    floored at ``floor`` (only reached when rounding makes a pivot of a near-singular H non-positive)."""
    m = H.shape[0]
    L = np.array(H, dtype=np.float64)
    d = np.zeros(m)
    r = np.zeros(m)
    for j in range(m):
        dj = L[j, j]
        for k in range(j):
            dj -= (L[j, k] * L[j, k]) * d[k]
        if not dj > floor:
            dj = floor
        d[j] = dj
        r[j] = 1.0 / dj
        for i in range(j + 1, m):
            s = L[i, j]
            for k in range(j):
                s -= (L[i, k] * L[j, k]) * d[k]
            L[i, j] = s * r[j]
    x = np.zeros(m)
    for i in range(m):  # L y = rhs
        s = rhs[i]
        for k in range(i):
            s -= L[i, k] * x[k]
        x[i] = s
    for i in range(m):  # D z = y
        x[i] = x[i] * r[i]
    for i in range(m - 1, -1, -1):  # L^T x = z
        s = x[i]
        for k in range(i + 1, m):
            s -= L[k, i] * x[k]
        x[i] = s
    return x


def critic_fit_single(A, b, w0, lo, hi, stats=None, w_start=None):
    """Bounded least squares of the TD stack for ONE env.

    The reference minimises ``Jc(w) = 1/2 |A w - b|^2`` over the box ``[Wmin, Wmax]`` with SLSQP started
    at ``w_critic_init`` (= ones, never updated: controllers.py:1041-1042, 1264, 1474).  SLSQP's iterates
    are path dependent (and on the 3wrobot features it returns its start point unchanged, see
    tests/golden/F8_slsqp_critic_3wrobot.npz), so the build defines the fit as the unique minimiser of

        1/2 |A w - b|^2 + mu/2 |w - w_init|^2   s.t.  lo <= w <= hi,   mu = FIT_MU_REL * trace(A A^T) / m

    (for mu -> 0 and inactive bounds: the least-squares solution closest to w_init, which is also where
    a quasi-Newton method started at w_init with an identity Hessian converges).  It is computed by a
    primal active-set method (bounded-variable least squares in the manner of Stark & Parker): every
    iteration solves the equality-constrained problem on the free variables exactly through the m x m
    system ``(A_F A_F^T + mu I) lam = b - A_B w_B - A_F w0_F``, ``z_F = w0_F + A_F^T lam``; if ``z``
    leaves the box the iterate moves towards it until the first bound is hit and that variable is fixed,
    otherwise ``w_F = z`` and the bound variable whose multiplier has the most wrong sign is released
    (none: optimal).  The objective never increases, the iterate is always feasible, and a released
    variable that is pushed straight back (rounding on a degenerate stack) is kept fixed until the
    iterate moves.  Safeguard: ``w_init`` is returned if the result does not have ``Jc <= Jc(w_init)``
    (non-finite buffers).  The HIP kernel k_critic_fit mirrors this statement by statement, in float64
    whatever the handle's dtype.  ``stats`` (a list) receives the number of iterations used.

    ``w_start`` (default ``w_init``): the feasible point the active-set walk starts from.  The minimiser is unique (the
    problem is strictly convex), so the start only decides how many iterations the walk takes: the kernel starts from
    the previous fit's weights, whose active set is usually already the optimal one (1-2 iterations instead of one per
    bound the cold walk has to find).
    """
    A = np.asarray(A, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    w0 = np.asarray(w0, dtype=np.float64)
    lo = np.asarray(lo, dtype=np.float64)
    hi = np.asarray(hi, dtype=np.float64)
    m, dc = A.shape
    tr = 0.0
    for r in range(m):
        for i in range(dc):
            tr += A[r, i] * A[r, i]
    mu = FIT_MU_REL * (tr / m)
    if not mu > 1e-30:
        mu = 1e-30

    w = np.minimum(np.maximum(w0 if w_start is None else np.asarray(w_start, dtype=np.float64), lo), hi)
    free = (w > lo) & (w < hi)
    at_hi = (~free) & (w >= hi)
    blocked = np.zeros(dc, dtype=bool)
    last_freed = -1
    z = np.zeros(dc)
    iters = 0
    for _ in range(fit_max_iters(dc)):
        iters += 1
        rhs = np.zeros(m)
        M = np.zeros((m, m))
        for r in range(m):
            s = b[r]
            for i in range(dc):
                s -= A[r, i] * (w0[i] if free[i] else w[i])
            rhs[r] = s
            for q in range(r + 1):
                acc = 0.0
                for i in range(dc):
                    if free[i]:
                        acc += A[r, i] * A[q, i]
                M[r, q] = M[q, r] = acc + (mu if r == q else 0.0)
        lam = _chol_solve(M, rhs, floor=mu * 1e-6)
        # ratio test: the step to the bound z_i crosses is a_i = n_i / d_i, n_i = |bound_i - w_i| <= d_i = |z_i - w_i|; the
        # smallest one (first index on ties) is found by cross-multiplication, alpha is the ONE division (as the kernel)
        nb, db, jmin = 2.0, 1.0, -1
        for i in range(dc):
            if not free[i]:
                continue
            c = 0.0
            for r in range(m):
                c += A[r, i] * lam[r]
            z[i] = w0[i] + c
            if z[i] < lo[i] or z[i] > hi[i]:
                ni = abs((lo[i] if z[i] < lo[i] else hi[i]) - w[i])
                di = abs(z[i] - w[i])
                if ni * db < nb * di:
                    nb, db, jmin = ni, di, i
        if jmin >= 0:  # move towards z until the first bound, fix that variable
            alpha = nb / db
            if not alpha > 0.0:
                alpha = 0.0
            for i in range(dc):
                if free[i]:
                    w[i] = min(max(w[i] + alpha * (z[i] - w[i]), lo[i]), hi[i])
            at_hi[jmin] = z[jmin] > hi[jmin]
            w[jmin] = hi[jmin] if at_hi[jmin] else lo[jmin]
            free[jmin] = False
            if alpha > 0.0:
                blocked[:] = False
            elif jmin == last_freed:
                blocked[jmin] = True
            last_freed = -1
            continue
        res = -b.copy()
        for i in range(dc):
            if free[i]:
                w[i] = z[i]
            for r in range(m):
                res[r] += A[r, i] * w[i]
        best, best_score = -1, 0.0
        for i in range(dc):
            if free[i] or blocked[i]:
                continue
            g = mu * (w[i] - w0[i])
            scale = abs(g)
            for r in range(m):
                t = A[r, i] * res[r]
                g += t
                scale += abs(t)
            score = g if at_hi[i] else -g
            if score > FIT_KKT_TOL * scale and score > best_score:
                best, best_score = i, score
        if best < 0:
            break
        free[best] = True
        last_freed = best
    if stats is not None:
        stats.append(iters)

    def primal(v):
        r = A @ v - b
        return 0.5 * float(r @ r)

    wi = np.minimum(np.maximum(w0, lo), hi)
    return w if primal(w) <= primal(wi) else wi


def critic_fit(cfg: OracleCfg, w_prev, obs_buf, act_buf, w_init=None, w_start=None, stats=None):
    """Batched wrapper: ``w_prev [B, dc]``, buffers ``[B, buffer_size, d]`` -> fitted ``w [B, dc]``; ``w_start [B, dc]``
    as in :func:`critic_fit_single`."""
    A, b = critic_td_system(w_prev, obs_buf, act_buf, cfg)
    B = A.shape[0]
    lo, hi = critic_bounds(cfg.critic_struct, cfg.dc)
    w0 = np.ones(cfg.dc) if w_init is None else np.asarray(w_init, dtype=np.float64)
    return np.stack([critic_fit_single(A[i], b[i], w0, lo, hi, stats=stats,
                                       w_start=None if w_start is None else np.asarray(w_start)[i]) for i in range(B)])


def control_tick_opt(cfg: OracleCfg, env: EnvBatch, iters: int, warm_start: bool = False, action_init=None,
                     memory=OPT_MEMORY):
    """``control_tick`` with :func:`actor_optimize` as the decision (twin of rcg_control_tick_opt): sim_step ->
    [RQL/SQL: buffer push + critic fit, as ``control_tick``] -> optimise from ``action_sqn_init`` (or, with
    ``warm_start`` after the first tick, from the previous optimum shifted by one step with the last step repeated) ->
    action = first ``du`` entries -> accum, step_idx."""
    sim_substeps(cfg, env, cfg.substeps_per_tick)
    if cfg.mode != MODE_MPC:
        every = max(int(cfg.critic_every_ticks), 1)
        critic_update(cfg, env, do_fit=((env.tick_count + 1) % every) == 0)
    obs = env.state
    state_sys = env.state_prev if cfg.ref_lag else env.state
    B = obs.shape[0]
    prev = getattr(env, "action_sqn", None)
    if warm_start and env.tick_count > 0 and prev is not None:
        u0 = np.concatenate([prev[:, 1:], prev[:, -1:]], axis=1)
    else:
        u0 = np.broadcast_to(action_sqn_init(cfg, action_init), (B, cfg.n_actor, cfg.du))
    env.tick_count += 1
    U, J, its = actor_optimize(cfg, obs, state_sys, u0, iters, pars=env.pars if env.pars.ndim == 2 else None,
                               w_critic=None if cfg.mode == MODE_MPC else env.w_critic, memory=memory)
    env.action_sqn, env.best_J, env.best_idx = U, J, its
    env.action = U[:, 0, :].copy()
    if not cfg.accum_every_substep:
        env.accum = env.accum + stage_obj(obs, env.action, cfg) * cfg.sampling_time
    env.step_idx = env.step_idx + np.int32(1)


def critic_update(cfg: OracleCfg, env: EnvBatch, do_fit=True):
    """RQL/SQL bookkeeping of compute_action (controllers.py:1458-1477): push (action_curr, obs), refit."""
    env.act_buf = push_vec(env.act_buf, env.action)
    env.obs_buf = push_vec(env.obs_buf, env.state)
    if do_fit and cfg.n_critic - 1 < 1:
        # empty TD stack (Ncritic = 1): _critic_cost is identically 0 (controllers.py:1227-1245), SLSQP returns its
        # start point w_critic_init = ones, clipped into [Wmin, Wmax]
        lo, hi = critic_bounds(cfg.critic_struct, cfg.dc)
        env.w_critic = np.broadcast_to(np.clip(np.ones(cfg.dc), lo, hi), env.w_prev.shape).copy()
        env.w_prev = env.w_critic
    elif do_fit:
        env.w_critic = critic_fit(cfg, env.w_prev, env.obs_buf, env.act_buf)
        env.w_prev = env.w_critic
    else:
        env.w_critic = env.w_prev
