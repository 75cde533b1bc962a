"""TEST INFRASTRUCTURE - CPU restatement (numpy) of the disturbance model, SURVEY.md 8f row f4.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Reference (rcognita/systems.py): with ``is_disturb=1`` the full state is ``[state, disturb]``;
  _disturb_dyn (:325-345, :384-394)   dq_k/dt = -tau_k * (q_k + sigma_k * (randn() + mu_k))   (2tank: 0, :421-424)
  _state_dyn   (:308-323)  3wrobot    dv/dt = (F + q_0)/m,  domega/dt = (M + q_1)/I
               (:370-382)  3wrobotNI  dx/dt += q_0,  dy/dt += q_0 (sic),  dalpha/dt += q_1
               (:412-419)  2tank      unaffected
The reference calls the unseeded global ``randn()`` inside every right-hand-side evaluation, so its trajectories are
not reproducible and depend on the ODE solver's stage count.  Pinned here: the right-hand side itself, with the noise
as an explicit input (tests/golden/F11_disturb_*.npz, oracle/gen_disturb_fixtures.py).

Build-defined (what the reference leaves undefined): the noise xi is drawn ONCE per RK4 substep and env and held over
the four stages; it comes from the counter-based generator Philox4x32-10 (Salmon et al., SC'11) with
counter = (env_id lo, env_id hi, episode_idx, substep_idx), key = (seed lo, seed hi): the stream of an env depends only
on (seed, global env id, episode, substep), never on the batch size, the sharding over GPUs or the launch geometry.
Two normals per draw by Box-Muller from 24-bit uniforms.
"""
import numpy as np

from . import rcg_oracle as O

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)
DIM_DISTURB = {O.SYS_3WROBOT: 2, O.SYS_3WROBOT_NI: 2, O.SYS_2TANK: 1}


def philox4x32_10(counter, key, rounds=10):
    """Philox4x32 with 10 rounds (``rounds``: the candidate draws of oracle/search_oracle.py use 7).  counter [..., 4] uint32,
    key [..., 2] uint32 -> [..., 4] uint32."""
    c = [np.asarray(counter[..., i], dtype=np.uint32).copy() for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint32).copy()
    k1 = np.asarray(key[..., 1], dtype=np.uint32).copy()
    with np.errstate(over="ignore"):
        for _ in range(int(rounds)):
            p0 = M0 * c[0].astype(np.uint64)
            p1 = M1 * c[2].astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK).astype(np.uint32)
            c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
            k0 = (k0 + W0).astype(np.uint32)
            k1 = (k1 + W1).astype(np.uint32)
    return np.stack(c, axis=-1)


def noise_bits(seed, env_id, episode_idx, substep_idx):
    """The four 32-bit words of one draw.  env_id int64 [B] (global id), episode_idx / substep_idx int32 [B]."""
    env_id = np.asarray(env_id, dtype=np.int64).astype(np.uint64)
    ctr = np.stack([(env_id & MASK).astype(np.uint32), (env_id >> np.uint64(32)).astype(np.uint32),
                    np.asarray(episode_idx).astype(np.uint32), np.asarray(substep_idx).astype(np.uint32)], axis=-1)
    s = np.uint64(seed)
    key = np.broadcast_to(np.array([np.uint32(s & MASK), np.uint32(s >> np.uint64(32))], dtype=np.uint32),
                          ctr.shape[:-1] + (2,))
    return philox4x32_10(ctr, key)


def normals_from_bits(bits):
    """Box-Muller on words 0, 1: u = ((w >> 8) + 0.5) * 2^-24 in (0, 1); xi = sqrt(-2 ln u0) * (cos, sin)(2 pi u1)."""
    u0 = ((bits[..., 0] >> np.uint32(8)).astype(np.float64) + 0.5) * 2.0 ** -24
    u1 = ((bits[..., 1] >> np.uint32(8)).astype(np.float64) + 0.5) * 2.0 ** -24
    r = np.sqrt(-2.0 * np.log(u0))
    th = 2.0 * np.pi * u1
    return np.stack([r * np.cos(th), r * np.sin(th)], axis=-1)


def disturb_noise(seed, env_id, episode_idx, substep_idx):
    return normals_from_bits(noise_bits(seed, env_id, episode_idx, substep_idx))


def state_dyn_disturbed(sys_id, x, u, q, pars):
    """_state_dyn with a disturbance (systems.py:308-323, 370-382, 412-419)."""
    d = O.state_dyn(sys_id, x, u, pars)
    if sys_id == O.SYS_3WROBOT:
        m, I = pars[..., 0], pars[..., 1]
        d[..., 3] = 1 / m * (u[..., 0] + q[..., 0])
        d[..., 4] = 1 / I * (u[..., 1] + q[..., 1])
    elif sys_id == O.SYS_3WROBOT_NI:
        d[..., 0] = d[..., 0] + q[..., 0]
        d[..., 1] = d[..., 1] + q[..., 0]
        d[..., 2] = d[..., 2] + q[..., 1]
    return d


def disturb_dyn(sys_id, q, xi, sigma, mu, tau):
    """_disturb_dyn (systems.py:325-345, 384-394; 2tank :421-424 returns zeros)."""
    if sys_id == O.SYS_2TANK:
        return np.zeros_like(q)
    dd = q.shape[-1]
    return -tau[:dd] * (q + sigma[:dd] * (xi[..., :dd] + mu[:dd]))


def closed_loop_rhs_full(sys_id, x, q, u, xi, pars, ctrl_bnds, sigma, mu, tau):
    """closed_loop_rhs on the full state (systems.py:213-253).  Returns (d_state, d_disturb, clipped action)."""
    a = O.clip_action(u, ctrl_bnds)
    return state_dyn_disturbed(sys_id, x, a, q, pars), disturb_dyn(sys_id, q, xi, sigma, mu, tau), a


def rk4_step_full(sys_id, x, q, u, xi, pars, ctrl_bnds, sigma, mu, tau, dt):
    """Classical RK4 on [state, disturb], noise held over the stages; combination order of rcg_oracle.rk4_step."""
    f = lambda xx, qq: closed_loop_rhs_full(sys_id, xx, qq, u, xi, pars, ctrl_bnds, sigma, mu, tau)[:2]
    k1x, k1q = f(x, q)
    k2x, k2q = f(x + 0.5 * dt * k1x, q + 0.5 * dt * k1q)
    k3x, k3q = f(x + 0.5 * dt * k2x, q + 0.5 * dt * k2q)
    k4x, k4q = f(x + dt * k3x, q + dt * k3q)
    return (x + dt / 6 * (((k1x + 2 * k2x) + 2 * k3x) + k4x), q + dt / 6 * (((k1q + 2 * k2q) + 2 * k3q) + k4q))


class DisturbCfg:
    def __init__(self, sigma, mu, tau, seed=0, env_id_base=0, disturb_init=None):
        self.sigma, self.mu, self.tau = (np.asarray(v, dtype=np.float64) for v in (sigma, mu, tau))
        self.seed, self.env_id_base = int(seed), int(env_id_base)
        self.disturb_init = None if disturb_init is None else np.asarray(disturb_init, dtype=np.float64)


def attach(cfg, env, dcfg: DisturbCfg):
    """Give an rcg_oracle.EnvBatch its disturbance state (q = disturb_init or 0, substep counter 0)."""
    B, dd = env.state.shape[0], DIM_DISTURB[cfg.sys_id]
    env.disturb = np.zeros((B, dd)) if dcfg.disturb_init is None else np.tile(dcfg.disturb_init[:dd], (B, 1))
    env.substep_idx = np.zeros(B, dtype=np.int32)
    if not hasattr(env, "episode_idx") or env.episode_idx is None:
        env.episode_idx = np.zeros(B, dtype=np.int32)


def sim_substeps(cfg, env, dcfg: DisturbCfg, n_substeps):
    """Twin of rcg_sim_step on a handle created with RCG_FLAG_DISTURB."""
    B = env.state.shape[0]
    ids = dcfg.env_id_base + np.arange(B, dtype=np.int64)
    for _ in range(n_substeps):
        xi = disturb_noise(dcfg.seed, ids, env.episode_idx, env.substep_idx)
        env.state_prev = env.state
        env.state, env.disturb = rk4_step_full(cfg.sys_id, env.state, env.disturb, env.action, xi, env.pars, cfg.ctrl_bnds,
                                               dcfg.sigma, dcfg.mu, dcfg.tau, cfg.dt_sim)
        env.substep_idx = env.substep_idx + np.int32(1)
        if cfg.accum_every_substep:
            env.accum = env.accum + O.stage_obj(env.state, env.action, cfg) * cfg.sampling_time
