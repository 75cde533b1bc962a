"""TEST INFRASTRUCTURE - CPU restatement (numpy) of the device-side candidate search, rcg_actor_search /
rcg_control_tick_search / rcg_candidates_sample (rcognita_amd/csrc/rcg_search.hpp).  Only tests/, __graft_entry__.smoke() and
bench.py's checker legs may import this.

What the search replaces: the SLSQP call of ``CtrlOptPred._actor_optimizer`` (rcognita/controllers.py:1330-1427), by ``rounds``
rounds of ``K`` candidate action sequences evaluated with the pinned ``_actor_cost`` (rcg_oracle.actor_cost) and refined
around each round's winner.  The reference has no counterpart of the sampling rule - it is build-defined, and stated once in
rcg_search.hpp; this file mirrors it statement by statement:

    centre   round 0: the caller's sequence; later rounds: the previous round's winner
    k = 0    the centre itself;   round 0, k = 1: action_sqn_init
    else     clip(centre + sigma_r xi, lo, hi),  sigma_r = 0.5 (hi - lo) 2^-round
             k < K // 2: one draw per input held over the horizon;  k >= K // 2: one draw per input and step
    xi       key of (env, tick) = words 0, 1 of Philox4x32-10(counter = (env id lo, env id hi, episode_idx, step_idx),
             key = (seed lo ^ 0x43414E44, seed hi));  chunk j of candidate k in round r = Philox(counter = (k, j, r, 0), that
             key) -> u_i = float32((m_i + 0.5) 2^-24), m_i = word_i >> 8 -> (xi_0, xi_1) = sqrt(-2 ln u_0) (cos, sin)(2 pi u_1),
             (xi_2, xi_3) likewise from (u_2, u_3): the normals of row elements 4 j .. 4 j + 3.

The integer stream and the uniforms are bit-exact twins of the kernel's; the kernel evaluates ln / sqrt / sin / cos with the
hardware's float32 instructions, this file in float64, so a candidate agrees to ~1e-6 sigma (the tests state the tolerance) -
and a test that checks a DECISION feeds the oracle the device's own candidates (rcg_candidates_sample), as every other
closed-loop parity test of the build feeds it the device's own inputs.
"""
import numpy as np

from . import rcg_oracle as O
from .disturb_oracle import MASK, philox4x32_10

CAND_DOMAIN = np.uint32(0x43414E44)


def cand_subkey(seed, env_id, episode_idx, step_idx):
    """[B, 2] uint32: the key of each env's candidate stream at its current (episode, step)."""
    env_id = np.asarray(env_id, dtype=np.int64).astype(np.uint64)
    ctr = np.stack([(env_id & MASK).astype(np.uint32), (env_id >> np.uint64(32)).astype(np.uint32),
                    np.asarray(episode_idx).astype(np.uint32), np.asarray(step_idx).astype(np.uint32)], axis=-1)
    s = np.uint64(int(seed) & 0xFFFFFFFFFFFFFFFF)
    key = np.broadcast_to(np.array([np.uint32(s & MASK) ^ CAND_DOMAIN, np.uint32(s >> np.uint64(32))], dtype=np.uint32),
                          ctr.shape[:-1] + (2,))
    return philox4x32_10(ctr, key)[..., :2]


def cand_uniforms(key, K, n_chunks, round_):
    """[B, K, n_chunks, 4] float32 uniforms (bit-exact twin of the kernel's)."""
    B = key.shape[0]
    k, j = np.meshgrid(np.arange(K, dtype=np.uint32), np.arange(n_chunks, dtype=np.uint32), indexing="ij")
    ctr = np.stack([k, j, np.full_like(k, np.uint32(round_)), np.zeros_like(k)], axis=-1)  # [K, n_chunks, 4]
    ctr = np.broadcast_to(ctr[None], (B,) + ctr.shape)
    kk = np.broadcast_to(key[:, None, None, :], (B, K, n_chunks, 2))
    bits = philox4x32_10(ctr, kk)
    m = (bits >> np.uint32(8)).astype(np.float32)
    return m * np.float32(2.0 ** -24) + np.float32(2.0 ** -25)  # float32: exact product, one rounding in the sum


def cand_normals(key, K, n_chunks, round_):
    """[B, K, n_chunks, 4] float64 normals: Box-Muller on the pairs (u0, u1), (u2, u3)."""
    u = cand_uniforms(key, K, n_chunks, round_).astype(np.float64)
    out = np.empty_like(u)
    for p in range(2):
        r = np.sqrt(-2.0 * np.log(u[..., 2 * p]))
        th = 2.0 * np.pi * u[..., 2 * p + 1]
        out[..., 2 * p] = r * np.cos(th)
        out[..., 2 * p + 1] = r * np.sin(th)
    return out


def candidates_sample(cfg: O.OracleCfg, seed, env_id, episode_idx, step_idx, K, round_, centre=None, action_init=None):
    """The K candidate rows of one round -> ``[B, K, N, du]`` (twin of rcg_candidates_sample)."""
    N, du = cfg.n_actor, cfg.du
    R = N * du
    B = len(np.atleast_1d(env_id))
    lo, hi = cfg.ctrl_bnds[:, 0], cfg.ctrl_bnds[:, 1]
    u0 = O.action_sqn_init(cfg, action_init)  # [N, du]
    c = np.broadcast_to(u0, (B, N, du)) if centre is None else np.asarray(centre, dtype=np.float64).reshape(B, N, du)
    n_chunks = (R + 3) // 4
    key = cand_subkey(seed, env_id, episode_idx, step_idx)
    xi = cand_normals(key, K, n_chunks, round_)  # [B, K, n_chunks, 4]
    per_step = xi.reshape(B, K, n_chunks * 4)[..., :R].reshape(B, K, N, du)
    held = np.broadcast_to(xi[:, :, 0, :du][:, :, None, :], (B, K, N, du))  # chunk 0's first du normals at every step
    noise = np.where((np.arange(K) >= (K >> 1))[None, :, None, None], per_step, held)
    sigma = (0.5 * (hi - lo)) * 2.0 ** (-float(round_))
    cand = np.minimum(np.maximum(c[:, None] + sigma * noise, lo), hi)
    cand[:, 0] = c
    if round_ == 0 and K > 1:
        cand[:, 1] = u0
    return cand


def actor_search(cfg: O.OracleCfg, obs, state_sys, K, rounds, seed, env_id, episode_idx, step_idx, centre=None,
                 action_init=None, pars=None, w_critic=None, sampler=None):
    """``rounds`` rounds of the search (twin of rcg_actor_search) -> ``(u_best [B, N, du], best_J [B], best_idx [B])``.

    ``sampler(round, centre) -> [B, K, N, du]`` overrides the oracle's own candidates: the GPU tests pass the DEVICE's
    producer (Engine.candidates_sample), so that the decision is checked on identical inputs."""
    obs = np.asarray(obs, dtype=np.float64).reshape(-1, cfg.ds)
    xs = np.asarray(state_sys, dtype=np.float64).reshape(-1, cfg.ds)
    B = obs.shape[0]
    u0 = O.action_sqn_init(cfg, action_init)
    c = np.array(np.broadcast_to(u0, (B, cfg.n_actor, cfg.du)) if centre is None else
                 np.asarray(centre, dtype=np.float64).reshape(B, cfg.n_actor, cfg.du))
    bj = bi = None
    for r in range(int(rounds)):
        cand = (sampler(r, c) if sampler is not None else
                candidates_sample(cfg, seed, env_id, episode_idx, step_idx, K, r, centre=c, action_init=action_init))
        cand = np.asarray(cand, dtype=np.float64)
        J = O.actor_cost(cand, obs[:, None, :], xs[:, None, :], cfg,
                         pars=None if pars is None else (pars[:, None, :] if np.ndim(pars) == 2 else pars),
                         w_critic=None if w_critic is None else np.asarray(w_critic)[:, None, :])
        bj, bi = O.argmin_first(J)
        c = cand[np.arange(B), bi]
    return c, bj, bi
