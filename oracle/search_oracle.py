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
             k < K - K // 4: one draw per input held over the horizon;  the last quarter: one draw per input and step
    xi       key of (env, tick) = words 0, 1 of Philox4x32-7(counter = (env id lo, env id hi, episode_idx, step_idx),
             key = (seed lo ^ 0x43414E44, seed hi)).  A DRAW = Philox4x32-7(counter = (a, b, r, kind), that key) -> four
             words -> eight normals: word p gives (n_2p, n_2p+1) = sqrt(-2 ln u_r) (cos, sin)(2 pi u_a) with
             u_r = float32(((w & 0xffff) + 0.5) 2^-16), u_a = float32(((w >> 16) + 0.5) 2^-16).
             per-step candidate k (k >= K - K // 4): row element e <- draw (k, e // 8, kind 0), normal e % 8;
             held candidate k (k < K - K // 4): l = k % 64, t = k // 64, TPC = 8 // du: input c <- draw (l, t // TPC, kind 1),
             normal (t % TPC) du + c.

The integer stream and the uniforms are bit-exact twins of the kernel's; the kernel evaluates ln / sqrt / sin / cos with the
hardware's float32 instructions, this file in float64, so a candidate agrees to ~1e-6 sigma (the tests state the tolerance) -
and a test that checks a DECISION feeds the oracle the device's own candidates (rcg_candidates_sample), as every other
closed-loop parity test of the build feeds it the device's own inputs.
"""
import numpy as np

from . import rcg_oracle as O
from .disturb_oracle import MASK, philox4x32_10

CAND_DOMAIN = np.uint32(0x43414E44)
CAND_ROUNDS = 7  # Philox rounds of the candidate stream (key derivation and draws)


def cand_subkey(seed, env_id, episode_idx, step_idx):
    """[B, 2] uint32: the key of each env's candidate stream at its current (episode, step)."""
    env_id = np.asarray(env_id, dtype=np.int64).astype(np.uint64)
    ctr = np.stack([(env_id & MASK).astype(np.uint32), (env_id >> np.uint64(32)).astype(np.uint32),
                    np.asarray(episode_idx).astype(np.uint32), np.asarray(step_idx).astype(np.uint32)], axis=-1)
    s = np.uint64(int(seed) & 0xFFFFFFFFFFFFFFFF)
    key = np.broadcast_to(np.array([np.uint32(s & MASK) ^ CAND_DOMAIN, np.uint32(s >> np.uint64(32))], dtype=np.uint32),
                          ctr.shape[:-1] + (2,))
    return philox4x32_10(ctr, key, rounds=CAND_ROUNDS)[..., :2]


def ps_first(K):
    """Index of the first per-step candidate: the last quarter of the K candidates."""
    return K - (K >> 2)


def cand_draws(key, a, b, round_, kind):
    """The words of draws (a, b, round, kind) under each env's key: ``a``, ``b`` integer arrays of one shape S -> [B, *S, 4]
    uint32 (bit-exact twin of the kernel's)."""
    a, b = np.broadcast_arrays(np.asarray(a, dtype=np.uint32), np.asarray(b, dtype=np.uint32))
    ctr = np.stack([a, b, np.full_like(a, np.uint32(round_)), np.full_like(a, np.uint32(kind))], axis=-1)
    B = key.shape[0]
    ctr = np.broadcast_to(ctr[None], (B,) + ctr.shape)
    kk = np.broadcast_to(key.reshape((B,) + (1,) * a.ndim + (2,)), ctr.shape[:-1] + (2,))
    return philox4x32_10(ctr, kk, rounds=CAND_ROUNDS)


def uniforms_of(bits):
    """[..., 4] words -> (u_r, u_a) [..., 4] float32 each: the 16-bit uniforms of the four Box-Muller pairs."""
    mr = (bits & np.uint32(0xFFFF)).astype(np.float32)
    ma = (bits >> np.uint32(16)).astype(np.float32)
    c, h = np.float32(2.0 ** -16), np.float32(2.0 ** -17)
    return mr * c + h, ma * c + h  # float32: exact products, exact sums (17 significant bits)


def normals_of(bits):
    """[..., 4] words -> [..., 8] float64 normals."""
    ur, ua = uniforms_of(bits)
    r = np.sqrt(-2.0 * np.log(ur.astype(np.float64)))
    th = 2.0 * np.pi * ua.astype(np.float64)
    out = np.empty(bits.shape[:-1] + (8,))
    out[..., 0::2] = r * np.cos(th)
    out[..., 1::2] = r * np.sin(th)
    return out


def cand_uniforms(key, K, n_draws, round_):
    """[B, K, n_draws, 8] float32: the uniforms (u_r, u_a interleaved pair by pair) of the per-step draws (k, j)."""
    k, j = np.meshgrid(np.arange(K), np.arange(n_draws), indexing="ij")
    ur, ua = uniforms_of(cand_draws(key, k, j, round_, 0))
    out = np.empty(ur.shape[:-1] + (8,), dtype=np.float32)
    out[..., 0::2], out[..., 1::2] = ur, ua
    return out


def cand_normals(key, K, n_draws, round_):
    """[B, K, n_draws, 8] float64 normals of the per-step draws (k, j)."""
    k, j = np.meshgrid(np.arange(K), np.arange(n_draws), indexing="ij")
    return normals_of(cand_draws(key, k, j, round_, 0))


def held_normals(key, K, du, round_):
    """[B, K, du] float64: the normals a HELD candidate k applies at every step (only rows k < ps_first(K) are used)."""
    tpc = 8 // du
    k = np.arange(K)
    lane, t = k % 64, k // 64
    n8 = normals_of(cand_draws(key, lane, t // tpc, round_, 1))  # [B, K, 8]
    idx = ((t % tpc) * du)[:, None] + np.arange(du)[None, :]     # [K, du]
    return np.take_along_axis(n8, np.broadcast_to(idx[None], (n8.shape[0],) + idx.shape), axis=-1)


def candidates_sample(cfg: O.OracleCfg, seed, env_id, episode_idx, step_idx, K, round_, centre=None, action_init=None):
    """The K candidate rows of one round -> ``[B, K, N, du]`` (twin of rcg_candidates_sample)."""
    N, du = cfg.n_actor, cfg.du
    R = N * du
    B = len(np.atleast_1d(env_id))
    lo, hi = cfg.ctrl_bnds[:, 0], cfg.ctrl_bnds[:, 1]
    u0 = O.action_sqn_init(cfg, action_init)  # [N, du]
    c = np.broadcast_to(u0, (B, N, du)) if centre is None else np.asarray(centre, dtype=np.float64).reshape(B, N, du)
    n_draws = (R + 7) // 8
    key = cand_subkey(seed, env_id, episode_idx, step_idx)
    xi = cand_normals(key, K, n_draws, round_)  # [B, K, n_draws, 8]
    per_step = xi.reshape(B, K, n_draws * 8)[..., :R].reshape(B, K, N, du)
    held = np.broadcast_to(held_normals(key, K, du, round_)[:, :, None, :], (B, K, N, du))
    noise = np.where((np.arange(K) >= ps_first(K))[None, :, None, None], per_step, held)
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
