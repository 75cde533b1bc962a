"""Device-side candidate search (rcg_candidates_sample / rcg_actor_search / rcg_control_tick_search, rcg_search.hpp) against
its oracle twin (oracle/search_oracle.py).  ``gpu`` marked.

Two kinds of statement:
  * the PRODUCER: the integer stream (Philox) is bit-exact by construction on both sides and the float32 uniforms are the
    same bits; ln / sqrt / sin / cos are the hardware's float32 instructions on the device and float64 libm in the oracle,
    so a candidate agrees to 1e-5 sigma_r (+ one float32 rounding of the value itself) - asserted per element;
  * the DECISION: the oracle is fed the device's own candidates (Engine.candidates_sample) round by round, so the argmin,
    the cost and the refined sequence are checked on identical inputs: best_idx exact in float64, a float32 index that
    differs must be a near-tie (the rule of oracle/parity.py).
"""
import numpy as np
import pytest

from oracle import rcg_oracle as O
from oracle import search_oracle as S
from tests.conftest import load_golden
from tests.helpers import SYSTEMS, TOL, assert_kernel, both, rand_states, rel_err_norm

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("name", SYSTEMS)
def test_candidate_producer_vs_oracle(name, dtype):
    from rcognita_amd import _native as N

    rng = np.random.default_rng(3)
    B, K, Nh = 37, 160, 7
    eng, cfg = both(name, B, dtype, n_actor=Nh, engine_only=dict(seed=2026, env_id_base=10_000_000_000))
    lo, hi = cfg.ctrl_bnds[:, 0], cfg.ctrl_bnds[:, 1]
    ep, st = rng.integers(0, 5, B).astype(np.int32), rng.integers(0, 900, B).astype(np.int32)
    eng.set_field(N.FIELD_EPISODE_IDX, ep)
    eng.set_field(N.FIELD_STEP_IDX, st)
    centre = rng.uniform(lo, hi, (B, Nh, cfg.du)).astype(eng.real)
    ids = 10_000_000_000 + np.arange(B)
    for r, ce in ((0, None), (0, centre), (3, centre)):
        c = eng.candidates_sample(K, round=r, centre=ce)
        c_or = S.candidates_sample(cfg, 2026, ids, ep, st, K, r, centre=None if ce is None else ce.astype(np.float64))
        sigma = 0.5 * (hi - lo) * 2.0 ** -r
        tol = 1e-5 * sigma + (0 if dtype == "f64" else 1.2e-7 * np.maximum(np.abs(lo), np.abs(hi)))
        assert np.all(np.abs(c - c_or) <= tol), float(np.max(np.abs(c - c_or) / sigma))
        assert np.all(c >= lo.astype(eng.real)) and np.all(c <= hi.astype(eng.real))
        if ce is not None:
            np.testing.assert_array_equal(c[:, 0], ce)  # candidate 0 is the centre, bit for bit
        if r == 0:
            np.testing.assert_array_equal(c[:, 1], np.broadcast_to(O.action_sqn_init(cfg).astype(eng.real), c[:, 1].shape))
    # the stream belongs to (seed, global env id, episode, step): a shard holding envs 20 .. 29 reproduces those rows
    sub, _ = both(name, 10, dtype, n_actor=Nh, engine_only=dict(seed=2026, env_id_base=10_000_000_020))
    sub.set_field(N.FIELD_EPISODE_IDX, ep[20:30])
    sub.set_field(N.FIELD_STEP_IDX, st[20:30])
    np.testing.assert_array_equal(sub.candidates_sample(K, round=3, centre=centre[20:30]), c[20:30])
    # ... and another tick draws other candidates
    eng.set_field(N.FIELD_STEP_IDX, st + 1)
    c2 = eng.candidates_sample(K, round=3, centre=centre)
    assert np.mean(c2[:, 2:] == c[:, 2:]) < 0.35  # equal only where the clip put both on a bound


def _device_sampler(eng, K):
    return lambda r, centre: eng.candidates_sample(K, round=r, centre=centre.astype(eng.real)).astype(np.float64)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("name,mode,cs", [("3wrobot", "MPC", "quad-nomix"), ("3wrobotNI", "RQL", "quad-mix"),
                                          ("2tank", "SQL", "quadratic"), ("2tank", "MPC", "quad-nomix")])
def test_actor_search_vs_oracle_on_the_device_candidates(name, mode, cs, dtype):
    from rcognita_amd import _native as N

    rng = np.random.default_rng(8)
    B, K, Nh, rounds = 29, 192, 6, 3
    eng, cfg = both(name, B, dtype, n_actor=Nh, mode=O.MODE_IDS[mode], critic_struct=O.CRITIC_IDS[cs], buffer_size=6,
                    gamma=0.96, engine_only=dict(seed=7))
    x = rand_states(rng, name, B)
    obs = x + rng.uniform(-0.02, 0.02, x.shape)
    w = rng.uniform(0.1, 2, (B, cfg.dc))
    eng.set_field(N.FIELD_W_CRITIC, w)
    eng.set_field(N.FIELD_STEP_IDX, np.arange(B, dtype=np.int32))
    r_ = eng.real
    obs_r, x_r, w_r = (a.astype(r_).astype(np.float64) for a in (obs, x, w))
    act, U, J, bi = eng.actor_search(K=K, rounds=rounds, obs=obs, state_sys=x)
    assert_kernel(eng, "k_actor_search")
    np.testing.assert_array_equal(act, U[:, 0, :])
    # the reported cost is the oracle's _actor_cost of the reported sequence
    J_chk = O.actor_cost(U.astype(np.float64), obs_r, x_r, cfg, w_critic=w_r)
    assert rel_err_norm(J, J_chk) < TOL[dtype]
    # the same search on the device's own candidates
    U_or, J_or, bi_or = S.actor_search(cfg, obs_r, x_r, K, rounds, 7, np.arange(B), np.zeros(B, int), np.arange(B),
                                       w_critic=w_r, sampler=_device_sampler(eng, K))
    if dtype == "f64":
        np.testing.assert_array_equal(bi, bi_or)
        np.testing.assert_array_equal(U, U_or)
        assert rel_err_norm(J, J_or) < 1e-11
    else:  # a float32 argmin may take the other side of a near-tie in some round: the cost reached must agree
        same = np.all(U.astype(np.float64) == U_or, axis=(1, 2))
        assert np.mean(same) > 0.8
        assert np.all(np.abs(J - J_or) <= 4 * TOL[dtype] * np.maximum(np.abs(J_or), 1.0) + 1e-3 * np.abs(J_or) * ~same)
    # more rounds never hurt (candidate 0 is the incumbent), one round never ends above action_sqn_init's cost
    _, _, J1, _ = eng.actor_search(K=K, rounds=1, obs=obs, state_sys=x)
    u0 = np.broadcast_to(O.action_sqn_init(cfg), (B, Nh, cfg.du))
    J0 = O.actor_cost(u0, obs_r, x_r, cfg, w_critic=w_r)
    slack = 4 * TOL[dtype] * np.maximum(np.abs(J0), 1.0)
    assert np.all(J1 <= J0 + slack) and np.all(J <= J1 + slack)


@pytest.mark.parametrize("name,mode", [("3wrobot", "MPC"), ("2tank", "RQL")])
def test_control_tick_search_closed_loop_vs_oracle(name, mode):
    """T ticks of rcg_control_tick_search (warm start on) in float64, every tick checked as a map from the device's own
    pre-tick values, the oracle searching over the device's own candidates."""
    from rcognita_amd import _native as N

    rng = np.random.default_rng(12)
    B, K, Nh, rounds, T = 9, 128, 5, 2, 5
    ai = [0.5] if name == "2tank" else None
    eng, cfg = both(name, B, "f64", n_actor=Nh, mode=O.MODE_IDS[mode], critic_struct=O.CRITIC_QUADRATIC, n_critic=4,
                    buffer_size=6, engine_only=dict(seed=99), **({"action_init": ai} if ai else {}))
    x0 = rand_states(rng, name, B) * 0.4
    eng.set_state(x0)
    env = O.new_batch(cfg, x0, action0=ai)
    prev = None
    for t in range(T):
        step_before = eng.get_field(N.FIELD_STEP_IDX).copy()
        # oracle side of the tick up to the decision
        O.sim_substeps(cfg, env, cfg.substeps_per_tick)
        if cfg.mode != O.MODE_MPC:
            O.critic_update(cfg, env, do_fit=True)
        # the device's candidates depend on its post-step state only through the centre: sample them on a twin handle
        # state by running the tick, then reproduce the rounds from the stored fields
        eng.control_tick_search(K=K, rounds=rounds, warm_start=True)
        assert_kernel(eng, "k_actor_search")
        assert rel_err_norm(eng.get_state(), env.state) < 1e-9, t
        if cfg.mode != O.MODE_MPC:
            assert rel_err_norm(eng.get_field(N.FIELD_W_CRITIC), env.w_critic, floor=1.0) < 1e-6, t
            env.w_critic = eng.get_field(N.FIELD_W_CRITIC).astype(np.float64)
            env.w_prev = eng.get_field(N.FIELD_W_PREV).astype(np.float64)
        centre = None if (t == 0 or prev is None) else np.concatenate([prev[:, 1:], prev[:, -1:]], axis=1)
        # candidates of this tick: STEP_IDX was `step_before` when they were drawn
        eng.set_field(N.FIELD_STEP_IDX, step_before)
        U_or, J_or, bi_or = S.actor_search(cfg, env.state, env.state, K, rounds, 99, np.arange(B), np.zeros(B, int),
                                           step_before, centre=centre, action_init=ai,
                                           w_critic=None if cfg.mode == O.MODE_MPC else env.w_critic,
                                           sampler=_device_sampler(eng, K))
        eng.set_field(N.FIELD_STEP_IDX, step_before + 1)
        U = eng.get_field(N.FIELD_ACTION_SQN)
        np.testing.assert_array_equal(eng.get_field(N.FIELD_BEST_IDX), bi_or)
        np.testing.assert_array_equal(U, U_or)
        assert rel_err_norm(eng.get_field(N.FIELD_BEST_J), J_or) < 1e-10
        np.testing.assert_array_equal(eng.get_field(N.FIELD_ACTION), U[:, 0, :])
        env.action = U[:, 0, :].astype(np.float64)
        env.accum = env.accum + O.stage_obj(env.state, env.action, cfg) * cfg.sampling_time
        assert rel_err_norm(eng.get_field(N.FIELD_ACCUM), env.accum, floor=float(np.max(np.abs(env.accum)) + 1e-9)) < 1e-9
        env.tick_count += 1
        env.state = eng.get_state().astype(np.float64)
        env.state_prev = eng.get_field(N.FIELD_STATE_PREV).astype(np.float64)
        if cfg.mode != O.MODE_MPC:
            env.obs_buf = eng.get_field(N.FIELD_OBS_BUF).astype(np.float64)
            env.act_buf = eng.get_field(N.FIELD_ACT_BUF).astype(np.float64)
        prev = U.astype(np.float64)
    np.testing.assert_array_equal(eng.get_field(N.FIELD_STEP_IDX), np.full(B, T, np.int32))


@pytest.mark.parametrize("name", SYSTEMS)
def test_search_quality_vs_reference_slsqp(name):
    """F8 states: six rounds of 256 device-generated candidates against the cost the reference's SLSQP reaches."""
    meta, z = load_golden(f"F8_slsqp_actor_{name}")
    x = z["state"]
    ai = [0.5] if name == "2tank" else None
    eng, cfg = both(name, x.shape[0], "f64", n_actor=meta["N"], gamma=meta["gamma"], pred_step_size=meta["pred_step_size"],
                    **({"action_init": ai} if ai else {}))
    eng.set_state(x)
    act, U, J, bi = eng.actor_search(K=256, rounds=6)
    assert np.all(J <= z["J_init"] * (1 + 1e-12))
    ratio = J / z["J_opt"]
    print(f"\nsearch {name}: J / J_slsqp median {np.median(ratio):.5f} max {np.max(ratio):.5f}")
    assert np.median(ratio) < 1.002 and np.max(ratio) < 1.02


def test_search_refusals_leave_the_handle_untouched():
    from rcognita_amd import _native as N

    eng, _ = both("3wrobotNI", 8, "f32", n_actor=4)
    eng.set_state(np.ones((8, 3)))
    before = eng.get_state().copy()
    for kw in (dict(K=32, rounds=2), dict(K=128, rounds=0), dict(K=128, rounds=65)):
        with pytest.raises(N.NativeError) as ei:
            eng.control_tick_search(**kw)
        assert ei.value.code == N.ERR_BAD_ARG
    np.testing.assert_array_equal(eng.get_state(), before)
    np.testing.assert_array_equal(eng.get_field(N.FIELD_STEP_IDX), np.zeros(8, np.int32))


def test_full_size_search_tick():
    """configs[1]'s batch: 65 536 envs x 256 generated candidates x 2 rounds in one launch; counters exact, sequences in the
    box, a sample's reported cost = the oracle's cost of the reported sequence, never above the start sequence's."""
    from rcognita_amd import _native as N

    B, Nh = 65536, 10
    rng = np.random.default_rng(1)
    eng, cfg = both("3wrobot", B, "f32", n_actor=Nh, engine_only=dict(seed=5))
    eng.set_state(rand_states(rng, "3wrobot", B))
    eng.control_tick_search(K=256, rounds=2)
    np.testing.assert_array_equal(eng.get_field(N.FIELD_STEP_IDX), np.ones(B, np.int32))
    U = eng.get_field(N.FIELD_ACTION_SQN)
    lo, hi = cfg.ctrl_bnds[:, 0], cfg.ctrl_bnds[:, 1]
    assert np.all(U >= lo) and np.all(U <= hi)
    sel = np.sort(rng.choice(B, 64, replace=False))
    st = eng.get_state()[sel].astype(np.float64)
    J_chk = O.actor_cost(U[sel].astype(np.float64), st, st, cfg)
    assert rel_err_norm(eng.get_field(N.FIELD_BEST_J)[sel], J_chk) < 1e-5
    J0 = O.actor_cost(np.broadcast_to(O.action_sqn_init(cfg), (64, Nh, 2)), st, st, cfg)
    assert np.all(J_chk <= J0 * (1 + 1e-5))
