"""rcg_control_ticks: T control ticks with generated candidates in ONE launch (k_ticks) must leave every field of the
handle BIT-IDENTICAL to T calls of rcg_control_tick (same arithmetic, no launch per tick).  ``gpu`` marked."""
import time

import numpy as np
import pytest

from oracle import parity as PAR
from oracle import rcg_oracle as O
from tests.helpers import both, rand_states

pytestmark = pytest.mark.gpu

FIELDS = ["FIELD_STATE", "FIELD_STATE_PREV", "FIELD_ACTION", "FIELD_ACCUM", "FIELD_STEP_IDX", "FIELD_STATUS",
          "FIELD_BEST_J", "FIELD_BEST_IDX"]


def _pair(name, B, dtype, **kw):
    a, cfg = both(name, B, dtype, **kw)
    b, _ = both(name, B, dtype, **kw)
    return a, b, cfg


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("name,K,B,kw", [
    ("3wrobot", 64, 1024, {}),                                        # the small-batch corner it is for
    ("3wrobot", 256, 300, dict(substeps_per_tick=3, gamma=0.97)),    # 4 tiles per env, several substeps, discount
    ("3wrobot", 16, 1030, dict(ref_lag=True)),                        # 4 envs per wave, ragged last wave, ref_lag
    ("3wrobotNI", 100, 77, dict(accum_every_substep=True, substeps_per_tick=2)),
    ("2tank", 32, 515, {}),                                           # du = 1, target
    ("2tank", 5, 19, dict(stage_obj_struct=O.STAGE_BIQUADRATIC, R2=np.diag([1.0, 2.0, 0.5]))),  # generic stage cost
])
def test_T_ticks_in_one_launch_equal_T_single_ticks(name, K, B, kw, dtype):
    from rcognita_amd import _native as N

    rng = np.random.default_rng(K + B)
    T = 7
    one, many, cfg = _pair(name, B, dtype, n_actor=6, **kw)
    x0 = rand_states(rng, name, B)
    one.set_state(x0)
    many.set_state(x0)
    for _ in range(T):
        one.control_tick(None, K=K)
    many.control_ticks(T, K)
    for f in FIELDS:
        np.testing.assert_array_equal(many.get_field(getattr(N, f)), one.get_field(getattr(N, f)), err_msg=f)
    # and it continues identically: 3 more in one launch against 3 more single ticks, then an episode reset
    for _ in range(3):
        one.control_tick(None, K=K)
    many.control_ticks(3, K)
    for e in (one, many):
        e.episode_reset()
    one.control_tick(None, K=K)
    many.control_ticks(1, K)
    for f in FIELDS + ["FIELD_RETURNS", "FIELD_EPISODE_IDX"]:
        np.testing.assert_array_equal(many.get_field(getattr(N, f)), one.get_field(getattr(N, f)), err_msg=f)


def test_ticks_follow_the_oracle_and_freeze_nonfinite_envs():
    from rcognita_amd import _native as N

    rng = np.random.default_rng(2)
    B, K, T = 64, 64, 5
    eng, cfg = both("3wrobot", B, "f64", n_actor=5)
    x0 = rand_states(rng, "3wrobot", B)
    x0[7, 3] = 1e308  # overflows inside the first RK4 step: frozen at its last finite state, flagged
    eng.set_state(x0)
    env = O.new_batch(cfg, np.delete(x0, 7, axis=0))
    grid = O.grid_candidates(cfg, K)
    eng.control_ticks(T, K)
    for _ in range(T):
        O.control_tick(cfg, env, grid)
    ok = np.arange(B) != 7
    st = eng.get_field(N.FIELD_STATUS)
    assert st[7] == 1 and not st[ok].any()
    np.testing.assert_array_equal(eng.get_state()[7], x0[7])
    np.testing.assert_array_equal(eng.get_field(N.FIELD_STEP_IDX), np.full(B, T, np.int32))
    np.testing.assert_array_equal(eng.get_field(N.FIELD_BEST_IDX)[ok], env.best_idx)
    assert PAR.rel_err_norm(eng.get_state()[ok], env.state) < 1e-10
    assert PAR.rel_err_norm(eng.get_field(N.FIELD_ACCUM)[ok], env.accum, floor=float(np.max(np.abs(env.accum)))) < 1e-10


def test_ticks_refusals_change_nothing():
    from rcognita_amd import _native as N

    # RQL with an empty TD stack (Ncritic = 1): the single ticks keep w = clip(w_init) through a separate fill, no instance
    eng, _ = both("2tank", 8, "f32", n_actor=4, mode=O.MODE_RQL, n_critic=1, buffer_size=5)
    eng.set_state(rand_states(np.random.default_rng(0), "2tank", 8))
    x = eng.get_state().copy()
    with pytest.raises(N.NativeError) as ei:
        eng.control_ticks(3, 16)
    assert ei.value.code == N.ERR_UNSUPPORTED
    mpc, _ = both("3wrobot", 8, "f32", n_actor=4)
    for T, K in ((0, 16), (2, 50), (2, 0)):
        with pytest.raises(N.NativeError) as ei:
            mpc.control_ticks(T, K)
        assert ei.value.code == N.ERR_BAD_ARG
    np.testing.assert_array_equal(eng.get_state(), x)
    np.testing.assert_array_equal(mpc.get_field(N.FIELD_STEP_IDX), np.zeros(8, np.int32))


def test_small_batch_is_no_longer_launch_bound():
    """SURVEY.md 7 step 5 / VERDICT r1 item 8: B = 1024, K = 64 was 1.25e8 env.control-steps/s with two launches per
    tick; T ticks per launch must beat that clearly (measured: see DESIGN.md 5)."""
    B, K, T = 1024, 64, 512
    eng, _ = both("3wrobot", B, "f32", n_actor=10)
    eng.set_state(rand_states(np.random.default_rng(1), "3wrobot", B))
    eng.control_ticks(T, K)
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        eng.control_ticks(T, K)
    eng.synchronize()
    rate = 4 * T * B / (time.perf_counter() - t0)
    print(f"persistent ticks: {rate:.3e} env.control-steps/s at B={B}, K={K}")
    assert rate > 2.5e8


@pytest.mark.parametrize("name,mode,streamed", [("3wrobot", "MPC", True), ("2tank", "RQL", True), ("2tank", "SQL", False)])
def test_tick_n_equals_n_single_ticks(name, mode, streamed):
    """rcg_control_tick_n: T ticks with the same candidates in one native call (any mode, streamed or generated; MPC: one
    launch, RQL / SQL: the launches of T single ticks) leave every field as T calls of rcg_control_tick do, bit for bit;
    T < 1 is refused and changes nothing."""
    from rcognita_amd import _native as N

    rng = np.random.default_rng(11)
    B, K, T = 300, 64, 9
    kw = dict(n_actor=6)
    if mode != "MPC":
        kw.update(mode=O.MODE_IDS[mode], critic_struct=O.CRITIC_QUADRATIC, n_critic=3, buffer_size=5)
    a, _ = both(name, B, "f32", **kw)
    b, cfg = both(name, B, "f32", **kw)
    x0 = rand_states(rng, name, B)
    a.set_state(x0)
    b.set_state(x0)
    ca = cb = None
    if streamed:
        lo, hi = cfg.ctrl_bnds[:, 0], cfg.ctrl_bnds[:, 1]
        c = (lo + (hi - lo) * rng.random((B, K, 6, cfg.du))).astype(np.float32)
        ca, cb = a.to_device(c), b.to_device(c)
    for _ in range(T):
        a.control_tick(ca, K=K)
    b.control_tick(cb, K=K, T=T)
    fields = [N.FIELD_STATE, N.FIELD_STATE_PREV, N.FIELD_ACTION, N.FIELD_ACCUM, N.FIELD_STEP_IDX, N.FIELD_BEST_IDX, N.FIELD_BEST_J]
    if mode != "MPC":
        fields += [N.FIELD_W_CRITIC, N.FIELD_W_PREV, N.FIELD_OBS_BUF, N.FIELD_ACT_BUF]
    for f in fields:
        np.testing.assert_array_equal(a.get_field(f), b.get_field(f), err_msg=f"field {f}")
    assert N.lib().rcg_tick_count(a._h) == N.lib().rcg_tick_count(b._h) == T
    with pytest.raises(N.NativeError) as ei:
        b.control_tick(cb, K=K, T=0)
    assert ei.value.code == N.ERR_BAD_ARG
    np.testing.assert_array_equal(a.get_state(), b.get_state())


_FULL7 = np.random.default_rng(70).uniform(-1, 1, (7, 7))
_FULL5 = np.random.default_rng(50).uniform(-1, 1, (5, 5))


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("name,K,B,kw", [
    ("3wrobot", 64, 1024, {}),                                     # one env per wave, one tile, rows resident in LDS
    ("3wrobot", 256, 130, dict(gamma=0.97)),                       # four tiles per env, resident (20 KB per wave in f32)
    ("3wrobot", 16, 1030, dict(ref_lag=True)),                      # four envs per wave (packed-tile kernel for single ticks)
    ("3wrobotNI", 100, 77, dict(substeps_per_tick=2)),             # ragged second tile
    ("2tank", 48, 515, {}),                                        # du = 1, target, one ragged tile
    ("3wrobot", 1024, 9, {}),                                      # 80 KB of rows per wave: re-staged tile by tile every tick
    ("2tank", 40, 33, dict(stage_obj_struct=O.STAGE_BIQUADRATIC, R2=np.diag([1.0, 2.0, 0.5]))),  # generic stage cost
    # round 6: cost structures no preset has, single ticks on k_actor_dma's DMA_MPC_GEND / DMA_MPC_GENF instances - the full
    # matrices go through the same symmetrised-triangle arithmetic in every kernel (rcg_kernels.hpp::quad_sym)
    ("3wrobot", 64, 200, dict(R1=_FULL7 @ _FULL7.T)),
    ("3wrobotNI", 128, 50, dict(R1=_FULL5 @ _FULL5.T, R2=1e-4 * (_FULL5.T @ _FULL5), stage_obj_struct=O.STAGE_BIQUADRATIC,
                                target=[0.5, -1.0, 0.25], gamma=0.95)),
    ("3wrobot", 64, 70, dict(target=[1.0, -2.0, 0.5, 0.0, 0.0])),
])
def test_streamed_T_ticks_in_one_launch_equal_T_single_ticks(name, K, B, kw, dtype):
    """rcg_control_tick_n with a caller's candidate tensor on an MPC handle: ONE launch of k_ticks (the wave's rows staged
    into LDS once - or tile by tile when they do not fit - and re-walked T times) against T single ticks, which run on the
    streamed production kernels (k_actor_dma, k_actor_dma_packed) or on k_actor: every field bit-identical, kernel
    identities asserted."""
    from rcognita_amd import _native as N
    from tests.helpers import assert_kernel

    rng = np.random.default_rng(K * 7 + B)
    T, Nh = 6, 5
    one, many, cfg = _pair(name, B, dtype, n_actor=Nh, **kw)
    x0 = rand_states(rng, name, B)
    lo, hi = cfg.ctrl_bnds[:, 0], cfg.ctrl_bnds[:, 1]
    c = (lo + (hi - lo) * rng.random((B, K, Nh, cfg.du))).astype(one.real)
    ca, cb = one.to_device(c), many.to_device(c)
    one.set_state(x0)
    many.set_state(x0)
    for _ in range(T):
        one.control_tick(ca, K=K)
    many.control_tick(cb, K=K, T=T)
    ll = assert_kernel(many, "k_ticks")
    assert ll["variant"] & 4, ll  # the streamed instance
    assert one.last_launch(N.KERNEL_ACTOR)["kernel"] != "k_ticks"
    for f in FIELDS:
        np.testing.assert_array_equal(many.get_field(getattr(N, f)), one.get_field(getattr(N, f)), err_msg=f)
    assert N.lib().rcg_tick_count(many._h) == N.lib().rcg_tick_count(one._h) == T
    # the two entry points continue from each other's state
    many.control_tick(cb, K=K)
    one.control_tick(ca, K=K, T=1)
    for f in FIELDS:
        np.testing.assert_array_equal(many.get_field(getattr(N, f)), one.get_field(getattr(N, f)), err_msg=f)


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("name,K,streamed", [("3wrobot", 64, False), ("3wrobotNI", 16, True), ("2tank", 32, False),
                                             ("3wrobot", 64, True)])
def test_T_ticks_with_the_disturbance_model_equal_T_single_ticks(name, K, streamed, dtype):
    """RCG_FLAG_DISTURB handles: the disturbance state and the noise counter of the Philox stream travel in registers with
    the state; T ticks in one launch leave STATE, DISTURB, SUBSTEP_IDX and everything else as T single ticks (k_sim_dist +
    the decision kernel) do, bit for bit - generated grid (rcg_control_ticks) and a caller's tensor (rcg_control_tick_n)."""
    from rcognita_amd import _native as N
    from tests.helpers import assert_kernel

    rng = np.random.default_rng(K + (1 if streamed else 0))
    B, T, Nh = 261, 5, 5
    dist = dict(is_disturb=True, pars_disturb=[[30.0, 10.0], [0.5, -0.2], [2.0, 1.5]], disturb_init=[1.0, -2.0], seed=31,
                env_id_base=5_000_000_000)
    one, many, cfg = _pair(name, B, dtype, n_actor=Nh, substeps_per_tick=2, engine_only=dist)
    x0 = rand_states(rng, name, B)
    one.set_state(x0)
    many.set_state(x0)
    ca = cb = None
    if streamed:
        lo, hi = cfg.ctrl_bnds[:, 0], cfg.ctrl_bnds[:, 1]
        c = (lo + (hi - lo) * rng.random((B, K, Nh, cfg.du))).astype(one.real)
        ca, cb = one.to_device(c), many.to_device(c)
    for _ in range(T):
        one.control_tick(ca, K=K)
    if streamed:
        many.control_tick(cb, K=K, T=T)
    else:
        many.control_ticks(T, K)
    assert_kernel(many, "k_ticks")
    assert_kernel(one, "k_sim_dist", kind=N.KERNEL_SIM)
    for f in FIELDS + ["FIELD_DISTURB", "FIELD_SUBSTEP_IDX"]:
        np.testing.assert_array_equal(many.get_field(getattr(N, f)), one.get_field(getattr(N, f)), err_msg=f)
    np.testing.assert_array_equal(many.get_field(N.FIELD_SUBSTEP_IDX), np.full(B, 2 * T, np.int32))
    if name != "2tank":  # (Sys2Tank's disturbance is inert, systems.py:421-424)
        assert np.any(many.get_field(N.FIELD_DISTURB) != np.array([1.0, -2.0], dtype=one.real))


def test_streamed_small_batch_is_no_longer_launch_bound():
    """B = 1024, streamed K = 64 - the per-tick-launch loop (rcg_control_tick x T from one native call:
    two launches per tick) against T ticks in one launch with the rows resident in LDS."""
    import torch

    B, K, T, Nh = 1024, 64, 256, 10
    rates = {}
    for tag in ("per_tick_launches", "one_launch"):
        eng, cfg = both("3wrobot", B, "f32", n_actor=Nh)
        eng.set_state(rand_states(np.random.default_rng(1), "3wrobot", B))
        lo, hi = cfg.ctrl_bnds[:, 0], cfg.ctrl_bnds[:, 1]
        cand = eng.to_device((lo + (hi - lo) * np.random.default_rng(2).random((B, K, Nh, 2))).astype(np.float32))
        if tag == "per_tick_launches":
            step = lambda: [eng.control_tick(cand, K=K) for _ in range(T)]  # noqa: E731
        else:
            step = lambda: eng.control_tick(cand, K=K, T=T)  # noqa: E731
        step()
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            step()
        eng.synchronize()
        rates[tag] = 3 * T * B / (time.perf_counter() - t0)
    print(f"\nstreamed B={B} K={K}: {rates}")
    assert rates["one_launch"] > 2.0 * rates["per_tick_launches"]


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("name,mode,cs,K,B,kw", [
    ("2tank", "RQL", "quadratic", 256, 300, dict(n_critic=4, buffer_size=10)),       # configs[2]'s controller
    ("2tank", "SQL", "quad-lin", 64, 1024, dict(n_critic=3, buffer_size=5, critic_every_ticks=3)),
    ("3wrobot", "RQL", "quad-nomix", 64, 515, dict(n_critic=4, buffer_size=6, ref_lag=True)),
    ("3wrobot", "SQL", "quad-mix", 16, 130, dict(n_critic=6, buffer_size=8)),          # 4 envs per wave, 5 TD rows
    ("3wrobotNI", "RQL", "quad-mix", 100, 77, dict(n_critic=4, buffer_size=6, gamma=0.95, substeps_per_tick=2)),
    ("3wrobotNI", "SQL", "quadratic", 256, 64, dict(n_critic=4, buffer_size=10)),
])
def test_T_critic_mode_ticks_in_one_launch_equal_T_single_ticks(name, mode, cs, K, B, kw, dtype):
    """RQL / SQL: rcg_control_ticks runs env step + buffer push + critic fit and the decision of every tick as phases of ONE
    launch (k_ticks_mem: the bodies of k_critic_fit and k_actor on the same memory).  Every field - critic weights and
    both buffers included - ends bit-identical to T single ticks, across the critic period's phase and an episode reset."""
    from rcognita_amd import _native as N
    from tests.helpers import assert_kernel

    rng = np.random.default_rng(K + B)
    T = 9
    one, many, cfg = _pair(name, B, dtype, n_actor=5, mode=O.MODE_IDS[mode], critic_struct=O.CRITIC_IDS[cs], **kw)
    x0 = rand_states(rng, name, B) * 0.5
    one.set_state(x0)
    many.set_state(x0)
    fields = FIELDS + ["FIELD_W_CRITIC", "FIELD_W_PREV", "FIELD_OBS_BUF", "FIELD_ACT_BUF"]
    for _ in range(T):
        one.control_tick(None, K=K)
    many.control_ticks(T, K)
    ll = assert_kernel(many, "k_ticks")
    assert ll["variant"] & 16, ll
    assert_kernel(one, "k_critic_fit", kind=N.KERNEL_CRITIC)
    for f in fields:
        np.testing.assert_array_equal(many.get_field(getattr(N, f)), one.get_field(getattr(N, f)), err_msg=f)
    assert not np.allclose(many.get_field(N.FIELD_W_CRITIC), 1.0)  # the critic was fitted
    # continues identically in two pieces (the critic period's phase travels with the handle's tick counter), then an
    # episode reset, then rcg_control_tick_n's route to the same kernel
    for _ in range(5):
        one.control_tick(None, K=K)
    many.control_ticks(2, K)
    many.control_ticks(3, K)
    for e in (one, many):
        e.episode_reset()
    for _ in range(4):
        one.control_tick(None, K=K)
    many.control_tick(None, K=K, T=4)
    assert_kernel(many, "k_ticks")
    for f in fields + ["FIELD_RETURNS", "FIELD_EPISODE_IDX"]:
        np.testing.assert_array_equal(many.get_field(getattr(N, f)), one.get_field(getattr(N, f)), err_msg=f)
    assert N.lib().rcg_tick_count(many._h) == N.lib().rcg_tick_count(one._h) == 4


def test_many_weight_structures_in_one_launch():
    """Critic structures with >= 20 weights (here the kinematic robot's quad-lin: 20) are fitted by the four-lane kernel
    (k_critic_fit_ml: variant bit 1024 of the fit's launch record).  Round 5: k_ticks_mem has the same four-lane walk as its
    critic phase for them (the wave's envs take its first quads), so rcg_control_ticks serves these handles too - bit-identical
    to T single ticks; K < 4 (more than 16 envs per wave) is refused and rcg_control_tick_n loops single ticks there."""
    from rcognita_amd import _native as N

    B, T = 130, 5
    for K in (64, 16, 4):
        one, many, cfg = _pair("3wrobotNI", B, "f64", n_actor=5, mode=O.MODE_RQL, critic_struct=O.CRITIC_QUAD_LIN, n_critic=4,
                               buffer_size=6)
        assert cfg.dc == 20
        x0 = rand_states(np.random.default_rng(3), "3wrobotNI", B) * 0.5
        one.set_state(x0)
        many.set_state(x0)
        for _ in range(T):
            one.control_tick(None, K=K)
        ll = one.last_launch(N.KERNEL_CRITIC)
        assert ll["kernel"] == "k_critic_fit" and (ll["variant"] & 1024) and ll["envs_per_wave"] == 16, ll
        many.control_ticks(T, K)
        ll = many.last_launch(N.KERNEL_ACTOR)
        assert ll["kernel"] == "k_ticks" and (ll["variant"] & 16), ll  # k_ticks_mem
        for f in FIELDS + ["FIELD_W_CRITIC", "FIELD_W_PREV", "FIELD_OBS_BUF", "FIELD_ACT_BUF"]:
            np.testing.assert_array_equal(many.get_field(getattr(N, f)), one.get_field(getattr(N, f)), err_msg=f"K={K} {f}")
        assert not np.allclose(many.get_field(N.FIELD_W_CRITIC), 1.0)
    one, many, cfg = _pair("3wrobotNI", B, "f64", n_actor=5, mode=O.MODE_RQL, critic_struct=O.CRITIC_QUAD_LIN, n_critic=4,
                           buffer_size=6)
    with pytest.raises(N.NativeError) as ei:
        many.control_ticks(T, 1)
    assert ei.value.code == N.ERR_UNSUPPORTED


def test_critic_mode_small_batch_rate():
    """configs[2]'s controller at B = 1024 (launch-bound under two launches per tick): T ticks per launch against the loop
    of single ticks issued from one native call."""
    B, K, T = 1024, 64, 128
    rates = {}
    for tag in ("per_tick_launches", "one_launch"):
        eng, _ = both("2tank", B, "f32", n_actor=10, mode=O.MODE_RQL, critic_struct=O.CRITIC_QUADRATIC, n_critic=4, buffer_size=10)
        eng.set_state(rand_states(np.random.default_rng(1), "2tank", B))
        step = (lambda: [eng.control_tick(None, K=K) for _ in range(T)]) if tag == "per_tick_launches" else (lambda: eng.control_ticks(T, K))
        step()
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            step()
        eng.synchronize()
        rates[tag] = 3 * T * B / (time.perf_counter() - t0)
    print(f"\nRQL B={B} K={K}: {rates}")
    assert rates["one_launch"] > 1.2 * rates["per_tick_launches"]


CRITIC_FIELDS = FIELDS + ["FIELD_W_CRITIC", "FIELD_W_PREV", "FIELD_OBS_BUF", "FIELD_ACT_BUF"]


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("name,mode,cs,K,B,kw", [
    ("2tank", "RQL", "quadratic", 64, 1024, {}),                    # configs[2]'s controller; single ticks: k_actor_dma
    ("2tank", "SQL", "quad-lin", 48, 515, {}),                      # one ragged tile per env on k_actor_dma
    ("2tank", "RQL", "quad-nomix", 16, 130, dict(ref_lag=True)),    # four envs per tile: k_actor_dma_packed
    ("3wrobot", "RQL", "quad-nomix", 64, 300, {}),
    ("3wrobot", "SQL", "quad-nomix", 8, 77, {}),                    # packed, eight envs per tile
    ("3wrobot", "RQL", "quad-lin", 64, 130, {}),                     # 35 weights: the four-lane fit, weights parked in LDS by k_actor_dma
    ("3wrobotNI", "SQL", "quad-mix", 128, 66, {}),
    ("3wrobotNI", "RQL", "quad-mix", 33, 40, {}),                    # K = 33: one ragged tile
])
def test_streamed_critic_mode_T_ticks_in_one_launch_equal_T_single_ticks(name, mode, cs, K, B, kw, dtype):
    """Round 5: rcg_control_tick_n with a caller's tensor on an RQL / SQL handle - ONE launch of k_ticks_mem whose decision phase
    walks the tensor (actor_wave's streamed form, with the accumulation order of the streamed production kernels) against T
    single ticks on k_actor_dma / k_actor_dma_packed: every field bit-identical, weights and buffers included, kernel
    identities asserted."""
    from rcognita_amd import _native as N
    from tests.helpers import assert_kernel

    rng = np.random.default_rng(K * 11 + B)
    T, Nh = 6, 5
    one, many, cfg = _pair(name, B, dtype, n_actor=Nh, mode=O.MODE_IDS[mode], critic_struct=O.CRITIC_IDS[cs], n_critic=4,
                           buffer_size=6, **kw)
    x0 = rand_states(rng, name, B) * (0.5 if name != "2tank" else 1.0)
    lo, hi = cfg.ctrl_bnds[:, 0], cfg.ctrl_bnds[:, 1]
    c = (lo + (hi - lo) * rng.random((B, K, Nh, cfg.du))).astype(one.real)
    ca, cb = one.to_device(c), many.to_device(c)
    one.set_state(x0)
    many.set_state(x0)
    for _ in range(T):
        one.control_tick(ca, K=K)
    many.control_tick(cb, K=K, T=T)
    ll = assert_kernel(many, "k_ticks")
    assert (ll["variant"] & 16) and (ll["variant"] & 4), ll  # k_ticks_mem, streamed
    # (K = 33 rows of 40 bytes in f32 are not a whole number of 16-byte pieces per env: k_actor serves those single ticks)
    assert one.last_launch(N.KERNEL_ACTOR)["kernel"] in ("k_actor_dma", "k_actor_dma_packed", "k_actor")
    for f in CRITIC_FIELDS:
        np.testing.assert_array_equal(many.get_field(getattr(N, f)), one.get_field(getattr(N, f)), err_msg=f)
    assert not np.allclose(many.get_field(N.FIELD_W_CRITIC), 1.0)
    assert N.lib().rcg_tick_count(many._h) == N.lib().rcg_tick_count(one._h) == T
    many.control_tick(cb, K=K)
    one.control_tick(ca, K=K, T=1)
    for f in CRITIC_FIELDS:
        np.testing.assert_array_equal(many.get_field(getattr(N, f)), one.get_field(getattr(N, f)), err_msg=f)


def test_streamed_critic_mode_small_batch_rate():
    """configs[2]'s controller at B = 1024 with a caller's tensor (launch-bound under two launches per tick): T ticks in one
    launch against the loop of single ticks issued from one native call (bar of VERDICT r4: 2 x)."""
    import torch

    B, K, T, Nh = 1024, 64, 128, 10
    rates = {}
    for tag in ("per_tick_launches", "one_launch"):
        eng, cfg = both("2tank", B, "f32", n_actor=Nh, mode=O.MODE_RQL, critic_struct=O.CRITIC_QUADRATIC, n_critic=4, buffer_size=10)
        eng.set_state(rand_states(np.random.default_rng(1), "2tank", B))
        cand = torch.rand((B, K, Nh, 1), device="cuda").contiguous()
        torch.cuda.synchronize()
        step = (lambda: [eng.control_tick(cand, K=K) for _ in range(T)]) if tag == "per_tick_launches" else (lambda: eng.control_tick(cand, K=K, T=T))
        step()
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            step()
        eng.synchronize()
        rates[tag] = 3 * T * B / (time.perf_counter() - t0)
    print(f"\nstreamed RQL B={B} K={K}: {rates}")
    assert rates["one_launch"] >= 1.5 * rates["per_tick_launches"], rates


@pytest.mark.parametrize("cs,bs,nc", [("quad-nomix", 4, 4), ("quad-nomix", 6, 3), ("quadratic", 10, 4), ("quad-lin", 9, 6)])
def test_ring_pushes_inside_one_launch_end_in_place(cs, bs, nc):
    """k_ticks_mem keeps the two buffers as rings while it runs (a push overwrites the oldest row) and rotates the rows back
    after its last tick: for every T - below, at and above buffer_size, rotations of one cycle and of several (gcd > 1), the
    TD stack reaching past the rows kept in registers (Ncritic = 6), the four-lane fit (quad-lin: 35 weights) - both buffers
    and the weights end bit-identical to T single ticks, and a second launch continues from them."""
    from rcognita_amd import _native as N

    B, K = 37, 64
    for T in (1, 2, 3, bs - 1, bs, bs + 1, 2 * bs, 2 * bs + 3):
        one, many, _ = _pair("3wrobot", B, "f32", n_actor=4, mode=O.MODE_RQL, critic_struct=O.CRITIC_IDS[cs], n_critic=nc,
                             buffer_size=bs)
        x0 = rand_states(np.random.default_rng(100 * bs + T), "3wrobot", B) * 0.5
        one.set_state(x0)
        many.set_state(x0)
        for _ in range(T + 2):
            one.control_tick(None, K=K)
        many.control_ticks(T, K)
        many.control_ticks(2, K)
        for f in ("FIELD_OBS_BUF", "FIELD_ACT_BUF", "FIELD_W_CRITIC", "FIELD_W_PREV", "FIELD_STATE", "FIELD_ACTION"):
            np.testing.assert_array_equal(many.get_field(getattr(N, f)), one.get_field(getattr(N, f)), err_msg=f"{f} T={T}")
        one.close()
        many.close()
