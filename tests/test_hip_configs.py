"""BASELINE.json configs[2] and [4] at their full per-GPU sizes, through size-independent properties plus an
oracle check on a sample of envs; the mixed pool against per-type oracles.  ``gpu`` marked."""
import numpy as np
import pytest

from oracle import rcg_oracle as O
from tests.helpers import PRESETS, assert_kernel, oracle_cfg, rand_states, rel_err_norm

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("streamed", [False, True])
def test_full_size_C3_2tank_rql_critic(streamed, dtype):
    """configs[2]: Sys2Tank, B = 131072, Nactor = 20, RQL with the quadratic critic refit every tick, K = 256
    candidates per env - generated level grid, and streamed from a [B][K][N][du] tensor (2.7 GB, k_actor_dma's critic
    instances).  Properties: integer counters exact for every env; a fit never increases Jc over w_init; weights stay
    in the box; every env is independent of its position (first 16 envs alone == inside the batch); a random sample
    of envs follows the oracle on EVERY tick at 1e-5 (state, action, best_J, accum, critic weights, buffers;
    oracle/parity.py)."""
    from oracle import parity as PAR
    from rcognita_amd import Engine
    from rcognita_amd import _native as N
    from rcognita_amd.pool import preset_engine_config

    B, K, Nh, T = 131072, 256, 20, 5
    rng = np.random.default_rng(1234)
    real = np.float32 if dtype == "f32" else np.float64  # (f64: the reference's width, 5.4 GB of streamed candidates)
    x0 = np.stack([rng.uniform(0, 2, B), rng.uniform(-2, 2, B)], axis=-1).astype(real)
    kw = dict(Nactor=Nh, mode="RQL", critic_struct="quadratic", Ncritic=4, buffer_size=10, gamma=1.0, dtype=dtype)
    eng = Engine(preset_engine_config("2tank", B, **kw))
    eng.set_state(x0)
    small = Engine(preset_engine_config("2tank", 16, **kw))
    small.set_state(x0[:16])
    sel = np.sort(rng.choice(B, 24, replace=False))
    cfg = oracle_cfg("2tank", n_actor=Nh, mode=O.MODE_RQL, critic_struct=O.CRITIC_QUADRATIC, n_critic=4,
                     buffer_size=10, gamma=1.0)
    env = O.new_batch(cfg, x0[sel].astype(np.float64))
    if streamed:
        cand1 = rng.random((K, Nh, 1), dtype=real)  # one candidate set, streamed per env
        cand = eng.to_device(np.ascontiguousarray(np.broadcast_to(cand1, (B, K, Nh, 1))))
        cand_small = small.to_device(np.ascontiguousarray(np.broadcast_to(cand1, (16, K, Nh, 1))))
        cand_or = cand1.astype(np.float64)
    else:
        cand = cand_small = None
        cand_or = O.grid_candidates(cfg, K)
    rep = PAR.TickReport()
    for t in range(T):
        eng.control_tick(cand, K=K)
        small.control_tick(cand_small, K=K)
        dev = {k: v[sel] for k, v in PAR.device_fields(eng, N, critic=True).items()}
        env = PAR.check_tick(cfg, env, cand_or, dev, tol=1e-5 if dtype == "f32" else 1e-9,
                             tol_over=None if dtype == "f32" else {"w_critic": 1e-6, "best_J": 1e-7}, report=rep, what=f"C3 t={t}")
    assert rep.ticks == T
    np.testing.assert_array_equal(eng.get_field(N.FIELD_STEP_IDX), np.full(B, T, np.int32))
    np.testing.assert_array_equal(eng.get_field(N.FIELD_STATUS), np.zeros(B, np.uint32))
    w = eng.get_field(N.FIELD_W_CRITIC)
    assert np.all(w >= 0.0) and np.all(w <= 1e3)  # Wmin / Wmax of the quadratic critic (controllers.py:1030-1031)
    # position independence, bit for bit
    np.testing.assert_array_equal(small.get_state(), eng.get_state()[:16])
    np.testing.assert_array_equal(small.get_field(N.FIELD_W_CRITIC), w[:16])
    np.testing.assert_array_equal(small.get_field(N.FIELD_ACCUM), eng.get_field(N.FIELD_ACCUM)[:16])
    # the fitted critic explains its own TD stack at least as well as the start point
    Jc_fit = eng.critic_cost()
    Jc_init = eng.critic_cost(np.ones((B, 6)))
    assert np.all(Jc_fit <= Jc_init * (1 + 1e-5) + 1e-6)
    summ, _ = eng.episode_stats(from_accum=True)
    assert summ["count"] == B and summ["n_failed"] == 0 and np.isfinite(summ["sum"])


@pytest.mark.parametrize("world,rank", [(1, 0), (8, 3)])
def test_mixed_pool_C5_shard(world, rank):
    """configs[4]: 3wrobot + 3wrobot_NI + 2tank, Nactor = 15, 16 x 16 (or 256-level) generated candidate grid;
    (8, 3): the shard rank 3 of 8 owns out of a 524288-env pool = 65536 envs, sharded within each type."""
    from rcognita_amd import _native as N
    from rcognita_amd.parallel import shard_by_type
    from rcognita_amd.pool import MixedPool

    total = 65536 * world
    counts = {"3wrobot": total // 3 + total % 3, "3wrobotNI": total // 3, "2tank": total // 3}
    pool = MixedPool(counts, rank=rank, world=world, Nactor=15, dtype="f32")
    spans = shard_by_type(counts, rank, world)
    assert pool.n_envs == sum(hi - lo for lo, hi in spans.values())
    assert abs(pool.n_envs - 65536) <= 3
    rng = np.random.default_rng(1234 + rank)
    states = {s.name: rand_states(rng, s.name, s.hi - s.lo) for s in pool.segments}
    pool.set_states(states)
    K, T = 256, 3
    from oracle import parity as PAR

    sels, envs, cfgs, grids = {}, {}, {}, {}
    for s in pool.segments:  # a sample of every segment against the oracle of its own system type, every tick, 1e-5
        n = s.hi - s.lo
        sels[s.name] = np.sort(rng.choice(n, 16, replace=False))
        cfgs[s.name] = oracle_cfg(s.name, n_actor=15)
        envs[s.name] = O.new_batch(cfgs[s.name], states[s.name][sels[s.name]].astype(np.float32).astype(np.float64))
        grids[s.name] = O.grid_candidates(cfgs[s.name], K)
    for t in range(T):
        pool.control_tick(K)
        for s in pool.segments:
            dev = {k: v[sels[s.name]] for k, v in PAR.device_fields(s.engine, N).items()}
            envs[s.name] = PAR.check_tick(cfgs[s.name], envs[s.name], grids[s.name], dev, tol=1e-5,
                                          what=f"C5 {s.name} t={t}")
    pool.synchronize()
    # the launch decisions behind the pool's rate (profiles/r04_ab_pool_gpw.txt): the 3-wheel robot's tick is ONE launch of the
    # hand-packed persistent kernel, 8 envs per wave at this size; the kinematic robot's stays two launches (its fused kernel
    # is slower), the decision on k_actor's packed instance; the tank has no packed rollout
    expect = {"3wrobot": ("k_ticks", 8, 8), "3wrobotNI": ("k_actor", 8, 1), "2tank": ("k_actor", 2, None)}
    for s in pool.segments:
        ll = s.engine.last_launch(N.KERNEL_ACTOR)
        kernel, variant, epw = expect[s.name]
        assert ll["kernel"] == kernel and ll["variant"] == variant, (s.name, ll)
        if epw is not None:
            assert ll["envs_per_wave"] == epw, (s.name, ll)
    total_summ, per = pool.episode_stats(from_accum=True)
    assert total_summ["count"] == pool.n_envs and total_summ["n_failed"] == 0
    for s in pool.segments:
        n = s.hi - s.lo
        np.testing.assert_array_equal(s.engine.get_field(N.FIELD_STEP_IDX), np.full(n, T, np.int32))
        assert abs(per[s.name]["count"] - n) == 0
    pool.close()


@pytest.mark.parametrize("B,K", [(32773, 64), (65536 + 9, 64), (32773, 100)])
def test_streamed_tick_with_several_envs_per_wave_and_a_ragged_tail(B, K):
    """Production kernel geometry: for B >= 16384 a wave owns several consecutive envs and writes their results once,
    coalesced; B is chosen so that the LAST wave owns fewer envs than the others.  Every env of the batch against the
    float64 oracle (streamed candidates shared by all envs), integer fields exact."""
    from rcognita_amd import Engine, _native as N
    from rcognita_amd.pool import preset_engine_config

    rng = np.random.default_rng(B)
    Nh, T = 5, 2  # (K = 100: every env's second tile is ragged, 36 rows - masked loads, env rows no longer tile-aligned)
    eng = Engine(preset_engine_config("3wrobot", B, Nactor=Nh))
    x0 = rand_states(rng, "3wrobot", B).astype(np.float32)
    eng.set_state(x0)
    cfg = oracle_cfg("3wrobot", n_actor=Nh)
    lo, hi = cfg.ctrl_bnds[:, 0], cfg.ctrl_bnds[:, 1]
    cand1 = (lo + (hi - lo) * rng.random((K, Nh, 2))).astype(np.float32)          # one candidate set ...
    cand = eng.to_device(np.ascontiguousarray(np.broadcast_to(cand1, (B, K, Nh, 2))))  # ... streamed per env
    env = O.new_batch(cfg, x0.astype(np.float64))
    # operator mode first (rcg_actor_cost: J of every candidate, staged per wave in LDS and written in one burst)
    J = eng.actor_cost(cand)
    ll = assert_kernel(eng, "k_actor_dma", N.DMA_MPC_G1)
    assert ll["envs_per_wave"] > 1, ll
    J_or = O.actor_cost(cand1.astype(np.float64)[None], x0.astype(np.float64)[:, None, :], x0.astype(np.float64)[:, None, :],
                        cfg, pars=env.pars)
    assert J.shape == (B, K) and rel_err_norm(J, J_or) < 1e-5
    from oracle import parity as PAR

    rep = PAR.TickReport()
    for t in range(T):  # EVERY env of the batch, every tick, 1e-5; near-ties (a handful in 32k argmins) are followed
        eng.control_tick(cand, K=K)
        env = PAR.check_tick(cfg, env, cand1.astype(np.float64), PAR.device_fields(eng, N), tol=1e-5, report=rep,
                             what=f"B={B} t={t}")
    assert rep.ties <= 0.002 * B * T
    np.testing.assert_array_equal(eng.get_field(N.FIELD_STEP_IDX), np.full(B, T, np.int32))
    summ, _ = eng.episode_stats(from_accum=True)
    assert summ["count"] == B and summ["n_failed"] == 0


@pytest.mark.parametrize("dtype,tol", [("f32", 1e-5), ("f64", 1e-11)])
@pytest.mark.parametrize("name,Nh,ref_lag", [("3wrobot", 20, False), ("3wrobotNI", 19, True), ("2tank", 40, False)])
def test_long_rows_closed_loop_on_the_production_kernel(name, Nh, ref_lag, dtype, tol):
    """Rows of up to 40 reals (the robots' Nactor = 20; f32 160 B, f64 320 B - four 64-row tiles of a block are then
    80 KB of LDS) on k_actor_dma: a streamed closed loop, a ragged last block, every env and every
    tick against the oracle (near-tied argmins followed), with and without the reference's one-step state lag."""
    from oracle import parity as PAR
    from rcognita_amd import Engine, _native as N
    from rcognita_amd.pool import preset_engine_config

    rng = np.random.default_rng(Nh)
    B, K, T = 2048 + 37, 128, 3
    eng = Engine(preset_engine_config(name, B, Nactor=Nh, dtype=dtype, ref_lag=ref_lag))
    x0 = rand_states(rng, name, B).astype(eng.real)
    eng.set_state(x0)
    cfg = oracle_cfg(name, n_actor=Nh, ref_lag=ref_lag)
    du = cfg.du
    lo, hi = cfg.ctrl_bnds[:, 0], cfg.ctrl_bnds[:, 1]
    cand1 = (lo + (hi - lo) * rng.random((K, Nh, du))).astype(eng.real)
    cand = eng.to_device(np.ascontiguousarray(np.broadcast_to(cand1, (B, K, Nh, du))))
    env = O.new_batch(cfg, x0.astype(np.float64))
    rep = PAR.TickReport()
    for t in range(T):
        eng.control_tick(cand, K=K)
        env = PAR.check_tick(cfg, env, cand1.astype(np.float64), PAR.device_fields(eng, N), tol=tol, report=rep,
                             what=f"{name} N={Nh} {dtype} t={t}")
    assert_kernel(eng, "k_actor_dma", N.DMA_MPC_G1)
    assert rep.ties <= 0.002 * B * T
    np.testing.assert_array_equal(eng.get_field(N.FIELD_STEP_IDX), np.full(B, T, np.int32))
    eng.close()


def test_C4_job_sharded_eight_ways_equals_the_unsharded_job():
    """configs[3] at FULL size by construction on one GPU: the 524288-env Sys3WRobot job (Nactor = 10, K = 256 streamed
    candidates: a 10.7 GB tensor, generated on the device) as ONE handle, and as the 8 shards `shard_range` gives 8 ranks
    (65536 envs each, run one after the other here, each on its slice of the same tensor).  The concatenation of the
    shards' per-env returns - what the ranks all_gather - must equal the unsharded job's bit for bit, the merged 6-number
    summaries must agree, integer counters exact, and a sample follows the oracle at 1e-5."""
    import torch

    from oracle import parity as PAR
    from rcognita_amd import Engine, _native as N
    from rcognita_amd.parallel import merge_summaries, shard_config, shard_range
    from rcognita_amd.pool import preset_engine_config

    total, world, K, Nh, T = 524288, 8, 256, 10, 3
    rng = np.random.default_rng(524288)
    x0 = rand_states(rng, "3wrobot", total).astype(np.float32)
    cfg = oracle_cfg("3wrobot", n_actor=Nh)
    g = torch.Generator(device="cuda")
    g.manual_seed(524288)
    lo = torch.tensor(cfg.ctrl_bnds[:, 0], device="cuda", dtype=torch.float32)
    hi = torch.tensor(cfg.ctrl_bnds[:, 1], device="cuda", dtype=torch.float32)
    cand = (torch.rand((total, K, Nh, 2), generator=g, device="cuda", dtype=torch.float32) * (hi - lo) + lo).contiguous()
    torch.cuda.synchronize()  # the handles below run on the legacy default stream, torch wrote on its own current stream
    base = preset_engine_config("3wrobot", total, Nactor=Nh)
    whole = Engine(base)
    whole.set_state(x0)
    sel = np.sort(rng.choice(total, 64, replace=False))
    cand_sel = cand[torch.as_tensor(sel, device="cuda")].cpu().numpy().astype(np.float64)
    env = O.new_batch(cfg, x0[sel].astype(np.float64))
    for t in range(T):
        whole.control_tick(cand, K=K)
        dev = {k: v[sel] for k, v in PAR.device_fields(whole, N).items()}
        env = PAR.check_tick(cfg, env, cand_sel, dev, tol=1e-5, what=f"C4 t={t}")
    ll = assert_kernel(whole, "k_actor_dma", N.DMA_MPC_G1)
    assert ll["envs_per_wave"] == 16, ll
    summ_whole, ret_whole = whole.episode_stats(from_accum=True, want_returns=True)
    np.testing.assert_array_equal(whole.get_field(N.FIELD_STEP_IDX), np.full(total, T, np.int32))
    act_whole = whole.get_field(N.FIELD_ACTION)
    whole.close()
    parts, rets, acts = [], [], []
    for r in range(world):
        ecfg, (a, b) = shard_config(base, total, r, world)
        assert (a, b) == shard_range(total, r, world) and b - a == 65536 and ecfg.env_id_base == a
        e = Engine(ecfg)
        e.set_state(x0[a:b])
        d = cand[a:b]  # this rank's rows of the job's tensor (contiguous: envs are the leading axis)
        for _ in range(T):
            e.control_tick(d, K=K)
        assert_kernel(e, "k_actor_dma", N.DMA_MPC_G1)
        s, ret = e.episode_stats(from_accum=True, want_returns=True)
        parts.append(s)
        rets.append(ret)
        acts.append(e.get_field(N.FIELD_ACTION))
        e.close()
    np.testing.assert_array_equal(np.concatenate(rets), ret_whole)   # the all_gather payload
    np.testing.assert_array_equal(np.concatenate(acts), act_whole)
    merged = merge_summaries(parts)
    assert merged["count"] == total == summ_whole["count"] and merged["n_failed"] == 0
    assert merged["min"] == summ_whole["min"] and merged["max"] == summ_whole["max"]
    np.testing.assert_allclose(merged["sum"], summ_whole["sum"], rtol=1e-12)
    del cand
    torch.cuda.empty_cache()


def test_f32_production_kernel_free_running_against_the_f64_build():
    """The float32 production path left to run FREE for 200 control ticks (streamed K = 256, Nactor = 10; no re-sync to an
    oracle, unlike the per-tick map checks of oracle/parity.py) beside the float64 build of the same kernel on the same
    states and the same (float32-exact) candidate rows.  What is claimed, and why:
      * a closed loop under an argmin is discontinuous: where two candidates are within rounding of each other the two
        element types may choose differently and the env's trajectories part for good.  Every such first divergence must
        BE a near-tie: at that tick both builds' costs of all K rows are read back (operator mode at the handles' own
        states); they agree within 1e-3 of the env's largest cost (rounding + the drift so far; measured 1.8e-4), and the float64 margin
        between the two chosen rows is no larger than twice the difference between the builds on those rows - i.e. the flip
        is explained by the float32 error, not by anything else.  Divergences must be rare (< 3 % of the envs in 200 ticks);
      * every env whose 200 decisions agree stays within 1e-4 of the float64 trajectory, relative to the largest magnitude
        its state reaches: float32 rounding (6e-8 per operation, ~100 operations per RK4 step) compounds over 200 steps of
        an integrator chain (position <- speed <- force); measured 9.1e-6, with 78 of 4096 envs (1.9 %) parting at a near-tie;
      * integer counters are exact."""
    import torch

    from rcognita_amd import Engine, _native as N
    from rcognita_amd.pool import preset_engine_config

    B, K, Nh, T = 4096, 256, 10, 200
    rng = np.random.default_rng(4096)
    x0 = rand_states(rng, "3wrobot", B).astype(np.float32)
    cfg = oracle_cfg("3wrobot", n_actor=Nh)
    lo, hi = cfg.ctrl_bnds[:, 0].astype(np.float32), cfg.ctrl_bnds[:, 1].astype(np.float32)
    cand32 = (lo + (hi - lo) * rng.random((B, K, Nh, 2), dtype=np.float32))
    e32 = Engine(preset_engine_config("3wrobot", B, Nactor=Nh, dtype="f32"))
    e64 = Engine(preset_engine_config("3wrobot", B, Nactor=Nh, dtype="f64"))
    e32.set_state(x0)
    e64.set_state(x0.astype(np.float64))
    d32, d64 = e32.to_device(cand32), e64.to_device(cand32.astype(np.float64))
    same = np.ones(B, dtype=bool)
    peak = np.abs(x0.astype(np.float64))
    worst = worst_J = 0.0
    n_div = 0
    J_TOL = 1e-3  # measured 1.8e-4: float32 rounding + up to 1e-5 of state drift in front of sin / cos of a heading of ~100 rad
    for t in range(T):
        e32.control_tick(d32, K=K)
        e64.control_tick(d64, K=K)
        b32, b64 = e32.get_field(N.FIELD_BEST_IDX), e64.get_field(N.FIELD_BEST_IDX)
        new = same & (b32 != b64)
        if new.any():  # first divergence of these envs
            # operator mode at each handle's (post-step) state = the costs this tick's decisions were taken on
            J64, J32 = e64.actor_cost(d64), e32.actor_cost(d32).astype(np.float64)
            idx = np.flatnonzero(new)
            scale = np.max(np.abs(J64[idx]), axis=1)
            agree = np.max(np.abs(J64[idx] - J32[idx]), axis=1) / scale
            worst_J = max(worst_J, float(agree.max()))
            assert np.all(agree <= J_TOL), (t, agree.max())
            margin = J64[idx, b32[idx]] - J64[idx, b64[idx]]  # >= 0: b64 is the float64 argmin
            explained = np.abs(J64[idx, b32[idx]] - J32[idx, b32[idx]]) + np.abs(J64[idx, b64[idx]] - J32[idx, b64[idx]])
            assert np.all(margin >= 0) and np.all(margin <= 2 * explained + 1e-300), (t, idx[:4], margin[:4], explained[:4])
            n_div += int(new.sum())
            same &= ~new
        s32, s64 = e32.get_state().astype(np.float64), e64.get_state()
        peak = np.maximum(peak, np.abs(s64))
        drift = np.abs(s32 - s64)[same] / np.maximum(peak[same].max(axis=1, keepdims=True), 1.0)
        worst = max(worst, float(drift.max()))
    assert_kernel(e32, "k_actor_dma", N.DMA_MPC_G1)
    assert_kernel(e64, "k_actor_dma", N.DMA_MPC_G1)
    print(f"\nFREE RUN f32 vs f64: {n_div} of {B} envs met a near-tie in {T} ticks; worst drift of the others {worst:.2e}; "
          f"worst cost disagreement at a divergence {worst_J:.2e} of the env's largest cost")
    assert n_div <= 0.03 * B, n_div
    assert worst <= 1e-4, worst
    np.testing.assert_array_equal(e32.get_field(N.FIELD_STEP_IDX), np.full(B, T, np.int32))
    np.testing.assert_array_equal(e64.get_field(N.FIELD_STEP_IDX), np.full(B, T, np.int32))
    a32, a64 = e32.get_field(N.FIELD_ACCUM).astype(np.float64), e64.get_field(N.FIELD_ACCUM)
    assert np.max(np.abs(a32 - a64)[same] / np.abs(a64[same])) <= 2e-4
    e32.close()
    e64.close()


def test_one_million_envs():
    """Size edge: B = 2^20 envs (16x the bench batch) through the tick with generated candidates, an episode reset in
    the middle.  Integer fields exact for every env, the summary consistent with the per-env returns, and a random
    sample of envs against the oracle."""
    from rcognita_amd import Engine, _native as N
    from rcognita_amd.pool import preset_engine_config

    rng = np.random.default_rng(2026)
    B, K, Nh, T = 1 << 20, 64, 5, 3
    eng = Engine(preset_engine_config("3wrobotNI", B, Nactor=Nh))
    x0 = rand_states(rng, "3wrobotNI", B).astype(np.float32)
    eng.set_state(x0)
    from oracle import parity as PAR

    sel = np.sort(rng.choice(B, 64, replace=False))
    cfg = oracle_cfg("3wrobotNI", n_actor=Nh)
    env = O.new_batch(cfg, x0[sel].astype(np.float64))
    grid = O.grid_candidates(cfg, K)
    for t in range(T):
        eng.control_tick(None, K=K)
        dev = {k: v[sel] for k, v in PAR.device_fields(eng, N).items()}
        env = PAR.check_tick(cfg, env, grid, dev, tol=1e-5, what=f"2^20 envs t={t}")
    summ, returns = eng.episode_stats(from_accum=True, want_returns=True)
    assert summ["count"] == B and summ["n_failed"] == 0
    np.testing.assert_allclose(summ["sum"], returns.astype(np.float64).sum(), rtol=1e-9)
    assert summ["min"] == returns.min() and summ["max"] == returns.max()
    np.testing.assert_array_equal(eng.get_field(N.FIELD_STEP_IDX), np.full(B, T, np.int32))
    eng.episode_reset()
    np.testing.assert_array_equal(eng.get_field(N.FIELD_EPISODE_IDX), np.ones(B, np.int32))
    np.testing.assert_array_equal(eng.get_field(N.FIELD_STEP_IDX), np.zeros(B, np.int32))
    np.testing.assert_array_equal(eng.get_state(), x0)
    np.testing.assert_array_equal(eng.get_field(N.FIELD_RETURNS), returns)
    eng.control_tick(None, K=K)
    np.testing.assert_array_equal(eng.get_field(N.FIELD_STEP_IDX), np.ones(B, np.int32))


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("name", ["3wrobot", "2tank"])
def test_vectorised_env_step_equals_the_scalar_kernel(name, dtype):
    """From 2^18 envs the 2tank env step runs on k_sim_v (16 bytes per lane and component, VEC consecutive envs per lane);
    below, and for the robots, on k_sim.  Same arithmetic: the first 4096 envs of a 2^18 batch must equal a 4096-env handle bit for bit,
    including an env that overflows in this step (frozen, flagged), an already frozen env, per-env parameters and several
    substeps; a sample follows the oracle."""
    from rcognita_amd import Engine, _native as N
    from rcognita_amd.pool import preset_engine_config

    rng = np.random.default_rng(18)
    B, Bs, S = 1 << 18, 4096, 3
    per_env = name == "3wrobot"
    big = Engine(preset_engine_config(name, B, Nactor=3, dtype=dtype, substeps_per_tick=S, per_env_pars=per_env))
    small = Engine(preset_engine_config(name, Bs, Nactor=3, dtype=dtype, substeps_per_tick=S, per_env_pars=per_env))
    x0 = rand_states(rng, name, B).astype(big.real)
    big_v = 3e38 if dtype == "f32" else 1.7e308  # goes non-finite inside the step (lane 1 of its vector)
    if name == "3wrobot":
        x0[5, 2], x0[5, 3] = 0.0, big_v  # k1 + 2 k2 + 2 k3 + k4 of x overflows
    else:
        x0[5, 1] = 1e30 if dtype == "f32" else 1e200  # K3 h2^2 overflows
    u0 = (rng.uniform(-1, 1, (B, big.du)) * (100 if name == "3wrobot" else 1)).astype(big.real)
    st = np.zeros(B, np.uint32)
    st[10] = 1  # frozen before the step
    for e, n in ((big, B), (small, Bs)):
        e.set_state(x0[:n])
        e.set_field(N.FIELD_ACTION, u0[:n])
        e.set_field(N.FIELD_STATUS, st[:n])
        if per_env:
            e.set_field(N.FIELD_PARS, np.stack([np.linspace(5, 20, B)[:n], np.linspace(0.5, 2, B)[:n]], axis=-1))
    for _ in range(2):
        big.sim_step(S)
        small.sim_step(S)
    for f in (N.FIELD_STATE, N.FIELD_STATE_PREV, N.FIELD_STATUS):
        np.testing.assert_array_equal(big.get_field(f)[:Bs], small.get_field(f), err_msg=str(f))
    s1 = big.get_field(N.FIELD_STATUS)
    assert s1[5] == 1 and s1[10] == 1 and s1.sum() == 2
    np.testing.assert_array_equal(big.get_state()[[5, 10]], x0[[5, 10]])  # frozen at their last finite state
    if not per_env:
        sel = np.sort(rng.choice(np.arange(64, B), 64, replace=False))
        cfg = oracle_cfg(name, n_actor=3, substeps_per_tick=S)
        env = O.new_batch(cfg, x0[sel].astype(np.float64), action0=u0[sel].astype(np.float64))
        for _ in range(2):
            O.sim_substeps(cfg, env, S)
        assert rel_err_norm(big.get_state()[sel], env.state) < (1e-5 if dtype == "f32" else 1e-11)


@pytest.mark.parametrize("name,mode,parts", [("2tank", "RQL", 2), ("3wrobot", "MPC", 3), ("3wrobotNI", "SQL", 2)])
def test_one_system_cut_into_parts_equals_the_single_handle(name, mode, parts):
    """MixedPool(parts=...): one system type as several handles on streams of their own (the critic fit of one part runs
    under the actor kernel of another).  Envs are independent, so every per-env field of the parts, concatenated, equals
    the single handle's bit for bit - streamed device candidates (ordered against the producer stream), ragged split."""
    import torch

    from rcognita_amd import Engine, _native as N
    from rcognita_amd.pool import MixedPool, preset_engine_config

    rng = np.random.default_rng(parts)
    B, K, Nh, T = 4099, 64, 6, 12
    kw = dict(mode=mode, critic_struct="quadratic", Ncritic=4, buffer_size=6) if mode != "MPC" else dict(mode="MPC")
    x0 = rand_states(rng, name, B).astype(np.float32)
    du = PRESETS[name]["sys_id"] == O.SYS_2TANK and 1 or 2
    bnds = np.array(PRESETS[name]["bnds"], dtype=np.float32)
    cand = torch.as_tensor(rng.uniform(bnds[:, 0], bnds[:, 1], (B, K, Nh, du)).astype(np.float32), device="cuda")
    torch.cuda.synchronize()
    one = Engine(preset_engine_config(name, B, Nactor=Nh, **kw))
    one.set_state(x0)
    pool = MixedPool({name: B}, Nactor=Nh, parts=parts, **kw)
    assert len(pool.segments) == parts and pool.n_envs == B and [s.off for s in pool.segments][0] == 0
    pool.set_states({name: x0})
    for _ in range(T):
        one.control_tick(cand, K=K)
        pool.control_tick(K, {name: cand}, producer_stream=torch.cuda.current_stream().cuda_stream)
    pool.synchronize()
    fields = [N.FIELD_STATE, N.FIELD_ACTION, N.FIELD_ACCUM, N.FIELD_BEST_IDX, N.FIELD_BEST_J, N.FIELD_STEP_IDX]
    if mode != "MPC":
        fields += [N.FIELD_W_CRITIC, N.FIELD_OBS_BUF, N.FIELD_ACT_BUF]
    for f in fields:
        np.testing.assert_array_equal(np.concatenate([s.engine.get_field(f) for s in pool.segments]), one.get_field(f))
    tot, _ = pool.episode_stats(from_accum=True)
    ref, _ = one.episode_stats(from_accum=True)
    assert tot["count"] == ref["count"] == B and tot["min"] == ref["min"] and tot["max"] == ref["max"]
    pool.close()
    one.close()


@pytest.mark.parametrize("use_device_array", [False, True])
def test_pool_with_candidates_refilled_every_tick(use_device_array):
    """The candidate tensor is REWRITTEN before every tick on the producer's stream while the pool's segments run on
    streams of their own: control_tick orders producer -> segments (rcg_wait_stream) AND segments -> producer
    (rcg_release_stream), so tick t reads tick t's candidates and the refill for tick t + 1 cannot overtake a segment
    that is still reading.  Checked against one handle driven on the producer's own stream (stream order = program order).
    Large rows, long ticks and three parts make an unordered refill visible; a DeviceArray input (no torch) is sliced per
    part as a view."""
    import torch

    from rcognita_amd import Engine, _native as N
    from rcognita_amd.pool import MixedPool, preset_engine_config

    rng = np.random.default_rng(7)
    name, B, K, Nh, T, parts = "3wrobot", 8193, 128, 10, 6, 3
    x0 = rand_states(rng, name, B).astype(np.float32)
    bnds = np.array(PRESETS[name]["bnds"], dtype=np.float32)
    stream = torch.cuda.current_stream().cuda_stream
    one = Engine(preset_engine_config(name, B, Nactor=Nh))
    one.set_stream(stream)
    one.set_state(x0)
    pool = MixedPool({name: B}, Nactor=Nh, parts=parts)
    pool.set_states({name: x0})
    gen = torch.Generator(device="cuda").manual_seed(3)
    lo, w = torch.as_tensor(bnds[:, 0], device="cuda"), torch.as_tensor(bnds[:, 1] - bnds[:, 0], device="cuda")
    cand_one = torch.empty((B, K, Nh, 2), device="cuda", dtype=torch.float32)
    cand_pool = torch.empty_like(cand_one)
    dev = pool.segments[0].engine.empty((B, K, Nh, 2)) if use_device_array else None
    for t in range(T):
        fresh = lo + w * torch.rand((B, K, Nh, 2), device="cuda", generator=gen)  # this tick's candidates
        cand_one.copy_(fresh)
        one.control_tick(cand_one, K=K)
        cand_pool.copy_(fresh)  # overwrites what the segments read last tick: legal only behind the reverse edge
        if use_device_array:
            dev.upload(fresh.cpu().numpy())  # synchronous, on segment 0's stream
            pool.synchronize()  # order the other segments behind it
            pool.control_tick(K, {name: dev}, ordered=True)
            pool.synchronize()  # ... and the next upload behind every segment
        else:
            pool.control_tick(K, {name: cand_pool}, producer_stream=stream)
    pool.synchronize()
    torch.cuda.synchronize()
    for f in (N.FIELD_STATE, N.FIELD_ACTION, N.FIELD_ACCUM, N.FIELD_BEST_IDX, N.FIELD_BEST_J, N.FIELD_STEP_IDX):
        np.testing.assert_array_equal(np.concatenate([s.engine.get_field(f) for s in pool.segments]), one.get_field(f))
    with pytest.raises(ValueError, match="rows"):
        pool.control_tick(K, {name: cand_pool[: B - 1]}, producer_stream=stream)
    pool.close()
    one.close()


def test_C5_pool_sharded_eight_ways_equals_the_unsharded_pool():
    """configs[4] by construction on one GPU: the mixed pool of 8 x 65536 + 5 envs (ragged: the 5 make the per-type
    shards of the ranks differ by one env) as ONE pool, and as the 8 per-rank pools `shard_by_type` gives 8 ranks, run one
    after the other here.  Every env's state comes from the job-wide table, so env g is the same env in both runs.  The
    per-type concatenation of the ranks' per-env returns (what the ranks all_gather, in rank order) equals the unsharded
    pool's bit for bit - with the device-side candidate search as the decision (k_actor_search: its Philox streams are keyed
    by the GLOBAL env id, so this also checks env_id_base of every rank's handles) as well as with the generated grid -, the
    merged 6-number summaries agree, counters exact."""
    from rcognita_amd import _native as N
    from rcognita_amd.parallel import merge_summaries, shard_by_type
    from rcognita_amd.pool import MixedPool

    world, T, K = 8, 2, 256
    total = 65536 * world + 5
    counts = {"3wrobot": total // 3 + total % 3, "3wrobotNI": total // 3, "2tank": total // 3}
    assert sum(counts.values()) == total
    rng = np.random.default_rng(55)
    job_states = {name: rand_states(rng, name, n).astype(np.float32) for name, n in counts.items()}

    def run(pool, states, search):
        pool.set_states(states)
        for t in range(T):
            if search:
                for s in pool.segments:
                    s.engine.control_tick_search(K=64, rounds=2, warm_start=True)
            else:
                pool.control_tick(K)
        pool.synchronize()
        out = {}
        for s in pool.segments:
            summ, ret = s.engine.episode_stats(from_accum=True, want_returns=True)
            np.testing.assert_array_equal(s.engine.get_field(N.FIELD_STEP_IDX), np.full(s.hi - s.lo, T, np.int32))
            out[s.name] = (summ, ret, s.engine.get_field(N.FIELD_ACTION))
        return out

    for search in (False, True):
        whole_pool = MixedPool(counts, Nactor=15, dtype="f32", seed=9)
        whole = run(whole_pool, job_states, search)
        whole_pool.close()
        rets = {name: [] for name in counts}
        acts = {name: [] for name in counts}
        summs = []
        sizes = []
        for r in range(world):
            spans = shard_by_type(counts, r, world)
            pool = MixedPool(counts, rank=r, world=world, Nactor=15, dtype="f32", seed=9)
            for s in pool.segments:
                assert (s.lo, s.hi) == spans[s.name] and s.engine.cfg.env_id_base == s.lo
            sizes.append(pool.n_envs)
            res = run(pool, {name: job_states[name][lo:hi] for name, (lo, hi) in spans.items()}, search)
            for name, (summ, ret, act) in res.items():
                rets[name].append(ret)
                acts[name].append(act)
                summs.append(summ)
            pool.close()
        assert sum(sizes) == total and max(sizes) - min(sizes) <= 3  # ragged: per-type shards differ by one env
        for name in counts:
            np.testing.assert_array_equal(np.concatenate(rets[name]), whole[name][1], err_msg=f"{name} search={search}")
            np.testing.assert_array_equal(np.concatenate(acts[name]), whole[name][2])
        m, w = merge_summaries(summs), merge_summaries([v[0] for v in whole.values()])
        assert m["count"] == w["count"] == total and m["min"] == w["min"] and m["max"] == w["max"]
        np.testing.assert_allclose(m["sum"], w["sum"], rtol=1e-9)
