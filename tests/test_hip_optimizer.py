"""SURVEY 8f row f1 on the GPU: rcg_actor_optimize / rcg_control_tick_opt against the oracle twin and against the
cost the reference's SLSQP reaches (tests/golden/F8_slsqp_actor_*.npz).  ``gpu`` marked."""
import numpy as np
import pytest

from oracle import rcg_oracle as O
from tests.conftest import load_golden
from tests.helpers import PRESETS, SYSTEMS, assert_kernel, both, rand_states, rel_err_norm

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("name", SYSTEMS)
def test_optimizer_vs_oracle_and_reference_slsqp(name, dtype):
    meta, z = load_golden(f"F8_slsqp_actor_{name}")
    x = z["state"]
    B = x.shape[0]
    ai = [0.5] if name == "2tank" else None
    eng, cfg = both(name, B, dtype, n_actor=meta["N"], gamma=meta["gamma"], pred_step_size=meta["pred_step_size"],
                    **({"action_init": ai} if ai else {}))
    eng.set_state(x)
    act, U, J, its = eng.actor_optimize(iters=10)
    lo, hi = cfg.ctrl_bnds[:, 0], cfg.ctrl_bnds[:, 1]
    assert np.all(U >= lo - 1e-4) and np.all(U <= hi + 1e-4)
    np.testing.assert_array_equal(act, U[:, 0, :])
    # the reported cost is the oracle's _actor_cost of the reported sequence
    J_chk = O.actor_cost(U.astype(np.float64), x, x, cfg)
    assert rel_err_norm(J, J_chk) < (1e-10 if dtype == "f64" else 1e-5)
    # quality: never worse than the start, within 0.2 % of the reference's SLSQP optimum
    assert np.all(J <= z["J_init"] * (1 + 1e-6))
    ratio = J / z["J_opt"]
    assert np.median(ratio) < 1.0005 and np.max(ratio) < 1.002, (np.median(ratio), np.max(ratio))
    # same algorithm as the oracle twin
    U_or, J_or, its_or = O.actor_optimize(cfg, x, x, O.action_sqn_init(cfg, ai), iters=10)
    if dtype == "f64":
        # iteration counts may differ by trailing iterations whose improvement is at rounding level
        assert np.all(np.abs(its - its_or) <= 3)
        assert rel_err_norm(J, J_or) < 1e-9 and rel_err_norm(U, U_or, floor=float(np.max(hi))) < 1e-5
    else:  # f32 may take another branch of the discrete line search; the cost reached must agree
        assert rel_err_norm(J, J_or) < 2e-4


@pytest.mark.parametrize("dtype,N", [("f64", 16), ("f64", 24), ("f32", 30)])
def test_optimizer_long_horizons_large_lds(dtype, N):
    """Horizons whose per-block LDS (u, d, states of 16 envs x 4 waves) exceeds the 64 KB default limit: the launcher
    raises the kernel's dynamic-LDS attribute (f64: N = 16 needs 80 KB, N = 24 119 KB; f32: N = 30 74 KB).  Result
    against the oracle twin; a ragged batch (B = 37: a wave with 5 of its 16 envs)."""
    rng = np.random.default_rng(N)
    B = 37
    eng, cfg = both("3wrobot", B, dtype, n_actor=N)
    x = rand_states(rng, "3wrobot", B)
    eng.set_state(x)
    act, U, J, its = eng.actor_optimize(iters=4)
    xin = x.astype(eng.real).astype(np.float64)
    J_chk = O.actor_cost(U.astype(np.float64), xin, xin, cfg)
    assert rel_err_norm(J, J_chk) < (1e-10 if dtype == "f64" else 1e-5)
    U_or, J_or, its_or = O.actor_optimize(cfg, xin, xin, O.action_sqn_init(cfg, None), iters=4)
    assert rel_err_norm(J, J_or) < (1e-9 if dtype == "f64" else 5e-4)
    assert np.all(J <= O.actor_cost(np.tile(O.action_sqn_init(cfg, None), (B, 1, 1)), xin, xin, cfg) * (1 + 1e-6))


@pytest.mark.parametrize("warm", [False, True])
@pytest.mark.parametrize("name,N,ref_lag", [("3wrobot", 6, False), ("3wrobotNI", 4, True), ("2tank", 9, False)])
def test_control_tick_opt_closed_loop_vs_oracle(name, N, ref_lag, warm):
    from rcognita_amd import _native as Nn

    rng = np.random.default_rng(31)
    B, T = 9, 5
    eng, cfg = both(name, B, "f64", n_actor=N, substeps_per_tick=2, ref_lag=ref_lag, gamma=0.98)
    x0 = rand_states(rng, name, B)
    eng.set_state(x0)
    env = O.new_batch(cfg, x0)
    for t in range(T):
        eng.control_tick_opt(iters=6, warm_start=warm)
        O.control_tick_opt(cfg, env, 6, warm_start=warm)
        assert np.all(np.abs(eng.get_field(Nn.FIELD_BEST_IDX) - env.best_idx) <= 3)  # iterations used (see above)
        # the quasi-Newton direction divides by curvature estimates: the kernel's fused multiply-adds and numpy's separate
        # roundings move the iterates by ~1e-9 of the box, the cost reached by less (it is stationary there)
        assert rel_err_norm(eng.get_field(Nn.FIELD_BEST_J), env.best_J) < 1e-7, t
        assert rel_err_norm(eng.get_field(Nn.FIELD_ACTION_SQN), env.action_sqn,
                            floor=float(np.max(cfg.ctrl_bnds))) < 1e-4, t
        assert rel_err_norm(eng.get_state(), env.state) < 1e-9
        assert rel_err_norm(eng.get_field(Nn.FIELD_ACCUM), env.accum, floor=float(np.max(np.abs(env.accum)))) < 1e-6
        np.testing.assert_array_equal(eng.get_field(Nn.FIELD_STEP_IDX), env.step_idx)
        # every tick is checked as a map: the oracle continues from the device's values
        env.state = eng.get_state().astype(np.float64)
        env.state_prev = eng.get_field(Nn.FIELD_STATE_PREV).astype(np.float64)
        env.action = eng.get_field(Nn.FIELD_ACTION).astype(np.float64)
        env.accum = eng.get_field(Nn.FIELD_ACCUM).astype(np.float64)
        env.action_sqn = eng.get_field(Nn.FIELD_ACTION_SQN).astype(np.float64)


def test_optimizer_beats_grid_search_in_closed_loop():
    """Closed-loop quality on the 3wrobot preset: accumulated objective after 1 s with the on-device optimiser vs
    the 256-candidate constant-sequence grid (SLSQP 389.0 vs grid 435.5 after 3 s at N = 5)."""
    from rcognita_amd import _native as Nn

    p = PRESETS["3wrobot"]
    x0 = np.array([p["x0"]], dtype=float)
    acc = {}
    for tag in ("grid", "opt"):
        eng, cfg = both("3wrobot", 1, "f64", n_actor=5)
        eng.set_state(x0)
        for _ in range(100):
            if tag == "grid":
                eng.control_tick(None, K=256)
            else:
                eng.control_tick_opt(iters=8)
        acc[tag] = float(eng.get_field(Nn.FIELD_ACCUM)[0])
        assert np.all(np.isfinite(eng.get_state()))
    print(acc)
    assert acc["opt"] < acc["grid"]


CRITIC_CASES = [("3wrobotNI", "quad-nomix"), ("3wrobotNI", "quad-mix"), ("3wrobot", "quad-nomix"), ("2tank", "quad-nomix"),
                ("2tank", "quadratic"), ("2tank", "quad-lin")]


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("mode", ["RQL", "SQL"])
@pytest.mark.parametrize("name,cs", CRITIC_CASES)
def test_optimizer_in_the_critic_modes_vs_reference_slsqp(name, cs, mode, dtype):
    """Fixtures F8c: decisions of the reference's own closed loop in RQL / SQL (state_sys != obs, the critic weights the
    reference had fitted at that tick, SLSQP's result).  k_actor_opt, 30 iterations from action_sqn_init: the cost it
    reports is the oracle's _actor_cost of the sequence it reports; it never ends above the start and ends within 0.5 % of
    SLSQP's cost on every decision; it follows its oracle twin."""
    from rcognita_amd import _native as Nn

    meta, z = load_golden(f"F8c_slsqp_actor_{name}_{mode}_{cs}")
    B = z["state"].shape[0]
    ai = [0.5] if name == "2tank" else None
    eng, cfg = both(name, B, dtype, n_actor=meta["N"], mode=O.MODE_IDS[mode], gamma=meta["gamma"],
                    critic_struct=O.CRITIC_IDS[cs], pred_step_size=meta["pred_step_size"], buffer_size=10,
                    **({"action_init": ai} if ai else {}))
    eng.set_field(Nn.FIELD_W_CRITIC, z["w"])
    act, U, J, its = eng.actor_optimize(iters=30, obs=z["obs"], state_sys=z["state"])
    ll = assert_kernel(eng, "k_actor_opt")
    assert ll["variant"] & 1, ll  # the generic instance
    lo, hi = cfg.ctrl_bnds[:, 0], cfg.ctrl_bnds[:, 1]
    assert np.all(U >= lo - 1e-4) and np.all(U <= hi + 1e-4)
    np.testing.assert_array_equal(act, U[:, 0, :])
    r = eng.real
    obs_r, st_r, w_r = (z[k].astype(r).astype(np.float64) for k in ("obs", "state", "w"))
    J_chk = O.actor_cost(U.astype(np.float64), obs_r, st_r, cfg, w_critic=w_r)
    # J is a difference of large terms when the fitted weights have both signs (quad-lin, quad-mix: |w| up to 1e3): the
    # scale of a float32 evaluation is the cost with every term taken positive
    scale = np.maximum(np.maximum(np.abs(z["J_init"]), O.actor_cost(U.astype(np.float64), obs_r, st_r, cfg, w_critic=np.abs(w_r))), 1.0)
    assert np.max(np.abs(J - J_chk) / scale) < (1e-10 if dtype == "f64" else 1e-5)
    assert np.all(J <= z["J_init"] + (1e-9 if dtype == "f64" else 2e-5) * scale)
    gap = (J - z["J_opt"]) / np.maximum(np.abs(z["J_opt"]), 1e-6 * scale)
    print(f"\nF8c {name} {mode} {cs} {dtype}: J / J_slsqp - 1: median {np.median(gap):.2e} max {np.max(gap):.2e}; "
          f"accepted steps {its.min()}..{its.max()}")
    assert np.max(gap) < 5e-3, (np.median(gap), np.max(gap))
    U_or, J_or, its_or = O.actor_optimize(cfg, obs_r, st_r, O.action_sqn_init(cfg, ai), iters=30, w_critic=w_r)
    # the quasi-Newton path divides by curvature estimates: rounding-level differences between the kernel's fused
    # multiply-adds and numpy move the iterates, not the cost reached
    twin = float(np.max(np.abs(J - J_or) / scale))
    print(f"F8c {name} {mode} {cs} {dtype}: |J - J_twin| / scale after 30 iterations: {twin:.2e} (gap to SLSQP above: the real guard)")
    assert twin < (1e-7 if dtype == "f64" else 1e-3)  # measured (profiles/r05_optimizer_and_search_quality.txt): <= 3.9e-9 / 4.0e-4


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("name", SYSTEMS)
def test_optimizer_on_every_decision_of_the_reference_mpc_loop(name, dtype):
    """``mpc_tick_*`` of the F7c fixtures: every decision of the reference's own closed MPC loop - its observation, its
    one-step-lagged state_sys, SLSQP's cost.  k_actor_opt (the MPC instance) ends within 0.5 % of SLSQP at every tick: the
    per-decision anchor under the closed-loop bands of tests/test_hip_ref_traces.py."""
    files = {"3wrobotNI": "F7c_trace_3wrobotNI_RQL_quad-nomix", "3wrobot": "F7c_trace_3wrobot_RQL_quad-nomix",
             "2tank": "F7c_trace_2tank_RQL_quad-nomix"}
    meta, z = load_golden(files[name])
    B = z["mpc_tick_obs"].shape[0]
    ai = [0.5] if name == "2tank" else None
    eng, cfg = both(name, B, dtype, n_actor=meta["Nactor"], gamma=1.0, pred_step_size=meta["pred_step_size"],
                    **({"action_init": ai} if ai else {}))
    act, U, J, its = eng.actor_optimize(iters=30, obs=z["mpc_tick_obs"], state_sys=z["mpc_tick_state_sys"])
    ll = assert_kernel(eng, "k_actor_opt")
    assert not (ll["variant"] & 1), ll  # the MPC / diagonal instance
    gap = (J - z["mpc_tick_J"]) / np.abs(z["mpc_tick_J"])
    print(f"\nMPC ticks {name} {dtype}: J / J_slsqp - 1: median {np.median(gap):.2e} max {np.max(gap):.2e} over {B} decisions")
    assert np.max(gap) < 5e-3
    assert np.all(J <= z["mpc_tick_J_init"] * (1 + (1e-9 if dtype == "f64" else 1e-5)))


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("name", SYSTEMS)
@pytest.mark.parametrize("mode,stage", [("MPC", "full"), ("MPC", "biquad"), ("RQL", "full"), ("RQL", "biquad")])
def test_optimizer_with_full_and_biquadratic_stage_costs(name, mode, stage, dtype):
    """A full (SPD) R1 and the biquadratic structure (controllers.py:1076-1082) through the generic instance: reported
    cost = oracle cost of the reported sequence, monotone, the oracle twin's cost."""
    import zlib

    rng = np.random.default_rng(zlib.crc32(f"{name} {mode} {stage}".encode()))
    B, Nh = 21, 6
    p = PRESETS[name]
    n = len(p["R1"])
    A = rng.uniform(-1, 1, (n, n))
    kw = dict(n_actor=Nh, mode=O.MODE_IDS[mode], gamma=0.97, R1=A @ A.T + np.diag(p["R1"]), buffer_size=8,
              critic_struct=O.CRITIC_QUADRATIC)
    if stage == "biquad":
        Bm = rng.uniform(-1, 1, (n, n))
        kw.update(R2=1e-4 * (Bm @ Bm.T), stage_obj_struct=O.STAGE_BIQUADRATIC)
    eng, cfg = both(name, B, dtype, **kw)
    x = rand_states(rng, name, B)
    obs = x + rng.uniform(-0.02, 0.02, x.shape)
    w = rng.uniform(0, 2, (B, cfg.dc))
    from rcognita_amd import _native as Nn

    eng.set_field(Nn.FIELD_W_CRITIC, w)
    act, U, J, its = eng.actor_optimize(iters=12, obs=obs, state_sys=x)
    assert_kernel(eng, "k_actor_opt")
    r = eng.real
    obs_r, x_r, w_r = obs.astype(r).astype(np.float64), x.astype(r).astype(np.float64), w.astype(r).astype(np.float64)
    J_chk = O.actor_cost(U.astype(np.float64), obs_r, x_r, cfg, w_critic=w_r)
    # float32: chi R1 chi with a full matrix sums 49 products of both signs per step (inputs up to 300: single terms of 1e5
    # against a total of 1e3): the rounding of the terms, not of the total, sets the error - 5e-5 of |J| measured at most
    assert rel_err_norm(J, J_chk) < (1e-10 if dtype == "f64" else 5e-5)
    u0 = O.action_sqn_init(cfg, None)
    J0 = O.actor_cost(np.broadcast_to(u0, (B,) + u0.shape), obs_r, x_r, cfg, w_critic=w_r)
    assert np.all(J <= J0 * (1 + 1e-6) + 1e-9)
    U_or, J_or, _ = O.actor_optimize(cfg, obs_r, x_r, u0, iters=12, w_critic=w_r)
    assert rel_err_norm(J, J_or) < (1e-6 if dtype == "f64" else 5e-3)  # f32: another branch of the discrete line search


@pytest.mark.parametrize("name,mode,cs,N", [("2tank", "RQL", "quadratic", 8), ("3wrobotNI", "SQL", "quad-mix", 4),
                                           ("3wrobot", "RQL", "quad-nomix", 5)])
def test_control_tick_opt_closed_loop_in_the_critic_modes_vs_oracle(name, mode, cs, N):
    """rcg_control_tick_opt in RQL / SQL: env step + buffer push + critic fit (one launch), then the optimiser on the fitted
    weights - tick by tick against the oracle twin, every tick checked as a map from the device's own pre-tick values."""
    from rcognita_amd import _native as Nn

    rng = np.random.default_rng(77)
    B, T = 11, 6
    ai = [0.5] if name == "2tank" else None
    eng, cfg = both(name, B, "f64", n_actor=N, mode=O.MODE_IDS[mode], critic_struct=O.CRITIC_IDS[cs], n_critic=4,
                    buffer_size=6, **({"action_init": ai} if ai else {}))
    x0 = rand_states(rng, name, B) * 0.3
    eng.set_state(x0)
    env = O.new_batch(cfg, x0, action0=ai)
    for t in range(T):
        eng.control_tick_opt(iters=8)
        O.control_tick_opt(cfg, env, 8, action_init=ai)
        assert_kernel(eng, "k_actor_opt")
        assert rel_err_norm(eng.get_state(), env.state) < 1e-9, t
        # critic weights: a bounded least squares regularised at 1e-8 of its scale
        assert rel_err_norm(eng.get_field(Nn.FIELD_W_CRITIC), env.w_critic, floor=1.0) < 1e-6, t
        scale = np.maximum(np.abs(env.best_J), 1.0)
        assert np.max(np.abs(eng.get_field(Nn.FIELD_BEST_J) - env.best_J) / scale) < 1e-6, t
        assert rel_err_norm(eng.get_field(Nn.FIELD_ACCUM), env.accum, floor=float(np.max(np.abs(env.accum)) + 1e-9)) < 1e-5
        np.testing.assert_array_equal(eng.get_field(Nn.FIELD_STEP_IDX), env.step_idx)
        # continue the oracle from the device's values: each tick is checked as a map (oracle/parity.py's rule)
        env.state = eng.get_state().astype(np.float64)
        env.state_prev = eng.get_field(Nn.FIELD_STATE_PREV).astype(np.float64)
        env.action = eng.get_field(Nn.FIELD_ACTION).astype(np.float64)
        env.accum = eng.get_field(Nn.FIELD_ACCUM).astype(np.float64)
        env.w_critic = eng.get_field(Nn.FIELD_W_CRITIC).astype(np.float64)
        env.w_prev = eng.get_field(Nn.FIELD_W_PREV).astype(np.float64)
        env.obs_buf = eng.get_field(Nn.FIELD_OBS_BUF).astype(np.float64)
        env.act_buf = eng.get_field(Nn.FIELD_ACT_BUF).astype(np.float64)


def test_optimizer_memory_setting_and_refusals():
    """rcg_set_optimizer: 0 .. 8 pairs; memory 0 is round 3's steepest descent and equals the oracle's memory = 0; a horizon
    whose pairs do not fit the CU's LDS is refused before the env is stepped."""
    from rcognita_amd import _native as Nn

    rng = np.random.default_rng(3)
    B = 19
    eng, cfg = both("3wrobot", B, "f64", n_actor=7)
    x = rand_states(rng, "3wrobot", B)
    eng.set_state(x)
    for bad in (-2, 9):
        with pytest.raises(Nn.NativeError) as ei:
            eng.set_optimizer(bad)
        assert ei.value.code == Nn.ERR_BAD_ARG
    for mem in (0, 2, 8):
        eng.set_optimizer(mem)
        act, U, J, its = eng.actor_optimize(iters=6)
        U_or, J_or, _ = O.actor_optimize(cfg, x, x, O.action_sqn_init(cfg, None), iters=6, memory=mem)
        assert rel_err_norm(J, J_or) < 1e-8, mem
    big, _ = both("3wrobot", 4, "f64", n_actor=32)  # R = 64 doubles: 8 pairs need 16 * 20 * 64 * 8 B > 160 KB per wave
    big.set_state(x[:4])
    big.set_optimizer(8)
    before = big.get_state().copy()
    with pytest.raises(Nn.NativeError) as ei:
        big.control_tick_opt(iters=2)
    assert ei.value.code == Nn.ERR_UNSUPPORTED
    np.testing.assert_array_equal(big.get_state(), before)  # refused before the env step
    np.testing.assert_array_equal(big.get_field(Nn.FIELD_STEP_IDX), np.zeros(4, np.int32))
    big.set_optimizer(2)
    big.control_tick_opt(iters=2)


def test_full_size_optimizer_tick():
    """B = 65536 envs, Nactor = 10, 8 iterations: counters exact, sequences inside the box, cost never above the
    start sequence's cost, a sample follows the oracle."""
    from rcognita_amd import _native as Nn

    B, Nh = 65536, 10
    rng = np.random.default_rng(1234)
    eng, cfg = both("3wrobot", B, "f32", n_actor=Nh)
    x = rand_states(rng, "3wrobot", B)
    eng.set_state(x)
    eng.control_tick_opt(iters=8)
    np.testing.assert_array_equal(eng.get_field(Nn.FIELD_STEP_IDX), np.ones(B, np.int32))
    U = eng.get_field(Nn.FIELD_ACTION_SQN)
    lo, hi = cfg.ctrl_bnds[:, 0], cfg.ctrl_bnds[:, 1]
    assert np.all(U >= lo - 1e-3) and np.all(U <= hi + 1e-3)
    assert np.all(np.isfinite(eng.get_state()))
    its = eng.get_field(Nn.FIELD_BEST_IDX)
    assert its.min() >= 0 and its.max() <= 8
    # the reported cost is the oracle's cost of the reported sequence from the (post-step) state, and it is not
    # above the cost of the start sequence
    sel = np.sort(rng.choice(B, 64, replace=False))
    st = eng.get_state()[sel].astype(np.float64)
    J_chk = O.actor_cost(U[sel].astype(np.float64), st, st, cfg)
    assert rel_err_norm(eng.get_field(Nn.FIELD_BEST_J)[sel], J_chk) < 1e-5
    J_start = O.actor_cost(np.broadcast_to(O.action_sqn_init(cfg), (64, Nh, 2)), st, st, cfg)
    assert np.all(J_chk <= J_start * (1 + 1e-6))


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("name,mode", [("3wrobot", "MPC"), ("3wrobotNI", "MPC"), ("2tank", "MPC"), ("2tank", "RQL"), ("3wrobotNI", "SQL")])
def test_stopping_tolerance_vs_oracle(name, mode, dtype):
    """rcg_set_optimizer_tol: an env is done after an accepted step that gained <= ftol.  f64: the oracle twin's walk with the same
    ftol (same point, same number of accepted steps, fewer than without the test); ftol = 0 is the handle's default and changes
    nothing; a huge ftol stops every env after its first accepted step; a negative or non-finite one is refused.  f32: the gains
    are compared in float32 - the cost reached agrees with the f64 twin to the f32 tolerance and no env takes more steps."""
    from rcognita_amd import _native as Nn

    rng = np.random.default_rng(11)
    B = 37
    kw = dict(n_actor=6)
    if mode != "MPC":
        kw.update(mode=O.MODE_IDS[mode], critic_struct=O.CRITIC_IDS["quad-nomix"], buffer_size=8, n_critic=4)
    eng, cfg = both(name, B, dtype, **kw)
    x = rand_states(rng, name, B)
    eng.set_state(x)
    w = None
    if mode != "MPC":
        w = rng.uniform(0.1, 2.0, (B, cfg.dc))
        eng.set_field(Nn.FIELD_W_CRITIC, w)
    u0 = O.action_sqn_init(cfg, None)
    _, U_a, J_a, n_a = eng.actor_optimize(iters=30)
    eng.set_optimizer(-1, ftol=0.0)
    _, U_z, J_z, n_z = eng.actor_optimize(iters=30)
    np.testing.assert_array_equal(U_a, U_z)
    np.testing.assert_array_equal(n_a, n_z)
    for bad in (-1e-9, float("nan"), float("inf")):
        with pytest.raises(Nn.NativeError) as ei:
            eng.set_optimizer(-1, ftol=bad)
        assert ei.value.code == Nn.ERR_BAD_ARG
    ftol = 1e-5
    eng.set_optimizer(-1, ftol=ftol)
    _, U, J, n = eng.actor_optimize(iters=30)
    U_or, J_or, n_or = O.actor_optimize(cfg, x, x, u0, iters=30, w_critic=w, ftol=ftol)
    assert np.all(n <= n_a) and np.all(n >= np.minimum(n_a, 1))
    if dtype == "f64":
        same = n == n_or  # (a gain within rounding of ftol may fall on either side)
        assert same.mean() > 0.9, (n, n_or)
        assert rel_err_norm(J[same], J_or[same]) < 1e-9
        assert rel_err_norm(U[same], U_or[same], floor=float(np.max(cfg.ctrl_bnds[:, 1]))) < 1e-5
        assert np.any(n < n_a), "no walk was cut short: the test has no case"
    else:
        assert rel_err_norm(J, J_or) < 2e-4
    assert np.all(J >= J_a * (1 - (1e-12 if dtype == "f64" else 1e-5)))
    eng.set_optimizer(-1, ftol=1e30)
    _, _, _, n1 = eng.actor_optimize(iters=30)
    assert np.all(n1 <= 1)
    eng.set_optimizer(-1, ftol=0.0)
