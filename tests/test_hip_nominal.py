"""Nominal controllers on the GPU: rcg_nominal_action / rcg_control_tick_nominal and the mirror
classes CtrlNominal3WRobot / CtrlNominal3WRobotNI against oracle/nominal_oracle.py and the reference's own outputs
(tests/golden/F10_nominal_*.npz)."""
import numpy as np
import pytest

from oracle import nominal_oracle as NO
from oracle import rcg_oracle as O
from tests.conftest import load_golden
from tests.helpers import both, rand_states, rel_err_norm

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_ni_nominal_matches_reference_outputs(dtype):
    """Closed-form controller: the HIP result IS the reference's result (f64: 1e-10; f32: 1e-5 of the action scale)."""
    meta, z = load_golden("F10_nominal_3wrobotNI")
    x = z["state"]
    eng, _ = both("3wrobotNI", x.shape[0], dtype)
    xin = x.astype(eng.real).astype(np.float64)  # what the device sees
    a_van, L = eng.nominal_action(x, meta["ctrl_gain"], clip=False, want_lyap=True)
    a_clip = eng.nominal_action(x, meta["ctrl_gain"], clip=True)
    if dtype == "f64":
        np.testing.assert_allclose(a_van, z["action_vanila"], rtol=1e-10, atol=1e-12, equal_nan=True)
        np.testing.assert_allclose(a_clip, z["action"], rtol=1e-10, atol=1e-12, equal_nan=True)
        np.testing.assert_allclose(L, z["LF"], rtol=1e-10, atol=1e-12, equal_nan=True)
    else:
        ref = NO.nominal_action_ni(xin, meta["ctrl_gain"])
        ok = np.all(np.isfinite(ref), axis=1)
        assert np.array_equal(np.all(np.isfinite(a_van), axis=1), ok)
        # the kernel evaluates the law in float64 on the f32 states: only the final rounding differs
        assert rel_err_norm(a_van[ok], ref[ok]) < 1e-5
        assert rel_err_norm(a_clip[ok], NO.nominal_action_ni(xin, meta["ctrl_gain"], meta["bnds"])[ok]) < 1e-5
        assert rel_err_norm(L[ok], NO.lyapunov_ni(xin)[ok]) < 1e-5
    assert np.all(np.isnan(a_clip[8:12]))  # exact origin: NaN in the reference, NaN here


def test_endi_nominal_vs_oracle_and_reference_f64():
    """(a) HIP == oracle (same theta search, same arithmetic); (b) the reference's own minimiser (trust-constr from
    theta = 0) on > 90 % of the fixture states and Fc(theta*) not above the reference's on >= 95 %; (c) the clipped
    actions agree with the reference's on > 90 % of ALL states."""
    meta, z = load_golden("F10_nominal_3wrobot")
    x = z["state"]
    eng, _ = both("3wrobot", x.shape[0], "f64")
    pars = [meta["m"], meta["I"]]
    a, L = eng.nominal_action(x, meta["ctrl_gain"], ctrl_pars=pars, clip=True, want_lyap=True)
    a_or = NO.nominal_action_endi(x, meta["ctrl_gain"], meta["m"], meta["I"], meta["bnds"])
    L_or = NO.lyapunov_endi(x)
    np.testing.assert_allclose(L, L_or, rtol=1e-9)
    # the action is a cube root of theta-dependent terms: golden section pins theta to ~1e-9
    bad = np.abs(a - a_or) > 1e-5 * (np.abs(a_or) + 1)
    assert bad.mean() < 0.02, bad.mean()
    assert np.mean(L <= z["Fc_star"] * (1 + 1e-9) + 1e-12) >= 0.95
    xNI, eta = NO.cart2nh(x)
    same = np.abs(np.angle(np.exp(1j * (NO.theta_star(xNI, eta) - z["theta_star"])))) < 1e-3
    close = np.all(np.abs(a - z["action"]) <= 2e-2 * (np.abs(z["action"]) + 1), axis=1)
    assert same.mean() > 0.9 and close[same].mean() > 0.95 and close.mean() > 0.9
    # handle pars are the default controller parameters
    a2 = eng.nominal_action(x, meta["ctrl_gain"], clip=True)
    np.testing.assert_array_equal(a2, a)


def test_endi_nominal_f32_reaches_the_same_lyapunov_value():
    meta, z = load_golden("F10_nominal_3wrobot")
    x = z["state"]
    eng, _ = both("3wrobot", x.shape[0], "f32")
    xin = x.astype(np.float32).astype(np.float64)
    a, L = eng.nominal_action(x, meta["ctrl_gain"], clip=True, want_lyap=True)
    L_or = NO.lyapunov_endi(xin)
    assert np.all(np.isfinite(a))
    assert np.max(np.abs(L - L_or) / L_or) < 1e-6  # f64 law of the f32 states, rounded once
    a_or = NO.nominal_action_endi(xin, meta["ctrl_gain"], meta["m"], meta["I"], meta["bnds"])
    close = np.all(np.abs(a - a_or) <= 1e-5 * (np.abs(a_or) + 1), axis=1)
    assert close.mean() > 0.97, close.mean()


def test_nominal_unsupported_for_2tank():
    from rcognita_amd import _native as N

    eng, _ = both("2tank", 4, "f64")
    with pytest.raises(N.NativeError) as ei:
        eng.nominal_action(np.zeros((4, 2)), 1.0)
    assert ei.value.code == N.ERR_UNSUPPORTED
    s0 = eng.get_state().copy()
    with pytest.raises(N.NativeError):
        eng.control_tick_nominal(1.0)
    np.testing.assert_array_equal(eng.get_state(), s0)  # refused before the env was stepped


@pytest.mark.parametrize("name,gain", [("3wrobotNI", 0.5), ("3wrobot", 5.0)])
def test_control_tick_nominal_vs_oracle(name, gain):
    """Fused tick (sim_step -> nominal action -> accum, step_idx), f64, against the oracle tick."""
    from rcognita_amd import _native as N

    rng = np.random.default_rng(5)
    B, T = 37, 12
    eng, cfg = both(name, B, "f64")
    x0 = rand_states(rng, name, B)
    eng.set_state(x0)
    env = O.new_batch(cfg, x0)
    m, I = (10.0, 1.0)
    for t in range(T):
        eng.control_tick_nominal(gain)
        NO.control_tick_nominal(cfg, env, gain, m, I)
        a = eng.get_field(N.FIELD_ACTION)
        ok = np.all(np.abs(a - env.action) <= 1e-6 * (np.abs(env.action) + 1), axis=1)
        assert ok.all() if name == "3wrobotNI" else ok.mean() > 0.9, t
        env.action = a.astype(np.float64)  # keep the two loops on the same trajectory (theta ties may flip)
        assert rel_err_norm(eng.get_state(), env.state) < 1e-9
        env.accum = eng.get_field(N.FIELD_ACCUM).astype(np.float64) if not ok.all() else env.accum
        assert rel_err_norm(eng.get_field(N.FIELD_ACCUM), env.accum, floor=float(np.max(np.abs(env.accum)))) < 1e-9
        np.testing.assert_array_equal(eng.get_field(N.FIELD_STEP_IDX), env.step_idx)


def test_nominal_parking_at_bench_size():
    """BASELINE configs[1] batch (65 536 envs), 3wrobotNI parking controller with the preset gain: the Lyapunov
    function the controller is built on decreases along the closed loop; integer counters exact; nothing fails."""
    from rcognita_amd import Engine, _native as N
    from rcognita_amd.pool import preset_engine_config

    rng = np.random.default_rng(11)
    B, T = 65536, 300
    eng = Engine(preset_engine_config("3wrobotNI", B, Nactor=3))
    x0 = rand_states(rng, "3wrobotNI", B)
    eng.set_state(x0)
    L0 = eng.nominal_action(x0, 0.5, want_lyap=True)[1].astype(np.float64)
    for _ in range(T):
        eng.control_tick_nominal(0.5)
    x1 = eng.get_state()
    L1 = eng.nominal_action(x1, 0.5, want_lyap=True)[1].astype(np.float64)
    summ, _ = eng.episode_stats(from_accum=True)
    assert summ["n_failed"] == 0
    np.testing.assert_array_equal(eng.get_field(N.FIELD_STEP_IDX), np.full(B, T, dtype=np.int32))
    # sampled + clipped (v <= 25, omega <= 5), so not monotone for every env: 98.8 % are lower after 3 s, median 2e-4
    assert np.mean(L1 < L0) > 0.97 and np.median(L1 / L0) < 0.01, (np.mean(L1 < L0), np.median(L1 / L0))


def test_mirror_classes_and_ctrl_selector():
    """CtrlNominal3WRobotNI / CtrlNominal3WRobot with the reference's constructor calls
    (presets/main_3wrobot_NI.py:235, main_3wrobot.py:239), sampled through ctrl_selector."""
    from rcognita_amd import controllers

    meta, z = load_golden("F10_nominal_3wrobotNI")
    bnds = np.array(meta["bnds"], dtype=float)
    c = controllers.CtrlNominal3WRobotNI(ctrl_gain=0.5, ctrl_bnds=bnds, t0=0, sampling_time=0.01)
    x = z["state"][20]
    np.testing.assert_array_equal(c.action_curr, np.zeros(2))
    a0 = controllers.ctrl_selector(0.001, x, None, c, None, "nominal")      # before the first sample: held zeros
    np.testing.assert_array_equal(a0, np.zeros(2))
    a1 = controllers.ctrl_selector(0.01, x, None, c, None, "nominal")
    np.testing.assert_allclose(a1, z["action"][20], rtol=1e-10)
    a2 = c.compute_action(0.015, z["state"][21])                            # inside the sample: held
    np.testing.assert_array_equal(a2, a1)
    np.testing.assert_allclose(c.compute_action_vanila(x), z["action_vanila"][20], rtol=1e-10)
    np.testing.assert_allclose(c.compute_LF(x), z["LF"][20], rtol=1e-10)
    ab = c.compute_action_vanila(z["state"][16:48])                         # batched observation
    np.testing.assert_allclose(ab, z["action_vanila"][16:48], rtol=1e-10)
    c.reset(0)
    np.testing.assert_array_equal(c.action_curr, np.zeros(2))

    meta, z = load_golden("F10_nominal_3wrobot")
    c = controllers.CtrlNominal3WRobot(meta["m"], meta["I"], ctrl_gain=5, ctrl_bnds=np.array(meta["bnds"], dtype=float),
                                       t0=0, sampling_time=0.01)
    L = c.compute_LF(z["state"])
    assert np.mean(L <= z["Fc_star"] * (1 + 1e-9) + 1e-12) >= 0.95  # the local search from theta = 0, as the reference's
    assert np.mean(np.abs(L - z["Fc_star"]) <= 1e-6 * z["Fc_star"]) > 0.9  # ... and mostly the very same minimum
    a = c.compute_action(0.01, z["state"][5])
    assert a.shape == (2,) and np.all(np.abs(a) <= np.array(meta["bnds"])[:, 1])
