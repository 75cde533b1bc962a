"""Episode boundary and argument guards of the C ABI and of the class mirror (``gpu`` marked).

* ``CtrlOptPred.reset`` (controllers.py:1046-1054) resets ONLY the clock and ``action_curr``; the
  critic weights, both buffers and the accumulated objective survive - on the mirror class and in
  ``rcg_episode_reset`` for RQL/SQL handles.
* A refused tick leaves the handle untouched (arguments are checked before the env step and the buffer push).
* An empty TD stack (Ncritic = 1) keeps ``w = clip(w_init)`` like the reference's SLSQP call on a zero objective.
* Device-resident inputs are checked for dtype / shape / size before their pointer reaches a kernel.
"""
import numpy as np
import pytest

from oracle import parity as PAR
from oracle import rcg_oracle as O
from tests.helpers import PRESETS, both, rand_states

pytestmark = pytest.mark.gpu


def test_ctrloptpred_reset_keeps_critic_buffers_and_accum():
    from rcognita_amd import controllers, systems

    p = PRESETS["2tank"]
    bnds = np.array(p["bnds"], dtype=float)
    sysm = systems.Sys2Tank(sys_type="diff_eqn", dim_state=2, dim_input=1, dim_output=2, dim_disturb=1, pars=p["pars"],
                            ctrl_bnds=bnds, is_dyn_ctrl=0, is_disturb=0, pars_disturb=[], dtype="f64")
    ctrl = controllers.CtrlOptPred(1, 2, mode="RQL", ctrl_bnds=bnds, action_init=[0.5], t0=0, sampling_time=0.1, Nactor=4,
                                   pred_step_size=0.2, sys_rhs=sysm._state_dyn, sys_out=sysm.out, state_sys=np.array(p["x0"], float),
                                   buffer_size=6, gamma=1, Ncritic=3, critic_period=0.1, critic_struct="quadratic",
                                   stage_obj_struct="quadratic", stage_obj_pars=[np.diag(p["R1"]).astype(float)],
                                   observation_target=np.array(p["target"]), n_candidates=32, rounds=1, dtype="f64")
    obs = np.array(p["x0"], float)
    for k in range(1, 6):
        a = ctrl.compute_action(0.1 * k, obs)
        ctrl.upd_accum_obj(obs, a)
        obs = obs + 0.01 * k
    w, wp = np.array(ctrl.w_critic), np.array(ctrl.w_critic_prev)
    ob, ab, acc = ctrl.observation_buffer.copy(), ctrl.action_buffer.copy(), ctrl.accum_obj_val
    assert not np.allclose(w, 1.0) and np.any(ob != 0) and acc > 0  # something was learnt and accumulated
    ctrl.reset(0.0)
    assert ctrl.ctrl_clock == 0.0
    np.testing.assert_array_equal(ctrl.action_curr, bnds[:, 0] / 10)  # action_min / 10, whatever action_init was
    np.testing.assert_array_equal(ctrl.w_critic, w)
    np.testing.assert_array_equal(ctrl.w_critic_prev, wp)
    np.testing.assert_array_equal(ctrl.observation_buffer, ob)
    np.testing.assert_array_equal(ctrl.action_buffer, ab)
    assert ctrl.accum_obj_val == acc
    # the first sample of the new episode ticks again and keeps learning from the retained buffers
    a = ctrl.compute_action(0.1, obs)
    assert np.all(np.isfinite(a)) and ctrl.ctrl_clock == 0.1
    np.testing.assert_array_equal(ctrl.observation_buffer[:-1], ob[1:])


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("mode,every", [(O.MODE_RQL, 1), (O.MODE_SQL, 3)])
def test_episode_reset_keeps_critic_state_and_restarts_the_clock(mode, every, dtype):
    from rcognita_amd import _native as N

    rng = np.random.default_rng(5)
    B, K, Nh = 11, 16, 4
    eng, cfg = both("2tank", B, dtype, n_actor=Nh, mode=mode, critic_struct=O.CRITIC_QUADRATIC, n_critic=3, buffer_size=5,
                    critic_every_ticks=every)
    x0 = rand_states(rng, "2tank", B).astype(eng.real)
    eng.set_state(x0)
    env = O.new_batch(cfg, x0.astype(np.float64))
    cand = O.grid_candidates(cfg, K)
    kw = dict(tol=1e-9 if dtype == "f64" else 1e-5, tol_over={"w_critic": 1e-6, "best_J": 1e-7} if dtype == "f64" else None)
    for t in range(7):
        eng.control_tick(None, K=K)
        env = PAR.check_tick(cfg, env, cand, PAR.device_fields(eng, N, critic=True), what=f"ep0 t={t}", **kw)
    keep = {f: eng.get_field(f).copy() for f in (N.FIELD_W_CRITIC, N.FIELD_W_PREV, N.FIELD_OBS_BUF, N.FIELD_ACT_BUF)}
    accum = eng.get_field(N.FIELD_ACCUM).copy()
    assert not np.allclose(keep[N.FIELD_W_CRITIC], 1.0)
    eng.episode_reset()
    ret = O.episode_reset(cfg, env, x0.astype(np.float64))
    for f, v in keep.items():  # retained, bit for bit
        np.testing.assert_array_equal(eng.get_field(f), v)
    np.testing.assert_array_equal(eng.get_field(N.FIELD_RETURNS), accum)
    np.testing.assert_allclose(accum, ret, rtol=1e-4)
    np.testing.assert_array_equal(eng.get_field(N.FIELD_ACCUM), np.zeros(B, eng.real))
    np.testing.assert_array_equal(eng.get_state(), x0)
    np.testing.assert_array_equal(eng.get_field(N.FIELD_ACTION), np.full((B, 1), 0.0, eng.real))  # action_min / 10
    np.testing.assert_array_equal(eng.get_field(N.FIELD_STEP_IDX), np.zeros(B, np.int32))
    np.testing.assert_array_equal(eng.get_field(N.FIELD_EPISODE_IDX), np.ones(B, np.int32))
    w_hold = keep[N.FIELD_W_CRITIC]
    for t in range(2 * every + 1):  # the critic period restarts: fits on ticks every-1, 2*every-1 of the NEW episode
        eng.control_tick(None, K=K)
        env = PAR.check_tick(cfg, env, cand, PAR.device_fields(eng, N, critic=True), what=f"ep1 t={t}", **kw)
        w = eng.get_field(N.FIELD_W_CRITIC)
        if (t + 1) % every != 0:
            np.testing.assert_array_equal(w, w_hold)
        w_hold = w.copy()
    np.testing.assert_array_equal(eng.get_field(N.FIELD_EPISODE_IDX), env.episode_idx)


@pytest.mark.parametrize("name,mode,kw", [
    ("2tank", O.MODE_RQL, dict(critic_struct=O.CRITIC_QUADRATIC, n_critic=4, buffer_size=7, critic_every_ticks=3)),
    ("3wrobot", O.MODE_MPC, dict(per_env_pars=True)),
    ("3wrobotNI", O.MODE_SQL, dict(critic_struct=O.CRITIC_QUAD_MIX, n_critic=3, buffer_size=5, ref_lag=True)),
])
def test_checkpoint_restore_continues_bit_identically(name, mode, kw, tmp_path):
    """SURVEY.md 5 (checkpoint / resume): a handle is its per-env tensors plus one host counter.  Run 5 ticks, write
    a checkpoint, run 6 more; a FRESH handle restored from the file and run for the same 6 ticks ends with every field
    bit-identical (critic period phase, buffers, weights, counters included)."""
    from rcognita_amd import _native as N

    rng = np.random.default_rng(44)
    B, K = 33, 16
    a, _ = both(name, B, "f32", n_actor=5, mode=mode, **kw)
    b, _ = both(name, B, "f32", n_actor=5, mode=mode, **kw)
    if kw.get("per_env_pars"):
        pars = np.stack([rng.uniform(5, 20, B), rng.uniform(0.5, 2, B)], axis=-1)
        a.set_field(N.FIELD_PARS, pars)
    a.set_state(rand_states(rng, name, B))
    for _ in range(5):
        a.control_tick(None, K=K)
    path = str(tmp_path / "ck.npz")
    a.checkpoint(path)
    for _ in range(6):
        a.control_tick(None, K=K)
    b.restore(path)
    assert N.lib().rcg_tick_count(b._h) == 5
    for _ in range(6):
        b.control_tick(None, K=K)
    for f in range(N.FIELD_COUNT):
        if N.lib().rcg_field_bytes(a._h, f) > 0:
            np.testing.assert_array_equal(b.get_field(f), a.get_field(f), err_msg=f"field {f}")
    assert N.lib().rcg_tick_count(b._h) == N.lib().rcg_tick_count(a._h) == 11
    other, _ = both(name, B + 1, "f32", n_actor=5, mode=mode, **kw)
    with pytest.raises(ValueError):
        other.restore(path)


def test_warm_start_does_not_cross_an_episode_boundary():
    """rcg_control_tick_opt(warm_start=1): the first decision of an episode starts from action_sqn_init like the
    reference does on every call - the previous episode's optimum is not shifted into it."""
    from rcognita_amd import _native as N

    rng = np.random.default_rng(3)
    B, Nh = 9, 5
    eng, cfg = both("3wrobot", B, "f64", n_actor=Nh)
    x0 = rand_states(rng, "3wrobot", B)
    eng.set_state(x0)
    eng.control_tick_opt(iters=4, warm_start=True)
    first = {f: eng.get_field(f).copy() for f in (N.FIELD_ACTION_SQN, N.FIELD_ACTION, N.FIELD_BEST_J, N.FIELD_STATE)}
    for _ in range(3):
        eng.control_tick_opt(iters=4, warm_start=True)
    eng.episode_reset()
    eng.control_tick_opt(iters=4, warm_start=True)  # same state_init, same held action -> the same decision as tick 0
    for f, v in first.items():
        np.testing.assert_array_equal(eng.get_field(f), v)


@pytest.mark.parametrize("mode", [O.MODE_MPC, O.MODE_RQL])
def test_a_refused_tick_changes_nothing(mode):
    from rcognita_amd import _native as N

    rng = np.random.default_rng(1)
    B = 7
    eng, cfg = both("3wrobot", B, "f64", n_actor=4, mode=mode, n_critic=3, buffer_size=5)
    eng.set_state(rand_states(rng, "3wrobot", B))
    eng.control_tick(None, K=16)
    fields = [N.FIELD_STATE, N.FIELD_STATE_PREV, N.FIELD_ACTION, N.FIELD_ACCUM, N.FIELD_STEP_IDX]
    if mode != O.MODE_MPC:
        fields += [N.FIELD_OBS_BUF, N.FIELD_ACT_BUF, N.FIELD_W_CRITIC]
    snap = {f: eng.get_field(f).copy() for f in fields}
    for bad_K in (50, 0, -3):  # du = 2: the generated grid needs a square K >= 1
        with pytest.raises(N.NativeError) as ei:
            eng.control_tick(None, K=bad_K)
        assert ei.value.code == N.ERR_BAD_ARG
        for f, v in snap.items():
            np.testing.assert_array_equal(eng.get_field(f), v)
    eng.control_tick(None, K=16)  # and the handle is still usable
    np.testing.assert_array_equal(eng.get_field(N.FIELD_STEP_IDX), np.full(B, 2, np.int32))


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("ncritic,bs", [(1, 6), (4, 2)])
def test_empty_td_stack_keeps_the_initial_weights(ncritic, bs, dtype):
    """Ncritic = 1 (or buffer_size = 2, which clips Ncritic to 1): _critic_cost has no term, the reference's SLSQP
    returns w_critic_init; the tick must run (it used to return RCG_ERR_UNSUPPORTED after stepping the env)."""
    from rcognita_amd import _native as N

    rng = np.random.default_rng(8)
    B, K = 10, 16
    eng, cfg = both("3wrobotNI", B, dtype, n_actor=3, mode=O.MODE_RQL, critic_struct=O.CRITIC_QUAD_MIX, n_critic=ncritic,
                    buffer_size=bs)
    assert cfg.n_critic == 1
    x0 = rand_states(rng, "3wrobotNI", B).astype(eng.real)
    eng.set_state(x0)
    env = O.new_batch(cfg, x0.astype(np.float64))
    cand = O.grid_candidates(cfg, K)
    for t in range(4):
        eng.control_tick(None, K=K)
        env = PAR.check_tick(cfg, env, cand, PAR.device_fields(eng, N, critic=True), tol=1e-9 if dtype == "f64" else 1e-5,
                             what=f"t={t}")
        np.testing.assert_array_equal(eng.get_field(N.FIELD_W_CRITIC), np.ones((B, cfg.dc), eng.real))


def test_device_inputs_are_checked_before_they_reach_a_kernel():
    """torch CUDA tensors / DeviceArrays with a wrong dtype, shape, size or K are refused by the binding (tests/
    torch_input_probe.py; its own process: PyTorch-ROCm must initialise the GPU before librcg does)."""
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "torch_input_probe.py")], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "INPUT_CHECKS_OK" in out.stdout


def test_zero_weighted_component_overflow_convention():
    """A component that overflows under a ZERO stage-cost weight (Sys3WRobot's speed, R1 = diag[1, 10, 1, 0, 0, 0, 0]):
    numpy evaluates the full sum, 0 * inf = NaN, and the candidate counts as +inf - the reference's own behaviour.  The
    streamed kernels compute that sum; the generated grid, which does not accumulate zero-weighted terms, tests the
    zero-weighted state components on the observation and on the last rolled-out state instead: same outcome (round 3
    kept a finite cost there)."""
    from oracle import rcg_oracle as O
    from rcognita_amd import Engine, _native as N
    from rcognita_amd.pool import preset_engine_config
    from tests.helpers import oracle_cfg

    B, K, Nh = 4, 64, 4
    eng = Engine(preset_engine_config("3wrobot", B, Nactor=Nh, dtype="f32"))
    x = np.zeros((B, 5), dtype=np.float32)
    x[:, 0], x[:, 1] = 1.0, 2.0
    x[1, 3] = 3e19   # v^2 overflows float32, x + h v cos(alpha) stays finite over three Euler steps
    x[2, 4] = -2e19  # omega^2 overflows; the heading is then ~1e17 rad: finite, any value of sin / cos
    eng.set_state(x)
    cfg = oracle_cfg("3wrobot", n_actor=Nh)
    grid = O.grid_candidates(cfg, K)  # the generated level grid, as a tensor
    cand = np.ascontiguousarray(np.broadcast_to(grid.astype(np.float32)[None], (B, K, Nh, 2)))
    _, bj_gen, _ = eng.actor_argmin(None, K=K)
    _, bj_str, _ = eng.actor_argmin(eng.to_device(cand), K=K)
    assert np.all(np.isfinite(bj_gen[[0, 3]]))
    np.testing.assert_allclose(bj_gen[[0, 3]], bj_str[[0, 3]], rtol=1e-6)  # finite envs: the same costs on both paths
    assert np.all(np.isinf(bj_str[[1, 2]])), "streamed rows: 0 * inf = NaN = +inf for every candidate, as numpy"
    assert np.all(np.isinf(bj_gen[[1, 2]])), "generated grid: the same"
    # ... and T ticks in one launch (k_ticks runs the generated rollouts): the overflowing envs report +inf as well
    eng.control_ticks(T=1, K=K)
    bj_t = eng.get_field(N.FIELD_BEST_J)
    assert np.all(np.isinf(bj_t[[1, 2]])) and np.all(np.isfinite(bj_t[[0, 3]]))
    eng.close()
