"""Every control tick of the reference's critic-mode closed loops (fixtures F7c, oracle/gen_critic_fixtures.py) replayed
TEACHER-FORCED through the mirror classes on librcg - the closed-loop check in RQL / SQL that can fail.  ``gpu`` marked.

The free-running comparison (tests/test_hip_ref_traces.py) integrates every difference between two optimisers over the
run and can only be held to a band; here nothing accumulates: at each tick ``CtrlOptPred`` (rcognita_amd/controllers.py, wired
as the presets wire it) is put into the state the reference's controller was in - ``state_sys``, both buffers one push
behind, ``action_curr``, ``w_critic_prev``, both clocks - and ``compute_action(t, observation)`` runs the reference's own
sequence (controllers.py:1458-1477): push, ``_critic_optimizer`` (k_critic_fit), ``_actor_optimizer`` (k_actor_opt).  Then
the weights are forced to the reference's and ``_actor_optimizer`` decides again.  Asserted per tick, by
tests/teacher_forced.py: the buffers after the push are the reference's bit for bit; Jc of the device's weights against
SLSQP's on the reference's TD stack; J of the device's sequence against SLSQP's at the reference's weights (0.5 %); and the
FIRST ACTION against the reference's wherever the reference's own cost is measurably sharp in it (fixture field
tick_first_rise).  The CPU twin of this file is tests/test_teacher_forced_oracle.py."""
import numpy as np
import pytest

from tests import teacher_forced as TF
from tests.conftest import load_golden
from tests.test_critic_traces import CASES, MODES, trace_cfg
from tests.test_dropin_api import DIMS
from tests.test_hip_ref_traces import make_loop_objects

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name,cs", CASES)
def test_teacher_forced_replay_of_the_reference_loop_through_the_mirror_classes(name, cs, mode, dtype):
    """``dtype``: the width of the handles behind the mirror classes - float64 (their default, the reference's width) and
    float32 (the production width: buffers, weights and sequences pass through float32 on the device; the fit itself is
    float64 inside either way, its objective is compared at 1e-5)."""
    meta, z = load_golden(f"F7c_trace_{name}_{mode}_{cs}")
    cfg = trace_cfg(meta)
    ds, du, _ = DIMS[name]
    dt, N = meta["dt"], meta["Nactor"]
    _, ctrl, _ = make_loop_objects(name, mode, N, meta["t1"], x0=meta["x0"], critic_struct=cs, dtype=dtype)
    tally = TF.Tally(f"(mirror classes, {dtype}) {name} {mode} {cs}")
    for i in range(len(z["tick_t"])):
        t, obs = float(z["tick_t"][i]), z["tick_obs"][i]
        ob, ab = z["tick_obs_buf"][i], z["tick_act_buf"][i]
        fitted = bool(z["tick_fitted"][i])
        # the controller's state just before the reference's compute_action of this tick
        ctrl.state_sys = z["tick_state_sys"][i].copy()
        ctrl.action_curr = z["tick_action_prev"][i].copy()
        ctrl.observation_buffer = np.vstack([np.zeros((1, ds)), ob[:-1]])  # push_vec drops row 0 and appends: restores ob
        ctrl.action_buffer = np.vstack([np.zeros((1, du)), ab[:-1]])
        ctrl.w_critic_prev = z["tick_w_prev"][i].copy()
        ctrl.ctrl_clock = t - dt
        ctrl.critic_clock = t - dt if fitted else t  # the reference refits only where its float clock test passed
        ctrl.compute_action(t, obs)
        assert np.array_equal(ctrl.observation_buffer, ob) and np.array_equal(ctrl.action_buffer, ab)
        w_dev = np.array(ctrl.w_critic, dtype=float).reshape(-1) if fitted else None
        if not fitted:
            assert np.array_equal(np.asarray(ctrl.w_critic, dtype=float).reshape(-1), z["tick_w_prev"][i])
        if dtype == "f32" and w_dev is not None:  # what the fit saw is the float32 image of the stack: the weights stay in the box
            w_dev = np.clip(w_dev, *TF.O.critic_bounds(cfg.critic_struct, cfg.dc))
        # the actor on the reference's weights
        ctrl.w_critic = z["tick_w"][i].copy()
        a = ctrl._actor_optimizer(obs)
        u_dev = np.asarray(ctrl._prev_opt, dtype=float).reshape(N, du)
        assert np.array_equal(np.asarray(a, dtype=float).reshape(-1), u_dev[0])
        TF.check_tick(tally, cfg, z, i, w_dev, u_dev, meta["first_fracs"], p_tol=1e-9 if dtype == "f64" else 1e-5)
    print("\n" + tally.line())
    assert not tally.failures, "\n".join(tally.failures[:10])
    assert tally.n_sharp > 0, "no tick of this trace pins the first action: the fixture cannot falsify the actor"
