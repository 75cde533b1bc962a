"""rcg_loop_step: one iteration of the reference's headless loop (presets/main_3wrobot.py:419-429) in one native call and one
host wait, and the drop-in classes riding on it (Simulator.sim_step computes ahead what compute_action / stage_obj will be
asked for).  Everything must equal, bit for bit, what the separate calls leave - the loop order, the one-step lag of the
controller's state, the float clock tests stay the reference's.  ``gpu`` marked."""
import numpy as np
import pytest

from oracle import rcg_oracle as O
from tests.helpers import both, rand_states
from tests.test_hip_ref_traces import DIMS, make_loop_objects

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("name,mode,cs,B", [("3wrobot", "MPC", "quad-nomix", 1), ("3wrobotNI", "MPC", "quad-nomix", 7),
                                            ("2tank", "RQL", "quadratic", 3), ("3wrobot", "SQL", "quad-mix", 2)])
def test_loop_step_equals_the_separate_calls(name, mode, cs, B, dtype):
    """Engine.loop_step against rcg_set_field + rcg_sim_step_h + rcg_critic_update + rcg_actor_optimize + rcg_stage_obj on a
    twin handle: state, action, stage cost, best_J, critic weights, buffers - bit for bit, over ticks that decide, ticks that
    only push, and plain simulation steps."""
    from rcognita_amd import _native as N

    rng = np.random.default_rng(3)
    kw = dict(n_actor=5, mode=O.MODE_IDS[mode], critic_struct=O.CRITIC_IDS[cs], n_critic=4, buffer_size=6)
    a, cfg = both(name, B, dtype, **kw)
    b, _ = both(name, B, dtype, **kw)
    x0 = rand_states(rng, name, B)
    a.set_state(x0)
    b.set_state(x0)
    du = cfg.du
    act = np.tile(cfg.ctrl_bnds[:, 0] / 10.0, (B, 1))
    h = cfg.dt_sim / 2
    for it in range(9):
        decide, push, fit = it % 2 == 1, (mode != "MPC" and it % 2 == 1), (mode != "MPC" and it % 4 == 1)
        st, ac, stage, bj, w = a.loop_step(act, h, 1, decide=decide, push=push, fit=fit, iters=6)
        if decide:  # the plain MPC decision does the iteration's head and tail in its own launch (variant bit 8)
            ll = a.last_launch(N.KERNEL_ACTOR)
            assert ll["kernel"] == "k_actor_opt" and bool(ll["variant"] & 8) == (mode == "MPC"), ll
        # the separate calls
        b.set_field(N.FIELD_ACTION, act)
        if push:
            b.sim_step(1, step=h)
            b.critic_update(do_fit=fit)
        else:
            b.sim_step(1, step=h)
        xs = b.get_field(N.FIELD_STATE_PREV)
        st_b = b.get_state()
        if decide:
            a_b, u_b, bj_b, _ = b.actor_optimize(iters=6, obs=st_b, state_sys=xs)
            b.set_field(N.FIELD_ACTION, a_b)
        else:
            a_b = act.astype(b.real)
        stage_b = b.stage_obj(st_b, a_b)
        np.testing.assert_array_equal(st, st_b.astype(np.float64), err_msg=f"state it={it}")
        np.testing.assert_array_equal(ac, np.asarray(a_b, dtype=np.float64), err_msg=f"action it={it}")
        np.testing.assert_array_equal(stage, stage_b.astype(np.float64), err_msg=f"stage it={it}")
        if decide:
            np.testing.assert_array_equal(bj, bj_b.astype(np.float64))
            np.testing.assert_array_equal(np.asarray(a.get_field(N.FIELD_ACTION_SQN)).reshape(u_b.shape), u_b)
        else:
            assert np.all(np.isnan(bj))
        if mode != "MPC":
            np.testing.assert_array_equal(w, b.get_field(N.FIELD_W_CRITIC).astype(np.float64))
            for f in (N.FIELD_OBS_BUF, N.FIELD_ACT_BUF, N.FIELD_W_PREV):
                np.testing.assert_array_equal(a.get_field(f), b.get_field(f))
        else:
            assert w is None
        act = np.asarray(ac, dtype=np.float64).copy()  # the loop hands the decision back: System.receive_action


def _run(name, mode, cs, fuse, T, B=None, dtype="f64", tamper_at=None, Nactor=5, speculate=True, meddle=None):
    from rcognita_amd import controllers

    x0 = None
    if B is not None:
        from tests.helpers import PRESETS

        x0 = np.array(PRESETS[name]["x0"], dtype=float) + np.random.default_rng(9).uniform(-0.5, 0.5, (B, DIMS[name][0]))
    my_sys, my_ctrl, my_sim = make_loop_objects(name, mode, Nactor, 1.0, x0=x0, critic_struct=cs, dtype=dtype, opt_iters=8)
    my_sim.fuse = fuse
    my_ctrl.speculate = speculate
    du = DIMS[name][1]
    rows = []
    for k in range(T):  # presets/main_3wrobot.py:419-446
        if meddle is not None:  # something a caller does between two iterations of the loop
            meddle(k, my_sys, my_ctrl, my_sim, rows)
        my_sim.sim_step()
        t, state, observation, state_full = my_sim.get_sim_step_data()
        if tamper_at is not None and k == tamper_at:  # a caller that does NOT follow the loop: another state for the rollout
            my_ctrl.receive_sys_state(np.asarray(my_ctrl.state_sys, dtype=float) * 1.01)
        action = controllers.ctrl_selector(t, observation, np.zeros(du), None, my_ctrl, mode)
        my_sys.receive_action(action)
        my_ctrl.receive_sys_state(my_sys._state)
        my_ctrl.upd_accum_obj(observation, action)
        rows.append(np.concatenate([[t], np.ravel(state_full), np.ravel(action), np.ravel(my_ctrl.stage_obj(observation, action)),
                                    np.ravel(my_ctrl.accum_obj_val),
                                    np.ravel(np.broadcast_to(my_ctrl.w_critic, (my_ctrl.B, my_ctrl.dim_critic)))]))
    return np.stack(rows), my_ctrl


@pytest.mark.parametrize("name,mode,cs,B", [("3wrobot", "MPC", "quad-nomix", None), ("3wrobotNI", "MPC", "quad-nomix", 4),
                                            ("2tank", "RQL", "quadratic", None), ("2tank", "SQL", "quad-lin", 3),
                                            ("3wrobotNI", "RQL", "quad-mix", None)])
def test_drop_in_loop_is_the_same_with_and_without_the_fused_step(name, mode, cs, B):
    """The reference's loop on the mirror classes, once with Simulator.fuse (one native call per iteration) and once with the
    separate calls: every row [t, state, action, stage_obj, accum_obj, w_critic] identical, and the fused run really served
    every iteration by one call (after the start-up iteration of the critic modes, whose first push needs the controller's own
    action_curr)."""
    T = 30
    fused, c1 = _run(name, mode, cs, True, T, B=B)
    plain, c0 = _run(name, mode, cs, False, T, B=B)
    np.testing.assert_array_equal(fused, plain)
    assert c0.fused_steps == 0
    assert c1.fused_steps >= T - 2 and c1.fused_decisions >= T // 2 - 1, (c1.fused_steps, c1.fused_decisions)
    # the optimal sequence of the last decision is fetched on demand and is the same
    np.testing.assert_array_equal(np.asarray(c1._prev_opt).reshape(np.asarray(c0._prev_opt).shape), c0._prev_opt)
    np.testing.assert_array_equal(c1.observation_buffer, c0.observation_buffer)
    np.testing.assert_array_equal(c1.action_buffer, c0.action_buffer)


def test_a_caller_that_leaves_the_loop_order_still_gets_the_separate_calls_answer():
    """The fused step computes ahead for the inputs the reference's loop WILL pass (observation = the new state, state_sys = the
    state one iteration earlier).  A caller that passes something else gets what the separate calls give for ITS inputs - the
    decision computed ahead is dropped, the handle re-synchronised."""
    fused, c1 = _run("3wrobot", "MPC", "quad-nomix", True, 16, tamper_at=7)
    plain, c0 = _run("3wrobot", "MPC", "quad-nomix", False, 16, tamper_at=7)
    np.testing.assert_array_equal(fused, plain)
    assert c1.fused_steps == 16 and c1.fused_decisions == 7 and c0.fused_decisions == 0  # 8 samples, one of them tampered with


def test_loop_step_argument_checks():
    from rcognita_amd import _native as N

    eng, _ = both("3wrobot", 300, "f64", n_actor=5)  # 300 envs x 9 doubles do not fit the 16-KB pinned buffer
    with pytest.raises(N.NativeError) as ei:
        eng.loop_step(None, 0.01)
    assert ei.value.code == N.ERR_UNSUPPORTED
    eng, _ = both("3wrobot", 2, "f64", n_actor=5)
    with pytest.raises(N.NativeError) as ei:
        eng.loop_step(None, -1.0)
    assert ei.value.code == N.ERR_BAD_ARG


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("name,mode,cs,B", [("3wrobot", "MPC", "quad-nomix", None), ("3wrobotNI", "MPC", "quad-nomix", 4),
                                            ("2tank", "RQL", "quadratic", None), ("2tank", "SQL", "quad-lin", 3),
                                            ("3wrobotNI", "RQL", "quad-mix", None), ("3wrobot", "SQL", "quad-nomix", 2)])
def test_next_step_started_ahead_changes_no_number(name, mode, cs, B, dtype):
    """compute_action starts the next simulation step (rcg_loop_step_begin), Simulator.sim_step collects it
    (rcg_loop_step_end): the same rows as without - and as the separate calls - and all but the first iterations really were
    started ahead."""
    T = 30
    ahead, c2 = _run(name, mode, cs, True, T, B=B, dtype=dtype, speculate=True)
    fused, c1 = _run(name, mode, cs, True, T, B=B, dtype=dtype, speculate=False)
    plain, c0 = _run(name, mode, cs, False, T, B=B, dtype=dtype)
    np.testing.assert_array_equal(ahead, fused)
    np.testing.assert_array_equal(ahead, plain)
    assert c1.spec_hits == 0 and c0.spec_hits == 0
    # (one drop at the start: the loop's first stage_obj is asked for an action the first step did not hold - the controller's
    # action_curr against the system's action_init - and goes to the handle)
    assert c2.spec_hits >= T - 3 and c2.spec_drops <= 1, (c2.spec_hits, c2.spec_drops)
    drops = c2.spec_drops
    assert c2.fused_steps == c1.fused_steps and c2.fused_decisions == c1.fused_decisions
    # one step is still pending (started from the last compute_action): asking for the sequence drops it and gets the last
    # TAKEN decision's, not the pending one's
    assert c2._spec is not None
    np.testing.assert_array_equal(np.asarray(c2._prev_opt).reshape(np.asarray(c0._prev_opt).shape), c0._prev_opt)
    assert c2._spec is None and c2.spec_drops == drops + 1
    np.testing.assert_array_equal(c2.observation_buffer, c0.observation_buffer)
    np.testing.assert_array_equal(c2.action_buffer, c0.action_buffer)


MEDDLERS = {
    # the caller hands the system another action than compute_action returned
    "another-action": lambda k, s, c, m, rows: s.receive_action(np.asarray(s.action, dtype=float) * 0.5) if k in (5, 6, 12) else None,
    # the caller uses the controller's handle for something else (any engine access drops the step)
    "other-call": lambda k, s, c, m, rows: c._actor_cost(np.tile(np.asarray(c.action_curr, dtype=float).reshape(-1), c.Nactor),
                                                         m.observation) if k in (4, 9) else None,
    # the caller reads the last decision's sequence
    "read-sequence": lambda k, s, c, m, rows: rows.append(rows.pop() + 0 * np.sum(c._prev_opt)) if k in (3, 8, 9) and rows else None,
    # the caller moves the simulator's state by hand
    "set-state": lambda k, s, c, m, rows: _set_state(m, 1.001) if k in (7, 13) else None,
    # a step of another length on the reference's own time grid
    "other-step": lambda k, s, c, m, rows: (m.sim_step(t_next=m.t + 0.3 * m.dt), s.receive_action(s.action)) if k == 6 else None,
    # an episode boundary
    "reset": lambda k, s, c, m, rows: (m.reset(), c.reset(0)) if k == 10 else None,
}


def _set_state(sim, f):
    sim.state_full = np.asarray(sim.state_full, dtype=float) * f
    sim.state = sim.state_full[..., 0:sim.dim_state]
    sim.observation = sim.sys_out(sim.state)


@pytest.mark.parametrize("what", sorted(MEDDLERS))
@pytest.mark.parametrize("name,mode,cs", [("3wrobot", "MPC", "quad-nomix"), ("2tank", "RQL", "quadratic"), ("3wrobotNI", "SQL", "quad-mix")])
def test_a_step_started_ahead_that_is_not_asked_for_is_dropped(name, mode, cs, what):
    """Between compute_action and the next Simulator.sim_step the caller does something the loop does not: the step that was
    started ahead is waited for and dropped, the handle is uploaded anew, and every row equals the run without it."""
    T = 18
    ahead, c2 = _run(name, mode, cs, True, T, speculate=True, meddle=MEDDLERS[what])
    fused, c1 = _run(name, mode, cs, True, T, speculate=False, meddle=MEDDLERS[what])
    plain, c0 = _run(name, mode, cs, False, T, meddle=MEDDLERS[what])
    np.testing.assert_array_equal(ahead, fused)
    np.testing.assert_array_equal(ahead, plain)
    assert c2.spec_drops >= 1 and c2.spec_hits >= T // 2, (what, c2.spec_hits, c2.spec_drops)


@pytest.mark.parametrize("name,mode,cs,B", [("3wrobot", "MPC", "quad-nomix", 3), ("2tank", "RQL", "quadratic", 2)])
def test_loop_step_in_two_halves(name, mode, cs, B):
    """rcg_loop_step_begin + rcg_loop_step_end = rcg_loop_step, bit for bit; one step may be pending; a dropped step leaves ACTION_SQN
    (the last collected decision's sequence) alone and a collected one replaces it."""
    from rcognita_amd import _native as N

    rng = np.random.default_rng(5)
    kw = dict(n_actor=5, mode=O.MODE_IDS[mode], critic_struct=O.CRITIC_IDS[cs], n_critic=4, buffer_size=6)
    a, cfg = both(name, B, "f64", **kw)
    b, _ = both(name, B, "f64", **kw)
    x0 = rand_states(rng, name, B)
    a.set_state(x0)
    b.set_state(x0)
    act = np.tile(cfg.ctrl_bnds[:, 0] / 10.0, (B, 1))
    crit = mode != "MPC"
    with pytest.raises(N.NativeError) as ei:
        a.loop_step_end()
    assert ei.value.code == N.ERR_BAD_ARG
    for it in range(6):
        flags = dict(decide=it % 2 == 1, push=crit and it % 2 == 1, fit=crit and it % 2 == 1, iters=5)
        a.loop_step_begin(act, cfg.dt_sim / 2, 1, **flags)
        with pytest.raises(N.NativeError) as ei:
            a.loop_step_begin(act, cfg.dt_sim / 2, 1, **flags)
        assert ei.value.code == N.ERR_BAD_ARG
        ra = a.loop_step_end()
        rb = b.loop_step(act, cfg.dt_sim / 2, 1, **flags)
        for va, vb in zip(ra, rb):
            if va is None:
                assert vb is None
            else:
                np.testing.assert_array_equal(va, vb)
        np.testing.assert_array_equal(a.get_field(N.FIELD_ACTION_SQN), b.get_field(N.FIELD_ACTION_SQN))
        act = np.array(ra[1])
    sqn = a.get_field(N.FIELD_ACTION_SQN).copy()
    assert np.any(sqn != 0)
    a.loop_step_begin(act * 0.3, cfg.dt_sim / 2, 1, decide=True, push=crit, fit=crit, iters=5)
    assert a.loop_step_end(drop=True) is None
    np.testing.assert_array_equal(a.get_field(N.FIELD_ACTION_SQN), sqn)  # the dropped decision's sequence never became ACTION_SQN
    assert np.any(a.get_state() != b.get_state())                        # ... its simulation step did happen on the handle
