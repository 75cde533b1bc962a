"""Float32 (and float64) kernels at LARGE HEADING against numbers the REFERENCE produced (fixture F13,
oracle/gen_large_heading_fixture.py): |alpha| in {30, 72, 300, 1e3, 1e4} rad, both signs, every input float32-exact, so
the kernels read bit for bit what the reference evaluated (systems.py:308-323, 370-382; controllers.py:1290-1296;
SURVEY §7 hard part 3).  Through every f32 kernel that takes a heading: k_rhs, k_actor_dma<float>,
k_actor_dma_packed<float>, k_actor (streamed and generated), k_ticks_pk (the fused generated-grid tick), k_sim<float>.

Tolerance: the north star's 1e-5, as everywhere (|hip - ref| / max(|ref|, 1); costs relative to the env's largest cost).
What float32 CANNOT do is stated where it is measured (test_F13_free_run_f32): a float32 heading of 1e4 rad is stored to
+-4.9e-4 rad, so a FREE-RUNNING float32 trajectory leaves the reference's by that much per step in the heading whatever
the kernel does; each step as a map from float32-exact inputs stays inside 1e-5.  `REPORT_LARGE_HEADING=1 pytest -s` prints the
worst errors (kept as profiles/r06_large_heading.txt)."""
import os

import numpy as np
import pytest

from oracle import rcg_oracle as O
from tests.conftest import load_golden
from tests.helpers import TOL, assert_kernel, both, rel_err_norm

pytestmark = pytest.mark.gpu
ROBOTS = ["3wrobot", "3wrobotNI"]
DTYPES = ["f32", "f64"]


def _report(*a):
    if os.environ.get("REPORT_LARGE_HEADING"):
        print("[F13]", *a)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name", ROBOTS)
def test_F13_rhs(name, dtype):
    _, z = load_golden(f"F13_large_heading_{name}")
    eng, _ = both(name, 1, dtype)
    d, _ = eng.rhs(z["rhs__state"], z["rhs__action"], clip=False)
    e1 = rel_err_norm(d, z["rhs__state_dyn"])
    d, a = eng.rhs(z["rhs__state"], z["rhs__action"], clip=True)
    e2 = rel_err_norm(d, z["rhs__closed_loop_rhs"])
    _report(f"rhs {name} {dtype}: state_dyn {e1:.2e} closed_loop_rhs {e2:.2e}")
    assert max(e1, e2) <= TOL[dtype]
    np.testing.assert_array_equal(a, z["rhs__clipped_action"].astype(eng.real))


def _cost_err(J, J_ref):
    scale = np.max(np.abs(J_ref), axis=1, keepdims=True)
    return float(np.max(np.abs(J - J_ref) / scale)), scale


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name", ROBOTS)
def test_F13_actor_cost_through_the_streamed_kernels(name, dtype):
    """The reference's `_actor_cost` at every heading magnitude through k_actor_dma (K = 64), k_actor_dma_packed (the
    first 8 / 16 sequences of every env) and the generic k_actor (the first 3): operator and argmin."""
    from rcognita_amd import _native as N

    meta, z = load_golden(f"F13_large_heading_{name}")
    worst = {}
    for c in meta["cost_cases"]:
        tag = "cost_" + c["tag"]
        x, aseq, w, J_ref = z[f"{tag}__state"], z[f"{tag}__action_sqn"], z[f"{tag}__w"], z[f"{tag}__J"]
        for K, kernel in ((64, "k_actor_dma"), (16, "k_actor_dma_packed"), (8, "k_actor_dma_packed"), (3, "k_actor")):
            eng, _ = both(name, 2, dtype, n_actor=c["N"], mode=O.MODE_IDS[c["mode"]], gamma=c["gamma"],
                          critic_struct=O.CRITIC_IDS[c["critic_struct"]], pred_step_size=c["pred_step_size"], buffer_size=4)
            eng.set_state(x)
            if c["mode"] != "MPC":
                eng.set_field(N.FIELD_W_CRITIC, w)
            cand = eng.to_device(np.ascontiguousarray(aseq[:, :K]).astype(eng.real))
            J = eng.actor_cost(cand)
            assert_kernel(eng, kernel)
            err, scale = _cost_err(J, J_ref[:, :K])
            key = (kernel, c["A"])
            worst[key] = max(worst.get(key, 0.0), err)
            assert err <= TOL[dtype], f"{tag} K={K} ({kernel}): J rel err {err:.3e}"
            a, bj, bi = eng.actor_argmin(cand)
            ref_i = np.argmin(J_ref[:, :K], axis=1)
            for e in range(2):  # a float32 near-tie may take the runner-up: its reference cost within rounding of the best
                assert bi[e] == ref_i[e] or abs(J_ref[e, bi[e]] - J_ref[e, ref_i[e]]) <= 4 * TOL[dtype] * scale[e, 0], tag
                np.testing.assert_array_equal(a[e], aseq[e, bi[e], 0, :].astype(eng.real))
                assert abs(bj[e] - J_ref[e, bi[e]]) <= TOL[dtype] * scale[e, 0], tag
            eng.close()
    for (kernel, A), e in sorted(worst.items()):
        _report(f"actor_cost {name} {dtype} {kernel} |alpha|~{A:g}: worst J rel err {e:.2e}")


@pytest.mark.parametrize("name", ROBOTS)
def test_F13_generated_grid_tick_k_ticks_pk(name):
    """The fused generated-grid tick (k_ticks_pk: env step + 256-level grid + argmin in one launch, f32) and the
    generated-grid operator (k_actor's packed instance) from states at rest under a held zero action - the env step leaves
    them exactly in place, so the decision is made on the state the reference evaluated; the grid levels are the
    kernel's own float32 levels (asserted bit for bit through the winner's action)."""
    from rcognita_amd import _native as N

    _, z = load_golden(f"F13_large_heading_{name}")
    x, J_ref, lev = z["grid__state"], z["grid__J"], z["grid__levels"]
    B = len(x)
    eng, cfg = both(name, B, "f32", n_actor=10, pred_step_size=float(z["grid__pred_step_size"]), action_init=[0.0, 0.0])
    scale = np.max(np.abs(J_ref), axis=1)
    ref_i = np.argmin(J_ref, axis=1)
    # operator form: no env step
    eng.set_state(x)
    a, bj, bi = eng.actor_argmin(None, K=256)
    assert_kernel(eng, "k_actor")
    worst_op = 0.0
    for e in range(B):
        assert bi[e] == ref_i[e] or abs(J_ref[e, bi[e]] - J_ref[e, ref_i[e]]) <= 4 * TOL["f32"] * scale[e]
        np.testing.assert_array_equal(a[e], lev[bi[e]])
        worst_op = max(worst_op, abs(bj[e] - J_ref[e, bi[e]]) / scale[e])
    assert worst_op <= TOL["f32"]
    # the tick: rcg_control_tick (3wrobot: the fused k_ticks_pk; NI: k_sim + k_actor's packed instance) and
    # rcg_control_ticks with T = 1 (k_ticks_pk for both robots)
    worst = {}
    for entry in ("control_tick", "control_ticks"):
        eng, cfg = both(name, B, "f32", n_actor=10, pred_step_size=float(z["grid__pred_step_size"]), action_init=[0.0, 0.0])
        eng.set_state(x)
        if entry == "control_tick":
            eng.control_tick(None, K=256)
            ll = eng.last_launch(N.KERNEL_ACTOR)
            assert (ll["kernel"], ll["variant"]) == (("k_ticks", 8) if name == "3wrobot" else ("k_actor", 8)), ll
        else:
            eng.control_ticks(1, 256)
            ll = assert_kernel(eng, "k_ticks")
            assert ll["variant"] == 8, ll  # k_ticks_pk
        np.testing.assert_array_equal(eng.get_state(), x)  # at rest under a zero action: the env step is the identity
        bi2, bj2 = eng.get_field(N.FIELD_BEST_IDX), eng.get_field(N.FIELD_BEST_J)
        act = eng.get_field(N.FIELD_ACTION)
        w = 0.0
        for e in range(B):
            assert bi2[e] == ref_i[e] or abs(J_ref[e, bi2[e]] - J_ref[e, ref_i[e]]) <= 4 * TOL["f32"] * scale[e]
            np.testing.assert_array_equal(act[e], lev[bi2[e]])
            w = max(w, abs(bj2[e] - J_ref[e, bi2[e]]) / scale[e])
        assert w <= TOL["f32"]
        worst[entry] = w
        # upd_accum_obj of the tick: rho(obs, action) * sampling_time (controllers.py:1086-1093), oracle pinned on F2
        acc = O.stage_obj(x.astype(np.float64), act.astype(np.float64), cfg) * cfg.sampling_time
        assert rel_err_norm(eng.get_field(N.FIELD_ACCUM), acc) <= TOL["f32"]
        np.testing.assert_array_equal(eng.get_field(N.FIELD_STEP_IDX), np.ones(B, dtype=np.int32))
        eng.close()
    _report(f"generated grid {name} f32: k_actor (operator) best_J rel err {worst_op:.2e}, tick {worst}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name", ROBOTS)
def test_F13_sim_step_as_a_map(name, dtype):
    """k_sim on the reference's own time grid, every step a MAP from the reference's previous state rounded to the
    handle's element type (rcg_sim_step_h): against the float64 oracle from the same rounded state (1e-5; the oracle is
    pinned on the same trajectories in tests/test_large_heading_oracle.py) and, reported, against the reference's next
    state itself (which also holds the rounding of the input: up to 4.9e-4 rad of heading at 1e4 rad in float32)."""
    from rcognita_amd import _native as N

    meta, z = load_golden(f"F13_large_heading_{name}")
    for tr in meta["traj"]:
        t, y = z[f"traj_{tr['tag']}__t"], z[f"traj_{tr['tag']}__y"]
        u = np.array(tr["action"])
        n = len(t) - 1
        hs = np.diff(t)
        # the reference's step lengths differ along the grid: one handle per DISTINCT length would be wasteful - instead
        # all n steps are taken at once as n envs of one handle per distinct step length (regular part: one length)
        eng, cfg = both(name, n, dtype)
        x_in = y[:-1].astype(eng.real)
        out = np.zeros_like(x_in)
        for h in np.unique(hs):
            eng.set_state(x_in)
            eng.set_field(N.FIELD_ACTION, np.broadcast_to(u, (n, len(u))))
            eng.sim_step(1, step=float(h))
            sel = hs == h
            out[sel] = eng.get_state()[sel]
        orc = np.stack([O.rk4_step(cfg.sys_id, x_in[i].astype(np.float64), u, cfg.pars, cfg.ctrl_bnds, hs[i]) for i in range(n)])
        e_or = rel_err_norm(out, orc)
        e_ref = rel_err_norm(out, y[1:])
        _report(f"sim map {name} {dtype} {tr['tag']}: vs oracle (same rounded input) {e_or:.2e}, vs the reference's next state {e_ref:.2e}")
        assert e_or <= TOL[dtype], (tr["tag"], e_or)
        if dtype == "f64":
            assert e_ref <= 1e-5, (tr["tag"], e_ref)
        eng.close()


# Bound of a float32 FREE RUN (worst component, |x_hip - x_ref| / max(|x_ref|, 1); 0.5 s = 104 steps of the reference's
# grid).  A float32 heading is stored to half an ulp = 2^-24 |alpha| PER STEP, and an increment h * omega of a few ulps
# rounds the same way every step, so the heading may drift by n_steps * 2^-24 * |alpha| (6e-2 rad at 1e4 rad) and the
# position by speed * time * that drift: measured (profiles/r06_large_heading.txt) 1.3e-5 / 2.2e-4 / 2.3e-2 for the NI
# robot at 8 m/s and 8.8e-7 / 2.9e-5 / 3.3e-5 for the 3-wheel robot - a property of float32 STORAGE, stated in rcg.h
# (RCG_F32); float64 handles (the drop-in classes' default) stay inside 1e-5 at every heading.
FREE_RUN_BOUND_F32 = {72.0: 3e-5, 1e3: 5e-4, 1e4: 5e-2}


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name", ROBOTS)
def test_F13_free_run(name, dtype):
    """k_sim free-running along the reference's time grid (no re-synchronisation): float64 <= 1e-5 at every heading;
    float32 within the stated bound of what a float32 heading can hold."""
    from rcognita_amd import _native as N

    meta, z = load_golden(f"F13_large_heading_{name}")
    bad = []
    for tr in meta["traj"]:
        t, y = z[f"traj_{tr['tag']}__t"], z[f"traj_{tr['tag']}__y"]
        u = np.array(tr["action"])
        eng, cfg = both(name, 1, dtype)
        eng.set_state(y[0][None])
        eng.set_field(N.FIELD_ACTION, u[None])
        for h in np.diff(t):
            eng.sim_step(1, step=float(h))
        e = rel_err_norm(eng.get_state()[0], y[-1])
        bound = 1e-5 if dtype == "f64" else FREE_RUN_BOUND_F32[tr["A"]]
        _report(f"free run {name} {dtype} {tr['tag']} ({len(t) - 1} steps): {e:.2e} (bound {bound:.0e})")
        if e > bound:
            bad.append((tr["tag"], e, bound))
        eng.close()
    assert not bad, bad


def test_F6_hip_rk4_f32_vs_reference_rk45_to_minus_72_rad():
    """The -72-rad constant-torque trajectory of fixture F6 (3wrobot, u = (120, -35), 2 s) replayed by
    k_sim<float> on a regular dt / 2 grid: 400 float32 RK4 steps free-running against the reference's RK45 end point."""
    from rcognita_amd import _native as N

    meta, z = load_golden("F6_rk45_const_3wrobot")
    t, y = z["t"], z["y"]
    h = meta["dt"] / 2.0
    u = np.array(meta["action"])
    eng, cfg = both("3wrobot", 1, "f32", dt_sim=h)
    eng.set_state(y[0][None])
    eng.set_field(N.FIELD_ACTION, u[None])
    nsteps = int(round(t[-1] / h))
    eng.sim_step(nsteps)
    x_hip = eng.get_state()[0].astype(np.float64)
    x_end = O.rk4_step(cfg.sys_id, x_hip, u, cfg.pars, cfg.ctrl_bnds, t[-1] - nsteps * h)
    e = rel_err_norm(x_end, y[-1])
    _report(f"F6 3wrobot f32 free run to alpha = {y[-1][2]:.1f} rad, {nsteps} steps: {e:.2e}")
    assert y[-1][2] < -70
    assert e <= 3e-5, e
