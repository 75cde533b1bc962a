"""The HIP path against the REFERENCE's own closed-loop traces (fixtures F7 / F7_long: per sim step t, state, action,
stage_obj, accum_obj captured from the reference's loop, presets/main_3wrobot.py:419-446, SciPy RK45 + SLSQP).

The mirror classes are wired as the presets wire them and driven by the reference's loop body, verbatim call order
(sim_step -> ctrl_selector -> receive_action -> receive_sys_state(my_sys._state) -> upd_accum_obj), in float64, with the
reference's structure of time: simulation steps of dt / 2 (its RK45 runs at max_step = dt / 2, simulator.py:150), one
control tick per sampling time dt, upd_accum_obj at EVERY simulation step (App. A-5: accum_obj is twice the integral), the
rollout started from the state of the previous simulation step (App. A-2), every decision optimised from action_sqn_init
(no warm start, controllers.py:1379).  ``gpu`` marked.

What cannot agree bit for bit, and therefore the bands (hard parts 1 and 4):
  * the reference's tick instants sit on the irregular grid of its adaptive solver (first steps 1e-6 .. 3e-3, then
    strides of dt / 2 offset by ~1e-3): it makes ~97 decisions per simulated second, the fixed grid here 100, and each
    decision acts up to dt / 2 earlier or later;
  * the decision itself: SLSQP with finite-difference gradients there, the on-device adjoint-gradient optimiser here
    (within 0.2 % of SLSQP's cost on the F8 states, tests/test_hip_optimizer.py); RQL: candidate search + native fit.
So trajectories are compared as trajectories of the same closed loop, at the reference's end time:
  * accum_obj over [2 dt, t1] within 6 % (measured: 0.1 .. 4.1 %; the largest is the 3-second 3wrobot run, where this
    loop, with 3 % more decisions per second, parks the robot at a LOWER running cost than the reference);
  * position-like state components within 5 % of max(range the reference trajectory covers, 0.2 x its largest
    magnitude) (measured < 2 % of the range on every trace longer than 0.2 s);
  * the rate components of Sys3WRobot (v, omega: integrals of the bang-bang inputs F / m <= 30, M / I <= 100) within one
    sampling period of full actuation, a_max * dt = (0.3, 1.0): a decision taken dt / 2 earlier or later moves them by
    up to half of that, whatever the controller (measured 0.11, 0.29);
  * on the 3-second runs the SURVEY's section-6 datapoint (reference SLSQP: accum_obj 389.0 on 3wrobot) is the fixture's
    last row.
"""
import numpy as np
import pytest

from tests.conftest import load_golden
from tests.helpers import PRESETS
from tests.test_dropin_api import CLS, DIMS

pytestmark = pytest.mark.gpu

ACCUM_BAND = 0.06
STATE_BAND = 0.05
# rate components: index -> largest |d/dt| the bounded inputs can produce (presets/main_3wrobot.py:207-215: F <= 300, m = 10;
# M <= 100, I = 1)
RATE = {"3wrobot": {3: 30.0, 4: 100.0}}


def make_loop_objects(name, mode, Nactor, t1, x0=None, critic_struct="quad-nomix", dtype="f64", **ctrl_kw):
    """System, controller and simulator wired as the reference's presets wire them (presets/main_3wrobot.py:199-300)."""
    from rcognita_amd import controllers, simulator, systems

    p = PRESETS[name]
    ds, du, dd = DIMS[name]
    ctrl_bnds = np.array(p["bnds"], dtype=float)
    dt = p["dt"]
    my_sys = getattr(systems, CLS[name])(sys_type="diff_eqn", dim_state=ds, dim_input=du, dim_output=ds, dim_disturb=dd,
                                         pars=list(p["pars"]), ctrl_bnds=ctrl_bnds, is_dyn_ctrl=0, is_disturb=0,
                                         pars_disturb=[], dtype=dtype)
    x0 = np.array(p["x0"] if x0 is None else x0, dtype=float)
    my_ctrl = controllers.CtrlOptPred(du, ds, mode, ctrl_bnds=ctrl_bnds, action_init=[0.5] if name == "2tank" else [],
                                      t0=0, sampling_time=dt, Nactor=Nactor, pred_step_size=dt * p["mult"],
                                      sys_rhs=my_sys._state_dyn, sys_out=my_sys.out, state_sys=x0, prob_noise_pow=8,
                                      is_est_model=0, model_est_stage=2, model_est_period=dt, buffer_size=10,
                                      model_order=5, model_est_checks=0, gamma=1, Ncritic=4, critic_period=dt,
                                      critic_struct=critic_struct, stage_obj_struct="quadratic",
                                      stage_obj_pars=[np.diag(np.array(p["R1"], dtype=float))],
                                      observation_target=[] if p["target"] is None else np.array(p["target"]),
                                      dtype=dtype, **ctrl_kw)
    # simulation steps of dt / 2: what the reference's solver takes (max_step = dt / 2 hard-coded, simulator.py:150)
    my_sim = simulator.Simulator(sys_type="diff_eqn", closed_loop_rhs=my_sys.closed_loop_rhs, sys_out=my_sys.out,
                                 state_init=x0, disturb_init=[], action_init=np.zeros(du), t0=0, t1=t1, dt=dt / 2,
                                 max_step=dt / 2, first_step=1e-6, atol=1e-5, rtol=1e-3, is_disturb=0, is_dyn_ctrl=0,
                                 dtype=dtype)
    return my_sys, my_ctrl, my_sim


def run_reference_loop(name, mode, Nactor, t1, x0=None, critic_struct="quad-nomix", times=None, **ctrl_kw):
    """The reference's headless loop on the mirror classes; returns rows [t, state..., action..., stage_obj, accum_obj].

    ``times`` (the t column of a reference trace): walk THAT time grid - every simulation step ends where the reference's
    adaptive solver ended it (``Simulator.sim_step(t_next=...)``, rcg_sim_step_h) and the controller samples with the
    reference's bare float comparison (``clock_tol = 0``), so every decision is taken at the instant the reference took it.
    Default: the build's own fixed grid of dt / 2."""
    from rcognita_amd import controllers

    if times is not None:
        ctrl_kw = dict(ctrl_kw, clock_tol=0.0)
    my_sys, my_ctrl, my_sim = make_loop_objects(name, mode, Nactor, t1, x0=x0, critic_struct=critic_struct, **ctrl_kw)
    du = DIMS[name][1]
    rows = []
    k = 0
    while True:  # presets/main_3wrobot.py:419-446
        my_sim.sim_step(**({} if times is None else {"t_next": float(times[k])}))
        k += 1
        t, state, observation, state_full = my_sim.get_sim_step_data()
        action = controllers.ctrl_selector(t, observation, np.zeros(du), None, my_ctrl, mode)
        my_sys.receive_action(action)
        my_ctrl.receive_sys_state(my_sys._state)
        my_ctrl.upd_accum_obj(observation, action)
        rows.append(np.concatenate([[t], np.array(state_full, dtype=float), np.array(action, dtype=float),
                                    [my_ctrl.stage_obj(observation, action), my_ctrl.accum_obj_val]]))
        if (t >= t1 - 1e-12) if times is None else (k == len(times)):
            break
    return np.stack(rows)


def compare(rows, ref, ds, what, dt, name, skip=()):
    assert abs(rows[-1, 0] - ref[-1, 0]) < 1e-9, (rows[-1, 0], ref[-1, 0])  # same end time
    # accum_obj over the window [2 dt, t1].  The reference's solver starts with steps of 1e-6, 1e-5, 1e-4, 1e-3 ... and
    # upd_accum_obj adds a FULL stage_obj * sampling_time at each of them (controllers.py:1093 is called per simulation
    # step, App. A-5), so its accum_obj carries 4-5 extra increments of the initial stage cost that no fixed-step loop
    # has; past the start-up both loops add one increment per dt / 2.
    def window(r):
        i0 = int(np.argmin(np.abs(r[:, 0] - 2 * dt)))
        return r[-1, -1] - r[i0, -1]

    acc, acc_ref = window(rows), window(ref)
    rel = abs(acc - acc_ref) / abs(acc_ref)
    # per-component range of the reference trajectory: the scale a final-state difference is measured against
    span = np.maximum(ref[:, 1:1 + ds].max(0) - ref[:, 1:1 + ds].min(0), 0.2 * np.abs(ref[:, 1:1 + ds]).max(0))
    dx = np.abs(rows[-1, 1:1 + ds] - ref[-1, 1:1 + ds]) / span
    rate = RATE.get(name, {})
    pos = [c for c in range(ds) if c not in rate and c not in skip]
    drate = {c: abs(rows[-1, 1 + c] - ref[-1, 1 + c]) for c in rate if c not in skip}
    print(f"\nTRACE {what}: accum_obj over [2 dt, t1] {acc:.4f} vs reference {acc_ref:.4f} ({rel:.2%}); totals "
          f"{rows[-1, -1]:.3f} / {ref[-1, -1]:.3f}; final-state error / range {np.round(dx[pos], 4)}; rate components "
          f"{ {c: round(v, 3) for c, v in drate.items()} }")
    assert rel <= ACCUM_BAND, f"{what}: accum_obj {acc} vs the reference's {acc_ref}"
    assert np.all(dx[pos] <= STATE_BAND), f"{what}: final state {rows[-1, 1:1 + ds]} vs the reference's {ref[-1, 1:1 + ds]}"
    for c, amax in rate.items():
        if c in skip:
            continue
        assert drate[c] <= amax * dt, f"{what}: rate component {c} off by {drate[c]} > a_max dt = {amax * dt}"


# (F7_trace_2tank_RQL equals the MPC trace to 5e-11 - the input is saturated for the whole run - and is no longer replayed: it
# adds no evidence about the critic; the critic modes are F7c's business)
@pytest.mark.parametrize("name,mode", [("3wrobotNI", "MPC"), ("3wrobot", "MPC"), ("2tank", "MPC")])
def test_F7_reference_trace_through_the_mirror_classes(name, mode):
    meta, z = load_golden(f"F7_trace_{name}_{mode}")
    rows = run_reference_loop(name, mode, meta["Nactor"], meta["t1"])  # every decision is the device's: on-device optimiser
    compare(rows, z["rows"], DIMS[name][0], f"F7 {name} {mode}", meta["dt"], name)


@pytest.mark.parametrize("name", ["3wrobot", "3wrobotNI", "2tank"])
def test_F7_long_reference_trace_and_the_survey_quality_datapoint(name):
    meta, z = load_golden(f"F7_long_{name}_MPC")
    ref = z["rows"]
    if name == "3wrobot":  # "accum_obj after 3 s, 3wrobot MPC Nactor=5: SLSQP 389.0"
        assert abs(ref[-1, -1] - 389.0) < 0.05 and abs(ref[-1, 0] - 3.0) < 1e-9
    rows = run_reference_loop(name, "MPC", meta["Nactor"], meta["t1"])
    # Sys3WRobot's preset puts no weight on the speed (R1 = diag[1, 10, 1, 0, 0, 0, 0]): once y and alpha have settled the
    # robot swings through x = 0 under bang-bang thrust (reference at t = 3 s: x = -2.6, v = 14.3, F = 300) - an undamped
    # oscillation whose phase after 3 s depends on every decision instant.  x and v are therefore compared through the
    # running cost only; y, alpha, omega and every component of the other two systems at the end point.
    compare(rows, ref, DIMS[name][0], f"F7_long {name}", meta["dt"], name, skip=(0, 3) if name == "3wrobot" else ())
    # the running cost as a curve: at one and two thirds of the run as well
    for frac in (1 / 3, 2 / 3):
        t = frac * meta["t1"]
        i, j = int(np.argmin(np.abs(rows[:, 0] - t))), int(np.argmin(np.abs(ref[:, 0] - t)))
        i0, j0 = int(np.argmin(np.abs(rows[:, 0] - 2 * meta["dt"]))), int(np.argmin(np.abs(ref[:, 0] - 2 * meta["dt"])))
        a, b = rows[i, -1] - rows[i0, -1], ref[j, -1] - ref[j0, -1]
        assert abs(a - b) <= ACCUM_BAND * abs(b), (name, t, a, b)
    # the decision count: one per sampling time here, ~97 % of that in the reference (its irregular time grid)
    du = DIMS[name][1]
    n_ticks = int(round(meta["t1"] / meta["dt"]))
    assert rows.shape[0] == 2 * n_ticks


CRITIC_CASES = [("3wrobotNI", "quad-nomix"), ("3wrobotNI", "quad-mix"), ("3wrobot", "quad-nomix"), ("2tank", "quad-nomix"),
                ("2tank", "quadratic"), ("2tank", "quad-lin")]
@pytest.mark.parametrize("mode", ["RQL", "SQL"])
@pytest.mark.parametrize("name,cs", CRITIC_CASES)
def test_F7c_critic_mode_traces_where_the_critic_steers(name, cs, mode):
    """Fixtures F7c (oracle/gen_critic_fixtures.py): the reference's loop in RQL / SQL from starts where its critic decides
    differently from MPC - the generator keeps a start only if the reference's MPC run from it lies at least TWO bands
    away (meta ``discriminating``; one combination has no such start among 47 and says so).  Every decision here is the
    device's: k_critic_fit on the buffers the loop itself filled, k_actor_opt on the fitted weights.

    Two free runs per trace.
    (1) On the REFERENCE'S TIME GRID (``times`` = the fixture's t column): same step and decision instants, so what is left
    is the build's integrator (RK4 on the reference's steps), fit and optimiser.  band = max(6 %, 2 x the distance the
    reference's OWN loop moves when only SLSQP's tolerance changes) (F7c_sensitivity.json; 6 % being the MPC traces' band).
    Asserted at two thirds of the run and at its end:
      (a) within the band of the reference's loop WITH ITS CRITIC SOLVER EXCHANGED for the exact minimiser of the
          build-defined fit (``exact_critic`` of F7c_sensitivity.json: oracle/ref_loop.py, which reproduces every trace bit
          for bit, with that one ingredient swapped) - on every trace.  SLSQP leaves 4 .. 74 % of the robots' fits at their
          start point w_init, whatever its tolerance (fixture fields tick_w, tick_critic_status); what that habit does to a
          trace is a property of SciPy's SLSQP, not of the algorithm, and this is the loop a build with an exact fit follows;
      (b) within the band of the REFERENCE'S TRACE itself wherever that exchange moves the reference's loop by less than HALF
          the band (9 of 12 traces; the three where it moves it further - 3wrobotNI RQL quad-mix: 15.6 %, 2tank RQL quadratic:
          7.8 %, 3wrobot RQL quad-nomix: 5.1 % of a 6 % band - are named by measurement, not by hand:
          profiles/r05_critic_loop_attribution.txt shows the actor and grid exchanges at <= 1 % on the first two; on the third,
          whose preset puts no weight on the inputs, re-associating four sums of the optimiser moved the device's run from 4.5
          to 6.2 % of the reference while it stayed within 1 % of (a)'s loop);
      (c) whenever the reference's MPC run is more than 6 % away, whatever the band: on the critic's side - closer to the
          critic-mode run than the MPC run is.
    (2) On the build's own fixed grid of dt / 2 with one decision per dt (the reference's float clock test makes 25 decisions
    in 3 s on the tank where this grid makes 30, ~97 per 100 on the robots: a DIFFERENT sampled-data loop).
    Printed, and held to (c) only.
    What is held to 0.5 % / to the fit's own objective / to the first action are the decisions of EVERY tick given the
    reference's inputs: tests/test_hip_teacher_forced.py."""
    import json
    import os

    from tests.conftest import GOLDEN

    meta, z = load_golden(f"F7c_trace_{name}_{mode}_{cs}")
    with open(os.path.join(GOLDEN, "F7c_sensitivity.json")) as f:
        sj = json.load(f)["traces"][f"{name}_{mode}_{cs}"]
    sens, ex = sj["sensitivity"], sj["exact_critic"]
    band = max(ACCUM_BAND, 2.0 * sens)
    assert abs(band - meta["band"]) < 1e-9  # the band the generator chose the start by
    ref, mpc = z["rows"], z["rows_mpc"]
    dt, t1 = meta["dt"], meta["t1"]

    def window(r, t):
        i0, i1 = int(np.argmin(np.abs(r[:, 0] - 2 * dt))), int(np.argmin(np.abs(r[:, 0] - t)))
        return r[i1, -1] - r[i0, -1]

    b, m = window(ref, t1), window(mpc, t1)
    mpc_gap = abs(m - b) / abs(b)
    assert meta["discriminating"] == (mpc_gap >= 2 * band)
    assert abs(window(ref, 2 / 3 * t1) - ex["ref_window_23"]) < 1e-9 * abs(b)
    for grid in ("reference", "fixed"):
        rows = run_reference_loop(name, mode, meta["Nactor"], t1, x0=meta["x0"], critic_struct=cs,
                                  times=ref[:, 0] if grid == "reference" else None)
        assert abs(rows[-1, 0] - ref[-1, 0]) < 1e-9
        rel = {fr: abs(window(rows, fr * t1) - window(ref, fr * t1)) / abs(window(ref, fr * t1)) for fr in (1 / 3, 2 / 3, 1.0)}
        a = window(rows, t1)
        relx = {2 / 3: abs(window(rows, 2 / 3 * t1) - ex["window_23"]) / abs(ex["window_23"]), 1.0: abs(a - ex["window_1"]) / abs(ex["window_1"])}
        print(f"\nTRACE F7c {name} {mode} {cs} [{grid} grid]: accum_obj over [2 dt, t1] {a:.4f} vs reference {b:.4f} "
              f"({rel[1.0]:.2%}; at 1/3, 2/3: {rel[1 / 3]:.2%}, {rel[2 / 3]:.2%}), vs the reference's loop with an exact critic fit "
              f"{ex['window_1']:.4f} ({relx[1.0]:.2%}; at 2/3: {relx[2 / 3]:.2%}; that exchange alone: {ex['shift']:.2%}); band "
              f"{band:.1%} (sensitivity of the reference's own loop {sens:.2%}); the reference's MPC run: {m:.4f} ({mpc_gap:.2%} away)")
        if grid == "reference":
            assert rows.shape[0] == ref.shape[0] and np.array_equal(rows[:, 0], ref[:, 0])
            for fr in (2 / 3, 1.0):
                assert relx[fr] <= band, (f"{name} {mode} {cs}: running cost at {fr:.2f} t1 is {relx[fr]:.2%} > {band:.2%} away from "
                                          "the reference's loop with an exact critic fit")
                if ex["shift"] <= 0.5 * band:
                    assert rel[fr] <= band, f"{name} {mode} {cs}: running cost at {fr:.2f} t1 off by {rel[fr]:.2%} > {band:.2%}"
        else:
            assert rows.shape[0] == 2 * int(round(t1 / dt))
        if mpc_gap > ACCUM_BAND:  # the modes are told apart: the device's loop is on the critic's side
            assert abs(a - b) < abs(m - b), (grid, a, b, m)
