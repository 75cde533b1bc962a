"""Shared implementation of the preset scripts ``presets/main_{3wrobot,3wrobot_NI,2tank}.py``.

The scripts keep the reference presets' command-line flags and defaults (presets/main_3wrobot.py:55-163,
main_3wrobot_NI.py, main_2tank.py), build the objects in the same order with the same constructor calls
(presets/main_3wrobot.py:218-320) and run the reference's headless loop (presets/main_3wrobot.py:415-468)
against ``rcognita_amd``, including its console / CSV log contract (``rcognita_amd/loggers.py``; with ``--batch > 1``
env 0 is what gets printed and logged).  Out of scope here, as in SURVEY.md 2: visualisation (``--is_visualization``
is accepted and ignored, the loop is always headless), the JACS controller, model estimation.
Added flags: ``--batch`` (run B perturbed copies of the env through the same objects), ``--n_candidates``,
``--rounds``, ``--dtype``, ``--device``, ``--seed``.
"""
from __future__ import annotations

import argparse
import os
import pathlib

import numpy as np

from . import controllers, loggers, simulator, systems

SPEC = {
    "3wrobot": dict(cls=systems.Sys3WRobot, dim_state=5, dim_input=2, dim_disturb=2, pars=[10, 1],
                    ctrl_bnds=[[-300, 300], [-100, 100]], dt=0.01, t1=10.0, state_init=["5", "5", "-3*pi/4", "0", "0"],
                    action_manual=[-5, -3], Nactor=5, mult=2.0, R_diag=[1, 10, 1, 0, 0, 0, 0], target=[],
                    action_init=[], modes=["manual", "nominal", "MPC", "RQL", "SQL", "JACS"], ctrl_mode="nominal"),
    "3wrobotNI": dict(cls=systems.Sys3WRobotNI, dim_state=3, dim_input=2, dim_disturb=2, pars=[],
                      ctrl_bnds=[[-25, 25], [-5, 5]], dt=0.01, t1=10.0, state_init=["5", "5", "-3*pi/4"],
                      action_manual=[-5, -3], Nactor=3, mult=1.0, R_diag=[1, 10, 1, 0, 0], target=[], action_init=[],
                      modes=["manual", "nominal", "MPC", "RQL", "SQL", "JACS"], ctrl_mode="nominal"),
    "2tank": dict(cls=systems.Sys2Tank, dim_state=2, dim_input=1, dim_disturb=1, pars=[18.4, 24.4, 1.3, 1, 0.2],
                  ctrl_bnds=[[0, 1]], dt=0.1, t1=100.0, state_init=["2", "-2"], action_manual=[0.5], Nactor=10,
                  mult=2.0, R_diag=[10, 10, 1], target=[0.5, 0.5], action_init=[0.5],
                  modes=["manual", "MPC", "RQL", "SQL"], ctrl_mode="MPC"),
}


def build_parser(name: str) -> argparse.ArgumentParser:
    s = SPEC[name]
    p = argparse.ArgumentParser(description=f"rcognita_amd preset: {name} (flags of the reference preset)")
    # default: 'nominal' for the two robots, 'MPC' for the tanks (presets/main_3wrobot.py:57-64, main_2tank.py:55-60)
    p.add_argument("--ctrl_mode", type=str, choices=s["modes"], default=s["ctrl_mode"])
    p.add_argument("--dt", type=float, default=s["dt"])
    p.add_argument("--t1", type=float, default=s["t1"])
    p.add_argument("--Nruns", type=int, default=1)
    p.add_argument("--state_init", type=str, nargs="+", default=s["state_init"])
    p.add_argument("--is_log_data", type=bool, default=False)
    p.add_argument("--is_visualization", type=bool, default=True)
    p.add_argument("--is_print_sim_step", type=bool, default=True)
    p.add_argument("--is_est_model", type=bool, default=False)
    p.add_argument("--model_est_stage", type=float, default=1.0)
    p.add_argument("--model_est_period_multiplier", type=float, default=1)
    p.add_argument("--model_order", type=int, default=5)
    p.add_argument("--prob_noise_pow", type=float, default=False)
    p.add_argument("--action_manual", type=float, default=s["action_manual"], nargs="+")
    p.add_argument("--Nactor", type=int, default=s["Nactor"])
    p.add_argument("--pred_step_size_multiplier", type=float, default=s["mult"])
    p.add_argument("--buffer_size", type=int, default=10)
    p.add_argument("--stage_obj_struct", type=str, default="quadratic", choices=["quadratic", "biquadratic"])
    p.add_argument("--R1_diag", type=float, nargs="+", default=s["R_diag"])
    p.add_argument("--R2_diag", type=float, nargs="+", default=s["R_diag"])
    p.add_argument("--Ncritic", type=int, default=4)
    p.add_argument("--gamma", type=float, default=1.0)
    p.add_argument("--critic_period_multiplier", type=float, default=1.0)
    p.add_argument("--critic_struct", type=str, default="quad-nomix",
                   choices=["quad-lin", "quadratic", "quad-nomix", "quad-mix"])
    p.add_argument("--actor_struct", type=str, default="quad-nomix", choices=["quad-lin", "quadratic", "quad-nomix"])
    # build-specific
    p.add_argument("--batch", type=int, default=1, help="number of envs run through the same objects")
    p.add_argument("--state_spread", type=float, default=0.5, help="uniform perturbation of state_init per env (batch>1)")
    p.add_argument("--n_candidates", type=int, default=256)
    p.add_argument("--rounds", type=int, default=6)
    p.add_argument("--dtype", type=str, default="f64", choices=["f32", "f64"],
                   help="element type of the device handles; the reference computes in float64")
    p.add_argument("--device", type=int, default=0)
    p.add_argument("--seed", type=int, default=0)
    return p


def run(name: str, argv=None):
    """Build the objects as the reference preset does and run its headless loop.  Returns a dict with the
    final time, state, action and accumulated objective (per env when ``--batch > 1``)."""
    s = SPEC[name]
    args = build_parser(name).parse_args(argv)
    if args.ctrl_mode == "JACS":
        raise SystemExit("--ctrl_mode JACS: the stabilising RL controller is out of scope of the native path "
                         "(SURVEY.md 2, component 5); use manual, nominal, MPC, RQL or SQL")
    if args.is_est_model:
        raise SystemExit("--is_est_model: model estimation needs the absent `sippy` package (out of scope)")
    # arithmetic expressions with `pi`, as the reference accepts (presets/main_3wrobot.py:166-170), but evaluated
    # without builtins; the int/float type of each entry is kept because it shows in the log header cell
    state_init_as_given = np.array([eval(v.replace("pi", str(np.pi)), {"__builtins__": {}}, {})
                                    for v in args.state_init])
    state_init = state_init_as_given.astype(float)
    dim_state, dim_input = s["dim_state"], s["dim_input"]
    assert args.t1 > args.dt > 0.0
    assert state_init.size == dim_state
    pred_step_size = args.dt * args.pred_step_size_multiplier
    critic_period = args.dt * args.critic_period_multiplier
    R1, R2 = np.diag(np.array(args.R1_diag)), np.diag(np.array(args.R2_diag))
    ctrl_bnds = np.array(s["ctrl_bnds"], dtype=float)
    t0 = 0
    if args.batch > 1:
        rng = np.random.default_rng(args.seed)
        state_init = state_init + rng.uniform(-args.state_spread, args.state_spread, (args.batch, dim_state))

    # ---- system, controller, simulator: constructor calls of presets/main_3wrobot.py:218-320 ----------
    my_sys = s["cls"](sys_type="diff_eqn", dim_state=dim_state, dim_input=dim_input, dim_output=dim_state,
                      dim_disturb=s["dim_disturb"], pars=list(s["pars"]), ctrl_bnds=ctrl_bnds, is_dyn_ctrl=0,
                      is_disturb=0, pars_disturb=[], dtype=args.dtype, device=args.device)
    my_ctrl_benchm = controllers.CtrlOptPred(
        dim_input, dim_state, args.ctrl_mode if args.ctrl_mode not in ("manual", "nominal") else "MPC", ctrl_bnds=ctrl_bnds,
        action_init=s["action_init"], t0=t0, sampling_time=args.dt, Nactor=args.Nactor, pred_step_size=pred_step_size,
        sys_rhs=my_sys._state_dyn, sys_out=my_sys.out, state_sys=state_init, prob_noise_pow=args.prob_noise_pow,
        is_est_model=0, model_est_stage=args.model_est_stage, model_est_period=args.dt * args.model_est_period_multiplier,
        buffer_size=args.buffer_size, model_order=args.model_order, model_est_checks=0, gamma=args.gamma,
        Ncritic=args.Ncritic, critic_period=critic_period, critic_struct=args.critic_struct,
        stage_obj_struct=args.stage_obj_struct, stage_obj_pars=[R1, R2] if args.stage_obj_struct == "biquadratic" else [R1],
        observation_target=np.array(s["target"], dtype=float) if len(s["target"]) else [],
        n_candidates=args.n_candidates, rounds=args.rounds, seed=args.seed, dtype=args.dtype, device=args.device)
    # nominal controller: presets/main_3wrobot.py:239 (gain 5), main_3wrobot_NI.py:235 (gain 0.5); 2tank has none
    my_ctrl_nominal = None
    if name == "3wrobot":
        my_ctrl_nominal = controllers.CtrlNominal3WRobot(s["pars"][0], s["pars"][1], ctrl_gain=5, ctrl_bnds=ctrl_bnds,
                                                         t0=t0, sampling_time=args.dt, dtype=args.dtype, device=args.device)
    elif name == "3wrobotNI":
        my_ctrl_nominal = controllers.CtrlNominal3WRobotNI(ctrl_gain=0.5, ctrl_bnds=ctrl_bnds, t0=t0,
                                                           sampling_time=args.dt, dtype=args.dtype, device=args.device)
    my_simulator = simulator.Simulator(
        sys_type="diff_eqn", closed_loop_rhs=my_sys.closed_loop_rhs, sys_out=my_sys.out, state_init=state_init,
        disturb_init=[], action_init=np.zeros(dim_input) if not len(s["action_init"]) else np.array(s["action_init"]),
        t0=t0, t1=args.t1, dt=args.dt, max_step=args.dt / 2, first_step=1e-6, atol=1e-5, rtol=1e-3, is_disturb=0,
        is_dyn_ctrl=0, dtype=args.dtype, device=args.device)

    # ---- logger: presets/main_3wrobot.py:322-368 ------------------------------------------------------
    my_logger = loggers.LOGGERS[name]()
    in_presets_dir = os.path.basename(os.path.normpath(os.path.abspath(os.getcwd()))) == "presets"
    data_folder = "../simdata" if in_presets_dir else "simdata"
    datafiles = loggers.datafile_names(data_folder, my_sys.name, args.ctrl_mode, args.Nruns)
    if args.is_log_data:
        pathlib.Path(data_folder).mkdir(parents=True, exist_ok=True)
        settings = {k: getattr(args, k) for k in loggers.HEADER_KEYS if k != "state_init"}
        settings["state_init"] = state_init_as_given
        for f in datafiles:
            print("Logging data to:    " + f)
            loggers.write_header(f, my_sys.name, args.ctrl_mode, settings, my_logger.columns)
    datafile = datafiles[0]

    action_manual = np.array(args.action_manual, dtype=float)
    run_curr, ticks = 1, 0
    while True:  # presets/main_3wrobot.py:417-468
        my_simulator.sim_step()
        t, state, observation, state_full = my_simulator.get_sim_step_data()
        action = controllers.ctrl_selector(t, observation, action_manual, my_ctrl_nominal, my_ctrl_benchm, args.ctrl_mode)
        my_sys.receive_action(action)
        my_ctrl_benchm.receive_sys_state(my_sys._state)
        my_ctrl_benchm.upd_accum_obj(observation, action)
        stage_obj = my_ctrl_benchm.stage_obj(observation, action)
        accum_obj = my_ctrl_benchm.accum_obj_val
        ticks += 1
        # env 0 is the logged env; cells are Python floats (csv writes their shortest round-trip repr)
        first = lambda a: np.asarray(a, dtype=float).reshape(-1, np.asarray(a).shape[-1])[0] if np.ndim(a) else float(a)
        x0, u0 = first(state_full), first(action)
        so, ao = float(np.ravel(stage_obj)[0]), float(np.ravel(accum_obj)[0])
        cells = (float(t), *map(float, x0), u0, so, ao) if name == "2tank" else \
            (float(t), *map(float, x0), so, ao, u0)
        if args.is_print_sim_step:
            my_logger.print_sim_step(*cells)
        if args.is_log_data:
            my_logger.log_data_row(datafile, *cells)
        if t >= args.t1:
            if args.is_print_sim_step:
                print(".....................................Run {run:2d} done.....................................".format(
                    run=run_curr))
            run_curr += 1
            if run_curr > args.Nruns:
                break
            if args.is_log_data:
                datafile = datafiles[run_curr - 1]
            my_simulator.reset()
            if args.ctrl_mode != "nominal":  # presets/main_3wrobot.py:463-466
                my_ctrl_benchm.reset(t0)
            else:
                my_ctrl_nominal.reset(t0)
    return dict(t=t, state=np.array(state_full), action=np.array(action), accum_obj=np.array(accum_obj), ticks=ticks)
