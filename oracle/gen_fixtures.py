#!/usr/bin/env python3
"""Generate golden vectors under tests/golden/ by IMPORTING the reference (read-only).

Runs only in the build container, where /root/reference exists; the GPU box never sees the
reference, only the small ``.npz`` files this script writes.  The files hold data only: inputs and
the outputs the reference computed for them.

    python oracle/gen_fixtures.py            # writes tests/golden/F*.npz

Import recipe: SURVEY.md Appendix B (two GUI-only modules are stubbed before ``import rcognita``;
a ``TargetArray`` ndarray subclass restores the pre-NumPy-1.25 meaning of
``observation_target == []`` so that the 2tank path runs, without touching the reference).
"""
import json
import os
import sys
import types
import warnings

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def import_reference():
    sys.dont_write_bytecode = True
    m = types.ModuleType("mpldatacursor")
    m.datacursor = lambda *a, **k: None
    sys.modules["mpldatacursor"] = m
    s = types.ModuleType("svgpath2mpl")
    s.parse_path = lambda *a, **k: None
    sys.modules["svgpath2mpl"] = s
    import matplotlib

    matplotlib.use("Agg")
    sys.path.insert(0, REF)
    warnings.filterwarnings("ignore")
    from rcognita import controllers, simulator, systems

    return systems, simulator, controllers


class TargetArray(np.ndarray):
    def __new__(cls, a):
        return np.asarray(a, dtype=float).view(cls)

    def __eq__(self, o):
        if isinstance(o, list) and not o:
            return False
        return np.asarray(self) == o

    __hash__ = None


# preset constants (presets/main_3wrobot.py:45-47,207-215; main_3wrobot_NI.py:45-48,207-211;
# main_2tank.py:45-48,199-211)
PRESETS = {
    "3wrobot": dict(
        cls="Sys3WRobot", ds=5, du=2, dd=2, pars=[10.0, 1.0], bnds=[[-300, 300], [-100, 100]],
        R1=[1, 10, 1, 0, 0, 0, 0], dt=0.01, mult=2.0, x0=[5, 5, -3 * np.pi / 4, 0, 0], target=None, Nactor=5,
        action_init=None,
    ),
    "3wrobotNI": dict(
        cls="Sys3WRobotNI", ds=3, du=2, dd=2, pars=[], bnds=[[-25, 25], [-5, 5]],
        R1=[1, 10, 1, 0, 0], dt=0.01, mult=1.0, x0=[5, 5, -3 * np.pi / 4], target=None, Nactor=3,
        action_init=None,
    ),
    "2tank": dict(
        cls="Sys2Tank", ds=2, du=1, dd=1, pars=[18.4, 24.4, 1.3, 1.0, 0.2], bnds=[[0, 1]],
        R1=[10, 10, 1], dt=0.1, mult=2.0, x0=[2, -2], target=[0.5, 0.5], Nactor=10,
        action_init=[0.5],
    ),
}


def make_sys(systems, name):
    p = PRESETS[name]
    return getattr(systems, p["cls"])(
        sys_type="diff_eqn", dim_state=p["ds"], dim_input=p["du"], dim_output=p["ds"], dim_disturb=p["dd"],
        pars=list(p["pars"]), ctrl_bnds=np.array(p["bnds"], dtype=float), is_dyn_ctrl=0, is_disturb=0, pars_disturb=[],
    )


def make_ctrl(controllers, sys_obj, name, mode="MPC", Nactor=None, gamma=1.0, critic_struct="quad-nomix",
              stage="quadratic", R1=None, R2=None, target="preset", state_sys=None, Ncritic=4, buffer_size=10,
              pred_step_size=None, action_init="preset"):
    p = PRESETS[name]
    n = p["ds"] + p["du"]
    R1 = np.diag(np.array(p["R1"], dtype=float)) if R1 is None else np.asarray(R1, dtype=float)
    pars = [R1] if R2 is None else [R1, np.asarray(R2, dtype=float)]
    if isinstance(target, str) and target == "preset":
        target = p["target"]
    tgt = [] if target is None else TargetArray(target)
    if isinstance(action_init, str) and action_init == "preset":
        action_init = p["action_init"]
    ai = [] if action_init is None else np.asarray(action_init, dtype=float)
    return controllers.CtrlOptPred(
        p["du"], p["ds"], mode, ctrl_bnds=np.array(p["bnds"], dtype=float), action_init=ai, t0=0,
        sampling_time=p["dt"], Nactor=p["Nactor"] if Nactor is None else Nactor,
        pred_step_size=p["dt"] * p["mult"] if pred_step_size is None else pred_step_size,
        sys_rhs=sys_obj._state_dyn, sys_out=sys_obj.out,
        state_sys=np.asarray(p["x0"], dtype=float) if state_sys is None else state_sys,
        prob_noise_pow=8, is_est_model=0, model_est_stage=2, model_est_period=p["dt"], buffer_size=buffer_size,
        model_order=5, model_est_checks=0, gamma=gamma, Ncritic=Ncritic, critic_period=p["dt"],
        critic_struct=critic_struct, stage_obj_struct=stage, stage_obj_pars=pars, observation_target=tgt,
    )


def rand_states(rng, name, n):
    if name == "3wrobot":
        return np.stack([rng.uniform(-10, 10, n), rng.uniform(-10, 10, n), rng.uniform(-8, 8, n),
                         rng.uniform(-3, 3, n), rng.uniform(-3, 3, n)], axis=-1)
    if name == "3wrobotNI":
        return np.stack([rng.uniform(-10, 10, n), rng.uniform(-10, 10, n), rng.uniform(-8, 8, n)], axis=-1)
    return np.stack([rng.uniform(0, 2, n), rng.uniform(-2, 2, n)], axis=-1)


def rand_actions(rng, name, shape, overshoot=1.0):
    b = np.array(PRESETS[name]["bnds"], dtype=float)
    mid, half = b.mean(axis=1), 0.5 * (b[:, 1] - b[:, 0])
    return mid + overshoot * half * rng.uniform(-1, 1, tuple(shape) + (b.shape[0],))


def save(name, meta, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, meta=np.array(json.dumps(meta)), **arrays)
    print(f"wrote {path}  ({os.path.getsize(path)} bytes)")


def main():
    systems, simulator, controllers = import_reference()
    rng = np.random.default_rng(20261003)

    # ---------------------------------------------------------------- F1: RHS (+ clip)
    for name in PRESETS:
        sys_obj = make_sys(systems, name)
        n = 256
        x = rand_states(rng, name, n)
        u = rand_actions(rng, name, (n,), overshoot=1.6)  # ~40 % out of bounds
        dyn = np.stack([sys_obj._state_dyn(0.0, x[i], u[i].copy()) for i in range(n)])
        clrhs, clipped = [], []
        for i in range(n):
            sys_obj.receive_action(u[i].copy())
            clrhs.append(sys_obj.closed_loop_rhs(0.0, x[i]))
            clipped.append(sys_obj.action.copy())
        save(f"F1_rhs_{name}", dict(system=name, pars=PRESETS[name]["pars"], bnds=PRESETS[name]["bnds"]),
             state=x, action=u, state_dyn=dyn, closed_loop_rhs=np.stack(clrhs), clipped_action=np.stack(clipped))

    # ---------------------------------------------------------------- F2: stage objective
    for name in PRESETS:
        p = PRESETS[name]
        nn = p["ds"] + p["du"]
        sys_obj = make_sys(systems, name)
        n = 128
        y = rand_states(rng, name, n)
        u = rand_actions(rng, name, (n,))
        R1d = np.diag(np.array(p["R1"], dtype=float))
        A = rng.uniform(-1, 1, (nn, nn))
        R1f = A @ A.T  # full symmetric
        R1n = rng.uniform(-1, 1, (nn, nn))  # full, NOT symmetric: chi @ R1 @ chi is still defined
        R2f = np.diag(rng.uniform(0, 1e-3, nn)) + 1e-4 * (A.T @ A)
        tgt = rng.uniform(-1, 1, p["ds"])
        out = {}
        for tag, kw in {
            "quad_diag": dict(R1=R1d, target=None),
            "quad_full": dict(R1=R1f, target=None),
            "quad_nonsym": dict(R1=R1n, target=None),
            "quad_diag_tgt": dict(R1=R1d, target=tgt),
            "biquad_full_tgt": dict(R1=R1f, R2=R2f, target=tgt, stage="biquadratic"),
            "biquad_diag": dict(R1=R1d, R2=np.diag(np.diag(R2f)), target=None, stage="biquadratic"),
        }.items():
            c = make_ctrl(controllers, sys_obj, name, **kw)
            out[tag] = np.array([c.stage_obj(y[i], u[i]) for i in range(n)])
        save(f"F2_stage_{name}", dict(system=name), obs=y, act=u, R1_diag=R1d, R1_full=R1f, R1_nonsym=R1n,
             R2_full=R2f, target=tgt, **out)

    # ---------------------------------------------------------------- F3: critic value
    for name in PRESETS:
        p = PRESETS[name]
        sys_obj = make_sys(systems, name)
        n = 64
        y = rand_states(rng, name, n)
        u = rand_actions(rng, name, (n,))
        tgt = rng.uniform(-1, 1, p["ds"])
        out = {"target": tgt}
        for cs in ["quad-lin", "quadratic", "quad-nomix", "quad-mix"]:
            for ttag, t in (("", None), ("_tgt", tgt)):
                c = make_ctrl(controllers, sys_obj, name, critic_struct=cs, target=t)
                w = rng.uniform(-1, 1, (n, c.dim_critic))
                out[f"w_{cs}{ttag}"] = w
                out[f"Q_{cs}{ttag}"] = np.array([c._critic(y[i], u[i], w[i]) for i in range(n)])
        save(f"F3_critic_{name}", dict(system=name), obs=y, act=u, **out)

    # ---------------------------------------------------------------- F4: actor cost
    for name in PRESETS:
        p = PRESETS[name]
        sys_obj = make_sys(systems, name)
        out = {}
        meta = dict(system=name, cases=[])
        for N in (3, 5, 10, 15, 20):
            for mode in ("MPC", "RQL", "SQL"):
                for cs in (("quad-nomix",) if mode == "MPC" else ("quad-lin", "quadratic", "quad-nomix", "quad-mix")):
                    gamma = 1.0 if (N == 10 and mode == "MPC") else 0.95
                    n = 24
                    x = rand_states(rng, name, n)
                    obs = x + rng.uniform(-0.05, 0.05, x.shape)  # state_sys != obs (one-step lag)
                    aseq = rand_actions(rng, name, (n, N))
                    c = make_ctrl(controllers, sys_obj, name, mode=mode, Nactor=N, gamma=gamma, critic_struct=cs)
                    w = rng.uniform(0, 2, (n, c.dim_critic))
                    J = np.zeros(n)
                    for i in range(n):
                        c.state_sys = x[i]
                        c.w_critic = w[i]
                        J[i] = c._actor_cost(aseq[i].reshape(-1), obs[i])
                    tag = f"N{N}_{mode}_{cs}"
                    meta["cases"].append(dict(tag=tag, N=N, mode=mode, critic_struct=cs, gamma=gamma,
                                              pred_step_size=p["dt"] * p["mult"]))
                    out.update({f"{tag}__state_sys": x, f"{tag}__obs": obs, f"{tag}__action_sqn": aseq,
                                f"{tag}__w": w, f"{tag}__J": J})
        save(f"F4_actor_cost_{name}", meta, **out)

    # ---------------------------------------------------------------- F5: critic cost
    for name in PRESETS:
        sys_obj = make_sys(systems, name)
        out = {}
        meta = dict(system=name, cases=[])
        for cs in ["quad-lin", "quadratic", "quad-nomix", "quad-mix"]:
            for Ncritic, bs in ((4, 10), (6, 8), (30, 5)):  # last one clips Ncritic to buffer_size-1
                n = 16
                c = make_ctrl(controllers, sys_obj, name, mode="RQL", critic_struct=cs, gamma=0.9, Ncritic=Ncritic,
                              buffer_size=bs)
                yb = np.stack([rand_states(rng, name, bs) for _ in range(n)])
                ub = rand_actions(rng, name, (n, bs))
                w = rng.uniform(0, 2, (n, c.dim_critic))
                wp = rng.uniform(0, 2, (n, c.dim_critic))
                Jc = np.zeros(n)
                for i in range(n):
                    c.observation_buffer, c.action_buffer, c.w_critic_prev = yb[i], ub[i], wp[i]
                    Jc[i] = c._critic_cost(w[i])
                tag = f"{cs}_Nc{Ncritic}_bs{bs}"
                meta["cases"].append(dict(tag=tag, critic_struct=cs, Ncritic=Ncritic, buffer_size=bs, gamma=0.9,
                                          Ncritic_eff=int(c.Ncritic)))
                out.update({f"{tag}__obs_buf": yb, f"{tag}__act_buf": ub, f"{tag}__w": w, f"{tag}__w_prev": wp,
                            f"{tag}__Jc": Jc})
        save(f"F5_critic_cost_{name}", meta, **out)

    # ---------------------------------------------------------------- F6: constant-action RK45 trajectories
    const_u = {"3wrobot": [120.0, -35.0], "3wrobotNI": [8.0, -1.5], "2tank": [0.7]}
    t_end = {"3wrobot": 2.0, "3wrobotNI": 2.0, "2tank": 20.0}
    for name in PRESETS:
        p = PRESETS[name]
        sys_obj = make_sys(systems, name)
        x0 = np.asarray(p["x0"], dtype=float)
        sim = simulator.Simulator(sys_type="diff_eqn", closed_loop_rhs=sys_obj.closed_loop_rhs, sys_out=sys_obj.out,
                                  state_init=x0.copy(), disturb_init=[], action_init=np.zeros(p["du"]), t0=0,
                                  t1=t_end[name] + 1.0, dt=p["dt"], max_step=p["dt"] / 2, first_step=1e-6,
                                  atol=1e-5, rtol=1e-3, is_disturb=0, is_dyn_ctrl=0)
        sys_obj.receive_action(np.array(const_u[name]))
        ts, ys = [0.0], [x0.copy()]
        while sim.t < t_end[name]:
            sim.sim_step()
            t, state, obs, full = sim.get_sim_step_data()
            ts.append(float(t))
            ys.append(np.array(full, dtype=float))
        save(f"F6_rk45_const_{name}", dict(system=name, dt=p["dt"], action=const_u[name], pars=p["pars"],
                                           bnds=p["bnds"], atol=1e-5, rtol=1e-3),
             t=np.array(ts), y=np.stack(ys))

    # ---------------------------------------------------------------- F7: closed-loop traces of the reference loop
    # loop body = presets/main_3wrobot.py:419-446 (sim_step -> ctrl_selector -> receive_action ->
    # receive_sys_state -> upd_accum_obj)
    for name, mode, t1, Nactor in (("3wrobotNI", "MPC", 0.3, 5), ("3wrobot", "MPC", 0.2, 5),
                                   ("2tank", "MPC", 2.0, 10), ("2tank", "RQL", 2.0, 10)):
        p = PRESETS[name]
        sys_obj = make_sys(systems, name)
        x0 = np.asarray(p["x0"], dtype=float)
        ctrl = make_ctrl(controllers, sys_obj, name, mode=mode, Nactor=Nactor, state_sys=x0.copy())
        sim = simulator.Simulator(sys_type="diff_eqn", closed_loop_rhs=sys_obj.closed_loop_rhs, sys_out=sys_obj.out,
                                  state_init=x0.copy(), disturb_init=[], action_init=np.zeros(p["du"]), t0=0, t1=t1,
                                  dt=p["dt"], max_step=p["dt"] / 2, first_step=1e-6, atol=1e-5, rtol=1e-3,
                                  is_disturb=0, is_dyn_ctrl=0)
        rows = []
        while True:
            sim.sim_step()
            t, state, obs, full = sim.get_sim_step_data()
            action = controllers.ctrl_selector(t, obs, np.zeros(p["du"]), None, ctrl, mode)
            sys_obj.receive_action(action)
            ctrl.receive_sys_state(sys_obj._state)
            ctrl.upd_accum_obj(obs, action)
            rows.append(np.concatenate([[t], np.array(full, dtype=float), np.array(action, dtype=float),
                                        [ctrl.stage_obj(obs, action), ctrl.accum_obj_val]]))
            if t >= t1:
                break
        save(f"F7_trace_{name}_{mode}", dict(system=name, mode=mode, t1=t1, Nactor=Nactor, dt=p["dt"],
                                            columns="t,state...,action...,stage_obj,accum_obj"),
             rows=np.stack(rows))

    # ---------------------------------------------------------------- F8: optimiser quality reference (SLSQP)
    from scipy.optimize import Bounds, minimize
    for name, N in (("3wrobot", 10), ("3wrobotNI", 5), ("2tank", 10)):
        p = PRESETS[name]
        sys_obj = make_sys(systems, name)
        n = 12
        x = rand_states(rng, name, n)
        c = make_ctrl(controllers, sys_obj, name, mode="MPC", Nactor=N)
        Jopt, useq, nfev, Jinit = np.zeros(n), np.zeros((n, N * p["du"])), np.zeros(n, dtype=np.int64), np.zeros(n)
        for i in range(n):
            c.state_sys = x[i]
            init = np.reshape(c.action_sqn_init, [N * p["du"]])
            res = minimize(lambda a: c._actor_cost(a, x[i]), init, method="SLSQP", tol=1e-7,
                           bounds=Bounds(c.action_sqn_min, c.action_sqn_max, keep_feasible=True),
                           options={"maxiter": 300, "disp": False})  # controllers.py:1373-1398
            Jopt[i], useq[i], nfev[i] = res.fun, res.x, res.nfev
            Jinit[i] = c._actor_cost(init, x[i])
        save(f"F8_slsqp_actor_{name}", dict(system=name, N=N, gamma=1.0, pred_step_size=p["dt"] * p["mult"]),
             state=x, J_opt=Jopt, action_sqn_opt=useq, nfev=nfev, J_init=Jinit)

    for name in PRESETS:
        sys_obj = make_sys(systems, name)
        out, meta = {}, dict(system=name, cases=[])
        for cs in ["quadratic", "quad-nomix", "quad-lin", "quad-mix"]:
            n = 12
            c = make_ctrl(controllers, sys_obj, name, mode="RQL", critic_struct=cs, gamma=1.0, Ncritic=4, buffer_size=10)
            yb = np.stack([rand_states(rng, name, 10) for _ in range(n)])
            ub = rand_actions(rng, name, (n, 10))
            wp = rng.uniform(0.5, 1.5, (n, c.dim_critic))
            wfit, Jc, Jc_init = np.zeros((n, c.dim_critic)), np.zeros(n), np.zeros(n)
            for i in range(n):
                c.observation_buffer, c.action_buffer, c.w_critic_prev = yb[i], ub[i], wp[i]
                wfit[i] = c._critic_optimizer()  # controllers.py:1248-1271
                Jc[i] = c._critic_cost(wfit[i])
                Jc_init[i] = c._critic_cost(c.w_critic_init)
            meta["cases"].append(dict(tag=cs, critic_struct=cs, Ncritic=4, buffer_size=10, gamma=1.0))
            out.update({f"{cs}__obs_buf": yb, f"{cs}__act_buf": ub, f"{cs}__w_prev": wp, f"{cs}__w_fit": wfit,
                        f"{cs}__Jc_fit": Jc, f"{cs}__Jc_init": Jc_init})
        save(f"F8_slsqp_critic_{name}", meta, **out)

    # ---------------------------------------------------------------- KAT: known answers quoted in SURVEY.md §8c
    s3 = make_sys(systems, "3wrobot")
    x = np.array([5, 5, -3 * np.pi / 4, 0.3, -0.2])
    u = np.array([50.0, -20.0])
    kat = {"kat1": s3._state_dyn(0, x, u)}
    s3.receive_action(np.array([400.0, -150.0]))
    kat["kat2_rhs"] = s3.closed_loop_rhs(0, x)
    kat["kat2_action"] = s3.action.copy()
    c = make_ctrl(controllers, s3, "3wrobot", Nactor=5, gamma=0.9, pred_step_size=0.02)
    kat["kat3"] = np.array(c.stage_obj(x, u))
    aseq = np.array([[50, -20], [40, -10], [30, 0], [20, 10], [10, 20]], dtype=float)
    for mode in ("MPC", "RQL", "SQL"):
        c = make_ctrl(controllers, s3, "3wrobot", mode=mode, Nactor=5, gamma=0.9, pred_step_size=0.02,
                      critic_struct="quad-nomix", state_sys=x)
        c.w_critic = 0.5 * np.arange(1, 8)
        kat[f"kat4_{mode}"] = np.array(c._actor_cost(aseq.reshape(-1), x + 0.01))
    for cs in ("quad-lin", "quadratic", "quad-nomix", "quad-mix"):
        c = make_ctrl(controllers, s3, "3wrobot", critic_struct=cs)
        kat[f"kat5_{cs}"] = np.array(c._critic(x, u, np.linspace(0.1, 1, c.dim_critic)))
    s2 = make_sys(systems, "2tank")
    kat["kat8"] = s2._state_dyn(0, np.array([2.0, -2.0]), np.array([0.7]))
    c = make_ctrl(controllers, s2, "2tank", Nactor=4, pred_step_size=0.2, state_sys=np.array([2.0, -2.0]))
    kat["kat9"] = np.array(c.stage_obj(np.array([2.0, -2.0]), np.array([0.7])))
    kat["kat10"] = np.array(c._actor_cost(np.array([0.7, 0.1, 0.9, 0.4]), np.array([2.0, -2.0])))
    save("KAT", dict(note="known answers of SURVEY.md 8c, recomputed from the reference"), **kat)


if __name__ == "__main__":
    main()
