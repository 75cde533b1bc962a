#!/usr/bin/env python3
"""Generate tests/golden/F10_nominal_{3wrobot,3wrobotNI}.npz by calling the reference's nominal controllers
(rcognita/controllers.py:1495-1956) on seeded random states.  Build container only; data only (inputs + the
reference's outputs), recipe of gen_fixtures.py / SURVEY.md Appendix B.

    python oracle/gen_nominal_fixtures.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_fixtures import PRESETS, import_reference, rand_states, save  # noqa: E402


def main():
    systems, simulator, controllers = import_reference()
    rng = np.random.default_rng(20261010)

    # ---- CtrlNominal3WRobotNI: closed form (controllers.py:1757-1956); preset gain 0.5 (main_3wrobot_NI.py:235)
    p = PRESETS["3wrobotNI"]
    bnds = np.array(p["bnds"], dtype=float)
    c = controllers.CtrlNominal3WRobotNI(ctrl_gain=0.5, ctrl_bnds=bnds, t0=0, sampling_time=p["dt"])
    x = rand_states(rng, "3wrobotNI", 256)
    x[:8, 2] = 0.0          # alpha = 0 ...
    x[:8, 0] = 0.0          # ... and xc = 0  =>  xNI[0] = xNI[1] = 0: the nablaF branch of _zeta (controllers.py:1828)
    x[8:12] = 0.0           # origin: 0/0 in both branches
    x[12:16, 1] = 0.0       # yc = 0
    xNI = np.stack([c._Cart2NH(s) for s in x])
    with np.errstate(all="ignore"):
        zeta = np.stack([c._zeta(v) for v in xNI])
        kappa = np.stack([c._kappa(v) for v in xNI])
        act_van = np.stack([c.compute_action_vanila(s).copy() for s in x])
        act_clip = []
        for s in x:
            c.reset(0)
            act_clip.append(np.array(c.compute_action(10.0, s), dtype=float).copy())
        LF = np.array([c.compute_LF(s) for s in x])
    save("F10_nominal_3wrobotNI", dict(system="3wrobotNI", ctrl_gain=0.5, bnds=p["bnds"]),
         state=x, xNI=xNI, zeta=zeta, kappa=kappa, action_vanila=act_van, action=np.stack(act_clip), LF=LF)

    # ---- CtrlNominal3WRobot: backstepping with theta* = argmin Fc by SciPy trust-constr (controllers.py:1495-1755);
    # preset gain 5 (main_3wrobot.py:239)
    p = PRESETS["3wrobot"]
    bnds = np.array(p["bnds"], dtype=float)
    m, I = p["pars"]
    c = controllers.CtrlNominal3WRobot(m, I, ctrl_gain=5, ctrl_bnds=bnds, t0=0, sampling_time=p["dt"])
    n, nth = 96, 16
    x = rand_states(rng, "3wrobot", n)
    x[:4, 2] = 0.0
    xNI, eta = np.zeros((n, 3)), np.zeros((n, 2))
    theta = rng.uniform(-np.pi, np.pi, (n, nth))
    Fc, zeta_t, kappa_t = np.zeros((n, nth)), np.zeros((n, nth, 3)), np.zeros((n, nth, 2))
    th_star, Fc_star, act_van, act_clip, LF = np.zeros(n), np.zeros(n), np.zeros((n, 2)), np.zeros((n, 2)), np.zeros(n)
    uCart_probe = np.zeros((n, 2))
    with np.errstate(all="ignore"):
        for i in range(n):
            a, b = c._Cart2NH(x[i])
            xNI[i], eta[i] = a, b
            for j in range(nth):
                Fc[i, j] = c._Fc(a, b, theta[i, j])
                zeta_t[i, j] = c._zeta(a, theta[i, j])
                kappa_t[i, j] = c._kappa(a, theta[i, j])
            th_star[i] = float(np.ravel(c._minimizer_theta(a, b))[0])
            Fc_star[i] = c._Fc(a, b, th_star[i])
            act_van[i] = c.compute_action_vanila(x[i])
            c.reset(0)
            act_clip[i] = c.compute_action(10.0, x[i])
            LF[i] = c.compute_LF(x[i])
            uCart_probe[i] = c._NH2ctrl_Cart(a, b, np.array([0.3, -0.7]))
    save("F10_nominal_3wrobot", dict(system="3wrobot", ctrl_gain=5, bnds=p["bnds"], m=m, I=I, uNI_probe=[0.3, -0.7]),
         state=x, xNI=xNI, eta=eta, theta=theta, Fc=Fc, zeta=zeta_t, kappa=kappa_t, theta_star=th_star,
         Fc_star=Fc_star, action_vanila=act_van, action=act_clip, LF=LF, uCart_probe=uCart_probe)


if __name__ == "__main__":
    main()
