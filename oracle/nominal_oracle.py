"""TEST INFRASTRUCTURE - CPU restatement (numpy float64, batched) of the reference's nominal controllers,
SURVEY.md 8f row f3.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

  CtrlNominal3WRobotNI   rcognita/controllers.py:1757-1956   closed form (disassembled subgradient of a CLF)
  CtrlNominal3WRobot     rcognita/controllers.py:1495-1755   nonsmooth backstepping; needs theta* = argmin_theta Fc

Pinned on tests/golden/F10_nominal_*.npz (outputs of the reference, oracle/gen_nominal_fixtures.py).  The reference
finds theta* with scipy.optimize.minimize(method='trust-constr', tol=1e-6, maxiter=50) from theta = 0
(controllers.py:1625-1634): third-party, path dependent, local.  The build defines theta* instead by a compass search
from theta = 0 (:func:`theta_star`); F10 compares by the value of Fc reached and, where both land in the same basin, by
the action.

All functions take arrays with a leading batch axis.  0/0 at the exact origin yields NaN exactly as in the reference.
"""
import numpy as np

THETA_STEP0 = 0.25      # first step of the compass search from theta = 0
THETA_TOL = 1e-9        # the search stops once its step is this short
THETA_MAX_ITERS = 200   # (a period is at most 26 steps of 0.25 away, 28 halvings reach the tolerance)


def _scbrt(d):
    """sign(d) * |d|^(1/3) as the reference writes it (controllers.py:1604-1605)"""
    return np.abs(d) ** (1 / 3) * np.sign(d)


def cart2nh(x):
    """_Cart2NH (controllers.py:1636-1668, 1877-1893): Cartesian -> non-holonomic coordinates.
    x [B, 3] or [B, 5] -> xNI [B, 3] (and eta [B, 2] for the 5-state robot)."""
    xc, yc, al = x[:, 0], x[:, 1], x[:, 2]
    c, s = np.cos(al), np.sin(al)
    xNI = np.stack([al, xc * c + yc * s, -2 * (yc * c - xc * s) - al * (xc * c + yc * s)], axis=-1)
    if x.shape[1] == 3:
        return xNI
    v, om = x[:, 3], x[:, 4]
    eta = np.stack([om, (yc * c - xc * s) * om + v], axis=-1)
    return xNI, eta


def zeta_theta(xNI, theta):
    """_zeta(xNI, theta) (controllers.py:1551-1590): theta-dependent disassembled subgradient nablaF."""
    x1, x2, x3 = xNI[:, 0], xNI[:, 1], xNI[:, 2]
    ct, st = np.cos(theta), np.sin(theta)
    sq = np.sqrt(np.abs(x3))
    sig = x1 * ct + x2 * st + sq
    a3 = np.abs(x3) ** 3
    return np.stack([4 * x1 ** 3 - 2 * a3 * ct / sig ** 3,
                     4 * x2 ** 3 - 2 * a3 * st / sig ** 3,
                     (3 * x1 * ct + 3 * x2 * st + 2 * sq) * x3 ** 2 * np.sign(x3) / sig ** 3], axis=-1)


def zeta_ni(xNI):
    """CtrlNominal3WRobotNI._zeta (controllers.py:1780-1831): analytic nablaL, or nablaF(theta = 0) when
    xNI[0] == xNI[1] == 0."""
    x1, x2, x3 = xNI[:, 0], xNI[:, 1], xNI[:, 2]
    r = np.sqrt(x1 ** 2 + x2 ** 2)
    sigma = r + np.sqrt(np.abs(x3))
    a3 = np.abs(x3) ** 3
    nL = np.stack([4 * x1 ** 3 + a3 / sigma ** 3 * 1 / r ** 3 * 2 * x1,
                   4 * x2 ** 3 + a3 / sigma ** 3 * 1 / r ** 3 * 2 * x2,
                   3 * np.abs(x3) ** 2 * np.sign(x3) + a3 / sigma ** 3 * 1 / np.sqrt(np.abs(x3)) * np.sign(x3)], axis=-1)
    nF = zeta_theta(xNI, np.zeros_like(x1))
    return np.where(((x1 == 0) & (x2 == 0))[:, None], nF, nL)


def kappa_of(xNI, zeta):
    """_kappa (controllers.py:1592-1608, 1833-1849): -cbrt(zeta . g_k), g_1 = (1, 0, x2), g_2 = (0, 1, -x1).
    The products with the zero entries are kept (np.dot in the reference: inf * 0 = NaN)."""
    x1, x2 = xNI[:, 0], xNI[:, 1]
    d0 = zeta[:, 0] * 1 + zeta[:, 1] * 0 + zeta[:, 2] * x2
    d1 = zeta[:, 0] * 0 + zeta[:, 1] * 1 + zeta[:, 2] * (-x1)
    return np.stack([-_scbrt(d0), -_scbrt(d1)], axis=-1)


def Fc(xNI, eta, theta):
    """_Fc (controllers.py:1610-1623): marginal function of the backstepping CLF."""
    x1, x2, x3 = xNI[:, 0], xNI[:, 1], xNI[:, 2]
    sig = x1 * np.cos(theta) + x2 * np.sin(theta) + np.sqrt(np.abs(x3))
    F = x1 ** 4 + x2 ** 4 + np.abs(x3) ** 3 / sig ** 2
    z = eta - kappa_of(xNI, zeta_theta(xNI, theta))
    return F + 1 / 2 * (z[:, 0] * z[:, 0] + z[:, 1] * z[:, 1])


def _wrap(theta):
    return np.where(theta > np.pi, theta - 2 * np.pi, np.where(theta < -np.pi, theta + 2 * np.pi, theta))


def theta_star(xNI, eta):
    """Build-defined replacement of _minimizer_theta (controllers.py:1618-1627), which runs SciPy trust-constr from
    theta = 0: a LOCAL, path-dependent search.  Round 6: a compass search from theta = 0 (what trust-constr starts from):
      step s = THETA_STEP0 = 0.25; at most THETA_MAX_ITERS times: evaluate Fc at theta - s and theta + s (non-finite = +inf);
      move to the lower one if it is lower than Fc(theta) (the left one on a tie), else halve s; stop when s <= THETA_TOL;
      wrap into [-pi, pi).
    Only comparisons of Fc values decide, so the HIP kernel k_nominal, which mirrors this statement by statement, takes the same
    path.  On the F10 states this is the reference's minimiser on 94.8 % of them, its action (2 %) on 96.9 %, and Fc(theta*) is
    never above the reference's (rounds 2-5: downhill walk on a 64-point grid + golden section: 92.7 % / 94.8 % / 96.9 %;
    round 1: the GLOBAL minimum of the scan, 72 %).  The alternatives measured: oracle/experiments/theta_search_study.py."""
    B = xNI.shape[0]
    with np.errstate(all="ignore"):
        fin = lambda th: (lambda f: np.where(np.isfinite(f), f, np.inf))(Fc(xNI, eta, th))
        th = np.zeros(B)
        f = fin(th)
        s = np.full(B, THETA_STEP0)
        for _ in range(THETA_MAX_ITERS):
            live = s > THETA_TOL
            if not live.any():
                break
            fl, fr = fin(th - s), fin(th + s)
            go_l = live & (fl < f) & (fl <= fr)
            go_r = live & (~go_l) & (fr < f)
            th = np.where(go_l, th - s, np.where(go_r, th + s, th))
            f = np.where(go_l, fl, np.where(go_r, fr, f))
            s = np.where(live & ~(go_l | go_r), 0.5 * s, s)
        return th - 2 * np.pi * np.floor((th + np.pi) / (2 * np.pi))


def clip_bnds(u, bnds):
    bnds = np.asarray(bnds, dtype=np.float64)
    if not bnds.any():
        return u
    return np.minimum(np.maximum(u, bnds[:, 0]), bnds[:, 1])  # NaN stays NaN, as np.clip


def nominal_action_ni(x, gain, bnds=None):
    """CtrlNominal3WRobotNI.compute_action_vanila (+ the clip of compute_action when ``bnds`` is given)
    (controllers.py:1906-1947)."""
    with np.errstate(all="ignore"):
        xNI = cart2nh(x)
        kap = kappa_of(xNI, zeta_ni(xNI))
        uNI = gain * kap
        u = np.stack([uNI[:, 1] + 1 / 2 * uNI[:, 0] * (xNI[:, 2] + xNI[:, 0] * xNI[:, 1]), uNI[:, 0]], axis=-1)
    return u if bnds is None else clip_bnds(u, bnds)


def nh2ctrl_cart(xNI, eta, uNI, m, I):
    """_NH2ctrl_Cart (controllers.py:1670-1691)"""
    return np.stack([m * (uNI[:, 1] + xNI[:, 1] * eta[:, 0] ** 2
                          + 1 / 2 * (xNI[:, 0] * xNI[:, 1] * uNI[:, 0] + uNI[:, 0] * xNI[:, 2])),
                     I * uNI[:, 0]], axis=-1)


def nominal_action_endi(x, gain, m, I, bnds=None, theta=None):
    """CtrlNominal3WRobot.compute_action_vanila (+ clip) (controllers.py:1693-1749) with ``theta`` (default: the
    build's :func:`theta_star`)."""
    with np.errstate(all="ignore"):
        xNI, eta = cart2nh(x)
        th = theta_star(xNI, eta) if theta is None else np.asarray(theta, dtype=np.float64)
        z = eta - kappa_of(xNI, zeta_theta(xNI, th))
        u = nh2ctrl_cart(xNI, eta, -gain * z, m, I)
    return u if bnds is None else clip_bnds(u, bnds)


def lyapunov_ni(x):
    """CtrlNominal3WRobotNI.compute_LF (controllers.py:1949-1955)"""
    xNI = cart2nh(x)
    sigma = np.sqrt(xNI[:, 0] ** 2 + xNI[:, 1] ** 2) + np.sqrt(np.abs(xNI[:, 2]))
    with np.errstate(all="ignore"):
        return xNI[:, 0] ** 4 + xNI[:, 1] ** 4 + np.abs(xNI[:, 2]) ** 3 / sigma ** 2


def lyapunov_endi(x, theta=None):
    """CtrlNominal3WRobot.compute_LF (controllers.py:1750-1755): Fc at theta*"""
    xNI, eta = cart2nh(x)
    with np.errstate(all="ignore"):
        th = theta_star(xNI, eta) if theta is None else np.asarray(theta, dtype=np.float64)
        return Fc(xNI, eta, th)


def control_tick_nominal(cfg, env, gain, m=None, I=None):
    """Twin of rcg_control_tick_nominal: sim_step -> clipped nominal action of the new state -> accum, step_idx
    (the reference loop with ``ctrl_mode='nominal'``, presets/main_3wrobot.py:419-429)."""
    from . import rcg_oracle as O

    O.sim_substeps(cfg, env, cfg.substeps_per_tick)
    env.tick_count += 1
    x = env.state
    bn = cfg.ctrl_bnds  # clipped iff ctrl_bnds.any(), as compute_action (controllers.py:1712)
    if x.shape[1] == 3:
        env.action = nominal_action_ni(x, gain, bn)
    else:
        env.action = nominal_action_endi(x, gain, m, I, bn)
    if not cfg.accum_every_substep:
        env.accum = env.accum + O.stage_obj(x, env.action, cfg) * cfg.sampling_time
    env.step_idx = env.step_idx + np.int32(1)
