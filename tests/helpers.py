"""Shared test helpers: preset constants (the reference's own values) and oracle config builders."""
import numpy as np

from oracle import rcg_oracle as O

# presets/main_3wrobot.py:45-47,207-215; main_3wrobot_NI.py:45-48,207-211; main_2tank.py:45-48,199-211
PRESETS = {
    "3wrobot": dict(sys_id=O.SYS_3WROBOT, pars=[10.0, 1.0], bnds=[[-300, 300], [-100, 100]],
                    R1=[1, 10, 1, 0, 0, 0, 0], dt=0.01, mult=2.0, x0=[5, 5, -3 * np.pi / 4, 0, 0], target=None),
    "3wrobotNI": dict(sys_id=O.SYS_3WROBOT_NI, pars=[], bnds=[[-25, 25], [-5, 5]],
                      R1=[1, 10, 1, 0, 0], dt=0.01, mult=1.0, x0=[5, 5, -3 * np.pi / 4], target=None),
    "2tank": dict(sys_id=O.SYS_2TANK, pars=[18.4, 24.4, 1.3, 1.0, 0.2], bnds=[[0, 1]],
                  R1=[10, 10, 1], dt=0.1, mult=2.0, x0=[2, -2], target=[0.5, 0.5]),
}
SYSTEMS = list(PRESETS)


def oracle_cfg(name, **kw):
    p = PRESETS[name]
    base = dict(
        sys_id=p["sys_id"], pars=p["pars"], ctrl_bnds=np.array(p["bnds"], dtype=float),
        R1=np.diag(np.array(p["R1"], dtype=float)), target=p["target"], dt_sim=p["dt"], sampling_time=p["dt"],
        pred_step_size=p["dt"] * p["mult"],
    )
    base.update(kw)
    return O.OracleCfg(**base)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.max(np.abs(a - b) / (np.abs(b) + 1e-300)) if a.size else 0.0


def rel_err_norm(a, b, floor=1.0):
    """max |a-b| / max(|b|, floor): relative where values are large, absolute near zero."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))) if a.size else 0.0
