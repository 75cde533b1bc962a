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


# ---- HIP engine builders (import of rcognita_amd is deferred so CPU-only collection works) ----
def engine_cfg(name, batch, dtype="f64", **kw):
    """EngineConfig that matches ``oracle_cfg(name, **same kw)``."""
    from rcognita_amd import EngineConfig
    from rcognita_amd import _native as N

    p = PRESETS[name]
    inv_mode = {v: k for k, v in O.MODE_IDS.items()}
    inv_stage = {v: k for k, v in O.STAGE_IDS.items()}
    inv_critic = {v: k for k, v in O.CRITIC_IDS.items()}
    base = dict(
        sys_id=N.SYS_IDS[name], batch=batch, dtype=dtype, pars=p["pars"], ctrl_bnds=np.array(p["bnds"], dtype=float),
        R1=np.diag(np.array(p["R1"], dtype=float)), observation_target=p["target"], dt_sim=p["dt"],
        sampling_time=p["dt"], pred_step_size=p["dt"] * p["mult"],
    )
    ren = dict(n_actor="Nactor", n_critic="Ncritic", target="observation_target")
    for k, v in kw.items():
        k2 = ren.get(k, k)
        if k == "mode":
            v = inv_mode[v]
        elif k == "stage_obj_struct":
            v = inv_stage[v]
        elif k == "critic_struct":
            v = inv_critic[v]
        base[k2] = v
    return EngineConfig(**base)


def both(name, batch, dtype="f64", **kw):
    """(Engine, OracleCfg) built from the same keyword set (oracle naming); ``engine_only`` = a dict of EngineConfig
    fields the OracleCfg has no counterpart for (e.g. the disturbance model, which has its own oracle config)."""
    from rcognita_amd import Engine

    eonly = kw.pop("engine_only", {})
    okw = {k: v for k, v in kw.items() if k not in ("per_env_pars", "action_init")}
    ec = engine_cfg(name, batch, dtype, **kw)
    for k, v in eonly.items():
        setattr(ec, k, v)
    return Engine(ec), oracle_cfg(name, **okw)


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


TOL = {"f64": 1e-11, "f32": 1e-5}  # f32: the north star's stated tolerance (BASELINE.json)


def assert_kernel(eng, kernel, variant=None, kind=None):
    """The library's own record of which kernel served the handle's last launch of a kind (rcg_last_launch): tests that
    claim a kernel ask the library instead of re-deriving its dispatch rule."""
    from rcognita_amd import _native as N

    ll = eng.last_launch(N.KERNEL_ACTOR if kind is None else kind)
    assert ll["kernel"] == kernel, f"expected {kernel}, the library launched {ll}"
    if variant is not None:
        assert ll["variant"] == variant, f"expected variant {variant}, the library launched {ll}"
    return ll
