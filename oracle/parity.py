"""Tick-by-tick parity checker: a float32 (or float64) device run against the float64 oracle.

TEST INFRASTRUCTURE ONLY (as everything under ``oracle/``): used by ``tests/`` and by ``bench.py``'s
``cpu_baseline`` leg to verify the outputs of the run it has just timed.

A float32 run cannot be compared with a float64 one by simply running both for T ticks: when two candidates'
costs differ by less than float32 rounding, the argmin may legitimately pick the runner-up, the applied action
differs and the two trajectories fork.  Stopping the comparison there would leave every later tick unchecked.
Instead every tick is verified as a MAP from the same inputs:

  1. the oracle advances one tick from the SAME pre-tick values as the device (its env is re-synchronised to the
     device's post-tick values after each check, so rounding does not accumulate across ticks either);
  2. where ``best_idx`` differs, the device's choice must be a near-tie: its oracle cost within ``tie_rel`` of the
     oracle's best; the oracle then re-runs the tick with the device's choice forced (``control_tick(force_idx)``);
  3. state, action, best_J, accum must agree within ``tol`` (relative to max(|value|, floor)), the integer counters
     exactly.
"""
from __future__ import annotations

import copy
from dataclasses import dataclass, field
from typing import Dict

import numpy as np

from . import rcg_oracle as O


def rel_err_norm(a, b, floor=1.0):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))) if a.size else 0.0


@dataclass
class TickReport:
    ticks: int = 0
    envs: int = 0
    ties: int = 0          # envs x ticks where the device took a near-tied runner-up
    idx_equal: int = 0     # envs x ticks with identical best_idx
    worst: Dict[str, float] = field(default_factory=lambda: {"state": 0.0, "action": 0.0, "best_J": 0.0, "accum": 0.0})


    def as_dict(self):
        return {"ticks": self.ticks, "envs": self.envs, "best_idx_equal": self.idx_equal, "near_ties": self.ties,
                "max_rel_err": dict(self.worst)}


def check_tick(cfg: O.OracleCfg, env: O.EnvBatch, cand, dev: Dict[str, np.ndarray], tol: float, tie_rel: float = None,
               report: TickReport = None, resync: bool = True, what: str = "", tol_over: Dict[str, float] = None):
    """Advance ``env`` (oracle) by one control tick and compare with the device's post-tick fields
    ``dev = {state, action, best_idx, best_J, accum, step_idx}`` (host arrays, reference shapes).  Raises
    AssertionError on a mismatch; returns the (possibly re-synchronised) oracle env."""
    tie_rel = 4.0 * tol if tie_rel is None else tie_rel
    before = copy.deepcopy(env)
    J = O.control_tick(cfg, env, cand)
    bi = np.asarray(dev["best_idx"])
    flipped = bi != env.best_idx
    if flipped.any():
        Jc = np.where(np.isnan(J), np.inf, J)
        rows = np.flatnonzero(flipped)
        j_dev, j_best = Jc[rows, bi[rows]], Jc[rows, env.best_idx[rows]]
        scale = np.maximum(np.max(np.abs(np.where(np.isfinite(Jc[rows]), Jc[rows], 0.0)), axis=1), 1e-30)
        bad = ~(np.abs(j_dev - j_best) <= tie_rel * scale)
        assert not bad.any(), (f"{what}: best_idx differs from the oracle's without a near-tie for envs "
                               f"{rows[bad][:8]}: J[dev choice] {j_dev[bad][:8]}, J[best] {j_best[bad][:8]}")
        env = before
        O.control_tick(cfg, env, cand, force_idx=np.where(flipped, bi, -1))
    Jf = np.where(np.isfinite(J), J, 0.0)
    jscale = np.maximum(np.max(np.abs(Jf), axis=1), 1e-30)
    errs = {
        "state": rel_err_norm(dev["state"], env.state),
        "action": rel_err_norm(dev["action"], env.action),
        # J may be a difference of large terms (signed critic weights): measured against the env's largest |J|, which
        # is what the argmin over the row is sensitive to
        "best_J": float(np.max(np.abs(np.asarray(dev["best_J"], dtype=np.float64) - env.best_J) / jscale)),
        "accum": rel_err_norm(dev["accum"], env.accum, floor=max(float(np.max(np.abs(env.accum))), 1e-30)),
    }
    if "w_critic" in dev:  # RQL / SQL: the tick pushed (action_curr, obs) and refitted the critic before the decision
        errs["w_critic"] = rel_err_norm(dev["w_critic"], env.w_critic)
        errs["obs_buf"] = rel_err_norm(dev["obs_buf"], env.obs_buf)
        errs["act_buf"] = rel_err_norm(dev["act_buf"], env.act_buf)
    for k, v in errs.items():
        # `tol_over`: per-quantity tolerances, e.g. the critic weights - they solve a least-squares problem regularised
        # at 1e-8 of its scale, so last-bit differences of the two float64 solvers are amplified accordingly - and the
        # costs computed from them
        lim = (tol_over or {}).get(k, tol)
        assert v <= lim, f"{what}: {k} differs from the oracle by {v:.3e} (tolerance {lim:.1e})"
    assert np.array_equal(np.asarray(dev["step_idx"]), env.step_idx), f"{what}: step_idx (int32) must be bit-exact"
    if report is not None:
        report.ticks += 1
        report.envs = len(bi)
        report.ties += int(flipped.sum())
        report.idx_equal += int((~flipped).sum())
        for k, v in errs.items():
            report.worst[k] = max(report.worst.get(k, 0.0), v)
    if resync:  # the next tick starts from the device's values: every tick is checked as a map from the same inputs
        env.state = np.asarray(dev["state"], dtype=np.float64).copy()
        env.action = np.asarray(dev["action"], dtype=np.float64).copy()
        env.accum = np.asarray(dev["accum"], dtype=np.float64).copy()
        if dev.get("state_prev") is not None:
            env.state_prev = np.asarray(dev["state_prev"], dtype=np.float64).copy()
        if "w_critic" in dev:
            env.w_critic = np.asarray(dev["w_critic"], dtype=np.float64).copy()
            env.w_prev = np.asarray(dev["w_prev"], dtype=np.float64).copy()
            env.obs_buf = np.asarray(dev["obs_buf"], dtype=np.float64).copy()
            env.act_buf = np.asarray(dev["act_buf"], dtype=np.float64).copy()
    return env


def device_fields(eng, N, with_prev=False, critic=False):
    """Post-tick fields of an ``rcognita_amd.Engine`` in the layout ``check_tick`` expects (``N`` = rcognita_amd._native)."""
    d = {"state": eng.get_state(), "action": eng.get_field(N.FIELD_ACTION), "best_idx": eng.get_field(N.FIELD_BEST_IDX),
         "best_J": eng.get_field(N.FIELD_BEST_J), "accum": eng.get_field(N.FIELD_ACCUM),
         "step_idx": eng.get_field(N.FIELD_STEP_IDX)}
    if with_prev:
        d["state_prev"] = eng.get_field(N.FIELD_STATE_PREV)
    if critic:
        d.update(w_critic=eng.get_field(N.FIELD_W_CRITIC), w_prev=eng.get_field(N.FIELD_W_PREV),
                 obs_buf=eng.get_field(N.FIELD_OBS_BUF), act_buf=eng.get_field(N.FIELD_ACT_BUF))
    return d
