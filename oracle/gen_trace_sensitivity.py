#!/usr/bin/env python3
"""How far does the REFERENCE's own closed loop move when nothing but SLSQP's stopping tolerance changes?

    python oracle/gen_trace_sensitivity.py     # writes tests/golden/F7c_sensitivity.json

The critic modes feed an under-determined least squares (Ncritic - 1 = 3 rows, 3 .. 9 weights) and a non-convex actor
problem to SLSQP with ``tol = 1e-7`` hard-coded (rcognita/controllers.py:1264, 1396); where SLSQP stops inside the set of
near-minimisers decides the next action, the next buffer row, the next fit.  A different optimiser (the build's bounded
least squares + on-device quasi-Newton actor) can therefore be held to the reference's closed-loop trace only up to the
distance the reference itself travels under a change of that tolerance.  This script measures that distance for every F7c
trace with oracle/ref_loop.py - the restatement of the reference's loop that reproduces all twelve traces (and the seven
MPC ones) to 1e-6 at the reference's own tolerance, tests/test_critic_traces.py, tests/test_ref_loop.py - run at
(actor_tol, critic_tol) in {(1e-7, 1e-10), (1e-10, 1e-7), (1e-10, 1e-10), (1e-5, 1e-5)}.

Per trace: ``accum_window`` of the reference (accum_obj over [2 dt, t1], as tests/test_hip_ref_traces.py measures it) and
``rel_change`` = the four relative changes of that window; ``sensitivity`` = their maximum.  tests/test_hip_ref_traces.py holds
the HIP mirror classes to max(6 %, 2 x sensitivity) on each trace; tests/test_critic_traces.py recomputes three entries.
``exact_critic`` (round 5): the same loop with ONE ingredient exchanged - the critic's SLSQP call replaced by the exact
minimiser of the build-defined fit (RefLoop(critic="exact")).  A tolerance change does not show what SLSQP's habit of leaving
a fit at its start point does to a trace (it does so on 4 .. 74 % of the robots' ticks, whatever the tolerance); this
exchange does: 0.1 .. 16 % by trace.  The HIP loop is held to the band around THIS loop on every trace, and around the
reference's trace wherever the exchange itself stays inside the band.
oracle/gen_critic_fixtures.py calls ``sensitivity_of`` while it chooses its traces (it refuses one whose distance to the
reference's MPC run is below twice that band).
TEST INFRASTRUCTURE: needs no reference import (numpy + scipy only), so it also runs on the GPU box.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import rcg_oracle as O  # noqa: E402
from oracle.ref_loop import RefLoop  # noqa: E402

TOLS = [(1e-7, 1e-10), (1e-10, 1e-7), (1e-10, 1e-10), (1e-5, 1e-5)]
ACCUM_BAND = 0.06  # the band of the MPC traces (tests/test_hip_ref_traces.py)


def window(rows, dt, t_end=None):
    i0 = int(np.argmin(np.abs(rows[:, 0] - 2 * dt)))
    i1 = len(rows) - 1 if t_end is None else int(np.argmin(np.abs(rows[:, 0] - t_end)))
    return float(rows[i1, -1] - rows[i0, -1])


def band_of(sens):
    """The band a different optimiser's closed loop is held to on a trace whose own sensitivity is ``sens``."""
    return max(ACCUM_BAND, 2.0 * sens)


def sensitivity_of(cfg, x0, t1, dt, rows, action_init, exact_critic=False):
    ref = window(rows, dt)
    ch = []
    for at, ct in TOLS:
        r = RefLoop(cfg, np.array(x0, dtype=float), t1, action_init=action_init, actor_tol=at, critic_tol=ct).run()
        ch.append(abs(window(r, dt) - ref) / abs(ref))
    out = dict(accum_window=ref, rel_change=ch, sensitivity=max(ch))
    if exact_critic:
        # the reference's loop (its time grid, its SLSQP actor at its tolerance) with only the critic's SLSQP call exchanged
        # for the exact minimiser of the build-defined fit: the running cost at 2/3 t1 and t1, and how far that one
        # exchange moves the reference's own trace
        r = RefLoop(cfg, np.array(x0, dtype=float), t1, action_init=action_init, critic="exact").run()
        out["exact_critic"] = dict(window_23=window(r, dt, 2 / 3 * t1), window_1=window(r, dt), ref_window_23=window(rows, dt, 2 / 3 * t1),
                                   shift=abs(window(r, dt) - ref) / abs(ref))
    return out


def trace_keys():
    from tests.conftest import GOLDEN

    return sorted(f[len("F7c_trace_"):-len(".npz")] for f in os.listdir(GOLDEN) if f.startswith("F7c_trace_"))


def sensitivity(name, mode, cs):
    from tests.conftest import load_golden
    from tests.helpers import oracle_cfg

    meta, z = load_golden(f"F7c_trace_{name}_{mode}_{cs}")
    cfg = oracle_cfg(name, n_actor=meta["Nactor"], mode=O.MODE_IDS[mode], gamma=meta["gamma"],
                     critic_struct=O.CRITIC_IDS[cs], n_critic=meta["Ncritic"], buffer_size=meta["buffer_size"])
    return sensitivity_of(cfg, meta["x0"], meta["t1"], meta["dt"], z["rows"], [0.5] if name == "2tank" else None,
                          exact_critic=True)


def main():
    out = {"tols": TOLS, "traces": {}}
    for key in trace_keys():
        name, mode, cs = key.split("_")
        out["traces"][key] = s = sensitivity(name, mode, cs)
        print(f"{name} {mode} {cs}: window {s['accum_window']:.4f}, sensitivity {s['sensitivity']:.2%}; with the critic's "
              f"SLSQP exchanged for the exact fit the reference's loop ends on {s['exact_critic']['window_1']:.4f} "
              f"({s['exact_critic']['shift']:.2%} away)")
    with open(os.path.join(ROOT, "tests", "golden", "F7c_sensitivity.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
