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

CASES = [("3wrobotNI", "quad-nomix"), ("3wrobotNI", "quad-mix"), ("3wrobot", "quad-nomix"), ("2tank", "quad-nomix"),
         ("2tank", "quadratic"), ("2tank", "quad-lin")]
TOLS = [(1e-7, 1e-10), (1e-10, 1e-7), (1e-10, 1e-10), (1e-5, 1e-5)]


def window(rows, dt):
    i0 = int(np.argmin(np.abs(rows[:, 0] - 2 * dt)))
    return float(rows[-1, -1] - rows[i0, -1])


def sensitivity(name, mode, cs):
    from tests.conftest import load_golden
    from tests.helpers import oracle_cfg

    meta, z = load_golden(f"F7c_trace_{name}_{mode}_{cs}")
    cfg = oracle_cfg(name, n_actor=meta["Nactor"], mode=O.MODE_IDS[mode], gamma=meta["gamma"],
                     critic_struct=O.CRITIC_IDS[cs], n_critic=meta["Ncritic"], buffer_size=meta["buffer_size"])
    ref = window(z["rows"], meta["dt"])
    ch = []
    for at, ct in TOLS:
        rows = RefLoop(cfg, np.array(meta["x0"], dtype=float), meta["t1"], action_init=[0.5] if name == "2tank" else None,
                       actor_tol=at, critic_tol=ct).run()
        ch.append(abs(window(rows, meta["dt"]) - ref) / abs(ref))
    return dict(accum_window=ref, rel_change=ch, sensitivity=max(ch))


def main():
    out = {"tols": TOLS, "traces": {}}
    for name, cs in CASES:
        for mode in ("RQL", "SQL"):
            out["traces"][f"{name}_{mode}_{cs}"] = s = sensitivity(name, mode, cs)
            print(f"{name} {mode} {cs}: window {s['accum_window']:.4f}, sensitivity {s['sensitivity']:.2%}")
    with open(os.path.join(ROOT, "tests", "golden", "F7c_sensitivity.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
