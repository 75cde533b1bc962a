#!/usr/bin/env python3
"""Wall-clock rate of the drop-in preset scripts (rcognita_amd.presets.run = presets/main_*.py) driven exactly like the
reference's: one Python loop iteration per sim step, objects with the reference's names.  For DESIGN.md 5 ("the
reference at B = 1": 27.1 ctrl-steps/s at Nactor = 5, 3.4 at Nactor = 10 on one core)."""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from rcognita_amd.presets import run
import io, contextlib
for name, argv in (("3wrobot", ["--ctrl_mode", "MPC", "--t1", "3", "--Nactor", "5"]),
                   ("3wrobot", ["--ctrl_mode", "MPC", "--t1", "3", "--Nactor", "10"]),
                   ("3wrobotNI", ["--ctrl_mode", "MPC", "--t1", "3"]),
                   ("2tank", ["--ctrl_mode", "MPC", "--t1", "30"]),
                   ("2tank", ["--ctrl_mode", "RQL", "--t1", "30"]),
                   ("3wrobot", ["--ctrl_mode", "nominal", "--t1", "3"]),
                   ("3wrobot", ["--ctrl_mode", "MPC", "--t1", "1", "--Nactor", "10", "--batch", "4096"])):
    run(name, argv + ["--is_print_sim_step", "", "--t1", "0.2" if name != "2tank" else "2"])  # warm
    t0 = time.perf_counter()
    out = run(name, argv + ["--is_print_sim_step", ""])
    dt = time.perf_counter() - t0
    print(f"{name:10s} {' '.join(argv):60s} {out['ticks']:5d} sim steps in {dt:6.2f} s = {out['ticks']/dt:8.1f} steps/s  accum_obj[0] {float(out['accum_obj'].ravel()[0]):.2f}")
