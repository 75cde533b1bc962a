#!/usr/bin/env python3
"""Where a B = 1 loop iteration of the drop-in classes spends its time (GPU box): the native call alone (rcg_loop_step with and
without a decision), and the whole reference loop with and without the fused step.   python tools/b1_breakdown.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from tests.helpers import both
from tests.test_hip_ref_traces import run_reference_loop, make_loop_objects
from rcognita_amd import _native as N

eng, cfg = both("3wrobot", 1, "f64", n_actor=5)
eng.set_state(np.array([[5, 5, -2.3, 0, 0.0]]))
act = np.array([[-30.0, -10.0]])
for decide, iters in ((False, 0), (True, 30), (True, 10)):
    for _ in range(50):
        eng.loop_step(act, 0.005, 1, decide=decide, iters=iters)
    t0 = time.perf_counter()
    n = 2000
    for _ in range(n):
        eng.loop_step(act, 0.005, 1, decide=decide, iters=iters)
    print(f"loop_step decide={decide} iters={iters}: {(time.perf_counter() - t0) / n * 1e6:.1f} us per call")
for fuse in (True, False):
    import rcognita_amd.simulator as S
    orig = S.Simulator.__init__
    def patched(self, *a, **k):
        orig(self, *a, **k)
        self.fuse = fuse
    S.Simulator.__init__ = patched
    run_reference_loop("3wrobot", "MPC", 5, 0.2)
    t0 = time.perf_counter()
    rows = run_reference_loop("3wrobot", "MPC", 5, 2.0)
    dt = time.perf_counter() - t0
    print(f"reference loop, fuse={fuse}: {len(rows) / dt:.0f} sim steps/s ({dt / len(rows) * 1e6:.1f} us per step)")
    S.Simulator.__init__ = orig
