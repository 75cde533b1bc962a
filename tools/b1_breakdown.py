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
for decide in (False, True):  # the two halves: what the host pays when the device runs the step behind its own bookkeeping
    tb = te = 0.0
    n = 2000
    for i in range(n + 50):
        t0 = time.perf_counter()
        eng.loop_step_begin(act, 0.005, 1, decide=decide, iters=30)
        t1 = time.perf_counter()
        while time.perf_counter() - t1 < (40e-6 if decide else 15e-6):  # (the loop body's bookkeeping)
            pass
        t2 = time.perf_counter()
        eng.loop_step_end()
        t3 = time.perf_counter()
        if i >= 50:
            tb += t1 - t0
            te += t3 - t2
    print(f"loop_step_begin decide={decide}: {tb / n * 1e6:.1f} us, loop_step_end after the step has finished: {te / n * 1e6:.1f} us")
for fuse, spec in ((True, True), (True, False), (False, False)):
    import rcognita_amd.simulator as S
    import rcognita_amd.controllers as Cc
    orig_c = Cc.CtrlOptPred.__init__
    def patched_c(self, *a, **k):
        orig_c(self, *a, **k)
        self.speculate = spec
    Cc.CtrlOptPred.__init__ = patched_c
    orig = S.Simulator.__init__
    def patched(self, *a, **k):
        orig(self, *a, **k)
        self.fuse = fuse
    S.Simulator.__init__ = patched
    run_reference_loop("3wrobot", "MPC", 5, 0.2)
    t0 = time.perf_counter()
    rows = run_reference_loop("3wrobot", "MPC", 5, 2.0)
    dt = time.perf_counter() - t0
    print(f"reference loop, fuse={fuse}, next step started ahead={spec}: {len(rows) / dt:.0f} sim steps/s ({dt / len(rows) * 1e6:.1f} us per step)")
    S.Simulator.__init__ = orig
    Cc.CtrlOptPred.__init__ = orig_c
