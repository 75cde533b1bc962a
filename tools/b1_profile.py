#!/usr/bin/env python3
"""The reference's headless preset loop through the drop-in classes at B = 1 (the 3-wheel robot, MPC, Nactor = 5, simulation steps of
dt / 2): simulation steps per second and where the host time goes (cProfile, by own time).  GPU box only.   python tools/b1_profile.py"""
import cProfile, pstats, sys, time
sys.path.insert(0,'.')
import numpy as np
from tests.test_hip_ref_traces import run_reference_loop
run_reference_loop("3wrobot","MPC",5,0.1)
t0=time.perf_counter(); rows=run_reference_loop("3wrobot","MPC",5,2.0); dt=time.perf_counter()-t0
print("sim steps/s", len(rows)/dt, "steps", len(rows))
pr=cProfile.Profile(); pr.enable(); run_reference_loop("3wrobot","MPC",5,1.0); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
