#!/usr/bin/env python3
"""How long does the GPU take to reach its steady clock under this workload, and how fast does it fall back after an
idle gap?  Prints ms per tick over time for a continuous stream of C2 ticks, then for short bursts after idle gaps.
(Measurement tool behind bench.py's untimed pre-spin; run on the GPU box.)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

args = bench.parse([])
from rcognita_amd import Engine  # noqa: E402

B, K, Nh = 65536, 256, 10
ecfg, bnds = bench.c2_engine_config(args, 0, B)
eng = Engine(ecfg)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.set_state(bench.synth_state(1234, 0, B))
cand = (torch.rand((B, K, Nh, 2), device="cuda") * 600 - 300).contiguous()
torch.cuda.synchronize()
time.sleep(1.0)

evs = []
n_chunks, chunk = 80, 50
t0 = time.perf_counter()
for i in range(n_chunks):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    evs.append(e)
    for _ in range(chunk):
        eng.control_tick(cand, K=K)
e = torch.cuda.Event(enable_timing=True)
e.record()
evs.append(e)
torch.cuda.synchronize()
print("continuous stream: ms/tick per chunk of", chunk)
print(" ".join(f"{evs[i].elapsed_time(evs[i + 1]) / chunk:.4f}" for i in range(n_chunks)))

for gap in (0.0, 0.001, 0.01, 0.05, 0.2, 1.0):
    res = []
    for rep in range(3):
        for _ in range(1500):
            eng.control_tick(cand, K=K)
        torch.cuda.synchronize()
        if gap:
            time.sleep(gap)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t1 = time.perf_counter()
        a.record()
        for _ in range(20):
            eng.control_tick(cand, K=K)
        b.record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t1
        res.append((a.elapsed_time(b) / 20, wall / 20 * 1e3))
    print(f"after 0.3 s of work, sync, idle {gap * 1e3:.0f} ms -> 20 ticks: " +
          ", ".join(f"{x:.4f} (wall {w:.4f})" for x, w in res))
# the same with a warm-up burst of W ticks between the idle gap and the timed 20
for W in (5, 50, 200, 500):
    res = []
    for rep in range(3):
        torch.cuda.synchronize()
        time.sleep(0.5)
        for _ in range(W):
            eng.control_tick(cand, K=K)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(20):
            eng.control_tick(cand, K=K)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t1) / 20 * 1e3)
    print(f"idle 0.5 s, {W} warm-up ticks, sync, 20 timed ticks: wall ms/tick " + ", ".join(f"{x:.4f}" for x in res))
