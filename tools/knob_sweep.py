#!/usr/bin/env python3
"""A/B of the actor launcher's scheduling knobs (rcg_sysops.hpp::DevKnobs) on the bench workload: prints the mean
k_actor_dma launch time for the knobs set in the environment of THIS process (they are read once per process):

    for g in 8 16 32 64; do RCG_GPW=$g python tools/knob_sweep.py; done

bench.py refuses to run with RCG_* set; experiments go through this tool."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

a = bench.parse([])
B = int(os.environ.get("SWEEP_B", "65536"))
K = int(os.environ.get("SWEEP_K", "256"))
a.nactor = int(os.environ.get("SWEEP_N", "10"))
a.dtype = os.environ.get("SWEEP_DTYPE", "f32")
from rcognita_amd import _native as N  # noqa: E402

N.use_library(os.path.join(ROOT, "rcognita_amd", "lib", "librcg_dev.so"))  # the knobs exist only in the -DRCG_DEV twin
from rcognita_amd import Engine  # noqa: E402

ecfg, bnds = bench.c2_engine_config(a, 0, B)
eng = Engine(ecfg)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.set_state(bench.synth_state(1234, 0, B))
td = torch.float32 if a.dtype == "f32" else torch.float64
cand = (torch.rand((B, K, a.nactor, 2), device="cuda", dtype=td) * 600 - 300).contiguous()
for _ in range(300):
    eng.control_tick(cand, K=K)
eng.profile((N.KERNEL_ACTOR,), stride=8)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 600
for _ in range(n):
    eng.control_tick(cand, K=K)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ms, c = eng.profile_read(N.KERNEL_ACTOR)
knobs = {k: v for k, v in os.environ.items() if k.startswith("RCG_")}
esz = 4 if a.dtype == "f32" else 8
gb = B * (K * a.nactor * 2 * esz + 52) / (ms / c * 1e-3) / 1e12
print(f"{knobs} {a.dtype} B={B} K={K} N={a.nactor}: actor {ms / c * 1e3:.1f} us ({gb:.2f} TB/s), tick {dt / n * 1e3:.4f} ms")
