#!/usr/bin/env python3
"""Launch geometry of k_actor_dma_packed (dev build: RCG_GPW, RCG_PER_CU): one child process per setting.
    python tools/packed_sweep.py            (GPU box; needs rcognita_amd/lib/librcg_dev.so)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from rcognita_amd import _native as N
N.use_library(%r)
from rcognita_amd import Engine
from rcognita_amd.pool import preset_engine_config
B, Nh = 65536, 10
K = int(sys.argv[1])
eng = Engine(preset_engine_config("3wrobot", B, Nactor=Nh))
eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.set_state(np.random.default_rng(0).uniform(-2, 2, (B, 5)))
cand = (torch.rand((B, K, Nh, 2), device="cuda") * 200 - 100).contiguous()
for _ in range(300): eng.control_tick(cand, K=K)
eng.profile((N.KERNEL_ACTOR,), stride=3)
for _ in range(300): eng.control_tick(cand, K=K)
s = eng.profile_samples(N.KERNEL_ACTOR); ll = eng.last_launch()
byt = B * (K * Nh * 2 * 4 + 52)
print("RES", ll["kernel"], ll["envs_per_wave"], round(float(np.median(s)) * 1e3, 2), round(float(s.min()) * 1e3, 2), round(byt / (np.median(s) * 1e-3) / 8e12, 3))
''' % (ROOT, os.path.join(ROOT, "rcognita_amd", "lib", "librcg_dev.so"))
for K in (16, 32, 8):
    for knobs in ({}, {"RCG_GPW": str(64 // K)}, {"RCG_GPW": str(2 * (64 // K))}, {"RCG_GPW": "32"}, {"RCG_GPW": "64"},
                  {"RCG_PER_CU": "8"}, {"RCG_PER_CU": "2"}, {"RCG_GPW": "32", "RCG_PER_CU": "8"}):
        env = {k: v for k, v in os.environ.items() if not k.startswith("RCG_")}
        env.update(knobs)
        out = subprocess.run([sys.executable, "-c", CHILD, str(K)], capture_output=True, text=True, env=env, timeout=600)
        res = [l for l in out.stdout.splitlines() if l.startswith("RES")]
        print("K", K, knobs, res[-1] if res else out.stderr[-300:], flush=True)
