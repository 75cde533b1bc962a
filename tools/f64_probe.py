#!/usr/bin/env python3
"""float64 (the reference's arithmetic width) on the kernels that are not the streamed production path: generated-candidate
tick, on-device optimiser tick, generic streamed kernel (ragged K) - Sys3WRobot, B = 65536, Nactor = 10."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rcognita_amd import Engine  # noqa: E402
from rcognita_amd.pool import preset_engine_config  # noqa: E402

rng = np.random.default_rng(1)
B = 65536
x0 = np.stack([rng.uniform(-10, 10, B), rng.uniform(-10, 10, B), rng.uniform(-np.pi, np.pi, B), rng.uniform(-1, 1, B),
               rng.uniform(-1, 1, B)], -1)


def rate(tick, eng, n=40, warm=40):
    for _ in range(warm):
        tick()
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tick()
    eng.synchronize()
    return B * n / (time.perf_counter() - t0)


for dtype in ("f64", "f32"):
    e = Engine(preset_engine_config("3wrobot", B, Nactor=10, dtype=dtype))
    e.set_state(x0)
    print(f"{dtype} generated K=256 tick: {rate(lambda: e.control_tick(None, K=256), e):.3e} steps/s")
    print(f"{dtype} optimiser tick, 5 iterations: {rate(lambda: e.control_tick_opt(iters=5), e):.3e} steps/s")
    cand = e.to_device(((rng.random((B, 96, 10, 2)) - 0.5) * np.array([600, 200])).astype(e.real))
    print(f"{dtype} streamed K=96 (generic kernel): {rate(lambda: e.control_tick(cand, K=96), e):.3e} steps/s")
    e.close()
