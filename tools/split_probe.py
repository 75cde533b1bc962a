#!/usr/bin/env python3
"""configs[2] (Sys2Tank, 131072 envs, Nactor 20, RQL + quadratic critic fit every tick, K = 256 streamed) as ONE handle and as
S handles of 131072 / S envs on streams of their own: does the latency-bound critic fit of one part hide behind the
HBM-bound actor kernel of another?  GPU box only.   python tools/split_probe.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rcognita_amd import _native as N  # noqa: E402

if os.environ.get("PROBE_LIB"):  # a dev-build knob: PROBE_LIB=rcognita_amd/lib/librcg_dev.so RCG_GPW=16 SPLITS=2,2 python tools/split_probe.py
    N.use_library(os.path.join(ROOT, os.environ["PROBE_LIB"]))
from rcognita_amd import Engine  # noqa: E402
from rcognita_amd.pool import preset_engine_config  # noqa: E402

B, K, Nh = 131072, 256, 20
rng = np.random.default_rng(1)
x0 = np.stack([rng.uniform(0, 2, B), rng.uniform(-2, 2, B)], -1)
cand = torch.rand((B, K, Nh, 1), device="cuda").contiguous()
torch.cuda.synchronize()
for S in [int(v) for v in os.environ.get("SPLITS", "1,2,4,1,2").split(",")]:
    n = B // S
    engs, streams = [], []
    for i in range(S):
        e = Engine(preset_engine_config("2tank", n, Nactor=Nh, mode="RQL", critic_struct="quadratic", Ncritic=4, buffer_size=10))
        st = torch.cuda.Stream()
        e.set_stream(st.cuda_stream)
        e.set_tick_parts(int(os.environ.get("TICK_PARTS", "1")))  # 1: the handles' own split off (this tool measures handles on streams)
        e.set_state(x0[i * n:(i + 1) * n])
        engs.append(e)
        streams.append(st)
    parts = [cand[i * n:(i + 1) * n] for i in range(S)]

    def tick():
        for e, c in zip(engs, parts):
            e.control_tick(c, K=K)

    for _ in range(400):
        tick()
    for e in engs:  # (round 5) a handle that splits its own tick over two internal streams: order its stream behind them
        e.join()
    a = [torch.cuda.Event(enable_timing=True) for _ in streams]
    b = [torch.cuda.Event(enable_timing=True) for _ in streams]
    here = [torch.cuda.Event() for _ in streams]  # align the streams on the device: the timed ticks start together
    for ev, st in zip(here, streams):
        ev.record(st)
    for i, st in enumerate(streams):
        for j, ev in enumerate(here):
            if j != i:
                st.wait_event(ev)
    for ev, st in zip(a, streams):
        ev.record(st)
    T = 300
    for _ in range(T):
        tick()
    for e in engs:
        e.join()
    for ev, st in zip(b, streams):
        ev.record(st)
    torch.cuda.synchronize()
    ms = max(x.elapsed_time(y) for x in a for y in b) / T
    print(f"S = {S}: {ms:.4f} ms per tick of {B} envs = {B / ms * 1e3:.4g} env-control-steps/s", flush=True)
    for e in engs:
        e.close()
