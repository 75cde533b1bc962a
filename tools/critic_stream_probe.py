#!/usr/bin/env python3
"""Streamed RQL / SQL decisions of the robots and the tank in both element types: kernel that ran (rcg_last_launch), time per
launch from the dispatch's own stamps, algorithmic TB/s.  GPU box only.

    python tools/critic_stream_probe.py [f64|f32] [B] [all]
    PROBE_LIB=rcognita_amd/lib/librcg_dev.so RCG_PER_CU=4 python tools/critic_stream_probe.py f32 65536 all   # a dev-build knob
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rcognita_amd import _native as N  # noqa: E402

if os.environ.get("PROBE_LIB"):
    N.use_library(os.path.join(ROOT, os.environ["PROBE_LIB"]))
from rcognita_amd import Engine  # noqa: E402
from rcognita_amd.pool import PRESETS, preset_engine_config  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "f64"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
K, Nh = 256, 10
td = torch.float64 if dtype == "f64" else torch.float32
esz = 8 if dtype == "f64" else 4
rng = np.random.default_rng(0)
ALL = [(n, "MPC", "quad-nomix") for n in ("3wrobot", "3wrobotNI", "2tank")] + [
    (n, m, c) for n in ("3wrobot", "3wrobotNI", "2tank") for m in ("RQL", "SQL")
    for c in ("quad-lin", "quadratic", "quad-nomix", "quad-mix")]
for name, mode, cs in ALL if (len(sys.argv) > 3 and sys.argv[3] == "all") else (("3wrobot", "MPC", "quad-nomix"), ("3wrobot", "RQL", "quad-nomix"), ("3wrobot", "RQL", "quad-lin"),
                       ("3wrobot", "SQL", "quad-nomix"), ("3wrobot", "SQL", "quadratic"), ("3wrobot", "SQL", "quad-lin"),
                       ("3wrobotNI", "RQL", "quadratic"), ("3wrobotNI", "SQL", "quad-mix"), ("2tank", "RQL", "quadratic"),
                       ("2tank", "SQL", "quad-lin")):
    p = PRESETS[name]
    eng = Engine(preset_engine_config(name, B, Nactor=Nh, dtype=dtype, mode=mode, critic_struct=cs, Ncritic=4,
                                      buffer_size=10 if mode != "MPC" else 0))
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    ds, du = eng.ds, eng.du
    eng.set_state(rng.uniform(-2, 2, (B, ds)))
    if mode != "MPC":
        eng.set_field(N.FIELD_W_CRITIC, rng.uniform(0.1, 2.0, (B, eng.dc)))
    lo = torch.tensor(np.array(p["ctrl_bnds"])[:, 0], device="cuda", dtype=td)
    hi = torch.tensor(np.array(p["ctrl_bnds"])[:, 1], device="cuda", dtype=td)
    cand = (torch.rand((B, K, Nh, du), device="cuda", dtype=td) * (hi - lo) + lo).contiguous()
    act = torch.empty((du, B), device="cuda", dtype=td)
    for _ in range(60):
        eng.actor_argmin(cand, K=K) if False else N.check(N.lib().rcg_actor_argmin(eng._h, cand.data_ptr(), K, None, None, act.data_ptr(), None, None), eng._h)
    eng.profile((N.KERNEL_ACTOR,), stride=2)
    for _ in range(80):
        N.check(N.lib().rcg_actor_argmin(eng._h, cand.data_ptr(), K, None, None, act.data_ptr(), None, None), eng._h)
    s = eng.profile_samples(N.KERNEL_ACTOR)
    ll = eng.last_launch(N.KERNEL_ACTOR)
    byt = B * (K * Nh * du * esz + ds * esz + (eng.dc * esz if mode != "MPC" else 0) + du * esz)
    print(f"{dtype} {name:10s} {mode} {cs:10s} {ll['kernel']:12s} v{ll['variant']} gpw {ll['envs_per_wave']:2d}: "
          f"median {np.median(s) * 1e3:7.1f} us  min {s.min() * 1e3:7.1f}  {byt / (np.median(s) * 1e-3) / 1e12:5.2f} TB/s "
          f"({byt / (np.median(s) * 1e-3) / 8e12:.3f} of peak)", flush=True)
    eng.close()
    del cand
