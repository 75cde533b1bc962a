#!/usr/bin/env python3
"""Streamed decisions whose cost structure is NOT the presets' (VERDICT r5 missing 3 / next 6): the C2 shape (Sys3WRobot, 65 536
envs, Nactor = 10, K = 256 streamed) with a full R1, with the biquadratic stage cost (controllers.py:1063-1084), with an
observation target on the robot, and an env slab that is not a whole number of 16-byte pieces - beside the preset's diagonal
R1.  Prints the kernel that ran (rcg_last_launch), time per launch from the dispatch's own stamps, algorithmic TB/s.
GPU box only.

    python tools/generic_stream_probe.py [f32|f64] [B]
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

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
td = torch.float64 if dtype == "f64" else torch.float32
esz = 8 if dtype == "f64" else 4
rng = np.random.default_rng(0)
A = rng.uniform(-1, 1, (7, 7))
R1_full = A @ A.T + np.diag([1.0, 10, 1, 0, 0, 0, 0])
R2_diag = np.diag(rng.uniform(0, 1e-3, 7))
R2_full = R2_diag + 1e-4 * (A.T @ A)
A3 = rng.uniform(-1, 1, (3, 3))
CASES = [
    ("3wrobot", "diagonal R1 (preset)", 256, 10, {}),
    ("3wrobot", "full R1", 256, 10, dict(R1=R1_full)),
    ("3wrobot", "biquadratic, diagonal R1 R2", 256, 10, dict(stage_obj_struct="biquadratic", R2=R2_diag)),
    ("3wrobot", "biquadratic, full R1 R2", 256, 10, dict(stage_obj_struct="biquadratic", R1=R1_full, R2=R2_full)),
    ("3wrobot", "diagonal R1 + observation target", 256, 10, dict(observation_target=[1.0, -2.0, 0.5, 0.0, 0.0])),
    ("3wrobot", "RQL quad-nomix, full R1", 256, 10, dict(R1=R1_full, mode="RQL", critic_struct="quad-nomix", buffer_size=10)),
    ("3wrobot", "SQL quad-nomix, full R1", 256, 10, dict(R1=R1_full, mode="SQL", critic_struct="quad-nomix", buffer_size=10)),
    ("3wrobot", "RQL quad-lin (35 weights), biquadratic diagonal", 256, 10, dict(stage_obj_struct="biquadratic", R2=R2_diag, mode="RQL",
                                                                                critic_struct="quad-lin", buffer_size=10)),
    ("2tank", "diagonal R1, slab of 255 x 36 B (not 16-byte pieces)", 255, 9, {}),
    ("2tank", "full R1 (with the preset's target)", 256, 20, dict(R1=A3 @ A3.T + np.diag([10.0, 10, 1]))),
]
for name, what, K, Nh, kw in CASES:
    p = PRESETS[name]
    eng = Engine(preset_engine_config(name, B, Nactor=Nh, dtype=dtype, **kw))
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    ds, du = eng.ds, eng.du
    eng.set_state(rng.uniform(-2, 2, (B, ds)))
    if kw.get("mode", "MPC") != "MPC":
        eng.set_field(N.FIELD_W_CRITIC, rng.uniform(0.1, 2.0, (B, eng.dc)))
    lo = torch.tensor(np.array(p["ctrl_bnds"])[:, 0], device="cuda", dtype=td)
    hi = torch.tensor(np.array(p["ctrl_bnds"])[:, 1], device="cuda", dtype=td)
    cand = (torch.rand((B, K, Nh, du), device="cuda", dtype=td) * (hi - lo) + lo).contiguous()
    act = torch.empty((du, B), device="cuda", dtype=td)
    call = lambda: N.check(N.lib().rcg_actor_argmin(eng._h, cand.data_ptr(), K, None, None, act.data_ptr(), None, None), eng._h)
    for _ in range(60):
        call()
    eng.profile((N.KERNEL_ACTOR,), stride=2)
    for _ in range(80):
        call()
    s = eng.profile_samples(N.KERNEL_ACTOR)
    ll = eng.last_launch(N.KERNEL_ACTOR)
    byt = B * (K * Nh * du * esz + ds * esz + du * esz)
    print(f"{dtype} {name:8s} {what:52s} {ll['kernel']:18s} v{ll['variant']:<3d}: median {np.median(s) * 1e3:7.1f} us  "
          f"min {s.min() * 1e3:7.1f}  {byt / (np.median(s) * 1e-3) / 1e12:5.2f} TB/s ({byt / (np.median(s) * 1e-3) / 8e12:.3f} of peak)",
          flush=True)
    eng.close()
    del cand
