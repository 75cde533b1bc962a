#!/usr/bin/env python3
"""Steady-state cost of the critic fit (env step + buffer push + fit: one launch of k_critic_fit) on the configs[2] shape: RQL closed loop,
per-tick HIP-event time of the critic update after the buffers have filled."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rcognita_amd import Engine  # noqa: E402
from rcognita_amd import _native as N  # noqa: E402
from rcognita_amd.pool import preset_engine_config  # noqa: E402

rng = np.random.default_rng(1234)
B, K = 131072, 256
# FIT_ROWS=<Ncritic>,<buffer_size> (round 6): more TD rows, e.g. FIT_ROWS=12,20 -> 11 rows on k_critic_fit_gen; FIT_B=<envs>
NC, BS = (int(v) for v in os.environ.get("FIT_ROWS", "4,10").split(","))
B = int(os.environ.get("FIT_B", B))
for cs in (sys.argv[1:] or ("quadratic", "quad-lin", "quad-nomix", "quad-mix")):
    eng = Engine(preset_engine_config("2tank", B, Nactor=20, mode="RQL", critic_struct=cs, Ncritic=NC, buffer_size=BS))
    eng.set_state(np.stack([rng.uniform(0, 2, B), rng.uniform(-2, 2, B)], axis=-1))
    for _ in range(max(25, BS + 5)):
        eng.control_tick(None, K=K)
    eng.profile((N.KERNEL_CRITIC, N.KERNEL_ACTOR), stride=1)
    for _ in range(20):
        eng.control_tick(None, K=K)
    eng.synchronize()
    cm, cn = eng.profile_read(N.KERNEL_CRITIC)
    am, an = eng.profile_read(N.KERNEL_ACTOR)
    print(f"2tank RQL {cs:10s} B={B} Ncritic={NC} buffer={BS} ({eng.last_launch(N.KERNEL_CRITIC)}): critic push+fit {cm / cn * 1e3:.1f} us per tick, actor {am / an * 1e3:.1f} us")
    eng.close()
