#!/usr/bin/env python3
"""One structure's critic fit, a few launches, for a counter pass:
   rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -d gpurun_out/fit_pmc -o p --output-format csv -- python3 tools/fit_pmc_probe.py quad-lin"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: F401
from valu_probe import states
from rcognita_amd import Engine
from rcognita_amd.pool import preset_engine_config
cs = sys.argv[1] if len(sys.argv) > 1 else "quad-lin"
B = 65536
rng = np.random.default_rng(7)
e = Engine(preset_engine_config("3wrobot", B, Nactor=10, dtype="f32", mode="RQL", critic_struct=cs, buffer_size=10))
e.set_state(states(rng, "3wrobot", B))
for _ in range(12):
    e.control_tick_opt(iters=2)
for _ in range(5):
    e.critic_update(do_fit=True)
e.synchronize()
