"""Run by tests/test_hip_reset_and_guards.py in its own process (torch first, then librcg)."""
import os
import sys

import numpy as np
import torch  # noqa: F401  (before rcognita_amd: PyTorch-ROCm has to bring the GPU up first)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytest  # noqa: E402

from rcognita_amd import _native as N  # noqa: E402
from tests.helpers import both, rand_states  # noqa: E402

B, K, Nh = 8, 64, 5
eng, _ = both("3wrobot", B, "f32", n_actor=Nh)
eng.set_state(rand_states(np.random.default_rng(0), "3wrobot", B))
good = torch.zeros((B, K, Nh, 2), device="cuda", dtype=torch.float32)
eng.control_tick(good)  # baseline: accepted
with pytest.raises(ValueError, match="dtype"):
    eng.control_tick(good.double())
with pytest.raises(ValueError, match="shape"):
    eng.control_tick(good[: B - 1].contiguous())  # fewer envs than the handle owns
with pytest.raises(ValueError, match="shape"):
    eng.control_tick(torch.zeros((B, K, Nh + 1, 2), device="cuda"))
with pytest.raises(ValueError, match="K = 128"):
    eng.control_tick(good, K=128)  # the kernel would read twice the tensor
with pytest.raises(ValueError, match="contiguous"):
    eng.control_tick(good.transpose(0, 1))
with pytest.raises(ValueError):
    eng.control_tick(good.cpu())
with pytest.raises(ValueError, match="dtype"):
    eng.set_field(N.FIELD_STATE, torch.zeros((5, B), device="cuda", dtype=torch.float64))
with pytest.raises(ValueError, match="shape"):
    eng.set_field(N.FIELD_STATE, torch.zeros((5, B - 1), device="cuda"))
with pytest.raises(ValueError, match="shape"):
    eng.actor_argmin(good, obs=torch.zeros((B, 5), device="cuda"))  # device inputs must be [ds][B]
d64 = eng.empty((B, K, Nh, 2), np.float64)
with pytest.raises(ValueError, match="dtype"):
    eng.control_tick(d64)
# accepted device-resident per-env inputs: [ds][B] on the handle's device
a, bj, bi = eng.actor_argmin(good, obs=torch.zeros((5, B), device="cuda"))
assert a.shape == (B, 2) and bi.shape == (B,)
np.testing.assert_array_equal(eng.get_field(N.FIELD_STEP_IDX), np.ones(B, np.int32))  # only the good tick ran
print("INPUT_CHECKS_OK")
