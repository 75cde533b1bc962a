"""RCCL on real hardware: the N>1 exchange of `bench.py` and `rcognita_amd/parallel.py` over
``torch.distributed`` backend "nccl" (= RCCL on ROCm), exercised with the ONE GPU a test box has: communicator creation
bound to the device, the device-side ``all_gather`` of ``returns[B]`` and of the 6-double summary, ``all_reduce``,
``barrier``, ``destroy_process_group``.  Each case is a child process (a process group belongs to its process).
``gpu`` marked.  Reference unit being replicated per rank: presets/main_3wrobot.py:415-468 (one loop per env)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")

CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "%(port)d")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np
import torch
import torch.distributed as dist
from rcognita_amd import Engine, _native as N
from rcognita_amd.pool import preset_engine_config
from rcognita_amd.parallel import gather_returns, gather_summaries, shard_range

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
B, K = 4099, 64
lo, hi = shard_range(B, 0, 1)
eng = Engine(preset_engine_config("3wrobot", hi - lo, Nactor=10))
eng.set_stream(torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(3)
eng.set_state(rng.uniform(-3, 3, (B, 5)))
for _ in range(3):
    eng.control_tick(None, K=K)
# the device-side exchange: ACCUM -> a torch tensor on the GPU (device-to-device, stream-ordered) -> all_gather over RCCL
ret = torch.zeros(B, device=dev, dtype=torch.float32)
N.check(N.lib().rcg_get_field(eng._h, N.FIELD_ACCUM, ret.data_ptr(), N.DEVICE), eng._h)
allret = gather_returns(ret, dist, force=True)
assert allret.is_cuda and allret.shape == (B,)
summ, host_ret = eng.episode_stats(from_accum=True, want_returns=True)
assert np.array_equal(allret.cpu().numpy(), host_ret), "all_gather over RCCL must return this rank's returns bit for bit"
total = gather_summaries(summ, dist, force=True)   # 6 doubles through the communicator, on the device
assert total["count"] == B and abs(total["sum"] - float(host_ret.astype(np.float64).sum())) <= 1e-6 * abs(total["sum"])
m = torch.tensor([1.5], device=dev, dtype=torch.float64)
dist.all_reduce(m, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert m.item() == 1.5
eng.close()
dist.destroy_process_group()
print("RCCL_WORLD1_OK", float(total["sum"]))
"""


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env():
    env = {k: v for k, v in os.environ.items() if not k.startswith("RCG_")}
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


def test_rccl_world1_device_all_gather_of_returns_and_summary():
    code = CHILD % {"root": ROOT, "port": _free_port()}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=_env())
    assert out.returncode == 0, out.stderr[-3000:]
    assert "RCCL_WORLD1_OK" in out.stdout, out.stdout[-2000:]


def _bench(extra, timeout=900):
    cmd = [sys.executable, BENCH, "--steps", "6", "--warmup", "2", "--batch", "8192", "--candidates", "64",
           "--no-cpu-baseline", "--no-secondary"] + extra
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=_env())


def test_bench_force_dist_runs_every_collective_over_rccl_at_world_size_1():
    out = _bench(["--force-dist"])
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["rccl_ranks"] == 1 and line["dist_backend"] == "nccl"
    assert line["ranks"] == [{"rank": 0, "device": 0, "envs": 8192}]
    t = line["timing"]
    assert t["allgather_ms"] is not None and 0.0 < t["allgather_ms"] < 50.0
    assert t["value_compute_only"] >= line["value"] > 0.0
    assert line["parity"]["ok"]


def test_two_ranks_on_a_one_gpu_box_fail_loudly_over_rccl_and_run_over_gloo():
    import torch

    if torch.cuda.device_count() != 1:
        pytest.skip("needs a box with exactly one visible GPU")
    bad = _bench(["--gpus", "2", "--launch-timeout", "300"])
    assert bad.returncode != 0, "two RCCL ranks cannot share one GPU: the launch must fail, not alias cuda:0"
    assert "only 1 device(s) are visible" in bad.stderr, bad.stderr[-2000:]
    ok = _bench(["--gpus", "2", "--dist-backend", "gloo", "--single-device", "--launch-timeout", "600"])
    assert ok.returncode == 0, ok.stderr[-3000:]
    line = json.loads(ok.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["dist_backend"] == "gloo"
    assert line["timing"]["allgather_ms"] is not None
