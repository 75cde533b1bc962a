"""bench.py's own rank launcher (`python bench.py --gpus N` from a plain shell), exercised without a GPU:
`--dry-launch` makes every rank rendezvous over gloo on the CPU and report what the launcher handed to it."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if not k.startswith("RCG_") and k not in
           ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + argv, capture_output=True, text=True, env=env, timeout=timeout,
                          cwd=ROOT)


@pytest.mark.parametrize("n,extra,total", [(2, [], 2 * 65536), (3, ["--scaling", "strong", "--batch", "100"], 100),
                                           (2, ["--config", "C4"], 524288), (2, ["--config", "C3"], 2 * 131072),
                                           # the driver's N = 8 forms: C2 weak, and configs[4] strong with a ragged total
                                           (8, [], 8 * 65536),
                                           (8, ["--config", "C5", "--scaling", "strong", "--batch", str(8 * 65536 + 5)],
                                            8 * 65536 + 5)])
def test_gpus_n_spawns_n_ranks_that_tile_the_job(n, extra, total):
    out = _run(["--gpus", str(n), "--dry-launch"] + extra)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, out.stdout  # ONE JSON line, from rank 0
    j = json.loads(line[0])
    assert j["n_gpus"] == n and j["rccl_ranks"] == n and j["launcher"] == "bench.py self-spawn"
    ranks = j["ranks"]
    assert [r["rank"] for r in ranks] == list(range(n)) and [r["local_rank"] for r in ranks] == list(range(n))
    assert len({r["pid"] for r in ranks}) == n  # one process per rank
    assert ranks[0]["envs"][0] == 0 and ranks[-1]["envs"][1] == total
    for a, b in zip(ranks, ranks[1:]):  # contiguous, disjoint shards
        assert a["envs"][1] == b["envs"][0]


def test_a_failing_rank_fails_the_launch():
    """No GPU in this container: every rank exits with the 'needs an MI355X' message, the parent must not exit 0."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present: the ranks would run")
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert out.returncode != 0
    assert "MI355X" in out.stderr and "rank" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_under_an_external_launcher_the_process_is_a_rank():
    """WORLD_SIZE already set (torchrun): no children are spawned; here a world of one."""
    out = _run(["--gpus", "1", "--dry-launch"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
                                                 "MASTER_PORT": "29741"})
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["launcher"] == "single process"


@pytest.mark.parametrize("knob", ["RCG_DBG", "RCG_LIB", "RCG_GPW"])
def test_developer_knobs_are_refused(knob):
    out = _run(["--dry-launch"], {knob: "1"})
    assert out.returncode != 0 and knob in out.stderr
