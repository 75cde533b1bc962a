"""Multi-GPU side of the path: environments are independent (one System + one CtrlOptPred each,
SURVEY.md 8e), so the batch axis is sharded across ranks with NO data-path collective; ranks exchange
only episode statistics - a 6-double summary per shard, or the per-env returns - once per episode.

One process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests).  ``dist`` arguments are the ``torch.distributed`` module or None for a single process,
so nothing here imports torch on its own.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np

SUMMARY_KEYS = ("count", "sum", "sumsq", "min", "max", "n_failed")


def shard_range(n_envs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of envs owned by ``rank``: [lo, hi).  Sizes differ by at most one env and
    the blocks tile [0, n_envs) exactly (ragged totals are fine)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    base, rem = divmod(int(n_envs), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_config(cfg, n_envs: int, rank: int, world: int):
    """Copy of an ``EngineConfig`` for this rank's block of a job of ``n_envs`` envs: ``batch`` = the block size and
    ``env_id_base`` = its first GLOBAL env id, so that counter-based per-env streams (the disturbance noise,
    rcg_disturb.hpp) are the same whatever the number of ranks.  Returns ``(config, (lo, hi))``."""
    import dataclasses

    lo, hi = shard_range(n_envs, rank, world)
    return dataclasses.replace(cfg, batch=hi - lo, env_id_base=int(getattr(cfg, "env_id_base", 0)) + lo), (lo, hi)


def shard_by_type(type_counts, rank: int, world: int):
    """Mixed preset pool (BASELINE configs[4]): shard WITHIN each system type so that every rank gets
    the same type mix and load.  ``type_counts``: {type: n_envs} -> {type: (lo, hi)} for this rank."""
    return {t: shard_range(n, rank, world) for t, n in type_counts.items()}


def merge_summaries(parts) -> Dict[str, float]:
    """Combine per-shard (count, sum, sumsq, min, max, n_failed) into the whole-job summary."""
    parts = list(parts)
    out = {"count": 0.0, "sum": 0.0, "sumsq": 0.0, "min": math.inf, "max": -math.inf, "n_failed": 0.0}
    for p in parts:
        out["count"] += p["count"]
        out["sum"] += p["sum"]
        out["sumsq"] += p["sumsq"]
        out["n_failed"] += p["n_failed"]
        out["min"] = min(out["min"], p["min"])
        out["max"] = max(out["max"], p["max"])
    n = max(out["count"], 1.0)
    out["mean"] = out["sum"] / n
    out["var"] = max(out["sumsq"] / n - out["mean"] ** 2, 0.0)
    return out


def gather_summaries(summary: Dict[str, float], dist=None, device=None, force: bool = False) -> Dict[str, float]:
    """all_gather of one 6-double summary per rank (48 B per rank: latency-bound).  ``force``: run the
    collective even in a world of one (exercises the communicator on a single-GPU box)."""
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return merge_summaries([summary])
    import torch

    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    mine = torch.tensor([float(summary[k]) for k in SUMMARY_KEYS], dtype=torch.float64, device=device)
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    return merge_summaries(dict(zip(SUMMARY_KEYS, p.tolist())) for p in parts)


def gather_returns(returns, dist=None, force: bool = False):
    """all_gather of per-env episode returns (BASELINE configs[3]).  ``returns``: 1-D torch tensor on
    the rank's device (equal length on every rank) or numpy array.  Returns the concatenation in rank
    order, same kind as the input.  ``force``: as :func:`gather_summaries`."""
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return returns
    import torch

    is_np = isinstance(returns, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(returns)) if is_np else returns.contiguous()
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    out = torch.cat(parts)
    return out.numpy() if is_np else out
