"""Property tests (hypothesis) of the host-side sharding and summary logic of the multi-GPU path: the shards
tile the job exactly whatever the totals, the per-type sharding keeps every rank's type mix, and merging per-shard
summaries equals the summary of the whole.  CPU only."""
import math

import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

from rcognita_amd.parallel import merge_summaries, shard_by_type, shard_range


@settings(max_examples=300, deadline=None)
@given(n=st.integers(0, 3_000_000), world=st.integers(1, 64))
def test_shards_tile_the_job_exactly_and_evenly(n, world):
    spans = [shard_range(n, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == n
    for (lo, hi), (lo2, hi2) in zip(spans, spans[1:]):
        assert hi == lo2 and lo <= hi
    sizes = [hi - lo for lo, hi in spans]
    assert max(sizes) - min(sizes) <= 1 and sum(sizes) == n
    assert sizes == sorted(sizes, reverse=True)  # the larger shards come first: rank 0 is never the smaller one


@settings(max_examples=200, deadline=None)
@given(counts=st.dictionaries(st.sampled_from(["3wrobot", "3wrobotNI", "2tank"]), st.integers(0, 400_000), min_size=1),
       world=st.integers(1, 16))
def test_per_type_sharding_gives_every_rank_the_same_mix(counts, world):
    per_rank = [shard_by_type(counts, r, world) for r in range(world)]
    for t, n in counts.items():
        sizes = [per_rank[r][t][1] - per_rank[r][t][0] for r in range(world)]
        assert sum(sizes) == n and max(sizes) - min(sizes) <= 1
        assert per_rank[0][t][0] == 0 and per_rank[-1][t][1] == n
    loads = [sum(hi - lo for lo, hi in per_rank[r].values()) for r in range(world)]
    assert max(loads) - min(loads) <= len(counts)  # at most one env per type apart


def _summary(v, failed):
    v = np.asarray(v, dtype=np.float64)
    return {"count": float(len(v)), "sum": float(v.sum()), "sumsq": float((v * v).sum()),
            "min": float(v.min()) if len(v) else math.inf, "max": float(v.max()) if len(v) else -math.inf,
            "n_failed": float(failed)}


@settings(max_examples=200, deadline=None)
@given(values=st.lists(st.floats(-1e6, 1e6, allow_nan=False), min_size=1, max_size=200), world=st.integers(1, 9),
       failed=st.integers(0, 5))
def test_merged_shard_summaries_equal_the_summary_of_the_whole(values, world, failed):
    v = np.asarray(values)
    parts = []
    for r in range(world):
        lo, hi = shard_range(len(v), r, world)
        parts.append(_summary(v[lo:hi], failed if r == 0 else 0))
    got, want = merge_summaries(parts), _summary(v, failed)
    assert got["count"] == want["count"] and got["n_failed"] == want["n_failed"]
    assert got["min"] == want["min"] and got["max"] == want["max"]
    scale = max(float(np.abs(v).sum()), 1.0)
    assert abs(got["sum"] - want["sum"]) <= 1e-12 * scale
    assert abs(got["sumsq"] - want["sumsq"]) <= 1e-12 * max(float((v * v).sum()), 1.0)
    assert abs(got["mean"] - v.mean()) <= 1e-9 * max(abs(v.mean()), 1.0)
    assert got["var"] >= 0.0
