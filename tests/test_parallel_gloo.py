"""N > 1 path on CPU: world_size-2 (and 3) gloo process groups exercising the shard/gather logic of
rcognita_amd.parallel.  No GPU; the per-shard numbers come from the numpy oracle."""
import os
import socket

import numpy as np
import pytest

from rcognita_amd import parallel as P


def test_shard_range_tiles_exactly():
    for n in (0, 1, 7, 64, 65536, 524288, 100003):
        for w in (1, 2, 3, 8):
            spans = [P.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        P.shard_range(10, 2, 2)


def test_shard_config_carries_global_env_ids():
    """A rank's EngineConfig = its block size + the global id of its first env; the noise of an env is a function of
    (seed, global id, episode, substep) only, so the union of the shards' streams is the unsharded stream."""
    from oracle import disturb_oracle as DO
    from rcognita_amd import EngineConfig

    base = EngineConfig(sys_id=0, batch=1, pars=[10, 1], is_disturb=True, pars_disturb=[[1, 1], [0, 0], [1, 1]], seed=77)
    n, world = 1003, 4
    ids = []
    for r in range(world):
        cfg, (lo, hi) = P.shard_config(base, n, r, world)
        assert cfg.batch == hi - lo and cfg.env_id_base == lo and cfg.seed == 77 and base.batch == 1
        c = cfg.to_native()
        assert c.env_id_base == lo and c.seed == 77 and c.batch == hi - lo
        ids.append(cfg.env_id_base + np.arange(cfg.batch, dtype=np.int64))
    z = np.zeros(n, np.int32)
    whole = DO.disturb_noise(77, np.arange(n, dtype=np.int64), z, z + 5)
    parts = np.concatenate([DO.disturb_noise(77, i, z[: len(i)], z[: len(i)] + 5) for i in ids])
    np.testing.assert_array_equal(whole, parts)


def test_shard_by_type_keeps_the_mix():
    counts = {"3wrobot": 21846, "3wrobotNI": 21845, "2tank": 21845}
    tot = {t: 0 for t in counts}
    for r in range(8):
        for t, (lo, hi) in P.shard_by_type(counts, r, 8).items():
            tot[t] += hi - lo
    assert tot == counts


def test_merge_summaries_matches_numpy():
    rng = np.random.default_rng(0)
    x = rng.normal(3, 2, 1000)
    parts = []
    for lo, hi in (P.shard_range(1000, r, 3) for r in range(3)):
        s = x[lo:hi]
        parts.append(dict(count=len(s), sum=s.sum(), sumsq=(s * s).sum(), min=s.min(), max=s.max(), n_failed=1))
    m = P.merge_summaries(parts)
    assert m["count"] == 1000 and m["n_failed"] == 3
    np.testing.assert_allclose([m["mean"], m["var"], m["min"], m["max"]], [x.mean(), x.var(), x.min(), x.max()],
                               rtol=1e-10)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_envs, q):
    import torch.distributed as dist

    from oracle import rcg_oracle as O
    from tests.helpers import oracle_cfg, rand_states

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank builds the SAME global problem from one seed, then works only on its shard
        rng = np.random.default_rng(42)
        cfg = oracle_cfg("3wrobotNI", n_actor=3)
        x0 = rand_states(rng, "3wrobotNI", n_envs)
        cand = O.grid_candidates(cfg, 16)
        lo, hi = P.shard_range(n_envs, rank, world)
        env = O.new_batch(cfg, x0[lo:hi])
        for _ in range(3):
            O.control_tick(cfg, env, cand)
        r = env.accum
        summ = dict(count=float(len(r)), sum=float(r.sum()), sumsq=float((r * r).sum()),
                    min=float(r.min()) if len(r) else float("inf"), max=float(r.max()) if len(r) else float("-inf"),
                    n_failed=0.0)
        total = P.gather_summaries(summ, dist)
        # per-env gather needs equal shard sizes: pad to the largest shard, as the runner does
        width = max(P.shard_range(n_envs, q_, world)[1] - P.shard_range(n_envs, q_, world)[0] for q_ in range(world))
        padded = np.full(width, np.nan)
        padded[: len(r)] = r
        allr = P.gather_returns(padded, dist)
        q.put((rank, total, allr))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_envs", [(2, 64), (2, 37), (3, 50)])
def test_gloo_sharded_run_equals_single_process(world, n_envs):
    import torch.multiprocessing as mp

    from oracle import rcg_oracle as O
    from tests.helpers import oracle_cfg, rand_states

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_envs, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    rng = np.random.default_rng(42)
    cfg = oracle_cfg("3wrobotNI", n_actor=3)
    env = O.new_batch(cfg, rand_states(rng, "3wrobotNI", n_envs))
    for _ in range(3):
        O.control_tick(cfg, env, O.grid_candidates(cfg, 16))
    ref = env.accum
    for rank, total, allr in results:
        assert total["count"] == n_envs
        np.testing.assert_allclose(total["sum"], ref.sum(), rtol=1e-12)
        np.testing.assert_allclose(total["sumsq"], (ref * ref).sum(), rtol=1e-12)
        assert total["min"] == ref.min() and total["max"] == ref.max()
        got = allr[~np.isnan(allr)]
        np.testing.assert_array_equal(got, ref)  # rank order == env order: sharding is a pure partition


def _c5_counts(total):
    return {"3wrobot": total // 3 + total % 3, "3wrobotNI": total // 3, "2tank": total // 3}


def _c5_state(name, g):
    """Deterministic initial state of the job's env ``g`` of a type (a function of the global id only)."""
    from tests.helpers import rand_states

    return rand_states(np.random.default_rng([77, {"3wrobot": 0, "3wrobotNI": 1, "2tank": 2}[name], int(g)]), name, 1)[0]


def _c5_env_return(name, g, ticks=2):
    """The oracle as the engine: ``ticks`` control ticks of env ``g`` with the generated 16-candidate grid."""
    from oracle import rcg_oracle as O
    from tests.helpers import oracle_cfg

    cfg = oracle_cfg(name, n_actor=4)
    env = O.new_batch(cfg, _c5_state(name, g)[None])
    cand = O.grid_candidates(cfg, 16)
    for _ in range(ticks):
        O.control_tick(cfg, env, cand)
    return float(env.accum[0])


def _c5_sample(lo, hi):
    """Envs of a shard [lo, hi) that get computed: both ends (where ragged splits go wrong) and one in the middle."""
    n = hi - lo
    return sorted({lo + i for i in (0, 1, n // 2, n - 2, n - 1) if 0 <= i < n})


def _worker_c5(rank, world, port, total, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        counts = _c5_counts(total)
        spans = P.shard_by_type(counts, rank, world)
        # this rank's per-env returns, types in sorted order (the pool's segment order), NaN = not computed here
        local, summ_parts = [], []
        for name in sorted(counts):
            lo, hi = spans[name]
            r = np.full(hi - lo, np.nan)
            for g in _c5_sample(lo, hi):
                r[g - lo] = _c5_env_return(name, g)
            v = r[~np.isnan(r)]
            summ_parts.append(dict(count=float(hi - lo), sum=float(v.sum()), sumsq=float((v * v).sum()), min=float(v.min()),
                                   max=float(v.max()), n_failed=0.0))
            local.append(r)
        local = np.concatenate(local)
        total_summ = P.gather_summaries(P.merge_summaries(summ_parts), dist)
        width = max(sum(hi - lo for lo, hi in P.shard_by_type(counts, q_, world).values()) for q_ in range(world))
        padded = np.full(width, np.nan)
        padded[: len(local)] = local
        allr = P.gather_returns(padded, dist)  # the episode-end exchange of configs[4]: ragged shards, padded payload
        q.put((rank, total_summ, allr if rank == 0 else None, len(local)))
    finally:
        dist.destroy_process_group()


def test_gloo_world8_mixed_pool_with_ragged_totals():
    """Pre-flight of the 8-rank run of configs[4] that no node has been available for: 8 gloo ranks, the job of 8 x 65536 + 5
    envs sharded WITHIN each system type at full size (index arithmetic, payload widths and the padded all_gather are the
    real ones: 4 MB), the oracle as each rank's engine on the envs at both ends and in the middle of every shard.  Rank 0
    reassembles every type's global env order from the gathered payload and finds each computed env where the unsharded
    job has it, with the value the single-process oracle gives."""
    import torch.multiprocessing as mp

    world, total = 8, 65536 * 8 + 5
    counts = _c5_counts(total)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_c5, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sizes = [r[3] for r in results]
    assert sum(sizes) == total and max(sizes) - min(sizes) <= 3
    allr = results[0][2]
    width = max(sizes)
    assert allr.shape == (world * width,)
    n_checked = 0
    for rank in range(world):
        spans = P.shard_by_type(counts, rank, world)
        off = rank * width
        for name in sorted(counts):
            lo, hi = spans[name]
            seg = allr[off:off + hi - lo]
            for g in _c5_sample(lo, hi):
                assert seg[g - lo] == _c5_env_return(name, g), (rank, name, g)
                n_checked += 1
            assert np.isnan(seg).sum() == (hi - lo) - len(_c5_sample(lo, hi))
            off += hi - lo
        assert np.all(np.isnan(allr[off:(rank + 1) * width]))  # the padding of a short shard
    assert n_checked == world * 3 * 5
    for _, summ, _, _ in results:  # every rank holds the same merged summary
        assert summ["count"] == total and summ == results[0][1]
