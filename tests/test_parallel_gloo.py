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
