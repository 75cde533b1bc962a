"""The C oracle (oracle/oracle.c, the timed CPU baseline of bench.py) against the reference's golden
vectors and against the numpy oracle.  CPU-only."""
import numpy as np
import pytest

from oracle import c_oracle as CO
from oracle import rcg_oracle as O
from tests.conftest import load_golden
from tests.helpers import SYSTEMS, oracle_cfg, rand_actions, rand_states, rel_err


@pytest.mark.parametrize("name", SYSTEMS)
def test_c_actor_cost_golden(name):
    meta, z = load_golden(f"F4_actor_cost_{name}")
    for c in meta["cases"]:
        tag = c["tag"]
        cfg = oracle_cfg(name, n_actor=c["N"], mode=O.MODE_IDS[c["mode"]], gamma=c["gamma"],
                         critic_struct=O.CRITIC_IDS[c["critic_struct"]], pred_step_size=c["pred_step_size"])
        J = CO.actor_cost(cfg, z[f"{tag}__action_sqn"][:, None], z[f"{tag}__obs"], z[f"{tag}__state_sys"],
                          w=z[f"{tag}__w"], nthreads=2)
        assert rel_err(J[:, 0], z[f"{tag}__J"]) < 1e-11, tag


@pytest.mark.parametrize("name", SYSTEMS)
def test_c_actor_cost_golden_production_shape(name):
    """F4b (64 sequences per env, state_sys == obs): the C oracle is the timed CPU baseline of exactly this shape."""
    meta, z = load_golden(f"F4b_actor_cost_dma_{name}")
    for c in meta["cases"]:
        tag = c["tag"]
        cfg = oracle_cfg(name, n_actor=c["N"], mode=O.MODE_IDS[c["mode"]], gamma=c["gamma"],
                         critic_struct=O.CRITIC_IDS[c["critic_struct"]], pred_step_size=c["pred_step_size"])
        x = z[f"{tag}__state"].astype(np.float64)
        J = CO.actor_cost(cfg, z[f"{tag}__action_sqn"].astype(np.float64), x, x, w=z[f"{tag}__w"].astype(np.float64),
                          nthreads=2)
        assert J.shape == (2, 64) and rel_err(J, z[f"{tag}__J"]) < 1e-11, tag


@pytest.mark.parametrize("name", SYSTEMS)
@pytest.mark.parametrize("per_env", [False, True])
def test_c_control_tick_equals_numpy_oracle(name, per_env):
    rng = np.random.default_rng(21)
    B, K, N = 17, 20, 6
    cfg = oracle_cfg(name, n_actor=N, substeps_per_tick=2, gamma=0.98)
    x0 = rand_states(rng, name, B)
    pars = None
    if per_env and cfg.pars.size:
        pars = cfg.pars * rng.uniform(0.8, 1.2, (B, cfg.pars.size))
    cand = rand_actions(rng, name, (B, K, N), overshoot=1.1)
    env = O.new_batch(cfg, x0, pars=pars)
    cb = CO.CBatch(cfg, x0, pars=pars)
    for _ in range(6):
        O.control_tick(cfg, env, cand)
        cb.tick(cand, nthreads=3)
        np.testing.assert_array_equal(cb.best_idx, env.best_idx)
        np.testing.assert_allclose(cb.state, env.state, rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(cb.accum, env.accum, rtol=1e-12)
        np.testing.assert_array_equal(cb.step_idx, env.step_idx)
