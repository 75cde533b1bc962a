"""Oracle side of SURVEY 8f row f1 (on-device actor optimiser): the adjoint gradient against finite differences
of the pinned ``_actor_cost``, and the optimiser's quality against what the reference's SLSQP reaches on the same
states (tests/golden/F8_slsqp_actor_*.npz, produced with the reference's own ``_actor_cost``).  CPU-only."""
import numpy as np
import pytest

from oracle import rcg_oracle as O
from tests.conftest import load_golden
from tests.helpers import PRESETS, SYSTEMS, oracle_cfg, rand_actions, rand_states


@pytest.mark.parametrize("name", SYSTEMS)
@pytest.mark.parametrize("with_target,gamma", [(False, 1.0), (True, 0.9)])
def test_adjoint_gradient_matches_finite_differences(name, with_target, gamma):
    rng = np.random.default_rng(5)
    Nh = 7
    tgt = rng.uniform(-1, 1, PRESETS[name]["sys_id"] * 0 + oracle_cfg(name).ds) if with_target else PRESETS[name]["target"]
    cfg = oracle_cfg(name, n_actor=Nh, gamma=gamma, target=tgt)
    x = rand_states(rng, name, 1)[0]
    obs = x + 0.03  # state_sys != obs
    u = rand_actions(rng, name, (Nh,))
    J, g = O.actor_grad(u, obs, x, cfg)
    assert abs(J - O.actor_cost(u, obs, x, cfg)) <= 1e-12 * abs(J)
    gf = np.zeros_like(g)
    for i in range(Nh):
        for c in range(cfg.du):
            e = 1e-6 * max(1.0, abs(u[i, c]))
            up, um = u.copy(), u.copy()
            up[i, c] += e
            um[i, c] -= e
            gf[i, c] = (O.actor_cost(up, obs, x, cfg) - O.actor_cost(um, obs, x, cfg)) / (2 * e)
    assert np.max(np.abs(g - gf)) <= 2e-5 * max(np.max(np.abs(gf)), 1e-9)


@pytest.mark.parametrize("name", SYSTEMS)
def test_optimizer_reaches_reference_slsqp_cost(name):
    meta, z = load_golden(f"F8_slsqp_actor_{name}")
    cfg = oracle_cfg(name, n_actor=meta["N"], gamma=meta["gamma"], pred_step_size=meta["pred_step_size"])
    u0 = O.action_sqn_init(cfg, [0.5] if name == "2tank" else None)
    x = z["state"]
    U, J, its = O.actor_optimize(cfg, x, x, u0, iters=10)
    lo, hi = cfg.ctrl_bnds[:, 0], cfg.ctrl_bnds[:, 1]
    assert np.all(U >= lo - 1e-12) and np.all(U <= hi + 1e-12)
    np.testing.assert_allclose(J, O.actor_cost(U, x, x, cfg), rtol=1e-12)
    assert np.all(J <= z["J_init"] * (1 + 1e-12))
    ratio = J / z["J_opt"]
    assert np.median(ratio) < 1.0005 and np.max(ratio) < 1.002, (np.median(ratio), np.max(ratio))
    # monotone: more iterations never hurt
    _, J20, _ = O.actor_optimize(cfg, x, x, u0, iters=20)
    assert np.all(J20 <= J * (1 + 1e-12))
