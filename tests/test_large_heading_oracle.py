"""The CPU oracle against fixture F13 (oracle/gen_large_heading_fixture.py): the reference's robots at headings of
30 ... 1e4 rad (SURVEY §7 hard part 3).  CPU only; the GPU side is tests/test_hip_large_heading.py."""
import numpy as np
import pytest

from oracle import rcg_oracle as O
from tests.conftest import load_golden
from tests.helpers import oracle_cfg, rel_err, rel_err_norm

ROBOTS = ["3wrobot", "3wrobotNI"]


@pytest.mark.parametrize("name", ROBOTS)
def test_F13_spans_the_headings_it_claims(name):
    meta, z = load_golden(f"F13_large_heading_{name}")
    assert meta["mags"] == [30.0, 72.0, 300.0, 1e3, 1e4]
    a = np.abs(z["rhs__state"][:, 2])
    assert a.min() > 20 and a.max() > 9.9e3
    for key in z.files:  # inputs are float32 (exact), outputs float64
        if key.endswith(("__state", "__action", "__action_sqn", "__w", "__levels")):
            assert z[key].dtype == np.float32, key


@pytest.mark.parametrize("name", ROBOTS)
def test_F13_rhs(name):
    _, z = load_golden(f"F13_large_heading_{name}")
    cfg = oracle_cfg(name)
    x, u = z["rhs__state"].astype(np.float64), z["rhs__action"].astype(np.float64)
    assert rel_err_norm(O.state_dyn(cfg.sys_id, x, u, cfg.pars), z["rhs__state_dyn"]) < 1e-13
    d, a = O.closed_loop_rhs(cfg.sys_id, x, u, cfg.pars, cfg.ctrl_bnds)
    assert rel_err_norm(d, z["rhs__closed_loop_rhs"]) < 1e-13
    np.testing.assert_array_equal(a, z["rhs__clipped_action"])


@pytest.mark.parametrize("name", ROBOTS)
def test_F13_actor_cost(name):
    meta, z = load_golden(f"F13_large_heading_{name}")
    assert len(meta["cost_cases"]) == 15
    for c in meta["cost_cases"]:
        tag = "cost_" + c["tag"]
        cfg = oracle_cfg(name, n_actor=c["N"], mode=O.MODE_IDS[c["mode"]], gamma=c["gamma"],
                         critic_struct=O.CRITIC_IDS[c["critic_struct"]], pred_step_size=c["pred_step_size"])
        x = z[f"{tag}__state"].astype(np.float64)
        J = O.actor_cost(z[f"{tag}__action_sqn"].astype(np.float64), x[:, None, :], x[:, None, :], cfg,
                         w_critic=z[f"{tag}__w"].astype(np.float64)[:, None, :])
        assert rel_err(J, z[f"{tag}__J"]) < 1e-11, tag


@pytest.mark.parametrize("name", ROBOTS)
def test_F13_generated_grid(name):
    """The float32 kernels' own grid levels (stored) are the oracle's float64 levels rounded the kernel's way, and the
    oracle's cost on them is the reference's."""
    _, z = load_golden(f"F13_large_heading_{name}")
    cfg = oracle_cfg(name, n_actor=10, pred_step_size=float(z["grid__pred_step_size"]))
    lev = z["grid__levels"].astype(np.float64)
    g64 = O.grid_candidates(cfg, 256)
    assert np.max(np.abs(g64[:, 0, :] - lev)) < 2e-5
    cand = np.broadcast_to(lev[:, None, :], (256, 10, 2))
    x = z["grid__state"].astype(np.float64)
    J = O.actor_cost(cand[None], x[:, None, :], x[:, None, :], cfg)
    assert rel_err(J, z["grid__J"]) < 1e-11
    # a state at rest under a zero action is a fixed point of the env step (what the fused-tick GPU test relies on)
    x1 = O.rk4_step(cfg.sys_id, x, np.zeros((len(x), 2)), cfg.pars, cfg.ctrl_bnds, cfg.dt_sim)
    np.testing.assert_array_equal(x1, x)


@pytest.mark.parametrize("name", ROBOTS)
def test_F13_trajectories_rk4_vs_reference_rk45(name):
    """Fixed-step float64 RK4 on the reference's own time grid, each step a map from the reference's previous state, and
    as a free run: <= 1e-5 (the north star's tolerance) with the heading at 72 ... 1e4 rad."""
    meta, z = load_golden(f"F13_large_heading_{name}")
    cfg = oracle_cfg(name)
    for tr in meta["traj"]:
        t, y = z[f"traj_{tr['tag']}__t"], z[f"traj_{tr['tag']}__y"]
        u = np.array(tr["action"])
        worst, x = 0.0, y[0].copy()
        for i in range(len(t) - 1):
            h = t[i + 1] - t[i]
            worst = max(worst, rel_err_norm(O.rk4_step(cfg.sys_id, y[i], u, cfg.pars, cfg.ctrl_bnds, h), y[i + 1]))
            x = O.rk4_step(cfg.sys_id, x, u, cfg.pars, cfg.ctrl_bnds, h)
        assert worst < 1e-5, (tr["tag"], worst)
        assert rel_err_norm(x, y[-1]) < 1e-5, tr["tag"]
        assert abs(y[-1][2]) > 0.9 * tr["A"]
