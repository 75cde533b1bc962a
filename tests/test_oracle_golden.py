"""Pin the CPU oracle (oracle/rcg_oracle.py) against golden vectors produced by the reference itself
(oracle/gen_fixtures.py, run in the build container).  CPU-only."""
import numpy as np
import pytest

from oracle import rcg_oracle as O
from tests.conftest import load_golden
from tests.helpers import PRESETS, SYSTEMS, oracle_cfg, rel_err

TOL = 1e-12  # fp64 restatement vs fp64 reference


def test_known_answers():
    """literal values quoted in the survey AND recomputed in KAT.npz."""
    _, k = load_golden("KAT")
    x = np.array([5, 5, -3 * np.pi / 4, 0.3, -0.2])
    u = np.array([50.0, -20.0])
    cfg = oracle_cfg("3wrobot", n_actor=5, gamma=0.9, pred_step_size=0.02)
    d = O.state_dyn(O.SYS_3WROBOT, x, u, cfg.pars)
    np.testing.assert_allclose(d, [-0.212132034356, -0.212132034356, -0.2, 5, -20], rtol=1e-11)
    np.testing.assert_allclose(d, k["kat1"], rtol=TOL)
    rhs, a = O.closed_loop_rhs(O.SYS_3WROBOT, x, np.array([400.0, -150.0]), cfg.pars, cfg.ctrl_bnds)
    np.testing.assert_allclose(rhs, k["kat2_rhs"], rtol=TOL)
    np.testing.assert_allclose(a, [300, -100])
    np.testing.assert_allclose(a, k["kat2_action"])
    assert abs(O.stage_obj(x, u, cfg) - 280.55165247561274) < 1e-10
    aseq = np.array([[50, -20], [40, -10], [30, 0], [20, 10], [10, 20]], dtype=float)
    w = 0.5 * np.arange(1, 8)
    for mode, val in (("MPC", 1145.9245022338207), ("RQL", 2710.8144679266966), ("SQL", 20236.425402576948)):
        c = oracle_cfg("3wrobot", n_actor=5, gamma=0.9, pred_step_size=0.02, mode=O.MODE_IDS[mode],
                       critic_struct=O.CRITIC_QUAD_NOMIX)
        J = O.actor_cost(aseq.reshape(-1), x + 0.01, x, c, w_critic=w)
        assert abs(J - val) / val < TOL
        assert abs(J - float(k[f"kat4_{mode}"])) / val < TOL
    for cs, val in (("quad-lin", 1541.0303565305703), ("quadratic", 1839.4486539684447),
                    ("quad-nomix", 2536.048160990245), ("quad-mix", 2848.164345162161)):
        c = oracle_cfg("3wrobot", critic_struct=O.CRITIC_IDS[cs])
        Q = O.critic(x, u, np.linspace(0.1, 1, c.dc), c)
        assert abs(Q - val) / val < TOL
    c2 = oracle_cfg("2tank", n_actor=4, pred_step_size=0.2)
    np.testing.assert_allclose(O.state_dyn(O.SYS_2TANK, [2.0, -2.0], [0.7], c2.pars),
                               [-0.059239130435, 0.196721311475], rtol=1e-10)
    assert abs(O.stage_obj([2.0, -2.0], [0.7], c2) - 85.49) < 1e-12
    assert abs(O.actor_cost(np.array([0.7, 0.1, 0.9, 0.4]), [2.0, -2.0], [2.0, -2.0], c2) - 327.50511436623594) < 1e-10


@pytest.mark.parametrize("name", SYSTEMS)
def test_F1_rhs(name):
    _, z = load_golden(f"F1_rhs_{name}")
    cfg = oracle_cfg(name)
    d = O.state_dyn(cfg.sys_id, z["state"], z["action"], cfg.pars)
    assert rel_err(d, z["state_dyn"]) <= TOL or np.allclose(d, z["state_dyn"], rtol=TOL, atol=1e-15)
    rhs, a = O.closed_loop_rhs(cfg.sys_id, z["state"], z["action"], cfg.pars, cfg.ctrl_bnds)
    np.testing.assert_allclose(rhs, z["closed_loop_rhs"], rtol=TOL, atol=1e-15)
    np.testing.assert_array_equal(a, z["clipped_action"])
    assert np.any(a != z["action"])  # the fixture does exercise the clip


@pytest.mark.parametrize("name", SYSTEMS)
def test_F2_stage_obj(name):
    _, z = load_golden(f"F2_stage_{name}")
    y, u = z["obs"], z["act"]
    cases = {
        "quad_diag": dict(R1=z["R1_diag"], target=None),
        "quad_full": dict(R1=z["R1_full"], target=None),
        "quad_nonsym": dict(R1=z["R1_nonsym"], target=None),
        "quad_diag_tgt": dict(R1=z["R1_diag"], target=z["target"]),
        "biquad_full_tgt": dict(R1=z["R1_full"], R2=z["R2_full"], target=z["target"],
                                stage_obj_struct=O.STAGE_BIQUADRATIC),
        "biquad_diag": dict(R1=z["R1_diag"], R2=np.diag(np.diag(z["R2_full"])), target=None,
                            stage_obj_struct=O.STAGE_BIQUADRATIC),
    }
    for tag, kw in cases.items():
        cfg = oracle_cfg(name, **kw)
        np.testing.assert_allclose(O.stage_obj(y, u, cfg), z[tag], rtol=1e-11, atol=1e-12, err_msg=tag)


@pytest.mark.parametrize("name", SYSTEMS)
def test_F3_critic(name):
    _, z = load_golden(f"F3_critic_{name}")
    for cs, cid in O.CRITIC_IDS.items():
        for ttag, tgt in (("", None), ("_tgt", z["target"])):
            cfg = oracle_cfg(name, critic_struct=cid, target=tgt)
            w = z[f"w_{cs}{ttag}"]
            assert w.shape[1] == cfg.dc
            np.testing.assert_allclose(O.critic(z["obs"], z["act"], w, cfg), z[f"Q_{cs}{ttag}"], rtol=1e-10,
                                       atol=1e-9, err_msg=cs + ttag)


@pytest.mark.parametrize("name", SYSTEMS)
def test_F4_actor_cost(name):
    meta, z = load_golden(f"F4_actor_cost_{name}")
    assert len(meta["cases"]) == 5 * 9
    for c in meta["cases"]:
        tag = c["tag"]
        cfg = oracle_cfg(name, n_actor=c["N"], mode=O.MODE_IDS[c["mode"]], gamma=c["gamma"],
                         critic_struct=O.CRITIC_IDS[c["critic_struct"]], pred_step_size=c["pred_step_size"])
        J = O.actor_cost(z[f"{tag}__action_sqn"], z[f"{tag}__obs"], z[f"{tag}__state_sys"], cfg,
                         w_critic=z[f"{tag}__w"])
        assert rel_err(J, z[f"{tag}__J"]) < 1e-11, tag


@pytest.mark.parametrize("name", SYSTEMS)
def test_F4b_actor_cost_on_the_production_shape(name):
    """F4b: 2 envs x 64 sequences per case, state_sys == obs, MPC gamma in {1, 0.95} and RQL x 4 critic structures
    (oracle/gen_f4b_fixture.py) - the shape k_actor_dma serves."""
    meta, z = load_golden(f"F4b_actor_cost_dma_{name}")
    assert len(meta["cases"]) >= 16
    for c in meta["cases"]:
        tag = c["tag"]
        cfg = oracle_cfg(name, n_actor=c["N"], mode=O.MODE_IDS[c["mode"]], gamma=c["gamma"],
                         critic_struct=O.CRITIC_IDS[c["critic_struct"]], pred_step_size=c["pred_step_size"])
        x = z[f"{tag}__state"].astype(np.float64)
        J = O.actor_cost(z[f"{tag}__action_sqn"].astype(np.float64), x[:, None, :], x[:, None, :], cfg,
                         w_critic=z[f"{tag}__w"].astype(np.float64)[:, None, :])
        assert J.shape == (2, 64) and rel_err(J, z[f"{tag}__J"]) < 1e-11, tag


@pytest.mark.parametrize("name", SYSTEMS)
def test_F5_critic_cost(name):
    meta, z = load_golden(f"F5_critic_cost_{name}")
    for c in meta["cases"]:
        tag = c["tag"]
        cfg = oracle_cfg(name, mode=O.MODE_RQL, gamma=c["gamma"], critic_struct=O.CRITIC_IDS[c["critic_struct"]],
                         n_critic=c["Ncritic"], buffer_size=c["buffer_size"])
        assert cfg.n_critic == c["Ncritic_eff"]
        Jc = O.critic_cost(z[f"{tag}__w"], z[f"{tag}__w_prev"], z[f"{tag}__obs_buf"], z[f"{tag}__act_buf"], cfg)
        assert rel_err(Jc, z[f"{tag}__Jc"]) < 1e-10, tag
        # the same cost through the (A, b) form used by the critic fit
        A, b = O.critic_td_system(z[f"{tag}__w_prev"], z[f"{tag}__obs_buf"], z[f"{tag}__act_buf"], cfg)
        r = np.einsum("brc,bc->br", A, z[f"{tag}__w"]) - b
        assert rel_err(0.5 * np.sum(r * r, axis=-1), z[f"{tag}__Jc"]) < 1e-9, tag


@pytest.mark.parametrize("name", SYSTEMS)
def test_F6_rk4_vs_reference_rk45(name):
    """Fixed-step RK4 (build-defined) against the reference's scipy-RK45 loop under a constant action.
    Tolerance: the north star's 1e-5 relative (SURVEY.md §8c 'Trajectory-level parity evidence')."""
    meta, z = load_golden(f"F6_rk45_const_{name}")
    cfg = oracle_cfg(name)
    t, y = z["t"], z["y"]
    u = np.array(meta["action"])
    x = y[0].copy()
    tt = 0.0
    h = meta["dt"] / 2.0  # reference grid: max_step = dt/2 (rcognita/simulator.py:150)
    worst = 0.0
    for i in range(1, len(t)):
        # integrate to the reference's (irregular) output instant with RK4 steps no longer than dt/2
        span = t[i] - tt
        n = max(1, int(np.ceil(span / h - 1e-12)))
        for _ in range(n):
            x = O.rk4_step(cfg.sys_id, x, u, cfg.pars, cfg.ctrl_bnds, span / n)
        tt = t[i]
        worst = max(worst, np.max(np.abs(x - y[i]) / np.maximum(np.abs(y[i]), 1.0)))
    assert worst < 1e-5, worst


def test_push_vec_and_argmin():
    buf = np.arange(12.0).reshape(1, 4, 3)
    out = O.push_vec(buf, np.array([[100.0, 101.0, 102.0]]))
    np.testing.assert_array_equal(out[0, :3], buf[0, 1:])
    np.testing.assert_array_equal(out[0, 3], [100, 101, 102])
    J = np.array([[3.0, 1.0, 1.0, np.nan], [np.nan, np.nan, 5.0, 5.0]])
    bj, bi = O.argmin_first(J)
    np.testing.assert_array_equal(bi, [1, 2])
    assert bi.dtype == np.int32
    np.testing.assert_array_equal(bj, [1.0, 5.0])


@pytest.mark.parametrize("name", SYSTEMS)
def test_F8_critic_fit_quality_vs_reference_slsqp(name):
    """Build-defined critic fit vs the reference's SLSQP result on the same TD stacks: parity is on the
    achieved cost Jc, not on w (SURVEY.md hard part 1).  Ours must never be worse than SLSQP's, never
    worse than the start point, and must respect the box."""
    meta, z = load_golden(f"F8_slsqp_critic_{name}")
    for c in meta["cases"]:
        cs = c["tag"]
        cfg = oracle_cfg(name, mode=O.MODE_RQL, gamma=c["gamma"], critic_struct=O.CRITIC_IDS[cs],
                         n_critic=c["Ncritic"], buffer_size=c["buffer_size"])
        w = O.critic_fit(cfg, z[f"{cs}__w_prev"], z[f"{cs}__obs_buf"], z[f"{cs}__act_buf"])
        lo, hi = O.critic_bounds(cfg.critic_struct, cfg.dc)
        assert np.all(w >= lo) and np.all(w <= hi)
        Jc = O.critic_cost(w, z[f"{cs}__w_prev"], z[f"{cs}__obs_buf"], z[f"{cs}__act_buf"], cfg)
        J0, Js = z[f"{cs}__Jc_init"], z[f"{cs}__Jc_fit"]
        assert np.all(Jc <= J0 * (1 + 1e-12)), cs
        assert np.all(Jc <= Js * (1 + 1e-6) + 1e-7 * J0), (cs, np.max(Jc / J0 - Js / J0))
        if name == "2tank" and cs != "quadratic":
            # where SLSQP converges and no bound is active both land on the same point
            assert np.median(np.max(np.abs(w - z[f"{cs}__w_fit"]), axis=1)) < 1e-3
