"""Parity of the HIP path (through the C ABI) against the CPU oracle and the reference's golden
vectors.  Needs an MI355X: every test is marked ``gpu``.

Tolerances: f64 build 1e-11 relative (same arithmetic, different libm / FMA contraction);
f32 build 1e-5 relative - the tolerance BASELINE.json's north_star states - measured as
|hip - ref| / max(|ref|, 1).  Integer outputs (best_idx, step_idx, episode_idx) are bit-exact.
"""
import numpy as np
import pytest

from oracle import rcg_oracle as O
from tests.conftest import load_golden
from tests.helpers import (PRESETS, SYSTEMS, TOL, assert_kernel, both, rand_actions, rand_states, rel_err_norm)

pytestmark = pytest.mark.gpu

DTYPES = ["f64", "f32"]


def _close(a, b, dtype, floor=1.0, scale=1.0, msg=""):
    e = rel_err_norm(a, b, floor)
    assert e <= TOL[dtype] * scale, f"{msg} rel err {e:.3e} > {TOL[dtype] * scale:.1e} ({dtype})"


# ------------------------------------------------------------------------------------------------
# reference golden vectors through the HIP path
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
def test_known_answers(dtype):
    _, k = load_golden("KAT")
    x = np.array([[5, 5, -3 * np.pi / 4, 0.3, -0.2]])
    u = np.array([[50.0, -20.0]])
    eng, cfg = both("3wrobot", 1, dtype, n_actor=5, gamma=0.9, pred_step_size=0.02)
    d, _ = eng.rhs(x, u, clip=False)
    _close(d[0], k["kat1"], dtype, msg="KAT1")
    d, a = eng.rhs(x, np.array([[400.0, -150.0]]), clip=True)
    _close(d[0], k["kat2_rhs"], dtype, msg="KAT2")
    np.testing.assert_array_equal(a[0], [300, -100])
    _close(eng.stage_obj(x, u), 280.55165247561274, dtype, msg="KAT3")
    aseq = np.array([[50, -20], [40, -10], [30, 0], [20, 10], [10, 20]], dtype=float)
    w = 0.5 * np.arange(1, 8)
    for mode in ("MPC", "RQL", "SQL"):
        e2, _ = both("3wrobot", 1, dtype, n_actor=5, gamma=0.9, pred_step_size=0.02, mode=O.MODE_IDS[mode],
                     critic_struct=O.CRITIC_QUAD_NOMIX, buffer_size=4)
        J = e2.actor_cost(aseq[None, None], obs=x + 0.01, state_sys=x, w=w[None])
        _close(J[0, 0], float(k[f"kat4_{mode}"]), dtype, msg=f"KAT4 {mode}")
    for cs in ("quad-lin", "quadratic", "quad-nomix", "quad-mix"):
        e3, c3 = both("3wrobot", 1, dtype, critic_struct=O.CRITIC_IDS[cs])
        _close(e3.critic(x, u, np.linspace(0.1, 1, c3.dc)[None]), float(k[f"kat5_{cs}"]), dtype, msg=f"KAT5 {cs}")
    e4, _ = both("2tank", 1, dtype, n_actor=4, pred_step_size=0.2)
    d, _ = e4.rhs(np.array([[2.0, -2.0]]), np.array([[0.7]]))
    _close(d[0], k["kat8"], dtype, floor=0.05, msg="KAT8")
    _close(e4.stage_obj([[2.0, -2.0]], [[0.7]]), 85.49, dtype, msg="KAT9")
    J = e4.actor_cost(np.array([0.7, 0.1, 0.9, 0.4]).reshape(1, 1, 4, 1), obs=[[2.0, -2.0]], state_sys=[[2.0, -2.0]])
    _close(J[0, 0], float(k["kat10"]), dtype, msg="KAT10")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name", SYSTEMS)
def test_F1_rhs_golden(name, dtype):
    _, z = load_golden(f"F1_rhs_{name}")
    eng, _ = both(name, 1, dtype)
    d, _ = eng.rhs(z["state"], z["action"], clip=False)
    _close(d, z["state_dyn"], dtype, msg="state_dyn")
    d, a = eng.rhs(z["state"], z["action"], clip=True)
    _close(d, z["closed_loop_rhs"], dtype, msg="closed_loop_rhs")
    _close(a, z["clipped_action"], dtype, msg="clip")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name", SYSTEMS)
def test_F2_stage_obj_golden(name, dtype):
    _, z = load_golden(f"F2_stage_{name}")
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
        eng, _ = both(name, 1, dtype, **kw)
        # non-symmetric / full matrices mix signs: compare against the magnitude of the terms
        floor = float(np.max(np.abs(z[tag]))) if "full" in tag or "nonsym" in tag else 1.0
        _close(eng.stage_obj(z["obs"], z["act"]), z[tag], dtype, floor=floor, msg=tag)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name", SYSTEMS)
def test_F3_critic_golden(name, dtype):
    _, z = load_golden(f"F3_critic_{name}")
    for cs, cid in O.CRITIC_IDS.items():
        for ttag, tgt in (("", None), ("_tgt", z["target"])):
            eng, _ = both(name, 1, dtype, critic_struct=cid, target=tgt)
            ref = z[f"Q_{cs}{ttag}"]
            _close(eng.critic(z["obs"], z["act"], z[f"w_{cs}{ttag}"]), ref, dtype, floor=float(np.max(np.abs(ref))),
                   msg=cs + ttag)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name", SYSTEMS)
def test_F4_actor_cost_golden(name, dtype):
    """Every (N, mode, critic_struct) case of the reference's _actor_cost, state_sys != obs."""
    meta, z = load_golden(f"F4_actor_cost_{name}")
    for c in meta["cases"]:
        tag = c["tag"]
        n = z[f"{tag}__J"].shape[0]
        eng, _ = both(name, n, dtype, n_actor=c["N"], mode=O.MODE_IDS[c["mode"]], gamma=c["gamma"],
                      critic_struct=O.CRITIC_IDS[c["critic_struct"]], pred_step_size=c["pred_step_size"],
                      buffer_size=4)
        J = eng.actor_cost(z[f"{tag}__action_sqn"][:, None], obs=z[f"{tag}__obs"], state_sys=z[f"{tag}__state_sys"],
                           w=z[f"{tag}__w"])
        _close(J[:, 0], z[f"{tag}__J"], dtype, msg=tag)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name", SYSTEMS)
def test_F4b_reference_actor_cost_through_the_production_kernel(name, dtype):
    """Numbers the REFERENCE produced (fixture F4b: its _actor_cost, controllers.py:1273-1328, state_sys == obs, 64
    sequences per env) through the streamed production kernel: f32 K = 64, rows of N*du <= 40 reals, diagonal R1 ->
    k_actor_dma - the gamma == 1 per-component instance (G1), the discounted instance, and the critic instances (RQL,
    four structures).  Both the operator (J of every sequence, staged in LDS) and the argmin.  Inputs are stored as
    float32, so the kernel reads exactly what the reference evaluated."""
    meta, z = load_golden(f"F4b_actor_cost_dma_{name}")
    for c in meta["cases"]:
        tag = c["tag"]
        x, aseq, w, J_ref = z[f"{tag}__state"], z[f"{tag}__action_sqn"], z[f"{tag}__w"], z[f"{tag}__J"]
        B, K = J_ref.shape
        eng, _ = both(name, B, dtype, n_actor=c["N"], mode=O.MODE_IDS[c["mode"]], gamma=c["gamma"],
                      critic_struct=O.CRITIC_IDS[c["critic_struct"]], pred_step_size=c["pred_step_size"], buffer_size=4)
        from rcognita_amd import _native as N

        eng.set_state(x)
        if c["mode"] != "MPC":
            eng.set_field(N.FIELD_W_CRITIC, w)
        cand = eng.to_device(aseq.astype(eng.real))  # device-resident [B][K][N][du]: the streamed path
        J = eng.actor_cost(cand)  # obs = state_sys = the handle's STATE
        # every mode in both element types (f64 critic modes of the robots: weights parked in LDS, round 3)
        variant = {"MPC": N.DMA_MPC_G1 if c["gamma"] == 1.0 else N.DMA_MPC, "RQL": N.DMA_RQL_0 + N.CRITIC_IDS[c["critic_struct"]],
                   "SQL": N.DMA_SQL_0 + N.CRITIC_IDS[c["critic_struct"]]}[c["mode"]]
        assert_kernel(eng, "k_actor_dma", variant)
        scale = np.max(np.abs(J_ref), axis=1, keepdims=True)
        err = float(np.max(np.abs(J - J_ref) / scale))
        assert err <= TOL[dtype], f"{tag}: J rel err {err:.3e}"
        a, bj, bi = eng.actor_argmin(cand)
        ref_i = np.argmin(J_ref, axis=1)
        for e in range(B):  # a float32 near-tie may take the runner-up: its reference cost within rounding of the best
            assert bi[e] == ref_i[e] or abs(J_ref[e, bi[e]] - J_ref[e, ref_i[e]]) <= 4 * TOL[dtype] * scale[e, 0], tag
            np.testing.assert_array_equal(a[e], aseq[e, bi[e], 0, :].astype(eng.real))
            assert abs(bj[e] - J_ref[e, bi[e]]) <= TOL[dtype] * scale[e, 0], tag


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name", SYSTEMS)
def test_F5_critic_cost_golden(name, dtype):
    meta, z = load_golden(f"F5_critic_cost_{name}")
    from rcognita_amd import _native as N

    for c in meta["cases"]:
        tag = c["tag"]
        n = z[f"{tag}__Jc"].shape[0]
        eng, cfg = both(name, n, dtype, mode=O.MODE_RQL, gamma=c["gamma"],
                        critic_struct=O.CRITIC_IDS[c["critic_struct"]], n_critic=c["Ncritic"],
                        buffer_size=c["buffer_size"])
        eng.set_field(N.FIELD_OBS_BUF, z[f"{tag}__obs_buf"])
        eng.set_field(N.FIELD_ACT_BUF, z[f"{tag}__act_buf"])
        eng.set_field(N.FIELD_W_PREV, z[f"{tag}__w_prev"])
        np.testing.assert_allclose(eng.get_field(N.FIELD_OBS_BUF), z[f"{tag}__obs_buf"].astype(eng.real))
        Jc = eng.critic_cost(z[f"{tag}__w"])
        # 1/2 e^2 with e a difference of large terms: f32 tolerance applies to e, i.e. ~2x on Jc
        _close(Jc, z[f"{tag}__Jc"], dtype, floor=float(np.max(np.abs(z[f"{tag}__Jc"]))), scale=4.0, msg=tag)


@pytest.mark.parametrize("name", SYSTEMS)
def test_F6_hip_rk4_vs_reference_rk45(name):
    """HIP fixed-step RK4 (f64) vs the reference's scipy-RK45 loop under a constant action: <= 1e-5."""
    from rcognita_amd import _native as N

    meta, z = load_golden(f"F6_rk45_const_{name}")
    t, y = z["t"], z["y"]
    h = meta["dt"] / 2.0
    # sample the reference trajectory at its regular dt/2 strides (after the start-up transient)
    eng, cfg = both(name, 1, "f64", dt_sim=h)
    eng.set_state(y[0][None])
    eng.set_field(N.FIELD_ACTION, np.array(meta["action"])[None])
    # the reference grid is irregular at the start (first_step = 1e-6); integrate the oracle on the
    # exact grid (test_oracle_golden) and here compare HIP vs oracle on a regular grid + final value
    x_or = y[0].copy()
    nsteps = int(round(t[-1] / h))
    for _ in range(nsteps):
        x_or = O.rk4_step(cfg.sys_id, x_or, np.array(meta["action"]), cfg.pars, cfg.ctrl_bnds, h)
    eng.sim_step(nsteps)
    x_hip = eng.get_state()[0]
    assert rel_err_norm(x_hip, x_or) < 1e-10
    # and against the reference's own end point, advanced by the (tiny) remaining time difference
    rem = t[-1] - nsteps * h
    x_end = O.rk4_step(cfg.sys_id, x_hip, np.array(meta["action"]), cfg.pars, cfg.ctrl_bnds, rem)
    assert rel_err_norm(x_end, y[-1]) < 1e-5


# ------------------------------------------------------------------------------------------------
# HIP vs oracle on seeded random inputs: tile shapes, ragged sizes, ties, NaN
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name,N", [("3wrobot", 10), ("3wrobot", 7), ("3wrobotNI", 15), ("2tank", 20), ("2tank", 5),
                                    ("3wrobot", 5), ("3wrobotNI", 3), ("3wrobot", 16), ("2tank", 32), ("2tank", 1),
                                    ("3wrobot", 20), ("2tank", 40), ("3wrobotNI", 21)])  # preset defaults, the largest DMA
#                                                                   rows (160 bytes: 40 floats / 20 doubles), one beyond them
@pytest.mark.parametrize("K", [1, 3, 16, 33, 64, 100, 256])
def test_actor_cost_and_argmin_vs_oracle(name, N, K, dtype):
    """Streamed candidates: J vs oracle; argmin bit-exact vs numpy on the SAME J; winner's first action."""
    rng = np.random.default_rng(1000 + 7 * K + N)
    B = 37 if K < 64 else 5  # not a multiple of the envs-per-wave group: exercises ragged tails
    eng, cfg = both(name, B, dtype, n_actor=N, gamma=0.97)
    x = rand_states(rng, name, B)
    cand = rand_actions(rng, name, (B, K, N), overshoot=1.2)
    eng.set_state(x)
    J = eng.actor_cost(cand)
    J_or = O.actor_cost(cand, x[:, None, :], x[:, None, :], cfg)
    _close(J, J_or, dtype, msg="J")
    act, bj, bi = eng.actor_argmin(cand)
    # the argmin must be exactly numpy's first-occurrence argmin of the J the device itself computed
    np.testing.assert_array_equal(bi, np.argmin(J, axis=1).astype(np.int32))
    np.testing.assert_array_equal(bj, J[np.arange(B), bi])
    np.testing.assert_array_equal(act, cand[np.arange(B), bi, 0, :].astype(eng.real))
    if dtype == "f64":
        _, bi_or = O.argmin_first(J_or)
        np.testing.assert_array_equal(bi, bi_or)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mode,cs", [("MPC", "quad-nomix"), ("RQL", "quad-mix"), ("SQL", "quadratic")])
def test_tank_without_a_target_on_the_production_kernel(mode, cs, dtype):
    """Sys2Tank's production instances subtract the observation target; a handle WITHOUT one (the reference's
    observation_target == []) runs on them with a target of zeros - y - 0 = y exactly - and must equal the oracle with
    no target (the critic's quad-mix features use the raw observation either way, controllers.py:1212)."""
    from rcognita_amd import _native as N

    rng = np.random.default_rng(77)
    B, K, Nh = 6, 128, 9
    eng, cfg = both("2tank", B, dtype, n_actor=Nh, target=None, mode=O.MODE_IDS[mode], critic_struct=O.CRITIC_IDS[cs],
                    gamma=0.97, buffer_size=4)
    assert cfg.target is None
    x = rand_states(rng, "2tank", B)
    cand = rand_actions(rng, "2tank", (B, K, Nh))
    w = rng.uniform(0.1, 2.0, (B, cfg.dc))
    eng.set_state(x)
    if mode != "MPC":
        eng.set_field(N.FIELD_W_CRITIC, w)
    J = eng.actor_cost(eng.to_device(cand.astype(eng.real)))
    assert_kernel(eng, "k_actor_dma")
    J_or = O.actor_cost(cand, x[:, None, :], x[:, None, :], cfg, w_critic=w[:, None, :] if mode != "MPC" else None)
    _close(J, J_or, dtype, msg=f"no-target tank {mode}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("K", [131, 130, 128, 256, 132, 196])
def test_argmin_ties_and_nan(dtype, K):
    """Lower index wins ties; NaN costs count as +inf; an all-NaN env returns index 0.  K = 131 in f32: the generic kernel
    (rows of 40 bytes: 131 of them are not a whole number of 16-byte pieces); K = 128, 256: the production kernel's DPP (f32)
    / shuffle (f64) wave argmin; K = 130, 132, 196 (and 131 in f64: 80-byte rows): the production kernel with a ragged last
    tile (2 / 4 rows: masked direct-to-LDS loads, the lanes without a row sit out the argmin - including the LAST env of the
    tensor, whose tile ends at the allocation's end)."""
    name, N, B = "3wrobot", 5, 4
    rng = np.random.default_rng(5)
    eng, cfg = both(name, B, dtype, n_actor=N)
    x = rand_states(rng, name, B)
    cand = rand_actions(rng, name, (B, K, N))
    cand[0, 77] = cand[0, 3]  # exact duplicates -> equal J -> index 3 must win if it is the minimum
    cand[0, 3] = 0.0
    cand[0, 77] = 0.0
    cand[0, 100] = 0.0
    cand[1, 5, 2, 0] = np.nan  # NaN candidate is never selected
    cand[2] = np.nan  # every candidate NaN
    eng.set_state(x)
    J = eng.actor_cost(cand)
    assert_kernel(eng, "k_actor_dma" if (K * N * 2 * eng.real().itemsize) % 16 == 0 else "k_actor")
    act, bj, bi = eng.actor_argmin(cand)
    Jc = np.where(np.isnan(J), np.inf, J)
    np.testing.assert_array_equal(bi, np.argmin(Jc, axis=1).astype(np.int32))
    assert np.isnan(J[1, 5]) and bi[1] != 5
    assert bi[2] == 0 and np.isinf(bj[2])
    assert J[0, 3] == J[0, 77] == J[0, 100]
    # the last env of the tensor (its ragged tile must not read past the rows): a unique winner in its LAST row
    cand2 = cand.copy()
    cand2[2] = cand[3]
    cand2[3, K - 1] = 0.0
    act, bj, bi = eng.actor_argmin(cand2)
    J2 = eng.actor_cost(cand2)
    np.testing.assert_array_equal(bi, np.argmin(np.where(np.isnan(J2), np.inf, J2), axis=1).astype(np.int32))
    np.testing.assert_array_equal(bj, J2[np.arange(B), bi])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mode", ["MPC", "RQL", "SQL"])
@pytest.mark.parametrize("K", [33, 34, 35, 36, 37, 38, 39])
def test_one_ragged_tile_below_64(K, mode, dtype):
    """33 <= K <= 63 is ONE ragged tile of k_actor_dma (K lanes of 64 hold a row; profiles/r04_ab_min_k.txt: 1.5 - 3.1 x
    k_actor at K = 33 .. 39), for every K whose env slab K * R * esz is a whole number of 16-byte pieces; the others stay on
    k_actor.  Costs against the oracle, the argmin exact, the winner in the LAST row of the LAST env (the tile that ends at
    the allocation's end), NaN rows never selected."""
    from rcognita_amd import _native as N

    name, Nh, B = "3wrobot", 5, 67
    rng = np.random.default_rng(330 + K)
    kw = dict(n_actor=Nh, mode=O.MODE_IDS[mode])
    if mode != "MPC":
        kw.update(critic_struct=O.CRITIC_IDS["quad-nomix"], n_critic=3, buffer_size=5, gamma=0.95)
    eng, cfg = both(name, B, dtype, **kw)
    x = rand_states(rng, name, B)
    cand = rand_actions(rng, name, (B, K, Nh))
    cand[B - 1, K - 1] = 0.0
    cand[1, 5, 2, 0] = np.nan
    w = None
    eng.set_state(x)
    if mode != "MPC":
        w = rng.uniform(0.1, 2.0, (B, cfg.dc))
        eng.set_field(N.FIELD_W_CRITIC, w)
    J = eng.actor_cost(cand)
    assert_kernel(eng, "k_actor_dma" if (K * Nh * 2 * eng.real().itemsize) % 16 == 0 else "k_actor")
    J_or = O.actor_cost(cand, x[:, None, :], x[:, None, :], cfg, w_critic=None if w is None else w[:, None, :])
    ok = ~np.isnan(J_or)
    assert np.array_equal(np.isnan(J), ~ok)
    scale = np.max(np.abs(np.where(ok, J_or, 0)), axis=1, keepdims=True)
    assert float(np.max(np.abs(np.where(ok, J - J_or, 0)) / scale)) <= TOL[dtype]
    act, bj, bi = eng.actor_argmin(cand)
    Jc = np.where(np.isnan(J), np.inf, J)
    np.testing.assert_array_equal(bi, np.argmin(Jc, axis=1).astype(np.int32))
    np.testing.assert_array_equal(bj, Jc[np.arange(B), bi])
    np.testing.assert_array_equal(act, cand[np.arange(B), bi, 0, :].astype(eng.real))
    assert bi[1] != 5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name", SYSTEMS)
@pytest.mark.parametrize("K", [1, 9, 64, 256, 1024])
def test_generated_grid_vs_oracle(name, K, dtype):
    """cand == NULL: the generated level grid equals oracle.grid_candidates evaluated by the oracle.  K = 256, 1024 on
    the robots: four tiles per lane rolled out together with the shared heading sub-trajectory (rollout_mpc_gen_multi)."""
    if PRESETS[name]["sys_id"] != O.SYS_2TANK and int(np.sqrt(K)) ** 2 != K:
        pytest.skip("du = 2 needs a square K")
    rng = np.random.default_rng(77 + K)
    B, N = 11, 6
    eng, cfg = both(name, B, dtype, n_actor=N)
    x = rand_states(rng, name, B)
    eng.set_state(x)
    act, bj, bi = eng.actor_argmin(None, K=K)
    grid = O.grid_candidates(cfg, K)
    J_or = O.actor_cost(grid[None], x[:, None, :], x[:, None, :], cfg)
    bj_or, bi_or = O.argmin_first(J_or)
    _close(bj, bj_or, dtype, msg="best_J")
    if dtype == "f64":
        np.testing.assert_array_equal(bi, bi_or)
        np.testing.assert_allclose(act, grid[bi_or, 0, :], rtol=1e-13)
    else:  # f32 may pick another candidate only if the oracle sees it as a near-tie
        sel = J_or[np.arange(B), bi]
        assert np.all(sel <= bj_or + 1e-5 * np.maximum(np.abs(bj_or), 1.0))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name", SYSTEMS)
def test_sim_step_vs_oracle(name, dtype):
    rng = np.random.default_rng(3)
    from rcognita_amd import _native as N

    B = 300
    eng, cfg = both(name, B, dtype)
    x = rand_states(rng, name, B)
    u = rand_actions(rng, name, (B,), overshoot=1.5)  # part of them gets clipped
    eng.set_state(x)
    eng.set_field(N.FIELD_ACTION, u)
    eng.sim_step(3)
    xo, prev = x, x
    for _ in range(3):
        prev = xo
        xo = O.rk4_step(cfg.sys_id, xo, u, cfg.pars, cfg.ctrl_bnds, cfg.dt_sim)
    _close(eng.get_state(), xo, dtype, msg="state")
    _close(eng.get_field(N.FIELD_STATE_PREV), prev, dtype, msg="state_prev")
    assert not eng.get_field(N.FIELD_STATUS).any()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name,K,streamed", [("3wrobot", 64, True), ("3wrobot", 16, False), ("3wrobotNI", 100, True),
                                             ("2tank", 32, False), ("2tank", 5, True)])
@pytest.mark.parametrize("ref_lag", [False, True])
def test_control_tick_closed_loop_vs_oracle(name, K, streamed, ref_lag, dtype):
    """T ticks of the fused loop body against the oracle's control_tick: states, actions, best_J, accum, integer
    counters - every tick of both builds.  f64: the two runs are simply compared (exact argmin agreement, 1e-10);
    f32: every tick is checked as a map from the same inputs at 1e-5, a float32 near-tie may take the runner-up
    (its oracle cost within 4e-5 of the best) and the oracle follows it (oracle/parity.py)."""
    from oracle import parity as PAR
    from rcognita_amd import _native as N

    rng = np.random.default_rng(11 + K)
    B, Nh, S = 13, 6, 2
    T = 12 if dtype == "f64" else 8
    eng, cfg = both(name, B, dtype, n_actor=Nh, substeps_per_tick=S, ref_lag=ref_lag, gamma=0.99)
    x0 = rand_states(rng, name, B).astype(eng.real)
    eng.set_state(x0)
    env = O.new_batch(cfg, x0.astype(np.float64))
    cand = rand_actions(rng, name, (B, K, Nh)).astype(eng.real) if streamed else O.grid_candidates(cfg, K)
    dcand = eng.to_device(cand) if streamed else None
    rep = PAR.TickReport()
    for t in range(T):
        eng.control_tick(dcand if streamed else None, K=K)
        env = PAR.check_tick(cfg, env, np.asarray(cand, dtype=np.float64), PAR.device_fields(eng, N, with_prev=True),
                             tol=1e-10 if dtype == "f64" else TOL["f32"], report=rep, resync=dtype == "f32",
                             what=f"{name} K={K} t={t}")
    assert rep.ticks == T
    if dtype == "f64":
        assert rep.ties == 0


@pytest.mark.parametrize("name,K", [("3wrobot", 64), ("2tank", 32)])
def test_long_closed_loop_f64_vs_oracle(name, K):
    """400 consecutive ticks (with two episode resets on the way) of the float64 build against the oracle: the argmin
    sequence identical, state / accum / returns to 1e-9, int32 counters exact - no drift, no bookkeeping slip."""
    from rcognita_amd import _native as N

    rng = np.random.default_rng(77)
    B, Nh, T = 24, 10, 400
    eng, cfg = both(name, B, "f64", n_actor=Nh)
    x0 = rand_states(rng, name, B)
    eng.set_state(x0)
    env = O.new_batch(cfg, x0)
    grid = O.grid_candidates(cfg, K)
    for t in range(T):
        eng.control_tick(None, K=K)
        O.control_tick(cfg, env, grid)
        if t % 50 == 49:
            np.testing.assert_array_equal(eng.get_field(N.FIELD_BEST_IDX), env.best_idx)
            assert rel_err_norm(eng.get_state(), env.state) < 1e-9, t
        if t in (149, 299):  # episode boundary: returns := accum, state := state_init, counters
            ret = env.accum.copy()
            eng.episode_reset()
            env = O.new_batch(cfg, x0)
            env.episode_idx = np.full(B, 1 if t == 149 else 2, dtype=np.int32)
            np.testing.assert_allclose(eng.get_field(N.FIELD_RETURNS), ret, rtol=1e-9)
            np.testing.assert_array_equal(eng.get_field(N.FIELD_EPISODE_IDX), env.episode_idx)
    np.testing.assert_array_equal(eng.get_field(N.FIELD_STEP_IDX), np.full(B, 100, np.int32))
    assert rel_err_norm(eng.get_state(), env.state) < 1e-9
    assert rel_err_norm(eng.get_field(N.FIELD_ACCUM), env.accum, floor=float(np.max(np.abs(env.accum)))) < 1e-9


def test_accum_every_substep_flag():
    from rcognita_amd import _native as N

    rng = np.random.default_rng(2)
    B, K, Nh, S = 9, 16, 4, 3
    eng, cfg = both("2tank", B, "f64", n_actor=Nh, substeps_per_tick=S, accum_every_substep=True)
    x0 = rand_states(rng, "2tank", B)
    eng.set_state(x0)
    env = O.new_batch(cfg, x0)
    for _ in range(5):
        eng.control_tick(None, K=K)
        O.control_tick(cfg, env, O.grid_candidates(cfg, K))
    np.testing.assert_allclose(eng.get_field(N.FIELD_ACCUM), env.accum, rtol=1e-11)


@pytest.mark.parametrize("dtype", DTYPES)
def test_per_env_parameters(dtype):
    """Heterogeneous (m, I) per env: coalesced [np][B] parameter loads."""
    from rcognita_amd import _native as N

    rng = np.random.default_rng(8)
    B, K, Nh = 70, 64, 5
    eng, cfg = both("3wrobot", B, dtype, n_actor=Nh, per_env_pars=True)
    pars = np.stack([rng.uniform(5, 20, B), rng.uniform(0.5, 2, B)], axis=-1)
    eng.set_field(N.FIELD_PARS, pars)
    x0 = rand_states(rng, "3wrobot", B)
    eng.set_state(x0)
    env = O.new_batch(cfg, x0, pars=pars)
    cand = rand_actions(rng, "3wrobot", (B, K, Nh))
    J = eng.actor_cost(cand)
    J_or = O.actor_cost(cand, x0[:, None], x0[:, None], cfg, pars=pars[:, None, :])
    _close(J, J_or, dtype, msg="J per-env pars")
    eng.control_tick(cand)
    O.control_tick(cfg, env, cand)
    _close(eng.get_state(), env.state, dtype, msg="state per-env pars")


def test_nonfinite_state_freezes_env_and_is_reported():
    from rcognita_amd import _native as N

    B = 6
    eng, cfg = both("2tank", B, "f32", dt_sim=50.0)  # huge step: K3*h2^2 blows up within a few steps
    x0 = np.tile(np.array([[1.0, 1.0]]), (B, 1))
    x0[2] = [1.0, 1e30]
    eng.set_state(x0)
    eng.set_field(N.FIELD_ACTION, np.full((B, 1), 0.5))
    for _ in range(3):
        eng.sim_step(1)
    st = eng.get_field(N.FIELD_STATUS)
    assert st[2] & 1
    assert np.all(np.isfinite(eng.get_state()))
    summ, _ = eng.episode_stats(from_accum=True)
    assert summ["n_failed"] >= 1 and summ["count"] == B


@pytest.mark.parametrize("dtype", DTYPES)
def test_episode_reset_and_stats(dtype):
    from rcognita_amd import _native as N

    rng = np.random.default_rng(4)
    B, K = 1000, 16
    eng, cfg = both("3wrobotNI", B, dtype, n_actor=3)
    x0 = rand_states(rng, "3wrobotNI", B)
    eng.set_state(x0)
    for _ in range(3):
        eng.control_tick(None, K=K)
    acc = eng.get_field(N.FIELD_ACCUM).astype(np.float64)
    np.testing.assert_array_equal(eng.get_field(N.FIELD_STEP_IDX), np.full(B, 3, np.int32))
    eng.episode_reset()
    summ, ret = eng.episode_stats(want_returns=True)
    np.testing.assert_array_equal(ret.astype(np.float64), acc)
    assert summ["count"] == B and summ["n_failed"] == 0
    np.testing.assert_allclose(summ["sum"], acc.sum(), rtol=1e-12)
    np.testing.assert_allclose(summ["sumsq"], (acc * acc).sum(), rtol=1e-12)
    assert summ["min"] == acc.min() and summ["max"] == acc.max()
    np.testing.assert_array_equal(eng.get_field(N.FIELD_STEP_IDX), np.zeros(B, np.int32))
    np.testing.assert_array_equal(eng.get_field(N.FIELD_EPISODE_IDX), np.ones(B, np.int32))
    np.testing.assert_array_equal(eng.get_state(), x0.astype(eng.real))
    np.testing.assert_array_equal(eng.get_field(N.FIELD_ACCUM), np.zeros(B, eng.real))
    a0 = np.array(PRESETS["3wrobotNI"]["bnds"], dtype=float)[:, 0] / 10
    np.testing.assert_allclose(eng.get_field(N.FIELD_ACTION), np.broadcast_to(a0, (B, 2)))


# ------------------------------------------------------------------------------------------------
# full BASELINE size: size-independent properties
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_full_size_C2_properties(dtype):
    """BASELINE configs[1]: Sys3WRobot, B = 65536, Nactor = 10, K = 256 streamed candidates - in float64 (the reference's width
    and, since round 6, the bench headline: 2.7 GB of candidates) and in float32.
    Properties: (a) best_J == min_k J and best_idx == first argmin of the device's own J for every env;
    (b) permuting an env's candidates permutes its J and moves the argmin accordingly; (c) an env's
    result does not depend on its position in the batch (env e evaluated alone == in the batch);
    (d) a sample of envs agrees with the oracle to 1e-5."""
    from rcognita_amd import _native as N

    B, K, Nh = 65536, 256, 10
    rng = np.random.default_rng(1234)
    eng, cfg = both("3wrobot", B, dtype, n_actor=Nh)
    x = np.stack([rng.uniform(-10, 10, B), rng.uniform(-10, 10, B), rng.uniform(-np.pi, np.pi, B),
                  rng.uniform(-1, 1, B), rng.uniform(-1, 1, B)], axis=-1).astype(eng.real)
    eng.set_state(x)
    lo, hi = cfg.ctrl_bnds[:, 0].astype(eng.real), cfg.ctrl_bnds[:, 1].astype(eng.real)
    cand = (lo + (hi - lo) * rng.random((B, K, Nh, 2), dtype=eng.real)).astype(eng.real)
    dcand = eng.to_device(cand)
    J = eng.actor_cost(dcand)
    act, bj, bi = eng.actor_argmin(dcand)
    np.testing.assert_array_equal(bi, np.argmin(J, axis=1).astype(np.int32))  # (a)
    np.testing.assert_array_equal(bj, J.min(axis=1))
    np.testing.assert_array_equal(act, cand[np.arange(B), bi, 0, :])
    # (d) oracle sample
    sel = rng.choice(B, 64, replace=False)
    J_or = O.actor_cost(cand[sel].astype(np.float64), x[sel, None, :].astype(np.float64), x[sel, None, :].astype(np.float64), cfg)
    assert rel_err_norm(J[sel], J_or) < TOL[dtype]
    assert_kernel(eng, "k_actor_dma", 0)
    # (b) permutation of candidates
    perm = rng.permutation(K)
    cand_p = np.ascontiguousarray(cand[:, perm])
    dcand_p = eng.to_device(cand_p)
    Jp = eng.actor_cost(dcand_p)
    np.testing.assert_array_equal(Jp, J[:, perm])
    # (c) position independence: first 8 envs alone in a small batch
    e2, _ = both("3wrobot", 8, dtype, n_actor=Nh)
    e2.set_state(x[:8])
    np.testing.assert_array_equal(e2.actor_cost(cand[:8]), J[:8])
    # one full tick: counters exact, state finite
    eng.control_tick(dcand)
    np.testing.assert_array_equal(eng.get_field(N.FIELD_STEP_IDX), np.ones(B, np.int32))
    assert np.all(np.isfinite(eng.get_state()))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name,N", [("3wrobot", 32), ("2tank", 64), ("3wrobotNI", 1), ("2tank", 1),
                                    ("3wrobot", 33), ("2tank", 65), ("3wrobot", 40), ("3wrobotNI", 100), ("2tank", 300)])
def test_horizon_limits(name, N, dtype):
    """Size edges of the candidate row.  The reference's horizon is unbounded (controllers.py:965): rows of up to
    RCG_MAX_ROW = 64 reals (Nactor = 32 for the robots, 64 for the tanks) are staged in LDS tiles, ONE STEP BEYOND and far
    beyond (Nactor = 40 / 100 / 300: rows of 80 / 200 / 300 reals) the generic kernel walks them straight from HBM (k_actor,
    variant bit 4) - round 5 refused them; and the shortest (Nactor = 1: no rollout at all, J = rho(y_0, u_0)).  Streamed
    (K = 64 and a ragged K = 7) and generated candidates against the oracle, the operator, the argmin and a closed-loop tick."""
    from oracle import parity as PAR
    from rcognita_amd import _native as Nn

    rng = np.random.default_rng(N)
    B = 6
    eng, cfg = both(name, B, dtype, n_actor=N, gamma=0.9)
    x = rand_states(rng, name, B).astype(eng.real)
    eng.set_state(x)
    x64 = x.astype(np.float64)
    long_row = N * cfg.du > 64
    # (a long rollout from a random start amplifies the last bit of every step: the f32 tolerance applies to horizons the
    # presets use; beyond, the cost is compared at the conditioning of the rollout itself - the oracle re-run with float32-
    # rounded inputs perturbed by one ulp moves by up to 1e-4 of the cost at Nactor = 100)
    tol = TOL[dtype] * (1.0 if (dtype == "f64" or N <= 64) else 30.0)
    for K in (64, 7):
        cand = rand_actions(rng, name, (B, K, N), overshoot=0.3 if N > 64 else 1.0).astype(eng.real)
        J = eng.actor_cost(eng.to_device(cand))
        ll = eng.last_launch(Nn.KERNEL_ACTOR)
        assert bool(ll["variant"] & 16) == long_row and (not long_row or ll["kernel"] == "k_actor"), ll
        J_or = O.actor_cost(cand.astype(np.float64), x64[:, None, :], x64[:, None, :], cfg)
        scale = np.maximum(np.max(np.abs(J_or), axis=1, keepdims=True), 1.0)
        assert np.max(np.abs(J - J_or) / scale) <= tol, (name, N, K, float(np.max(np.abs(J - J_or) / scale)))
        a, bj, bi = eng.actor_argmin(cand)
        ref = np.argmin(J_or, axis=1)
        for e in range(B):
            assert bi[e] == ref[e] or abs(J_or[e, bi[e]] - J_or[e, ref[e]]) <= 4 * tol * scale[e, 0]
            np.testing.assert_array_equal(a[e], cand[e, bi[e], 0])
    Kg = 64 if cfg.du == 2 else 50
    a, bj, bi = eng.actor_argmin(None, K=Kg)
    grid = O.grid_candidates(cfg, Kg)
    Jg = O.actor_cost(grid[None], x64[:, None, :], x64[:, None, :], cfg)
    assert rel_err_norm(bj, np.min(Jg, axis=1)) <= tol
    # closed loop: three ticks over a caller's tensor, each a map from the same inputs; then T ticks by one call (long rows:
    # the loop of single ticks - no LDS tile to keep) end on the same fields
    cand = rand_actions(rng, name, (B, 64, N), overshoot=0.3 if N > 64 else 1.0).astype(eng.real)
    dcand = eng.to_device(cand)
    env = O.new_batch(cfg, x64)
    rep = PAR.TickReport()
    for t in range(3):
        eng.control_tick(dcand)
        env = PAR.check_tick(cfg, env, cand.astype(np.float64), PAR.device_fields(eng, Nn, critic=False), tol=tol, report=rep,
                             what=f"N={N} t={t}")
    eng2, _ = both(name, B, dtype, n_actor=N, gamma=0.9)
    eng2.set_state(x)
    eng2.control_tick(dcand, T=3)
    for f in (Nn.FIELD_STATE, Nn.FIELD_ACTION, Nn.FIELD_ACCUM, Nn.FIELD_BEST_IDX, Nn.FIELD_STEP_IDX):
        np.testing.assert_array_equal(eng2.get_field(f), eng.get_field(f))


def test_horizon_sanity_bound_and_lds_bound_decisions():
    """rcg_create keeps a sanity bound (RCG_MAX_NACTOR); the on-device optimiser and the search hold a wave's working set in LDS
    and refuse - before anything is touched - a horizon that does not fit, with a message that says so; inside the bound they
    run: Nactor = 40 on the 3-wheel robot lowers the cost of the start sequence."""
    from rcognita_amd import Engine, _native as Nn
    from tests.helpers import engine_cfg

    with pytest.raises(Nn.NativeError) as ei:
        Engine(engine_cfg("3wrobot", 2, "f64", n_actor=4097))
    assert ei.value.code == Nn.ERR_BAD_ARG and "Nactor" in str(ei.value)
    rng = np.random.default_rng(3)
    B = 20
    for dtype in ("f64", "f32"):
        eng, cfg = both("3wrobot", B, dtype, n_actor=40)
        x = rand_states(rng, "3wrobot", B).astype(eng.real)
        eng.set_state(x)
        u0 = O.action_sqn_init(cfg)
        J0 = O.actor_cost(np.broadcast_to(u0, (B, 1, 40, 2)), x[:, None, :].astype(np.float64), x[:, None, :].astype(np.float64), cfg)[:, 0]
        a, u, bj, nit = eng.actor_optimize(iters=30)
        Ju = O.actor_cost(u.astype(np.float64)[:, None], x[:, None, :].astype(np.float64), x[:, None, :].astype(np.float64), cfg)[:, 0]
        assert np.all(Ju <= J0) and np.mean(Ju) < 0.9 * np.mean(J0)
        assert rel_err_norm(bj, Ju) <= (1e-9 if dtype == "f64" else 1e-4)
        a2, ub, bj2, bi2 = eng.actor_search(K=64, rounds=2)
        assert np.all(bj2 <= J0 * (1 + 1e-5))
    eng, _ = both("3wrobot", 4, "f64", n_actor=200)  # 16 envs x 200 steps of state, sequence, gradient: beyond 160 KB of LDS
    eng.set_state(rand_states(rng, "3wrobot", 4))
    st0 = eng.get_state()
    for call in (lambda: eng.control_tick_opt(iters=3), lambda: eng.control_tick_search(K=64, rounds=1)):
        with pytest.raises(Nn.NativeError) as ei:
            call()
        assert ei.value.code == Nn.ERR_UNSUPPORTED and "LDS" in str(ei.value)
        np.testing.assert_array_equal(eng.get_state(), st0)  # refused before the env step
    eng.control_tick(None, K=64)  # the candidate decisions have no such bound
    assert np.all(eng.get_field(Nn.FIELD_STEP_IDX) == 1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("K", [4, 6, 8, 12, 16, 20, 28, 32, 36])
@pytest.mark.parametrize("name,Nh", [("3wrobot", 10), ("3wrobotNI", 3), ("2tank", 20)])
def test_few_candidates_per_env_on_packed_tiles(name, Nh, K, dtype):
    """Streamed candidates with 4 <= K <= 32 (an env's rows a whole number of 16-byte pieces): k_actor_dma_packed - 64 / K envs share one DMA tile, the env state is
    per-lane, the argmin segmented (rcg_actor_dma_packed.hpp).  Operator (J of every row), argmin (ties -> lower index, NaN =
    +inf, all-NaN -> 0) and a three-tick closed loop against the oracle on a batch whose last wave and last tile are ragged;
    gamma != 1 takes the discounted instance.  K = 36: no packing, one ragged tile per env on k_actor_dma."""
    from oracle import parity as PAR
    from rcognita_amd import _native as N

    rng = np.random.default_rng(100 * K + Nh)
    B = 1000 + 7
    gamma = 1.0 if K % 8 else 0.95
    eng, cfg = both(name, B, dtype, n_actor=Nh, gamma=gamma)
    x = rand_states(rng, name, B).astype(eng.real)
    cand = rand_actions(rng, name, (B, K, Nh)).astype(eng.real)
    clean = cand.copy()
    cand[5, min(3, K - 1)] = cand[5, 0]      # an exact tie: the lower index must win if it is the minimum
    cand[6, 1, 0, 0] = np.nan                # a NaN candidate is never selected
    cand[7] = np.nan                         # every candidate NaN: index 0, +inf
    eng.set_state(x)
    dc = eng.to_device(cand)
    J = eng.actor_cost(dc)
    slab16 = (K * Nh * cfg.du * eng.real().itemsize) % 16 == 0  # an env's rows = whole 16-byte pieces
    packed = K <= 32 and slab16
    kernel = "k_actor_dma_packed" if packed else ("k_actor_dma" if slab16 and K >= 33 else "k_actor")
    assert_kernel(eng, kernel, (N.DMA_MPC_G1 if gamma == 1.0 else N.DMA_MPC) if slab16 else None)
    x64, c64 = x.astype(np.float64), cand.astype(np.float64)
    J_or = O.actor_cost(c64, x64[:, None, :], x64[:, None, :], cfg)
    fin = np.isfinite(J_or)
    assert np.array_equal(np.isnan(J), ~fin)
    scale = np.max(np.abs(np.where(fin, J_or, 0.0)), axis=1, keepdims=True)
    scale = np.where(scale > 0, scale, 1.0)
    assert np.nanmax(np.abs(J - J_or)[fin] / np.broadcast_to(scale, J.shape)[fin]) <= TOL[dtype]
    act, bj, bi = eng.actor_argmin(dc)
    assert_kernel(eng, kernel)
    Jc = np.where(np.isnan(J), np.inf, J)  # the kernel's own costs: its argmin must be numpy's on them, bit for bit
    np.testing.assert_array_equal(bi, np.argmin(Jc, axis=1).astype(np.int32))
    np.testing.assert_array_equal(bj, Jc[np.arange(B), bi])
    assert bi[7] == 0 and np.isinf(bj[7]) and bi[6] != 1
    np.testing.assert_array_equal(act[:5], cand[np.arange(5), bi[:5], 0, :])
    # closed loop on the rows as drawn (no NaN): every env, every tick, as a map from the same inputs
    dclean = eng.to_device(clean)
    env = O.new_batch(cfg, x64)
    rep = PAR.TickReport()
    for t in range(3):
        eng.control_tick(dclean, K=K)
        env = PAR.check_tick(cfg, env, clean.astype(np.float64), PAR.device_fields(eng, N), tol=TOL[dtype], report=rep,
                             what=f"{name} K={K} {dtype} t={t}")
    assert_kernel(eng, kernel)
    np.testing.assert_array_equal(eng.get_field(N.FIELD_STEP_IDX), np.full(B, 3, np.int32))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name,Nh,K", [("3wrobot", 5, 16), ("2tank", 10, 12), ("3wrobotNI", 4, 32)])
def test_packed_tiles_with_state_lag_and_per_env_parameters(name, Nh, K, dtype):
    """k_actor_dma_packed with the reference's one-step state lag (rollout from STATE_PREV, y_0 = the observation: two
    state vectors per lane) and heterogeneous per-env parameters (per-lane loads, the model constants prepared per tile):
    a streamed closed loop on a ragged batch, every env and tick against the oracle."""
    from oracle import parity as PAR
    from rcognita_amd import _native as N

    rng = np.random.default_rng(K + Nh)
    B, T = 531, 4
    kw = dict(n_actor=Nh, ref_lag=True)
    pars = None
    if name != "3wrobotNI":  # (the kinematic robot has no parameters)
        kw["per_env_pars"] = True
    eng, cfg = both(name, B, dtype, **kw)
    if name == "3wrobot":
        pars = np.stack([rng.uniform(5, 20, B), rng.uniform(0.5, 2, B)], axis=-1)
    elif name == "2tank":
        pars = np.asarray(cfg.pars)[None] * rng.uniform(0.8, 1.25, (B, 5))
    if pars is not None:
        eng.set_field(N.FIELD_PARS, pars.astype(eng.real))
        pars = pars.astype(eng.real).astype(np.float64)
    x0 = rand_states(rng, name, B).astype(eng.real)
    eng.set_state(x0)
    cand = rand_actions(rng, name, (B, K, Nh)).astype(eng.real)
    dc = eng.to_device(cand)
    env = O.new_batch(cfg, x0.astype(np.float64), pars=pars)
    rep = PAR.TickReport()
    for t in range(T):
        eng.control_tick(dc, K=K)
        env = PAR.check_tick(cfg, env, cand.astype(np.float64), PAR.device_fields(eng, N, with_prev=True), tol=TOL[dtype],
                             report=rep, what=f"{name} K={K} {dtype} lag t={t}")
    assert_kernel(eng, "k_actor_dma_packed", N.DMA_MPC_G1 | 16)  # (+ 16: the env step of the tick ran inside the launch)
    np.testing.assert_array_equal(eng.get_field(N.FIELD_STEP_IDX), np.full(B, T, np.int32))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name,case", [("3wrobot", "full"), ("3wrobot", "biquad_full_tgt"), ("3wrobot", "biquad_diag"),
                                       ("3wrobot", "biquad_diag_tgt"), ("3wrobotNI", "nonsym"), ("2tank", "full"),
                                       ("2tank", "biquad_full_no_tgt")])
def test_streamed_cost_structures_no_preset_has_on_the_production_kernel(name, case, dtype):
    """Full R1 (symmetric and not), the biquadratic stage cost (controllers.py:1079-1082) with diagonal and full matrices, an
    observation target on a robot, the tank without its target: streamed K = 128 through k_actor_dma's DMA_MPC_GEND /
    DMA_MPC_GENF instances (round 6; round 5 ran them on k_actor's plain staging) - operator, argmin and three closed-loop
    ticks against the oracle, whose stage_obj is pinned on the reference's F2 values for exactly these structures."""
    from oracle import parity as PAR
    from rcognita_amd import _native as N

    rng = np.random.default_rng(len(case) + len(name))
    n = len(PRESETS[name]["R1"])
    A = rng.uniform(-1, 1, (n, n))
    ds = n - len(PRESETS[name]["bnds"])
    kw = {
        "full": dict(R1=A @ A.T),
        "nonsym": dict(R1=rng.uniform(-1, 1, (n, n)) + 2 * np.eye(n)),
        "biquad_full_tgt": dict(R1=A @ A.T, R2=np.diag(rng.uniform(0, 1e-3, n)) + 1e-4 * (A.T @ A),
                                stage_obj_struct=O.STAGE_BIQUADRATIC, target=rng.uniform(-1, 1, ds)),
        "biquad_full_no_tgt": dict(R1=A @ A.T, R2=np.diag(rng.uniform(0, 1e-3, n)) + 1e-4 * (A.T @ A),
                                   stage_obj_struct=O.STAGE_BIQUADRATIC, target=None),
        "biquad_diag": dict(R2=np.diag(rng.uniform(0, 1e-3, n)), stage_obj_struct=O.STAGE_BIQUADRATIC),
        "biquad_diag_tgt": dict(R2=np.diag(rng.uniform(0, 1e-3, n)), stage_obj_struct=O.STAGE_BIQUADRATIC,
                                target=rng.uniform(-1, 1, ds)),
    }[case]
    B, K, Nh = 21, 128, 7
    eng, cfg = both(name, B, dtype, n_actor=Nh, gamma=0.96, **kw)
    x = rand_states(rng, name, B).astype(eng.real)
    cand = rand_actions(rng, name, (B, K, Nh)).astype(eng.real)
    dcand = eng.to_device(cand)
    eng.set_state(x)
    x64, c64 = x.astype(np.float64), cand.astype(np.float64)
    J = eng.actor_cost(dcand)
    full = case in ("full", "nonsym", "biquad_full_tgt", "biquad_full_no_tgt")
    assert_kernel(eng, "k_actor_dma", N.DMA_MPC_GENF if full else N.DMA_MPC_GEND)
    J_or = O.actor_cost(c64, x64[:, None, :], x64[:, None, :], cfg)
    # full matrices mix signs: the cost is compared at the magnitude of the env's largest cost
    scale = np.maximum(np.max(np.abs(J_or), axis=1, keepdims=True), 1.0)
    assert np.max(np.abs(J - J_or) / scale) <= TOL[dtype] * (4 if (dtype == "f32" and full) else 1), case
    a, bj, bi = eng.actor_argmin(dcand)
    np.testing.assert_array_equal(bi, np.argmin(np.where(np.isnan(J), np.inf, J), axis=1).astype(np.int32))
    np.testing.assert_array_equal(a, cand[np.arange(B), bi, 0, :])
    env = O.new_batch(cfg, x64)
    rep = PAR.TickReport()
    for t in range(3):
        eng.control_tick(dcand)
        env = PAR.check_tick(cfg, env, c64, PAR.device_fields(eng, N, critic=False),
                             tol=TOL[dtype] * (4 if (dtype == "f32" and full) else 1), report=rep, what=f"{case} t={t}")
    assert_kernel(eng, "k_actor_dma", N.DMA_MPC_GENF if full else N.DMA_MPC_GEND)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name,mode,cs,case", [("3wrobot", "RQL", "quad-nomix", "full"), ("3wrobot", "RQL", "quad-lin", "biquad_diag"),
                                               ("3wrobotNI", "RQL", "quad-mix", "biquad_full_tgt"), ("2tank", "RQL", "quadratic", "full"),
                                               ("3wrobot", "SQL", "quad-nomix", "full"), ("2tank", "SQL", "quad-lin", "biquad_full")])
def test_streamed_critic_modes_with_cost_structures_no_preset_has(name, mode, cs, case, dtype):
    """RQL / SQL with a full R1 or the biquadratic stage cost, streamed K = 128: RQL on k_actor_dma's DMA_RQL_GEN_* instances
    (stage_any per step), SQL on its ordinary instances (no stage cost inside the rollout: only upd_accum_obj sees the structure)
    - operator, argmin and three closed-loop ticks (env step, push, fit, decision, accum) against the oracle."""
    from oracle import parity as PAR
    from rcognita_amd import _native as N

    rng = np.random.default_rng(len(case) + 3 * len(name) + len(cs))
    n = len(PRESETS[name]["R1"])
    ds = n - len(PRESETS[name]["bnds"])
    A = rng.uniform(-1, 1, (n, n))
    kw = {
        "full": dict(R1=A @ A.T),
        "biquad_diag": dict(R2=np.diag(rng.uniform(0, 1e-3, n)), stage_obj_struct=O.STAGE_BIQUADRATIC),
        "biquad_full": dict(R1=A @ A.T, R2=np.diag(rng.uniform(0, 1e-3, n)) + 1e-4 * (A.T @ A), stage_obj_struct=O.STAGE_BIQUADRATIC),
        "biquad_full_tgt": dict(R1=A @ A.T, R2=np.diag(rng.uniform(0, 1e-3, n)) + 1e-4 * (A.T @ A),
                                stage_obj_struct=O.STAGE_BIQUADRATIC, target=rng.uniform(-1, 1, ds)),
    }[case]
    B, K, Nh = 21, 128, 6
    eng, cfg = both(name, B, dtype, n_actor=Nh, gamma=0.96, mode=O.MODE_IDS[mode], critic_struct=O.CRITIC_IDS[cs], n_critic=4,
                    buffer_size=6, **kw)
    x = rand_states(rng, name, B).astype(eng.real)
    cand = rand_actions(rng, name, (B, K, Nh)).astype(eng.real)
    w = rng.uniform(0.1, 2.0, (B, cfg.dc)).astype(eng.real)
    dcand = eng.to_device(cand)
    eng.set_state(x)
    eng.set_field(N.FIELD_W_CRITIC, w)
    x64, c64 = x.astype(np.float64), cand.astype(np.float64)
    J = eng.actor_cost(dcand)
    want = (N.DMA_RQL_GEN_0 if mode == "RQL" else N.DMA_SQL_0) + N.CRITIC_IDS[cs]
    assert_kernel(eng, "k_actor_dma", want)
    J_or = O.actor_cost(c64, x64[:, None, :], x64[:, None, :], cfg, w_critic=w.astype(np.float64)[:, None, :])
    scale = np.maximum(np.max(np.abs(J_or), axis=1, keepdims=True), 1.0)
    tol = TOL[dtype] * (4 if dtype == "f32" else 1)
    assert np.max(np.abs(J - J_or) / scale) <= tol, case
    a, bj, bi = eng.actor_argmin(dcand)
    np.testing.assert_array_equal(bi, np.argmin(np.where(np.isnan(J), np.inf, J), axis=1).astype(np.int32))
    env = O.new_batch(cfg, x64)
    rep = PAR.TickReport()
    for t in range(3):
        eng.control_tick(dcand)
        env = PAR.check_tick(cfg, env, c64, PAR.device_fields(eng, N, critic=True), tol=tol if dtype == "f32" else 1e-9,
                             tol_over={"w_critic": 1e-6, "best_J": 1e-7} if dtype == "f64" else None, report=rep,
                             what=f"{mode} {case} t={t}")
    assert_kernel(eng, "k_actor_dma", want)
