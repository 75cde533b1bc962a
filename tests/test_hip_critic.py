"""RQL/SQL side of the path on the GPU: buffer push, critic fit (replacement of the reference's SLSQP
_critic_optimizer) and the fused control tick with a learned terminal/stacked critic.  ``gpu`` marked."""
import numpy as np
import pytest

from oracle import rcg_oracle as O
from tests.conftest import load_golden
from tests.helpers import SYSTEMS, assert_kernel, both, rand_actions, rand_states, rel_err_norm

pytestmark = pytest.mark.gpu


def _load_buffers(eng, N, obs_buf, act_buf, w_prev):
    """Arrange the handle so that ONE push (rcg_critic_update) leaves exactly (obs_buf, act_buf):
    rows shifted down by one, the newest row supplied through STATE / ACTION."""
    B, bs, _ = obs_buf.shape
    ob = np.concatenate([np.zeros_like(obs_buf[:, :1]), obs_buf[:, :-1]], axis=1)
    ab = np.concatenate([np.zeros_like(act_buf[:, :1]), act_buf[:, :-1]], axis=1)
    eng.set_field(N.FIELD_OBS_BUF, ob)
    eng.set_field(N.FIELD_ACT_BUF, ab)
    eng.set_field(N.FIELD_STATE, obs_buf[:, -1])
    eng.set_field(N.FIELD_ACTION, act_buf[:, -1])
    eng.set_field(N.FIELD_W_PREV, w_prev)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("name", SYSTEMS)
def test_critic_fit_vs_oracle_and_reference_slsqp(name, dtype):
    """TD stacks of the golden F8 fixture: (a) HIP fit == oracle fit (same algorithm, float64 inside);
    (b) the cost reached is never worse than what the reference's SLSQP reached, nor than the start."""
    from rcognita_amd import _native as N

    meta, z = load_golden(f"F8_slsqp_critic_{name}")
    for c in meta["cases"]:
        cs = c["tag"]
        ob, ab, wp = z[f"{cs}__obs_buf"], z[f"{cs}__act_buf"], z[f"{cs}__w_prev"]
        B = ob.shape[0]
        eng, cfg = both(name, B, dtype, mode=O.MODE_RQL, gamma=c["gamma"], critic_struct=O.CRITIC_IDS[cs],
                        n_critic=c["Ncritic"], buffer_size=c["buffer_size"])
        _load_buffers(eng, N, ob, ab, wp)
        eng.critic_update(do_fit=True)
        np.testing.assert_array_equal(eng.get_field(N.FIELD_OBS_BUF), ob.astype(eng.real))  # push_vec
        np.testing.assert_array_equal(eng.get_field(N.FIELD_ACT_BUF), ab.astype(eng.real))
        w = eng.get_field(N.FIELD_W_CRITIC).astype(np.float64)
        np.testing.assert_array_equal(eng.get_field(N.FIELD_W_PREV).astype(np.float64), w)
        # the oracle sees the same (dtype-rounded) buffers the device holds
        rb = lambda a: a.astype(eng.real).astype(np.float64)
        w_or = O.critic_fit(cfg, rb(wp), rb(ob), rb(ab))
        Jc = O.critic_cost(w, rb(wp), rb(ob), rb(ab), cfg)
        Jc_or = O.critic_cost(w_or, rb(wp), rb(ob), rb(ab), cfg)
        J0, Js = z[f"{cs}__Jc_init"], z[f"{cs}__Jc_fit"]
        lo, hi = O.critic_bounds(cfg.critic_struct, cfg.dc)
        assert np.all(w >= lo - 1e-4) and np.all(w <= hi + 1e-3), cs
        tolw = 1e-6 if dtype == "f64" else 2e-4  # f32: w is stored rounded to float
        assert rel_err_norm(w, w_or) < tolw, (cs, rel_err_norm(w, w_or))
        assert np.all(np.abs(Jc - Jc_or) <= 1e-5 * J0 + 1e-9), cs
        slack = 1e-6 if dtype == "f64" else 1e-4
        assert np.all(Jc <= Js * (1 + slack) + slack * J0), (cs, float(np.max((Jc - Js) / J0)))
        assert np.all(Jc <= J0 * (1 + slack)), cs


@pytest.mark.parametrize("m_rows", [1, 2, 5, 8])
def test_critic_fit_row_counts(m_rows):
    """Ncritic - 1 = 1 .. 8 rows (both kernel instantiations), random stacks, vs the oracle."""
    from rcognita_amd import _native as N

    rng = np.random.default_rng(40 + m_rows)
    B, bs = 33, 12
    eng, cfg = both("2tank", B, "f64", mode=O.MODE_SQL, gamma=0.95, critic_struct=O.CRITIC_QUADRATIC,
                    n_critic=m_rows + 1, buffer_size=bs)
    ob = np.stack([rand_states(rng, "2tank", bs) for _ in range(B)])
    ab = rand_actions(rng, "2tank", (B, bs))
    wp = rng.uniform(0.5, 1.5, (B, cfg.dc))
    _load_buffers(eng, N, ob, ab, wp)
    eng.critic_update(do_fit=True)
    w = eng.get_field(N.FIELD_W_CRITIC)
    w_or = O.critic_fit(cfg, wp, ob, ab)
    assert rel_err_norm(w, w_or) < 1e-6
    np.testing.assert_allclose(eng.critic_cost(), O.critic_cost(w_or, w_or, ob, ab, cfg), rtol=1e-5, atol=1e-12)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("name,cs,n_critic,bs,m_eff", [
    ("2tank", "quadratic", 12, 20, 11),     # `--Ncritic 12 --buffer_size 20`: a legal run of the reference (VERDICT r5 missing 1)
    ("2tank", "quad-lin", 19, 20, 18),
    ("3wrobotNI", "quad-mix", 30, 20, 18),  # Ncritic clipped to buffer_size - 1 = 19 (controllers.py:1015): 18 rows
    ("3wrobot", "quad-lin", 12, 20, 11),    # 35 weights, fewer rows than unknowns
    ("3wrobot", "quad-nomix", 10, 16, 9),   # the first row count beyond the register kernels
])
def test_critic_fit_with_more_than_eight_rows(name, cs, n_critic, bs, m_eff, dtype):
    """Ncritic - 1 > 8 TD rows: the reference only clips Ncritic to buffer_size - 1 (controllers.py:1015, class default
    buffer_size = 20, :828), so up to 19 rows are a legal run.  k_critic_fit_gen (stack and factor in a scratch tensor of the
    handle) against the oracle's fit of the same stack, its cost against the cost at the start point."""
    from rcognita_amd import _native as N

    rng = np.random.default_rng(400 + n_critic)
    B = 70  # two waves, the second ragged
    eng, cfg = both(name, B, dtype, mode=O.MODE_RQL, gamma=0.95, critic_struct=O.CRITIC_IDS[cs], n_critic=n_critic,
                    buffer_size=bs)
    assert cfg.n_critic - 1 == m_eff
    ob = np.stack([rand_states(rng, name, bs) for _ in range(B)])
    ab = rand_actions(rng, name, (B, bs))
    wp = rng.uniform(0.5, 1.5, (B, cfg.dc))
    _load_buffers(eng, N, ob, ab, wp)
    eng.critic_update(do_fit=True)
    ll = eng.last_launch(N.KERNEL_CRITIC)
    assert ll["kernel"] == "k_critic_fit" and (ll["variant"] & 2048), ll  # bit 11: the any-m form
    np.testing.assert_array_equal(eng.get_field(N.FIELD_OBS_BUF), ob.astype(eng.real))  # push_vec
    np.testing.assert_array_equal(eng.get_field(N.FIELD_ACT_BUF), ab.astype(eng.real))
    rb = lambda a: a.astype(eng.real).astype(np.float64)
    w = eng.get_field(N.FIELD_W_CRITIC).astype(np.float64)
    w_or = O.critic_fit(cfg, rb(wp), rb(ob), rb(ab))
    # more rows than weights: the walk's m x m system A_F A_F^T + mu I is rank deficient up to the Tikhonov term (condition
    # ~1e8), the last bits of two float64 evaluations of the same walk move the weights by up to 1e-5 - the objective agrees
    tolw = (1e-6 if m_eff <= cfg.dc else 5e-5) if dtype == "f64" else 2e-4  # f32: w is stored rounded to float
    assert rel_err_norm(w, w_or) < tolw, rel_err_norm(w, w_or)
    Jc, Jc_or = O.critic_cost(w, rb(wp), rb(ob), rb(ab), cfg), O.critic_cost(w_or, rb(wp), rb(ob), rb(ab), cfg)
    J0 = O.critic_cost(np.ones((B, cfg.dc)), rb(wp), rb(ob), rb(ab), cfg)
    assert np.all(np.abs(Jc - Jc_or) <= 1e-5 * J0 + 1e-9)
    assert np.all(Jc <= J0 * (1 + 1e-6))
    # a second fit on the same handle re-uses the scratch; a non-finite buffer keeps the start point (the walk's safeguard)
    ob2 = ob.copy()
    ob2[3, 2, 0] = np.nan
    _load_buffers(eng, N, ob2, ab, wp)
    eng.critic_update(do_fit=True)
    w2 = eng.get_field(N.FIELD_W_CRITIC).astype(np.float64)
    np.testing.assert_array_equal(w2[3], np.ones(cfg.dc))
    keep = np.arange(B) != 3
    assert rel_err_norm(w2[keep], w_or[keep]) < tolw


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_closed_loop_with_eleven_td_rows_vs_oracle(dtype):
    """`--Ncritic 12 --buffer_size 20` as a closed loop: every tick (env step, push, 11-row fit, decision) as a map from the
    same inputs against the oracle (oracle/parity.py), streamed and generated candidates, past the point where the 20-row
    buffers have filled."""
    from oracle import parity as PAR
    from rcognita_amd import _native as N

    rng = np.random.default_rng(77)
    B, Nh, K, T = 9, 6, 16, 24
    eng, cfg = both("2tank", B, dtype, n_actor=Nh, mode=O.MODE_RQL, critic_struct=O.CRITIC_QUADRATIC, gamma=0.95,
                    n_critic=12, buffer_size=20)
    x0 = rand_states(rng, "2tank", B).astype(eng.real)
    eng.set_state(x0)
    env = O.new_batch(cfg, x0.astype(np.float64))
    cand = O.grid_candidates(cfg, K)
    rep = PAR.TickReport()
    for t in range(T):
        eng.control_tick(None, K=K)
        env = PAR.check_tick(cfg, env, cand, PAR.device_fields(eng, N, critic=True), tol=1e-9 if dtype == "f64" else 1e-5,
                             tol_over={"w_critic": 5e-5, "best_J": 1e-6} if dtype == "f64" else None, report=rep,
                             what=f"11 rows t={t}")  # (11 rows, 6 weights: rank-deficient m x m system, see above)
    assert rep.ticks == T
    assert eng.last_launch(N.KERNEL_CRITIC)["variant"] & 2048
    # T ticks issued by one call loop single ticks for such a handle (no persistent instance) and end on the same fields
    eng2, _ = both("2tank", B, dtype, n_actor=Nh, mode=O.MODE_RQL, critic_struct=O.CRITIC_QUADRATIC, gamma=0.95, n_critic=12,
                   buffer_size=20)
    eng2.set_state(x0)
    eng2.control_tick(None, K=K, T=T)
    for f in (N.FIELD_STATE, N.FIELD_W_CRITIC, N.FIELD_OBS_BUF, N.FIELD_ACTION, N.FIELD_ACCUM):
        np.testing.assert_array_equal(eng2.get_field(f), eng.get_field(f))


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("name,mode,cs,K,every", [
    ("2tank", O.MODE_RQL, O.CRITIC_QUADRATIC, 32, 1),      # BASELINE configs[2] shape (small batch)
    ("2tank", O.MODE_SQL, O.CRITIC_QUAD_NOMIX, 16, 2),
    ("3wrobotNI", O.MODE_RQL, O.CRITIC_QUAD_MIX, 64, 1),
    ("3wrobot", O.MODE_SQL, O.CRITIC_QUAD_NOMIX, 16, 3),
])
def test_rql_sql_control_tick_vs_oracle(name, mode, cs, K, every, dtype):
    """Fused tick with critic: sim -> push -> fit (on ticks every-1, 2*every-1, ...: controllers.py:1466 with
    critic_clock = t0) -> actor argmin with Q_w.  EVERY tick of both builds is checked as a map from the same inputs
    (oracle/parity.py): the TD target feeds w_prev back into the next fit and the fit's conditioning amplifies last-bit
    differences, so the oracle continues from the device's buffers and weights after each check.  The fit reads the
    OLDEST buffer rows (controllers.py:1231-1234), which by then are the device's own values: the weights agree to
    1e-5 in the float32 build as well."""
    from oracle import parity as PAR
    from rcognita_amd import _native as N

    rng = np.random.default_rng(17 + K)
    B, Nh = 19, 5
    T = 9
    eng, cfg = both(name, B, dtype, n_actor=Nh, mode=mode, critic_struct=cs, gamma=0.95, n_critic=4,
                    buffer_size=6, critic_every_ticks=every)
    x0 = rand_states(rng, name, B).astype(eng.real)
    eng.set_state(x0)
    env = O.new_batch(cfg, x0.astype(np.float64))
    cand = O.grid_candidates(cfg, K)
    rep = PAR.TickReport()
    w_before = eng.get_field(N.FIELD_W_CRITIC).copy()
    for t in range(T):
        eng.control_tick(None, K=K)
        env = PAR.check_tick(cfg, env, cand, PAR.device_fields(eng, N, critic=True),
                             tol=1e-9 if dtype == "f64" else 1e-5,
                             tol_over={"w_critic": 1e-6, "best_J": 1e-7} if dtype == "f64" else None, report=rep,
                             what=f"{name} t={t}")
        w = eng.get_field(N.FIELD_W_CRITIC)
        if (t + 1) % every != 0:  # not a fit tick: the weights are held
            np.testing.assert_array_equal(w, w_before)
        w_before = w.copy()
    assert rep.ticks == T


@pytest.mark.parametrize("dtype,tol", [("f32", 1e-5), ("f64", 1e-11)])
@pytest.mark.parametrize("cs", [O.CRITIC_QUAD_LIN, O.CRITIC_QUADRATIC, O.CRITIC_QUAD_NOMIX, O.CRITIC_QUAD_MIX])
@pytest.mark.parametrize("mode", [O.MODE_RQL, O.MODE_SQL])
@pytest.mark.parametrize("name", SYSTEMS)
def test_streamed_rql_sql_on_the_production_kernel(name, mode, cs, dtype, tol):
    """Streamed candidates, K a multiple of 64: RQL and SQL run on k_actor_dma's critic instances in both element types
    (the env's critic weights travel with its state: in registers, or - float64 with more than 9 weights - parked in a
    per-wave LDS slot).  _actor_cost of every candidate and the argmin against the float64 oracle, per-env weights,
    gamma != 1, several envs per launch so that the weights change between the envs a wave owns."""
    from rcognita_amd import _native as N

    rng = np.random.default_rng(1000 * mode + 10 * cs + len(name))
    B, K, Nh = 9, 128, 6
    eng, cfg = both(name, B, dtype, n_actor=Nh, mode=mode, critic_struct=cs, gamma=0.9, n_critic=4, buffer_size=6)
    real = eng.real
    x0 = rand_states(rng, name, B).astype(real)
    eng.set_state(x0)
    lo, hi = O.critic_bounds(cs, cfg.dc)
    w = rng.uniform(np.maximum(lo, -2.0), np.minimum(hi, 2.0), (B, cfg.dc)).astype(real)
    eng.set_field(N.FIELD_W_CRITIC, w)
    cand = rand_actions(rng, name, (B, K, Nh)).astype(real)
    x64, w64, c64 = x0.astype(np.float64), w.astype(np.float64), cand.astype(np.float64)
    J_or = O.actor_cost(c64, x64[:, None, :], x64[:, None, :], cfg, pars=np.asarray(cfg.pars, dtype=np.float64),
                        w_critic=w64[:, None, :])
    J = eng.actor_cost(cand)  # W_CRITIC of the handle
    assert J.shape == (B, K)
    assert_kernel(eng, "k_actor_dma", (N.DMA_RQL_0 if mode == O.MODE_RQL else N.DMA_SQL_0) + cs)
    # signed critic weights make J a difference of large terms: the error is measured against the env's largest |J|
    # (what an argmin over the row is sensitive to), 1e-5 as everywhere else in f32
    scale = np.max(np.abs(J_or), axis=1, keepdims=True)
    assert np.max(np.abs(J - J_or) / scale) < tol
    a, bj, bi = eng.actor_argmin(cand)
    ref_i = np.argmin(J_or, axis=1)
    flipped = bi != ref_i  # a near-tie may pick the runner-up: then its cost must be within rounding of the best
    for e in np.flatnonzero(flipped):
        assert abs(J_or[e, bi[e]] - J_or[e, ref_i[e]]) <= 2 * tol * abs(J_or[e, ref_i[e]]) + (1e-6 if dtype == "f32" else 0.0)
    np.testing.assert_array_equal(a[~flipped], cand[np.arange(B), ref_i, 0, :][~flipped])


def _dc_of(name, cs):
    n = {"3wrobot": 5 + 2, "3wrobotNI": 3 + 2, "2tank": 2 + 1}[name]
    ds, du = {"3wrobot": (5, 2), "3wrobotNI": (3, 2), "2tank": (2, 1)}[name]
    return {O.CRITIC_QUAD_LIN: n * (n + 1) // 2 + n, O.CRITIC_QUADRATIC: n * (n + 1) // 2, O.CRITIC_QUAD_NOMIX: n,
            O.CRITIC_QUAD_MIX: ds + ds * du + du}[cs]


@pytest.mark.parametrize("dtype,tol", [("f32", 1e-5), ("f64", 1e-11)])
@pytest.mark.parametrize("cs", [O.CRITIC_QUAD_LIN, O.CRITIC_QUADRATIC, O.CRITIC_QUAD_NOMIX, O.CRITIC_QUAD_MIX])
@pytest.mark.parametrize("mode", [O.MODE_RQL, O.MODE_SQL])
@pytest.mark.parametrize("name,K", [("3wrobot", 16), ("3wrobot", 24), ("3wrobotNI", 8), ("3wrobotNI", 20), ("2tank", 32),
                                    ("2tank", 12)])
def test_few_candidates_rql_sql_on_packed_tiles(name, K, mode, cs, dtype, tol):
    """Streamed RQL / SQL with 4 <= K <= 32: k_actor_dma_packed's critic instances (64 / K envs per DMA tile, the critic
    weights of a lane's env are per-lane loads) wherever the weights fit 36 dwords - every structure in f32, up to 18
    weights in f64; beyond that one ragged tile per env on k_actor_dma from K = 20, k_actor below.  _actor_cost of every
    candidate against the float64 oracle with per-env weights that differ between the envs of ONE tile, the argmin on the
    kernel's own costs bit for bit, a ragged last wave and tile, then three closed-loop ticks (critic fit in between)
    checked tick by tick."""
    from oracle import parity as PAR
    from rcognita_amd import _native as N

    rng = np.random.default_rng(7000 + 100 * mode + 10 * cs + K)
    B, Nh = 1000 + 7, 6
    eng, cfg = both(name, B, dtype, n_actor=Nh, mode=mode, critic_struct=cs, gamma=0.9, n_critic=4, buffer_size=6)
    real = eng.real
    x0 = rand_states(rng, name, B).astype(real)
    eng.set_state(x0)
    lo, hi = O.critic_bounds(cs, cfg.dc)
    w = rng.uniform(np.maximum(lo, -2.0), np.minimum(hi, 2.0), (B, cfg.dc)).astype(real)
    eng.set_field(N.FIELD_W_CRITIC, w)
    cand = rand_actions(rng, name, (B, K, Nh)).astype(real)
    clean = cand.copy()
    cand[6, 1, 0, 0] = np.nan  # a NaN candidate is never selected
    x64, w64, c64 = x0.astype(np.float64), w.astype(np.float64), cand.astype(np.float64)
    J_or = O.actor_cost(c64, x64[:, None, :], x64[:, None, :], cfg, w_critic=w64[:, None, :])
    dcand = eng.to_device(cand)
    J = eng.actor_cost(dcand)
    variant = (N.DMA_RQL_0 if mode == O.MODE_RQL else N.DMA_SQL_0) + cs
    assert cfg.dc == _dc_of(name, cs)
    slab16 = (K * Nh * cfg.du * real().itemsize) % 16 == 0
    if slab16 and cfg.dc * real().itemsize <= 144:
        kernel = "k_actor_dma_packed"
    elif slab16 and K >= 20:
        kernel = "k_actor_dma"
    else:
        kernel = "k_actor"
    assert_kernel(eng, kernel, variant if kernel != "k_actor" else None)
    fin = np.isfinite(J_or)
    assert np.array_equal(np.isnan(J), ~fin)
    scale = np.max(np.abs(np.where(fin, J_or, 0.0)), axis=1, keepdims=True)
    assert np.max(np.abs(np.where(fin, J - J_or, 0.0)) / scale) < tol
    a, bj, bi = eng.actor_argmin(dcand)
    assert_kernel(eng, kernel)
    Jc = np.where(np.isnan(J), np.inf, J)
    np.testing.assert_array_equal(bi, np.argmin(Jc, axis=1).astype(np.int32))
    np.testing.assert_array_equal(bj, Jc[np.arange(B), bi])
    np.testing.assert_array_equal(a, cand[np.arange(B), bi, 0, :])
    assert bi[6] != 1
    # closed loop: env step + push + fit (k_critic_fit), then the packed decision; every env, every tick
    dclean = eng.to_device(clean)
    env = O.new_batch(cfg, x64)
    env.w_critic = w64.copy()
    env.w_prev = w64.copy()
    eng.set_field(N.FIELD_W_PREV, w)
    rep = PAR.TickReport()
    for t in range(3):
        eng.control_tick(dclean, K=K)
        env = PAR.check_tick(cfg, env, clean.astype(np.float64), PAR.device_fields(eng, N, critic=True),
                             tol=1e-9 if dtype == "f64" else 1e-5,
                             tol_over={"w_critic": 1e-6, "best_J": 1e-7} if dtype == "f64" else None, report=rep,
                             what=f"{name} K={K} mode {mode} cs {cs} {dtype} t={t}")
    assert_kernel(eng, kernel)
    assert rep.ticks == 3


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("mode", ["RQL", "SQL"])
@pytest.mark.parametrize("name,cs", [("3wrobotNI", "quad-nomix"), ("3wrobotNI", "quad-mix"), ("3wrobot", "quad-nomix"),
                                     ("2tank", "quad-nomix"), ("2tank", "quadratic"), ("2tank", "quad-lin")])
def test_critic_fit_on_the_td_stacks_of_the_reference_closed_loop(name, cs, mode, dtype):
    """Fixtures F7c: at every control tick of the reference's own RQL / SQL loop, the buffers and w_prev its
    _critic_optimizer saw and the Jc its SLSQP reached.  k_critic_fit on those stacks (one env per tick): equals the
    oracle twin, never above Jc(w_init), above SLSQP's Jc by at most 2e-2 Jc(w_init) (the mu term: measured 1.34e-2 on these
    stacks, oracle/experiments/fit_mu_study.py; the sharp statement - the fit's own objective is not above its value at
    SLSQP's weights - is asserted tick by tick in tests/test_hip_teacher_forced.py)."""
    from rcognita_amd import _native as N

    meta, z = load_golden(f"F7c_trace_{name}_{mode}_{cs}")
    ob, ab, wp = z["tick_obs_buf"], z["tick_act_buf"], z["tick_w_prev"]
    B = ob.shape[0]
    eng, cfg = both(name, B, dtype, mode=O.MODE_IDS[mode], gamma=meta["gamma"], critic_struct=O.CRITIC_IDS[cs],
                    n_critic=meta["Ncritic"], buffer_size=meta["buffer_size"], n_actor=meta["Nactor"])
    _load_buffers(eng, N, ob, ab, wp)
    eng.critic_update(do_fit=True)
    assert_kernel(eng, "k_critic_fit", kind=N.KERNEL_CRITIC)
    w = eng.get_field(N.FIELD_W_CRITIC).astype(np.float64)
    rb = lambda a: a.astype(eng.real).astype(np.float64)
    w_or = O.critic_fit(cfg, rb(wp), rb(ob), rb(ab))
    Jc = O.critic_cost(w, rb(wp), rb(ob), rb(ab), cfg)
    Jc_or = O.critic_cost(w_or, rb(wp), rb(ob), rb(ab), cfg)
    J0, Js = z["tick_Jc_init"], z["tick_Jc"]
    scale = np.maximum(J0, 1e-12)
    assert np.all(np.abs(Jc - Jc_or) <= (1e-7 if dtype == "f64" else 1e-4) * scale + 1e-9)
    slack = 1e-6 if dtype == "f64" else 1e-4
    assert np.all(Jc <= J0 * (1 + slack) + 1e-9)
    assert np.all(Jc <= Js + (2e-2 + slack) * scale + 1e-9), float(np.max((Jc - Js) / scale))
    lo, hi = O.critic_bounds(cfg.critic_struct, cfg.dc)
    assert np.all(w >= lo - 1e-4) and np.all(w <= hi + 1e-3)


@pytest.mark.parametrize("name,cs", [("3wrobotNI", "quad-lin"), ("3wrobot", "quadratic"), ("2tank", "quadratic")])
def test_non_finite_buffers_leave_the_fit_at_its_start_point(name, cs):
    """buffers that hold inf / NaN (an env that ran away before it was frozen).  The walk's comparisons are then
    comparisons with NaN; the four-lane form (>= 20 weights: the first two cases) must still take the same decisions in the
    four lanes of a quad (total tie rule, rcg_critic_fit_ml.hpp) and both forms end on the safeguard - clip(w_init) - for
    those envs, exactly as the oracle does, while every other env of the wave gets its fit."""
    from rcognita_amd import _native as N

    B = 96
    eng, cfg = both(name, B, "f64", mode=O.MODE_RQL, critic_struct=O.CRITIC_IDS[cs], n_critic=4, buffer_size=6, n_actor=5)
    rng = np.random.default_rng(11)
    ob = np.stack([rand_states(rng, name, B) for _ in range(6)], axis=1)
    ab = rand_actions(rng, name, (B, 6))
    wp = 1.0 + rng.random((B, cfg.dc))
    bad = np.arange(B) % 7 == 3
    ob[bad, 1, 0] = np.inf          # a TD row with an infinite feature
    ob[bad & (np.arange(B) % 2 == 1), 2, -1] = np.nan
    _load_buffers(eng, N, ob, ab, wp)
    eng.critic_update(do_fit=True)
    ll = eng.last_launch(N.KERNEL_CRITIC)
    assert bool(ll["variant"] & 1024) == (cfg.dc >= 20), ll
    w = eng.get_field(N.FIELD_W_CRITIC).astype(np.float64)
    lo, hi = O.critic_bounds(cfg.critic_struct, cfg.dc)
    assert np.all(np.isfinite(w))
    np.testing.assert_array_equal(w[bad], np.broadcast_to(np.clip(np.ones(cfg.dc), lo, hi), w[bad].shape))
    w_or = O.critic_fit(cfg, wp[~bad], ob[~bad], ab[~bad])
    assert rel_err_norm(w[~bad], w_or, floor=1.0) < 1e-6
    assert not np.allclose(w[~bad], 1.0)
