"""Seeded sweep over the streamed decision kernels: random (system, element type, mode, critic structure, Nactor, K, batch,
gamma, state lag) - whatever kernel the library's dispatch picks (k_actor_dma, k_actor_dma_packed, k_actor: rcg_last_launch
says which) must return the oracle's `_actor_cost` of every row (controllers.py:1273-1328) and numpy's argmin of its own
costs.  Shapes include rows of 1 .. 40 reals (1, 2 and 4 rows per lane), K from 2 to 300 (packed tiles, ragged tiles, several
tiles per env), batches that leave waves and tiles ragged.  ``gpu`` marked."""
import numpy as np
import pytest

from oracle import rcg_oracle as O
from tests.helpers import PRESETS, TOL, both, rand_actions, rand_states

pytestmark = pytest.mark.gpu

SEEN = set()


@pytest.mark.parametrize("seed", range(144))
def test_streamed_decision_of_a_random_shape(seed):
    from rcognita_amd import _native as N

    rng = np.random.default_rng(1000 + seed)
    name = ["3wrobot", "3wrobotNI", "2tank"][seed % 3]
    dtype = ["f32", "f64"][(seed // 3) % 2]
    mode = [O.MODE_MPC, O.MODE_MPC, O.MODE_RQL, O.MODE_SQL][(seed // 6) % 4]
    cs = int(rng.integers(0, 4))
    du = 1 if name == "2tank" else 2
    Nh = int(rng.integers(1, 40 // du + 1))
    K = int(rng.choice([2, 3, 4, 6, 8, 10, 12, 16, 20, 24, 31, 32, 33, 35, 36, 38, 39, 40, 44, 52, 63, 64, 65, 100, 128, 130, 192, 256, 300]))
    B = int(rng.choice([1, 2, 5, 17, 64, 65, 129, 300]))
    gamma = float(rng.choice([1.0, 0.9]))
    lag = bool(rng.integers(0, 2))
    kw = dict(n_actor=Nh, mode=mode, critic_struct=cs, gamma=gamma)
    if mode != O.MODE_MPC:
        kw.update(n_critic=3, buffer_size=5)
    eng, cfg = both(name, B, dtype, **kw)
    real = eng.real
    x = rand_states(rng, name, B).astype(real)
    xs = (x + rng.normal(0, 0.05, x.shape)).astype(real) if lag else x
    cand = rand_actions(rng, name, (B, K, Nh)).astype(real)
    w = None
    if mode != O.MODE_MPC:
        lo, hi = O.critic_bounds(cs, cfg.dc)
        w = rng.uniform(np.maximum(lo, -2.0), np.minimum(hi, 2.0), (B, cfg.dc)).astype(real)
        eng.set_field(N.FIELD_W_CRITIC, w)
    dc = eng.to_device(cand)
    J = eng.actor_cost(dc, obs=x, state_sys=xs)
    ll = eng.last_launch(N.KERNEL_ACTOR)
    SEEN.add(ll["kernel"])
    x64, xs64, c64 = x.astype(np.float64), xs.astype(np.float64), cand.astype(np.float64)
    J_or = O.actor_cost(c64, x64[:, None, :], xs64[:, None, :], cfg,
                        w_critic=None if w is None else w.astype(np.float64)[:, None, :])
    scale = np.maximum(np.max(np.abs(J_or), axis=1, keepdims=True), 1e-30)
    err = float(np.max(np.abs(J - J_or) / scale))
    what = f"seed {seed}: {name} {dtype} mode {mode} cs {cs} N {Nh} K {K} B {B} gamma {gamma} lag {lag} -> {ll}"
    assert err <= TOL[dtype], f"{what}: J rel err {err:.3e}"
    act, bj, bi = eng.actor_argmin(dc, obs=x, state_sys=xs)
    Jc = np.where(np.isnan(J), np.inf, J)
    np.testing.assert_array_equal(bi, np.argmin(Jc, axis=1).astype(np.int32), err_msg=what)
    np.testing.assert_array_equal(bj, Jc[np.arange(B), bi], err_msg=what)
    np.testing.assert_array_equal(act, cand[np.arange(B), bi, 0, :], err_msg=what)
    eng.close()


def test_the_sweep_reached_every_streamed_kernel():
    assert {"k_actor_dma", "k_actor_dma_packed", "k_actor"} <= SEEN, SEEN
