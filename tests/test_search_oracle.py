"""CPU tests of the candidate-search oracle (oracle/search_oracle.py, twin of rcg_search.hpp): the sampling rule's
invariants, the generator's statistics, monotonicity of the search and its quality against the reference's SLSQP."""
import numpy as np
import pytest

from oracle import rcg_oracle as O
from oracle import search_oracle as S
from tests.conftest import load_golden
from tests.helpers import SYSTEMS, oracle_cfg, rand_states


def test_uniforms_and_normals_of_the_counter_based_stream():
    key = S.cand_subkey(seed=11, env_id=np.arange(64), episode_idx=np.zeros(64, int), step_idx=np.arange(64))
    assert key.dtype == np.uint32 and key.shape == (64, 2) and len({tuple(k) for k in key}) == 64
    u = S.cand_uniforms(key, K=128, n_draws=3, round_=3)
    assert u.dtype == np.float32 and u.min() > 0.0 and u.max() < 1.0
    assert np.all(u * 65536 - 0.5 == np.round(u * 65536 - 0.5))  # 16-bit uniforms (m + 0.5) 2^-16, exact in float32
    xi = np.concatenate([S.cand_normals(key, 128, 3, 3).ravel(), S.held_normals(key, 512, 2, 3).ravel()])
    n = xi.size
    assert abs(xi.mean()) < 4 / np.sqrt(n) and abs(xi.std() - 1) < 4 / np.sqrt(2 * n)
    assert abs(np.mean(xi ** 4) - 3) < 0.1  # kurtosis of a normal
    assert np.max(np.abs(xi)) <= np.sqrt(-2 * np.log(2.0 ** -17)) + 1e-12  # 16-bit radius: |xi| <= 4.86
    # the seven-round generator itself: known answer of Philox4x32-7 (Random123's kat_vectors: counter = key = 0)
    from oracle.disturb_oracle import philox4x32_10
    z4, z2 = np.zeros((1, 4), np.uint32), np.zeros((1, 2), np.uint32)
    assert [int(v) for v in philox4x32_10(z4, z2, rounds=7)[0]] == [0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48]
    # a different round, tick or seed is a different stream; the same arguments give the same bits
    np.testing.assert_array_equal(u, S.cand_uniforms(key, 128, 3, 3))
    assert np.mean(u == S.cand_uniforms(key, 128, 3, 4)) < 1e-3
    key2 = S.cand_subkey(12, np.arange(64), np.zeros(64, int), np.arange(64))
    assert np.mean(key == key2) < 0.01


@pytest.mark.parametrize("name", SYSTEMS)
def test_sampling_rule(name):
    cfg = oracle_cfg(name, n_actor=7)
    B, K = 5, 96
    lo, hi = cfg.ctrl_bnds[:, 0], cfg.ctrl_bnds[:, 1]
    rng = np.random.default_rng(0)
    centre = rng.uniform(lo, hi, (B, 7, cfg.du))
    ids = 1000 + np.arange(B)
    for r in (0, 2):
        c = S.candidates_sample(cfg, 5, ids, np.ones(B, int), np.full(B, 9), K, r, centre=centre)
        assert c.shape == (B, K, 7, cfg.du) and np.all(c >= lo) and np.all(c <= hi)
        np.testing.assert_array_equal(c[:, 0], centre)
        if r == 0:
            np.testing.assert_array_equal(c[:, 1], np.broadcast_to(O.action_sqn_init(cfg), (B, 7, cfg.du)))
        p0 = S.ps_first(K)  # the last quarter of the candidates draws per step
        held = c[:, 2:p0] - centre[:, None]
        inside = np.all((c[:, 2:p0] > lo) & (c[:, 2:p0] < hi), axis=(2, 3))  # rows no clip touched
        assert inside.any()
        assert np.allclose(held[inside], held[inside][:, :1], atol=1e-9)  # one draw per input over the horizon
        per_step = c[:, p0:] - centre[:, None]
        assert np.std(per_step, axis=2).mean() > 0.01 * np.mean(hi - lo) * 2.0 ** -r
        # the spread halves with every round
        s = np.std(c[:, 2:] - centre[:, None])
        assert 0.15 * 2.0 ** -r < s / np.mean(hi - lo) < 0.6 * 2.0 ** -r
    # an env's candidates do not depend on the batch it is in (global env id), nor on its position
    one = S.candidates_sample(cfg, 5, ids[3:4], np.ones(1, int), np.full(1, 9), K, 2, centre=centre[3:4])
    np.testing.assert_array_equal(one[0], c[3])


@pytest.mark.parametrize("name", SYSTEMS)
def test_search_is_monotone_and_close_to_the_reference_slsqp(name):
    meta, z = load_golden(f"F8_slsqp_actor_{name}")
    cfg = oracle_cfg(name, n_actor=meta["N"], gamma=meta["gamma"], pred_step_size=meta["pred_step_size"])
    x = z["state"]
    B = len(x)
    ai = [0.5] if name == "2tank" else None
    args = (np.arange(B), np.zeros(B, int), np.zeros(B, int))
    Js = [S.actor_search(cfg, x, x, 256, r, 3, *args, action_init=ai)[1] for r in (1, 3, 6)]
    assert np.all(Js[1] <= Js[0]) and np.all(Js[2] <= Js[1])  # candidate 0 is the incumbent
    assert np.all(Js[0] <= z["J_init"] * (1 + 1e-12))          # candidate 1 of round 0 is the reference's start
    ratio = Js[2] / z["J_opt"]
    assert np.median(ratio) < 1.002 and np.max(ratio) < 1.02, (np.median(ratio), np.max(ratio))
    U, J, bi = S.actor_search(cfg, x, x, 256, 6, 3, *args, action_init=ai)
    np.testing.assert_allclose(J, O.actor_cost(U, x, x, cfg), rtol=1e-12)
