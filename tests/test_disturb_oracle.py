"""oracle/disturb_oracle.py: the disturbed right-hand side against the reference's outputs
(tests/golden/F11_disturb_*.npz), and the counter-based generator against the published Philox4x32-10 known answers
(Random123 kat_vectors).  CPU only."""
import numpy as np
import pytest

from oracle import disturb_oracle as DO
from oracle import rcg_oracle as O
from tests.conftest import load_golden
from tests.helpers import PRESETS, SYSTEMS


def test_philox4x32_10_known_answers():
    kat = [
        ((0x00000000,) * 4, (0x00000000,) * 2, (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
        ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
        ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
         (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
    ]
    ctr = np.array([k[0] for k in kat], dtype=np.uint32)
    key = np.array([k[1] for k in kat], dtype=np.uint32)
    out = DO.philox4x32_10(ctr, key)
    np.testing.assert_array_equal(out, np.array([k[2] for k in kat], dtype=np.uint32))


def test_noise_is_a_function_of_seed_env_episode_substep_only():
    ids = np.arange(1000, dtype=np.int64)
    z = np.zeros(1000, dtype=np.int32)
    a = DO.disturb_noise(7, ids, z, z + 3)
    # an env's draw does not depend on which batch / shard it sits in
    b = DO.disturb_noise(7, ids[500:], z[500:], z[500:] + 3)
    np.testing.assert_array_equal(a[500:], b)
    assert not np.array_equal(a, DO.disturb_noise(8, ids, z, z + 3))
    assert not np.array_equal(a, DO.disturb_noise(7, ids, z + 1, z + 3))
    assert not np.array_equal(a, DO.disturb_noise(7, ids, z, z + 4))
    # 64-bit env ids reach the second counter word
    assert not np.array_equal(DO.noise_bits(7, ids, z, z), DO.noise_bits(7, ids + (1 << 32), z, z))
    # standard normal moments over 2e5 draws
    big = DO.disturb_noise(1, np.arange(100000, dtype=np.int64), np.zeros(100000, np.int32), np.zeros(100000, np.int32))
    assert abs(big.mean()) < 0.01 and abs(big.std() - 1) < 0.01 and np.all(np.isfinite(big))
    assert abs(np.mean(big[:, 0] * big[:, 1])) < 0.01


@pytest.mark.parametrize("name", SYSTEMS)
def test_F11_disturbed_rhs_matches_reference(name):
    meta, z = load_golden(f"F11_disturb_{name}")
    sys_id = PRESETS[name]["sys_id"]
    pars = np.array(meta["pars"], dtype=np.float64) if len(meta["pars"]) else np.zeros(0)
    bnds = np.array(meta["bnds"], dtype=np.float64)
    dx, dq, a = DO.closed_loop_rhs_full(sys_id, z["state"], z["disturb"], z["action"], z["xi"], pars, bnds,
                                        z["sigma"], z["mu"], z["tau"])
    ds = z["state"].shape[1]
    np.testing.assert_allclose(dx, z["rhs_full"][:, :ds], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(dq, z["rhs_full"][:, ds:], rtol=1e-13, atol=1e-13)
    np.testing.assert_array_equal(a, z["action_clipped"])
    assert DO.DIM_DISTURB[sys_id] == meta["dim_disturb"] == z["disturb"].shape[1]
    if name == "2tank":
        assert np.all(z["rhs_full"][:, ds:] == 0)  # the reference's 2tank disturbance is inert
    else:
        assert np.any(np.abs(z["action"]) > bnds[:, 1])  # the clip was exercised


def test_ou_process_statistics_of_the_substep_scheme():
    """Holding xi over a substep makes q an AR(1) process; its stationary mean is -sigma*mu, as for the SDE."""
    cfg_sys = O.SYS_3WROBOT_NI
    sigma, mu, tau = np.array([2.0, 1.0]), np.array([0.5, -0.25]), np.array([1.5, 0.7])
    B, T, dt = 4000, 400, 0.05
    q = np.zeros((B, 2))
    x = np.zeros((B, 3))
    u = np.zeros((B, 2))
    ids = np.arange(B, dtype=np.int64)
    ep = np.zeros(B, dtype=np.int32)
    for t in range(T):
        xi = DO.disturb_noise(3, ids, ep, np.full(B, t, dtype=np.int32))
        x, q = DO.rk4_step_full(cfg_sys, x, q, u, xi, np.zeros(0), np.zeros((2, 2)), sigma, mu, tau, dt)
    np.testing.assert_allclose(q.mean(axis=0), -sigma * mu, atol=0.12)
    assert np.all(np.isfinite(x))
