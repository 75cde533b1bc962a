"""oracle/nominal_oracle.py against outputs of the reference's nominal controllers (tests/golden/F10_nominal_*.npz,
made by oracle/gen_nominal_fixtures.py).  CPU only."""
import numpy as np

from oracle import nominal_oracle as NO
from tests.conftest import load_golden


def _close(a, b, rtol=1e-11, atol=1e-12):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, equal_nan=True)


def test_F10_ni_closed_form_matches_reference():
    meta, z = load_golden("F10_nominal_3wrobotNI")
    x = z["state"]
    with np.errstate(all="ignore"):
        xNI = NO.cart2nh(x)
        _close(xNI, z["xNI"])
        zeta = NO.zeta_ni(xNI)
        _close(zeta, z["zeta"])
        _close(NO.kappa_of(xNI, zeta), z["kappa"])
    _close(NO.nominal_action_ni(x, meta["ctrl_gain"]), z["action_vanila"])
    _close(NO.nominal_action_ni(x, meta["ctrl_gain"], meta["bnds"]), z["action"])
    _close(NO.lyapunov_ni(x), z["LF"])
    # the special rows really exercise the other branch / the clip
    assert np.all((z["xNI"][:8, 0] == 0) & (z["xNI"][:8, 1] == 0))
    assert np.any(np.abs(z["action_vanila"]) > np.array(meta["bnds"])[:, 1])
    assert np.all(np.isnan(z["action"][8:12]))  # the reference returns NaN at the exact origin; so does the oracle


def test_F10_endi_operators_match_reference():
    meta, z = load_golden("F10_nominal_3wrobot")
    x = z["state"]
    xNI, eta = NO.cart2nh(x)
    _close(xNI, z["xNI"])
    _close(eta, z["eta"])
    for j in range(z["theta"].shape[1]):
        th = z["theta"][:, j]
        with np.errstate(all="ignore"):
            zt = NO.zeta_theta(xNI, th)
            _close(zt, z["zeta"][:, j], rtol=1e-10)
            _close(NO.kappa_of(xNI, zt), z["kappa"][:, j], rtol=1e-10)
            _close(NO.Fc(xNI, eta, th), z["Fc"][:, j], rtol=1e-10)
    uNI = np.tile(np.array(meta["uNI_probe"]), (x.shape[0], 1))
    _close(NO.nh2ctrl_cart(xNI, eta, uNI, meta["m"], meta["I"]), z["uCart_probe"])
    # with the REFERENCE's theta* the whole controller is reproduced
    _close(NO.nominal_action_endi(x, meta["ctrl_gain"], meta["m"], meta["I"], theta=z["theta_star"]),
           z["action_vanila"], rtol=1e-9)
    _close(NO.nominal_action_endi(x, meta["ctrl_gain"], meta["m"], meta["I"], meta["bnds"], theta=z["theta_star"]),
           z["action"], rtol=1e-9)
    _close(NO.lyapunov_endi(x, theta=z["theta_star"]), z["LF"], rtol=1e-9)


def test_F10_endi_theta_search_vs_reference_trust_constr():
    """Build-defined theta* (round 6: compass search from theta = 0) vs the reference's SciPy trust-constr from theta = 0: the
    same minimiser on 94.8 % of the fixture states (where trust-constr stopped short of its basin's minimum, or jumped a
    barrier, they differ), the same action on 96.9 % of ALL states, and Fc never above the reference's (rounds 2-5, grid walk
    + golden section: 92.7 % / 94.8 % / 96.9 %)."""
    meta, z = load_golden("F10_nominal_3wrobot")
    x = z["state"]
    xNI, eta = NO.cart2nh(x)
    th = NO.theta_star(xNI, eta)
    assert np.all((th >= -np.pi) & (th <= np.pi))
    with np.errstate(all="ignore"):
        F_ours, F_ref = NO.Fc(xNI, eta, th), z["Fc_star"]
    assert np.mean(F_ours <= F_ref * (1 + 1e-9) + 1e-12) >= 0.99
    d = np.abs(np.angle(np.exp(1j * (th - z["theta_star"]))))
    same = d < 1e-3
    assert same.mean() > 0.94, same.mean()
    u = NO.nominal_action_endi(x, meta["ctrl_gain"], meta["m"], meta["I"], meta["bnds"])
    close = np.all(np.abs(u - z["action"]) <= 2e-2 * (np.abs(z["action"]) + 1), axis=1)
    assert close[same].mean() > 0.95, close[same].mean()
    assert close.mean() > 0.96, close.mean()  # over ALL states, not only where the minimisers agree
    print("same minimiser:", same.mean(), " strictly better Fc:", np.mean(F_ours < F_ref * (1 - 1e-6)))
