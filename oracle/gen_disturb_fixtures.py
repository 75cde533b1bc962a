#!/usr/bin/env python3
"""Generate tests/golden/F11_disturb_<system>.npz: ``System.closed_loop_rhs`` of the reference with ``is_disturb=1``
(rcognita/systems.py:213-253, 308-345, 370-394, 412-426) on seeded random full states.  Build container only; data
only.  Recipe as gen_fixtures.py, plus:

* the reference draws its noise with the unseeded global ``numpy.random.randn()`` INSIDE the right-hand side
  (systems.py:343, 392).  The generator replaces the module attribute ``rcognita.systems.randn`` by a function that
  replays a recorded sequence, so the noise becomes an explicit input ``xi`` of the fixture;
* under NumPy >= 1.25 ``disturb != []`` (systems.py:317, 373) raises for an ndarray; the full state is handed over as
  an ndarray subclass whose slices compare with ``[]`` the way old NumPy did (unequal), arithmetic untouched.

    python oracle/gen_disturb_fixtures.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_fixtures import PRESETS, import_reference, rand_actions, rand_states, save  # noqa: E402


class OldEqArray(np.ndarray):
    def __new__(cls, a):
        return np.asarray(a, dtype=float).view(cls)

    def __eq__(self, o):
        if isinstance(o, list) and not o:
            return False
        return np.asarray(self) == o

    def __ne__(self, o):
        if isinstance(o, list) and not o:
            return True
        return np.asarray(self) != o

    __hash__ = None


def main():
    systems, simulator, controllers = import_reference()
    rng = np.random.default_rng(20261111)
    n = 256
    for name, p in PRESETS.items():
        dd = p["dd"]
        sigma, mu, tau = rng.uniform(0.5, 3.0, dd), rng.uniform(-1.0, 1.0, dd), rng.uniform(0.2, 2.0, dd)
        sys_obj = getattr(systems, p["cls"])(
            sys_type="diff_eqn", dim_state=p["ds"], dim_input=p["du"], dim_output=p["ds"], dim_disturb=dd,
            pars=list(p["pars"]), ctrl_bnds=np.array(p["bnds"], dtype=float), is_dyn_ctrl=0, is_disturb=1,
            pars_disturb=[sigma, mu, tau])
        x = rand_states(rng, name, n)
        q = rng.normal(0.0, 5.0, (n, dd))
        u = rand_actions(rng, name, (n,), overshoot=1.5)
        xi = rng.standard_normal((n, dd))
        rhs = np.zeros((n, p["ds"] + dd))
        clipped = np.zeros((n, p["du"]))
        for i in range(n):
            seq = iter(xi[i])
            systems.randn = lambda: next(seq)  # replayed noise, in the order the reference draws it (k = 0, 1, ...)
            a = u[i].copy()
            sys_obj.receive_action(a)
            rhs[i] = sys_obj.closed_loop_rhs(0.0, OldEqArray(np.concatenate([x[i], q[i]])))
            clipped[i] = sys_obj.action
            assert sys_obj._dim_full_state == p["ds"] + dd
        save(f"F11_disturb_{name}", dict(system=name, pars=p["pars"], bnds=p["bnds"], dim_disturb=dd,
                                        note="rhs_full = closed_loop_rhs(0, [state, disturb]) with randn() replaced by xi"),
             state=x, disturb=q, action=u, xi=xi, sigma=sigma, mu=mu, tau=tau, rhs_full=rhs, action_clipped=clipped)


if __name__ == "__main__":
    main()
