"""BASELINE configs[0] (the reference's own CPU-runnable case) and SURVEY 8a row 26: the reference's loop,
restated around the oracle (oracle/ref_loop.py: SciPy RK45 + SLSQP + the reference's loop quirks), must
reproduce the closed-loop traces captured from the reference itself (tests/golden/F7_trace_*.npz).  CPU-only.

Tolerance: the loop contains SLSQP with 2-point finite differences (step sqrt(eps)), which amplifies last-bit
differences of the cost by ~1e8; time stamps and pre-decision states agree to 1e-12, everything downstream of
the optimiser to 1e-6 relative."""
import numpy as np
import pytest

from oracle import rcg_oracle as O
from oracle.ref_loop import RefLoop
from tests.conftest import load_golden
from tests.helpers import PRESETS, oracle_cfg


@pytest.mark.parametrize("name,mode", [("3wrobotNI", "MPC"), ("3wrobot", "MPC"), ("2tank", "MPC"), ("2tank", "RQL")])
def test_reference_loop_trace(name, mode):
    meta, z = load_golden(f"F7_trace_{name}_{mode}")
    ref = z["rows"]
    p = PRESETS[name]
    cfg = oracle_cfg(name, n_actor=meta["Nactor"], mode=O.MODE_IDS[mode], gamma=1.0, critic_struct=O.CRITIC_QUAD_NOMIX,
                     n_critic=4, buffer_size=10)
    loop = RefLoop(cfg, np.array(p["x0"], dtype=float), meta["t1"], action_init=[0.5] if name == "2tank" else None)
    rows = loop.run()
    assert rows.shape == ref.shape, (rows.shape, ref.shape)  # same number of sim steps: same accept/reject history
    ds = cfg.ds
    first_tick = int(np.argmax(ref[:, 0] >= meta["dt"]))  # rows before it never saw an optimiser result
    # RK45 time grid and states up to the first decision: untouched by the optimiser -> exact
    np.testing.assert_allclose(rows[: first_tick + 1, 0], ref[: first_tick + 1, 0], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(rows[: first_tick + 1, 1:1 + ds], ref[: first_tick + 1, 1:1 + ds], rtol=1e-12, atol=1e-14)
    # afterwards the step-size controller sees the optimiser's actions: same tolerance as everything downstream
    np.testing.assert_allclose(rows[:, 0], ref[:, 0], rtol=1e-6, atol=1e-9)
    scale = np.maximum(np.abs(ref), 1.0)
    err = np.max(np.abs(rows - ref) / scale)
    assert err < 1e-6, err


@pytest.mark.parametrize("name", ["3wrobot", "3wrobotNI", "2tank"])
def test_reference_loop_long_trace_and_the_survey_quality_datapoint(name):
    """Whole seconds of the reference's loop (fixture F7_long, oracle/gen_f7_long_fixture.py: 3 s of the robots, 20 s of
    the tank - 400 to 600 simulation steps, 300 SLSQP decisions): the restated loop follows the reference's trace to the
    same tolerance as the short ones (measured here: bit for bit), and the 3wrobot run ends on the SURVEY's section-6
    datapoint, accum_obj 389.0."""
    meta, z = load_golden(f"F7_long_{name}_MPC")
    ref = z["rows"]
    cfg = oracle_cfg(name, n_actor=meta["Nactor"], mode=O.MODE_MPC, gamma=1.0, critic_struct=O.CRITIC_QUAD_NOMIX,
                     n_critic=4, buffer_size=10)
    loop = RefLoop(cfg, np.array(PRESETS[name]["x0"], dtype=float), meta["t1"],
                   action_init=[0.5] if name == "2tank" else None)
    rows = loop.run()
    assert rows.shape == ref.shape
    assert np.max(np.abs(rows - ref) / np.maximum(np.abs(ref), 1.0)) < 1e-6
    if name == "3wrobot":
        assert abs(ref[-1, -1] - 389.0) < 0.05 and abs(rows[-1, -1] - 389.0) < 0.05


@pytest.mark.parametrize("name", ["3wrobot", "3wrobotNI", "2tank"])
@pytest.mark.parametrize("mode", ["MPC", "RQL", "SQL"])
def test_ref_loop_callbacks_equal_the_oracle(name, mode):
    """The loop's reference-ordered single-env callbacks are the same functions as the batched oracle's."""
    rng = np.random.default_rng(3)
    cfg = oracle_cfg(name, n_actor=6, mode=O.MODE_IDS[mode], gamma=0.93, critic_struct=O.CRITIC_QUADRATIC,
                     n_critic=4, buffer_size=8)
    x = np.array(PRESETS[name]["x0"], dtype=float)
    loop = RefLoop(cfg, x, 1.0)
    loop.state_sys = x + 0.01
    loop.w = rng.uniform(0, 2, cfg.dc)
    loop.w_prev = rng.uniform(0, 2, cfg.dc)
    b = np.array(PRESETS[name]["bnds"], dtype=float)
    sqn = rng.uniform(b[:, 0], b[:, 1], (6, cfg.du)).reshape(-1)
    np.testing.assert_allclose(loop._actor_cost(sqn, x), O.actor_cost(sqn, x, x + 0.01, cfg, w_critic=loop.w), rtol=1e-12)
    loop.obs_buf = x + rng.uniform(-1, 1, (8, cfg.ds))
    loop.act_buf = rng.uniform(b[:, 0], b[:, 1], (8, cfg.du))
    w = rng.uniform(0, 2, cfg.dc)
    np.testing.assert_allclose(loop._critic_cost(w), O.critic_cost(w, loop.w_prev, loop.obs_buf, loop.act_buf, cfg),
                               rtol=1e-12)
