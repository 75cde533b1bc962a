"""CPU twin of tests/test_hip_teacher_forced.py: every control tick of the reference's critic-mode closed loops (F7c)
replayed teacher-forced through the ORACLE's fit (twin of k_critic_fit) and optimiser (twin of k_actor_opt) - the same
three assertions per tick (tests/teacher_forced.py): Jc against SLSQP's on the reference's TD stack, J against SLSQP's at
the reference's weights, and the first action wherever the reference's own cost is measurably sharp in it."""
import pytest

from oracle import rcg_oracle as O
from tests import teacher_forced as TF
from tests.conftest import load_golden
from tests.test_critic_traces import CASES, MODES, trace_cfg


@pytest.mark.parametrize("ftol", [0.0, 1e-7], ids=["run-to-the-end", "ftol-of-the-mirror-classes"])
@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name,cs", CASES)
def test_teacher_forced_replay_of_the_reference_loop_through_the_oracle(name, cs, mode, ftol):
    meta, z = load_golden(f"F7c_trace_{name}_{mode}_{cs}")
    cfg = trace_cfg(meta)
    u0 = O.action_sqn_init(cfg, [0.5] if name == "2tank" else None)
    tally = TF.Tally(f"(oracle) {name} {mode} {cs}")
    n = len(z["tick_t"])
    # the long runs: every third tick (the GPU test replays all of them)
    for i in range(0, n, 3 if n > 60 else 1):
        w = None
        if z["tick_fitted"][i]:
            w = O.critic_fit(cfg, z["tick_w_prev"][i][None], z["tick_obs_buf"][i][None], z["tick_act_buf"][i][None])[0]
        u, _, _ = O.actor_optimize_single(cfg, z["tick_obs"][i], z["tick_state_sys"][i], u0, 30, w_critic=z["tick_w"][i], ftol=ftol)
        TF.check_tick(tally, cfg, z, i, w, u, meta["first_fracs"])
    print("\n" + tally.line())
    assert not tally.failures, "\n".join(tally.failures[:10])
    assert tally.n_sharp > 0, "no tick of this trace pins the first action: the fixture cannot falsify the actor"
