"""Every kernel that steps an env leaves the same bits (GPU).  Simulator.sim_step is one device function (env_substeps -> rk4_step)
inlined into k_sim, k_actor_dma_packed (the fused env step of small-K ticks), k_ticks (T ticks per launch), k_critic_fit* (the critic
modes' tick), k_loop and k_actor_opt's LOOP instance (the drop-in loop): under -ffp-contract=fast the compiler once folded the last
multiply of the right-hand side into RK4's final sum in ONE of them (round 6, found by tools/fuzz_parity.py: one component in a
thousand one ulp apart).  The slopes are pinned now; this test holds every path to k_sim's bits on 4 096 random envs."""
import numpy as np
import pytest

from oracle import rcg_oracle as O
from tests.helpers import SYSTEMS, both, rand_actions, rand_states

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("name", SYSTEMS)
def test_env_step_bits_are_the_same_in_every_kernel(name, dtype):
    from rcognita_amd import _native as N

    rng = np.random.default_rng(5)
    B, Nh = 4096, 7
    x = rand_states(rng, name, B)
    a0 = rand_actions(rng, name, (B,), overshoot=1.2)

    def fresh(**kw):
        e, cfg = both(name, B, dtype, n_actor=Nh, **kw)
        e.set_state(x)
        e.set_field(N.FIELD_ACTION, a0)
        return e, cfg

    e, cfg = fresh()
    e.sim_step(1)
    ref = e.get_state().copy()
    ref_prev = e.get_field(N.FIELD_STATE_PREV).copy()
    e.close()
    np.testing.assert_array_equal(ref_prev, x.astype(ref.dtype))
    seen = set()
    for K in (8, 16, 40, 256):  # k_actor_dma_packed with its fused env step / k_sim + k_actor_dma
        cand = rand_actions(rng, name, (B, K, Nh), overshoot=1.2)
        e, _ = fresh()
        e.control_tick(cand, K=K)
        ll = e.last_launch()
        seen.add((ll["kernel"], bool(ll["variant"] & 16) if ll["kernel"] == "k_actor_dma_packed" else False))
        np.testing.assert_array_equal(e.get_state(), ref, err_msg=f"streamed tick K={K} ({ll})")
        e.close()
        e, _ = fresh()  # T ticks in one call: the state after the first step is gone, so compare two ticks against two ticks
        e.control_tick(cand, K=K, T=2)
        two = e.get_state().copy()
        ll2 = e.last_launch()
        e.close()
        e, _ = fresh()
        e.control_tick(cand, K=K)
        e.control_tick(cand, K=K)
        np.testing.assert_array_equal(two, e.get_state(), err_msg=f"2 ticks in one call K={K} ({ll2})")
        e.close()
    assert ("k_actor_dma_packed", True) in seen and ("k_actor_dma", False) in seen, seen
    # the generated grid (k_ticks_pk / k_actor), the critic modes' tick (k_critic_fit's env step), the loop step (k_loop, k_actor_opt LOOP)
    e, _ = fresh()
    e.control_tick(None, K=64 if cfg.du == 1 else 256)
    np.testing.assert_array_equal(e.get_state(), ref, err_msg=f"generated tick ({e.last_launch()})")
    e.close()
    for mode in ("RQL", "SQL"):
        e, _ = fresh(mode=O.MODE_IDS[mode], critic_struct=O.CRITIC_IDS["quad-mix"], buffer_size=6, n_critic=4)
        e.control_tick(rand_actions(rng, name, (B, 40, Nh), overshoot=1.2), K=40)
        np.testing.assert_array_equal(e.get_state(), ref, err_msg=f"{mode} tick ({e.last_launch(N.KERNEL_CRITIC)})")
        e.close()
    Bs = 128  # (rcg_loop_step is a small-batch entry point)
    e, _ = both(name, Bs, dtype, n_actor=Nh)
    e.set_state(x[:Bs])
    for decide in (False, True):
        e.set_state(x[:Bs])
        st, _, _, _, _ = e.loop_step(a0[:Bs], float(cfg.dt_sim), 1, decide=decide, iters=2)
        np.testing.assert_array_equal(st.astype(ref.dtype), ref[:Bs], err_msg=f"loop step decide={decide}")
    e.close()


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_vector_env_step_of_the_tank_is_the_same_bits(dtype):
    """k_sim_v (16 bytes per lane and component: the tank from 2^18 envs) against k_sim on the same envs."""
    from rcognita_amd import _native as N

    rng = np.random.default_rng(6)
    B = 1 << 18
    x = rand_states(rng, "2tank", B)
    a0 = rand_actions(rng, "2tank", (B,), overshoot=1.2)
    big, _ = both("2tank", B, dtype, n_actor=3)
    big.set_state(x)
    big.set_field(N.FIELD_ACTION, a0)
    big.sim_step(2)
    assert big.last_launch(N.KERNEL_SIM)["kernel"] == "k_sim_v", big.last_launch(N.KERNEL_SIM)
    got = big.get_state()
    big.close()
    Bs = 1 << 16
    small, _ = both("2tank", Bs, dtype, n_actor=3)
    for i in range(0, B, Bs):
        small.set_state(x[i:i + Bs])
        small.set_field(N.FIELD_STATUS, np.zeros(Bs, dtype=np.uint32))
        small.set_field(N.FIELD_ACTION, a0[i:i + Bs])
        small.sim_step(2)
        assert small.last_launch(N.KERNEL_SIM)["kernel"] == "k_sim"
        np.testing.assert_array_equal(got[i:i + Bs], small.get_state(), err_msg=f"envs {i} ..")
    small.close()


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("name", SYSTEMS)
def test_disturbed_env_step_bits_in_one_launch_and_in_single_ticks(name, dtype):
    """The env step with the disturbance model (rk4_step_full on [state, disturb]) inside k_ticks against k_sim_dist, 4 096 envs:
    STATE and DISTURB after 3 ticks in one launch = after 3 single ticks (tests/test_hip_ticks.py holds the same on 261 envs; a
    one-in-a-thousand rounding needs more)."""
    from rcognita_amd import _native as N

    rng = np.random.default_rng(9)
    B, T, Nh, K = 4096, 3, 5, 64
    dist = dict(is_disturb=True, pars_disturb=[[30.0, 10.0], [0.5, -0.2], [2.0, 1.5]], disturb_init=[1.0, -2.0], seed=31,
                env_id_base=5_000_000_000)
    x0 = rand_states(rng, name, B)
    out = []
    for one_launch in (True, False):
        e, _ = both(name, B, dtype, n_actor=Nh, substeps_per_tick=2, engine_only=dist)
        e.set_state(x0)
        if one_launch:
            e.control_ticks(T, K)
            assert e.last_launch()["kernel"] == "k_ticks"
        else:
            for _ in range(T):
                e.control_tick(None, K=K)
            assert e.last_launch(N.KERNEL_SIM)["kernel"] == "k_sim_dist"
        out.append((e.get_state().copy(), e.get_field(N.FIELD_DISTURB).copy()))
        e.close()
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])
