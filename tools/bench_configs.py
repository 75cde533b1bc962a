#!/usr/bin/env python3
"""Secondary measurements on one MI355X for the BASELINE configs that are parity cases rather than the
bench line: configs[2] (Sys2Tank B=131072, Nactor=20, RQL + quadratic critic TD fit every tick) and the
per-GPU shard of configs[4] (mixed pool, Nactor=15, 256 generated candidates).  Prints one JSON object.

    python tools/bench_configs.py [--steps 50] [--compare profiles/r02_bench_configs.json]

--compare: every rate / kernel time of this run against a stored run, on stderr; exit code 3 if anything is more than
10 % worse (a regression guard for changes to shared device code: in round 2 a rounding fix in Sys2Tank::rhs slowed the
generated-candidate kernels of configs[2] by 20 % and went unnoticed until the next profile run).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=120,
                    help="untimed ticks per case: the GPU needs ~20 ms of continuous work to reach its steady clock "
                         "(tools/clock_ramp.py) and the RQL / SQL buffers ~10 ticks to fill")
    ap.add_argument("--compare", default=None, help="stored output of this tool to compare with")
    a = ap.parse_args()
    from rcognita_amd import Engine
    from rcognita_amd import _native as N
    from rcognita_amd.pool import MixedPool, preset_engine_config

    out = {}
    rng = np.random.default_rng(1234)

    # ---- configs[2] ----------------------------------------------------------------------------------
    B, K, Nh = 131072, 256, 20
    for mode in ("MPC", "RQL", "SQL"):
        kw = dict(Nactor=Nh, mode=mode, critic_struct="quadratic", Ncritic=4, buffer_size=10 if mode != "MPC" else 0)
        eng = Engine(preset_engine_config("2tank", B, **kw))
        eng.set_state(np.stack([rng.uniform(0, 2, B), rng.uniform(-2, 2, B)], axis=-1))
        for _ in range(a.warmup):
            eng.control_tick(None, K=K)
        eng.profile(True)
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            eng.control_tick(None, K=K)
        eng.synchronize()
        dt = time.perf_counter() - t0
        am, an = eng.profile_read(N.KERNEL_ACTOR)
        cm, cn = eng.profile_read(N.KERNEL_CRITIC)
        sm, sn = eng.profile_read(N.KERNEL_SIM)
        summ, _ = eng.episode_stats(from_accum=True)
        out[f"C3_2tank_B{B}_N{Nh}_K{K}_{mode}"] = {
            "env_control_steps_per_s": B * a.steps / dt, "ms_per_tick": dt / a.steps * 1e3,
            "actor_ms": am / max(an, 1), "critic_push_fit_ms": cm / max(cn, 1) if cn else None,
            "sim_ms": sm / max(sn, 1), "n_failed": summ["n_failed"], "candidates": "generated 256-level grid"}
        eng.close()

    # ---- configs[2] with STREAMED candidates (2.7 GB per tick): MPC / RQL on k_actor_dma, SQL on k_actor --
    cand_host = rng.random((B, K, Nh, 1), dtype=np.float32)
    for mode in ("MPC", "RQL", "SQL"):
        kw = dict(Nactor=Nh, mode=mode, critic_struct="quadratic", Ncritic=4, buffer_size=10 if mode != "MPC" else 0)
        eng = Engine(preset_engine_config("2tank", B, **kw))
        eng.set_state(np.stack([rng.uniform(0, 2, B), rng.uniform(-2, 2, B)], axis=-1))
        cand = eng.to_device(cand_host)
        for _ in range(a.warmup + 5):
            eng.control_tick(cand, K=K)
        eng.profile(True)
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            eng.control_tick(cand, K=K)
        eng.synchronize()
        dt = time.perf_counter() - t0
        am, an = eng.profile_read(N.KERNEL_ACTOR)
        out[f"C3_2tank_B{B}_N{Nh}_K{K}_{mode}_streamed"] = {
            "env_control_steps_per_s": B * a.steps / dt, "ms_per_tick": dt / a.steps * 1e3, "actor_ms": am / max(an, 1),
            "actor_GBps": B * K * Nh * 4 / (am / max(an, 1) * 1e-3) / 1e9}
        eng.close()
        del cand
    del cand_host

    # ---- on-device optimiser tick on the C2 shape ------------------------------------
    B2 = 65536
    for iters in (5, 10):
        eng = Engine(preset_engine_config("3wrobot", B2, Nactor=10))
        eng.set_state(np.stack([rng.uniform(-10, 10, B2), rng.uniform(-10, 10, B2), rng.uniform(-np.pi, np.pi, B2),
                                rng.uniform(-1, 1, B2), rng.uniform(-1, 1, B2)], axis=-1))
        for _ in range(a.warmup):
            eng.control_tick_opt(iters=iters)
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            eng.control_tick_opt(iters=iters)
        eng.synchronize()
        dt = time.perf_counter() - t0
        used = eng.get_field(N.FIELD_BEST_IDX)
        out[f"C2_3wrobot_B{B2}_N10_optimizer_iters{iters}"] = {
            "env_control_steps_per_s": B2 * a.steps / dt, "ms_per_tick": dt / a.steps * 1e3,
            "mean_iterations_used": float(used.mean()),
            "note": "adjoint gradient + 16-way line search per iteration (17 _actor_cost-sized rollouts each)"}
        eng.close()

    # ---- the operator boundary itself (unit U1 of SURVEY 8d): rcg_actor_cost, J for every candidate ------
    import ctypes as C

    Kc, Nc = 256, 10
    eng = Engine(preset_engine_config("3wrobot", B2, Nactor=Nc))
    eng.set_state(np.stack([rng.uniform(-10, 10, B2), rng.uniform(-10, 10, B2), rng.uniform(-np.pi, np.pi, B2),
                            rng.uniform(-1, 1, B2), rng.uniform(-1, 1, B2)], axis=-1))
    bn = np.array([[-300, 300], [-100, 100]], dtype=np.float32)
    cand = eng.to_device(bn[:, 0] + (bn[:, 1] - bn[:, 0]) * rng.random((B2, Kc, Nc, 2), dtype=np.float32))
    Jd = eng.empty((B2, Kc))
    call = lambda: N.check(N.lib().rcg_actor_cost(eng._h, C.c_void_p(cand.ptr), Kc, None, None, None, C.c_void_p(Jd.ptr)),
                           eng._h)
    for _ in range(a.warmup + 20):
        call()
    eng.profile(True)
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps * 4):
        call()
    eng.synchronize()
    dt = time.perf_counter() - t0
    am, an = eng.profile_read(N.KERNEL_ACTOR)
    # the tick with heterogeneous parameters (m ~ U(5, 20), I ~ U(0.5, 2) per env, [np][B] loads)
    ech = preset_engine_config("3wrobot", B2, Nactor=Nc)
    ech.per_env_pars = True
    engh = Engine(ech)
    engh.set_field(N.FIELD_PARS, np.stack([rng.uniform(5, 20, B2), rng.uniform(0.5, 2, B2)], axis=-1))
    engh.set_state(eng.get_state())
    for _ in range(a.warmup + 20):
        engh.control_tick(cand, K=Kc)
    engh.profile(True)
    engh.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps * 4):
        engh.control_tick(cand, K=Kc)
    engh.synchronize()
    dth = time.perf_counter() - t0
    amh, anh = engh.profile_read(N.KERNEL_ACTOR)
    out[f"C2_3wrobot_B{B2}_N{Nc}_K{Kc}_streamed_per_env_pars"] = {
        "env_control_steps_per_s": B2 * a.steps * 4 / dth, "ms_per_tick": dth / (a.steps * 4) * 1e3,
        "actor_ms": amh / max(anh, 1)}
    engh.close()
    per_eval = 4 * (Nc * 2 + 5 + 5 + 1)  # 4 (N du + ds + dy + 1) = 124 B; obs == state_sys here: read once
    out[f"U1_actor_cost_operator_3wrobot_B{B2}_K{Kc}_N{Nc}_streamed"] = {
        "actor_cost_evals_per_s": B2 * Kc * a.steps * 4 / dt, "kernel_ms": am / max(an, 1),
        "kernel_GBps_moved": (B2 * Kc * (Nc * 2 * 4 + 4) + B2 * 20) / (am / max(an, 1) * 1e-3) / 1e9,
        "survey_bytes_per_eval": per_eval}
    eng.close()
    del cand, Jd

    # ---- SURVEY 8f rows f3 / f4 on the C2 batch ---------------------------------------------------------
    def timed(eng, tick):
        for _ in range(a.warmup):
            tick()
        eng.profile(True)
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            tick()
        eng.synchronize()
        dt = time.perf_counter() - t0
        am, an = eng.profile_read(N.KERNEL_ACTOR)
        sm, sn = eng.profile_read(N.KERNEL_SIM)
        summ, _ = eng.episode_stats(from_accum=True)
        return {"env_control_steps_per_s": eng.B * a.steps / dt, "ms_per_tick": dt / a.steps * 1e3,
                "decision_kernel_ms": am / max(an, 1), "sim_ms": sm / max(sn, 1), "n_failed": summ["n_failed"]}

    from tests.helpers import rand_states

    for name, gain in (("3wrobotNI", 0.5), ("3wrobot", 5.0)):  # preset gains (main_3wrobot_NI.py:235, main_3wrobot.py:239)
        eng = Engine(preset_engine_config(name, B2, Nactor=5))
        eng.set_state(rand_states(rng, name, B2))
        out[f"f3_nominal_tick_{name}_B{B2}"] = timed(eng, lambda: eng.control_tick_nominal(gain))
        eng.close()
    ec = preset_engine_config("3wrobot", B2, Nactor=10)
    ec.is_disturb, ec.pars_disturb, ec.seed = True, [[2.0, 1.0], [0.5, -0.25], [1.5, 0.7]], 4
    eng = Engine(ec)
    eng.set_state(rand_states(rng, "3wrobot", B2))
    out[f"f4_disturbed_MPC_tick_3wrobot_B{B2}_N10_K256_generated"] = timed(eng, lambda: eng.control_tick(None, K=256))
    eng.close()

    # ---- the pure env step (Simulator.sim_step = k_sim) where it is bandwidth-bound: 2^24 envs ----------------
    for name, ds, du in (("3wrobot", 5, 2), ("2tank", 2, 1)):
        Bs = 1 << 24
        eng = Engine(preset_engine_config(name, Bs, Nactor=3))
        x = rng.uniform(-1, 1, (Bs, ds)).astype(np.float32)
        eng.set_state(np.abs(x) + 0.1 if name == "2tank" else x)
        eng.set_field(N.FIELD_ACTION, np.full((Bs, du), 0.3, np.float32))
        del x
        for _ in range(20):
            eng.sim_step(1)
        eng.profile((N.KERNEL_SIM,), stride=1)
        eng.synchronize()
        for _ in range(50):
            eng.sim_step(1)
        eng.synchronize()
        ms, n = eng.profile_read(N.KERNEL_SIM)
        bytes_env = (3 * ds + du) * 4 + 4  # read state, action, status; write state, state_prev
        out[f"sim_step_only_{name}_B{Bs}"] = {"env_steps_per_s": Bs / (ms / n * 1e-3), "kernel_ms": ms / n,
                                              "kernel_GBps": Bs * bytes_env / (ms / n * 1e-3) / 1e9,
                                              "bytes_per_env_step": bytes_env}
        eng.close()

    # ---- configs[4], one GPU's shard -----------------------------------------------------------------
    total = 65536
    counts = {"3wrobot": total // 3 + total % 3, "3wrobotNI": total // 3, "2tank": total // 3}
    pool = MixedPool(counts, Nactor=15, dtype="f32")
    from tests.helpers import rand_states

    pool.set_states({s.name: rand_states(rng, s.name, s.hi - s.lo) for s in pool.segments})
    for _ in range(a.warmup):
        pool.control_tick(256)
    pool.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        pool.control_tick(256)
    pool.synchronize()
    dt = time.perf_counter() - t0
    summ, _ = pool.episode_stats(from_accum=True)
    out["C5_mixed_pool_65536_N15_K256_generated"] = {"env_control_steps_per_s": pool.n_envs * a.steps / dt,
                                                    "ms_per_tick": dt / a.steps * 1e3, "n_failed": summ["n_failed"]}
    pool.close()
    print(json.dumps(out, indent=1))
    if a.compare:
        sys.exit(compare(out, json.load(open(a.compare))))


def compare(new, old, tol=0.10):
    """Rates (higher is better) and times (lower is better) of `new` against `old`; 3 if any is > tol worse."""
    worse = 0
    for k, e in new.items():
        o = old.get(k)
        if not isinstance(e, dict) or not isinstance(o, dict):
            continue
        for f, v in e.items():
            ov = o.get(f)
            if not isinstance(v, (int, float)) or not isinstance(ov, (int, float)) or not v or not ov:
                continue
            if f.endswith("_per_s") or f.endswith("GBps") or f.endswith("GBps_moved"):
                r = v / ov
            elif f.endswith("_ms"):
                r = ov / v
            else:
                continue
            flag = "  <-- WORSE" if r < 1 - tol else ""
            worse += bool(flag)
            print(f"{k:62s} {f:26s} {ov:12.5g} -> {v:12.5g}  x{r:.3f}{flag}", file=sys.stderr)
    return 3 if worse else 0


if __name__ == "__main__":
    main()
