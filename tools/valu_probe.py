#!/usr/bin/env python3
"""The VALU-bound kernels of the path, a few launches each, for a counter pass:

    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES \
              SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU -d gpurun_out/prof_valu -o v --output-format csv \
              -- python3 tools/valu_probe.py

(the program itself after `--`; tools/prof_summary.py --valu condenses the CSV into profiles/<round>_valu_pmc.json and
profiles/valu_instr.json, which bench.py reads for `secondary.generated_grid.roofline_valu`).  Prints, per workload,
the units one launch processes - the denominators of "instructions per evaluation".

Workloads: generated-grid rollout (k_actor, C2 shape and configs[2] shape, MPC / RQL / SQL), the on-device optimiser
(k_actor_opt), the critic fit (k_critic_fit), the nominal controllers (k_nominal), the persistent multi-tick kernel
(k_ticks, k_ticks_pk), the device-side search (k_actor_search), the streamed SQL instance with 35 features, the per-GPU
shard of configs[4].
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def states(rng, name, n):
    if name == "3wrobot":
        return np.stack([rng.uniform(-10, 10, n), rng.uniform(-10, 10, n), rng.uniform(-np.pi, np.pi, n),
                         rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)], axis=-1)
    if name == "3wrobotNI":
        return np.stack([rng.uniform(-10, 10, n), rng.uniform(-10, 10, n), rng.uniform(-np.pi, np.pi, n)], axis=-1)
    return np.stack([rng.uniform(0, 2, n), rng.uniform(-2, 2, n)], axis=-1)


def main():
    import torch  # noqa: F401  (before the first Engine: torch's HIP runtime must be the one that initialises the GPU)

    from rcognita_amd import Engine
    from rcognita_amd.pool import MixedPool, preset_engine_config

    rng = np.random.default_rng(1234)
    n = int(os.environ.get("VALU_PROBE_LAUNCHES", "6"))
    part = sys.argv[1] if len(sys.argv) > 1 else "main"  # "ticks": B = 1024 persistent kernels; "pool": configs[4] alone (it re-uses kernel instances of "main"); "c3rql" / "c3sql"
    units = {}
    if part == "pool":
        total = 65536
        counts = {"3wrobot": total // 3 + total % 3, "3wrobotNI": total // 3, "2tank": total // 3}
        pool = MixedPool(counts, Nactor=15, dtype="f32")
        pool.set_states({s.name: states(rng, s.name, s.hi - s.lo) for s in pool.segments})
        for _ in range(n):
            pool.control_tick(256)
        pool.synchronize()
        for s in pool.segments:
            units[f"k_actor_generated_{s.name}_N15_f32_C5"] = {"evals": (s.hi - s.lo) * 256, "envs": s.hi - s.lo}
        pool.close()
        print(json.dumps({"launches_each": n, "units_per_launch": units}))
        return

    if part == "search":  # round 5: the device-side candidate search and the stand-alone producer, alone
        import time

        B, K = 65536, 256
        e = Engine(preset_engine_config("3wrobot", B, Nactor=10))
        e.set_stream(torch.cuda.current_stream().cuda_stream)
        e.set_state(states(rng, "3wrobot", B))
        cbuf = torch.empty((B, K, 10, 2), device="cuda", dtype=torch.float32)

        def timed(fn, reps):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3

        out = {}
        for rounds in (1, 4):
            out[f"search_tick_rounds_{rounds}_ms"] = timed(lambda: e.control_tick_search(K=K, rounds=rounds, warm_start=True), n * 4)
        out["cand_sample_ms"] = timed(lambda: e.candidates_sample(K, round=1, out=cbuf), n * 4)
        units["k_actor_search_3wrobot_N10_K256_f32"] = {"evals": B * K, "envs": B}
        print(json.dumps({"launches_each": n * 4 + 1, "units_per_launch": units, "timings": out}))
        return

    if part in ("c3rql", "c3sql"):  # configs[2] generated, ONE critic mode (RQL and SQL share a kernel instance)
        mode = part[2:].upper()
        B3, K = 131072, 256
        kw = dict(Nactor=20, mode=mode, critic_struct="quadratic", Ncritic=4, buffer_size=10)
        e = Engine(preset_engine_config("2tank", B3, **kw))
        e.set_state(states(rng, "2tank", B3))
        for _ in range(n):
            e.control_tick(None, K=K)
        e.synchronize()
        units[f"k_actor_generated_2tank_N20_{mode}_f32"] = {"evals": B3 * K, "envs": B3}
        e.close()
        print(json.dumps({"launches_each": n, "units_per_launch": units}))
        return

    def run(key, eng, tick, per_launch):
        for _ in range(n):
            tick()
        eng.synchronize()
        units[key] = per_launch
        eng.close()

    if part == "ticks":  # the persistent kernels at a small batch, alone (k_ticks_pk also serves the C2 tick of "main")
        Bs, T = 1024, 64
        e = Engine(preset_engine_config("3wrobot", Bs, Nactor=10))
        e.set_state(states(rng, "3wrobot", Bs))
        run("k_ticks_3wrobot_B1024_K64_T64_f32", e, lambda: e.control_ticks(T, 64), {"evals": Bs * 64 * T, "envs": Bs, "ticks": T})
        e = Engine(preset_engine_config("3wrobot", Bs, Nactor=10))
        e.set_state(states(rng, "3wrobot", Bs))
        run("k_ticks_pk_3wrobot_B1024_K256_T64_f32", e, lambda: e.control_ticks(T, 256),
            {"evals": Bs * 256 * T, "envs": Bs, "ticks": T})
        print(json.dumps({"launches_each": n, "units_per_launch": units}))
        return

    B, K = 65536, 256
    e = Engine(preset_engine_config("3wrobot", B, Nactor=10))
    e.set_state(states(rng, "3wrobot", B))
    run("k_actor_generated_3wrobot_N10_f32", e, lambda: e.control_tick(None, K=K), {"evals": B * K, "envs": B})

    B3 = 131072
    for mode in ("MPC", "RQL", "SQL"):
        kw = dict(Nactor=20, mode=mode, critic_struct="quadratic", Ncritic=4, buffer_size=10 if mode != "MPC" else 0)
        e = Engine(preset_engine_config("2tank", B3, **kw))
        e.set_state(states(rng, "2tank", B3))
        run(f"k_actor_generated_2tank_N20_{mode}_f32", e, lambda: e.control_tick(None, K=K), {"evals": B3 * K, "envs": B3})

    for iters in (5,):
        e = Engine(preset_engine_config("3wrobot", B, Nactor=10))
        e.set_state(states(rng, "3wrobot", B))
        run(f"k_actor_opt_3wrobot_N10_iters{iters}_f32", e, lambda: e.control_tick_opt(iters=iters),
            {"evals": B * iters * 17, "envs": B, "iters": iters})

    for name, gain in (("3wrobotNI", 0.5), ("3wrobot", 5.0)):
        e = Engine(preset_engine_config(name, B, Nactor=5))
        e.set_state(states(rng, name, B))
        run(f"k_nominal_{name}_f32", e, lambda: e.control_tick_nominal(gain), {"envs": B})

    # round 4: the device-side candidate search (one round of K = 256 Philox rows per env), the quasi-Newton optimiser in
    # a critic mode (memory 4), the streamed SQL instance with 35 features (VALU-bound although it streams)
    e = Engine(preset_engine_config("3wrobot", B, Nactor=10))
    e.set_state(states(rng, "3wrobot", B))
    run("k_actor_search_3wrobot_N10_K256_f32", e, lambda: e.control_tick_search(K=K, rounds=1, warm_start=True),
        {"evals": B * K, "envs": B})
    e = Engine(preset_engine_config("3wrobot", B, Nactor=10, mode="RQL", critic_struct="quad-nomix", buffer_size=10))
    e.set_state(states(rng, "3wrobot", B))
    run("k_actor_opt_3wrobot_N10_RQL_iters5_m4_f32", e, lambda: e.control_tick_opt(iters=5),
        {"evals": B * 5 * 17, "envs": B, "iters": 5})
    e = Engine(preset_engine_config("3wrobot", B, Nactor=10, mode="SQL", critic_struct="quad-lin", buffer_size=10))
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    e.set_state(states(rng, "3wrobot", B))
    cand = (torch.rand((B, K, 10, 2), device="cuda") * torch.tensor([600.0, 200.0], device="cuda")
            - torch.tensor([300.0, 100.0], device="cuda")).contiguous()
    run("k_actor_dma_sql_quadlin_3wrobot_N10_K256_f32", e, lambda: e.control_tick(cand, K=K), {"evals": B * K, "envs": B})

    print(json.dumps({"launches_each": n, "units_per_launch": units}))


if __name__ == "__main__":
    main()
