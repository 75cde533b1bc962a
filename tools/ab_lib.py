#!/usr/bin/env python3
"""A/B of two builds of librcg on one GPU, interleaved child processes (cdna_hip_programming.md 5.4 rule 24: perf deltas
come from interleaved rounds on ONE device - devices differ by up to 12 % on VALU-bound kernels):

    make ab ABFLAGS="-DRCG_AB_..."          # rcognita_amd/lib/librcg_ab.so
    python tools/ab_lib.py [--a lib/librcg.so] [--b lib/librcg_ab.so] [--rounds 3] [workload ...]
    python tools/ab_lib.py --a rcognita_amd/lib/librcg_dev.so --b rcognita_amd/lib/librcg_dev.so --b-env RCG_NO_PK=1 gen ticks

Workloads (median / min of the per-launch durations the dispatches carry, us):
  gen      generated 256-level grid, C2 shape (k_actor)          gen_c3   2tank N = 20 RQL generated
  opt0/opt4 k_actor_opt, C2 shape, 5 iterations, memory 0 / 4    ticks    k_ticks B = 1024, K = 64, T = 64 (ticks256: K = 256)
  search   k_actor_search C2 shape, one round                    fit      k_critic_fit, configs[2] in closed loop
  stream   k_actor_dma, C2 (the headline kernel)                 sql      streamed SQL quad-lin (VALU-bound instance)
"""
import argparse
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ALL = ["gen", "gen_c3", "opt0", "opt4", "ticks", "ticks256", "search", "fit", "stream", "sql"]


def child(lib, workload):
    from rcognita_amd import _native as N

    N.use_library(lib)
    import torch

    from rcognita_amd import Engine
    from rcognita_amd.pool import preset_engine_config

    rng = np.random.default_rng(1)
    kind = N.KERNEL_ACTOR

    def st3(n):
        return np.stack([rng.uniform(-10, 10, n), rng.uniform(-10, 10, n), rng.uniform(-np.pi, np.pi, n),
                         rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)], -1)

    B, K, Nh = 65536, int(os.environ.get("AB_K", "256")), 10
    if workload == "gen":
        B = int(os.environ.get("AB_B", B))
    if workload in ("gen", "opt0", "opt4", "search", "stream", "sql"):
        kw = {}
        if workload == "sql":
            kw = dict(mode="SQL", critic_struct="quad-lin", buffer_size=10)
        if os.environ.get("AB_MODE"):  # e.g. AB_MODE=RQL AB_K=36 ... stream
            kw = dict(mode=os.environ["AB_MODE"], critic_struct=os.environ.get("AB_CS", "quad-nomix"), buffer_size=10)
        name = os.environ.get("AB_SYS", "3wrobot") if workload in ("stream", "gen") else "3wrobot"  # AB_SYS / AB_B / AB_N: stream, gen
        if workload in ("gen", "search", "opt0", "opt4"):
            Nh = int(os.environ.get("AB_N", Nh))
        if workload == "stream":
            B, Nh = int(os.environ.get("AB_B", B)), int(os.environ.get("AB_N", Nh))
            if name == "2tank" and "Ncritic" not in kw and kw:
                kw["Ncritic"] = 4
        if workload in ("stream", "gen", "search", "opt0", "opt4") and os.environ.get("AB_DTYPE"):  # AB_DTYPE=f64: the reference's width
            kw["dtype"] = os.environ["AB_DTYPE"]
        eng = Engine(preset_engine_config(name, B, Nactor=Nh, **kw))
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        eng.set_state({"3wrobot": st3(B), "3wrobotNI": st3(B)[:, :3],
                       "2tank": np.stack([rng.uniform(0, 2, B), rng.uniform(-2, 2, B)], -1)}[name])
        if workload in ("stream", "sql"):
            from rcognita_amd.pool import PRESETS

            td = torch.float64 if kw.get("dtype") == "f64" else torch.float32
            bnd = torch.tensor(np.array(PRESETS[name]["ctrl_bnds"]), device="cuda", dtype=td)
            cand = (torch.rand((B, K, Nh, eng.du), device="cuda", dtype=td) * (bnd[:, 1] - bnd[:, 0]) + bnd[:, 0]).contiguous()
            step = lambda: eng.control_tick(cand, K=K)
        elif workload == "gen":
            step = lambda: eng.control_tick(None, K=K)
        elif workload == "search":
            step = lambda: eng.control_tick_search(K=K, rounds=1, warm_start=True)
        else:
            eng.set_optimizer(int(workload[3:]))
            step = lambda: eng.control_tick_opt(iters=5)
    elif workload in ("gen_c3", "fit"):
        B = 131072
        eng = Engine(preset_engine_config("2tank", B, Nactor=20, mode="RQL", critic_struct="quadratic", Ncritic=4,
                                          buffer_size=10))
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        eng.set_state(np.stack([rng.uniform(0, 2, B), rng.uniform(-2, 2, B)], -1))
        step = lambda: eng.control_tick(None, K=K)
        if workload == "fit":
            kind = N.KERNEL_CRITIC
    elif workload in ("ticks", "ticks256"):
        B = int(os.environ.get("AB_B", 1024))
        name = os.environ.get("AB_SYS", "3wrobot")  # (a robot)
        eng = Engine(preset_engine_config(name, B, Nactor=Nh))
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        eng.set_state(st3(B)[:, :eng.ds])
        step = lambda: eng.control_ticks(T=64, K=256 if workload == "ticks256" else 64)
    elif workload == "pool":  # configs[4]'s per-GPU shard: three handles on three streams, WALL time per tick (overlap included)
        import time

        from rcognita_amd.pool import MixedPool

        total = 65536
        counts = {"3wrobot": total // 3 + total % 3, "3wrobotNI": total // 3, "2tank": total // 3}
        pool = MixedPool(counts, Nactor=15, dtype="f32")
        pool.set_states({sg.name: {"3wrobot": st3, "3wrobotNI": lambda n: st3(n)[:, :3],
                                   "2tank": lambda n: np.stack([rng.uniform(0, 2, n), rng.uniform(-2, 2, n)], -1)}[sg.name](sg.hi - sg.lo)
                         for sg in pool.segments})
        for _ in range(300):
            pool.control_tick(256)
        pool.synchronize()
        meds = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(400):
                pool.control_tick(256)
            pool.synchronize()
            meds.append((time.perf_counter() - t0) / 400 * 1e6)
        print("RES " + json.dumps({"workload": workload, "kernel": "pool tick (wall)", "median_us": float(np.median(meds)),
                                   "min_us": float(min(meds)), "n": 5}))
        return
    else:
        raise SystemExit(f"unknown workload {workload}")
    for _ in range(150):
        step()
    eng.profile((kind,), stride=2)
    for _ in range(120):
        step()
    s = eng.profile_samples(kind) * 1e3
    print("RES " + json.dumps({"workload": workload, "kernel": eng.last_launch(kind)["kernel"], "median_us": float(np.median(s)),
                               "min_us": float(s.min()), "n": int(s.size)}))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--a", default=os.path.join(ROOT, "rcognita_amd", "lib", "librcg.so"))
    p.add_argument("--b", default=os.path.join(ROOT, "rcognita_amd", "lib", "librcg_ab.so"))
    p.add_argument("--rounds", type=int, default=3)
    p.add_argument("--a-env", default="", help="KEY=VAL[,KEY=VAL] set for the A children (dev-build knobs: RCG_NO_PK=1 ...)")
    p.add_argument("--b-env", default="", help="the same for the B children")
    p.add_argument("--child", nargs=2, default=None)
    p.add_argument("workloads", nargs="*", default=["gen"])
    a = p.parse_args()
    if a.child:
        return child(*a.child)
    env = {k: v for k, v in os.environ.items() if not k.startswith("RCG_")}  # (AB_K / AB_MODE / AB_CS pass through)
    for w in (ALL if a.workloads == ["all"] else a.workloads):
        res = {"A": [], "B": []}
        for _ in range(a.rounds):
            for tag, lib, extra in (("A", a.a, a.a_env), ("B", a.b, a.b_env)):
                cenv = dict(env)
                cenv.update(kv.split("=", 1) for kv in extra.split(",") if kv)
                out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib, w], capture_output=True,
                                     text=True, env=cenv, timeout=900)
                line = [l for l in out.stdout.splitlines() if l.startswith("RES ")]
                if not line:
                    print(w, tag, "FAILED", out.stderr[-400:], flush=True)
                    continue
                res[tag].append(json.loads(line[-1][4:]))
        fmt = lambda rs: "[" + ", ".join(f"{r['median_us']:.1f}" for r in rs) + "]"
        ka = res["A"][0]["kernel"] if res["A"] else "?"
        ma = np.median([r["median_us"] for r in res["A"]]) if res["A"] else float("nan")
        mb = np.median([r["median_us"] for r in res["B"]]) if res["B"] else float("nan")
        print(f"AB {w:8s} {ka:18s} A medians {fmt(res['A'])} B medians {fmt(res['B'])}  B/A = {mb / ma:.3f}", flush=True)


if __name__ == "__main__":
    main()
