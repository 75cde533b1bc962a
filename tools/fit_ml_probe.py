#!/usr/bin/env python3
"""k_critic_fit (lane = env) against k_critic_fit_ml (four lanes per env, dev build: RCG_FIT_LANES=4) in closed loops: each child
process runs the same loop and dumps the critic weights; the parent compares them and prints the fit kernel's time.

    python tools/fit_ml_probe.py            # GPU box, librcg_dev.so built
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = [("2tank", "RQL", "quadratic", 131072, 20), ("2tank", "SQL", "quad-lin", 65536, 10), ("3wrobotNI", "RQL", "quad-mix", 65536, 5),
         ("3wrobot", "SQL", "quad-nomix", 65536, 5), ("3wrobot", "RQL", "quad-lin", 32768, 5), ("3wrobotNI", "SQL", "quadratic", 65536, 5)]


def child(out):
    from rcognita_amd import _native as N

    N.use_library(os.path.join(ROOT, "rcognita_amd", "lib", "librcg_dev.so"))
    import torch  # noqa: F401

    from rcognita_amd import Engine
    from rcognita_amd.pool import preset_engine_config

    res, arrs = {}, {}
    for name, mode, cs, B, Nh in CASES:
        rng = np.random.default_rng(7)
        eng = Engine(preset_engine_config(name, B, Nactor=Nh, mode=mode, critic_struct=cs, Ncritic=4, buffer_size=10, dtype="f32"))
        x0 = {"3wrobot": lambda: np.stack([rng.uniform(-5, 5, B), rng.uniform(-5, 5, B), rng.uniform(-3, 3, B), rng.uniform(-1, 1, B),
                                           rng.uniform(-1, 1, B)], -1),
              "3wrobotNI": lambda: np.stack([rng.uniform(-5, 5, B), rng.uniform(-5, 5, B), rng.uniform(-3, 3, B)], -1),
              "2tank": lambda: np.stack([rng.uniform(0, 2, B), rng.uniform(-2, 2, B)], -1)}[name]()
        eng.set_state(x0)
        for _ in range(25):
            eng.control_tick(None, K=64)
        eng.profile((N.KERNEL_CRITIC,), stride=1)
        for _ in range(15):
            eng.control_tick(None, K=64)
        s = eng.profile_samples(N.KERNEL_CRITIC) * 1e3
        key = f"{name}_{mode}_{cs}"
        res[key] = {"fit_us_mean": float(s.mean()), "fit_us_min": float(s.min()), "kernel": eng.last_launch(N.KERNEL_CRITIC)}
        arrs[key + "_w"] = eng.get_field(N.FIELD_W_CRITIC)
        arrs[key + "_accum"] = eng.get_field(N.FIELD_ACCUM)
        eng.close()
    np.savez(out, **arrs)
    print("RES " + json.dumps(res))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    outs = {}
    for tag, extra in (("lane_env", {}), ("four_lanes", {"RCG_FIT_LANES": "4"})):
        env = {k: v for k, v in os.environ.items() if not k.startswith("RCG_")}
        env.update(extra)
        path = os.path.join(ROOT, "gpurun_out", f"fit_ml_{tag}.npz")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        o = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", path], capture_output=True, text=True, env=env, timeout=900)
        line = [l for l in o.stdout.splitlines() if l.startswith("RES ")]
        if not line:
            print(tag, "FAILED", o.stderr[-800:])
            return 1
        outs[tag] = (json.loads(line[-1][4:]), np.load(path))
    a, b = outs["lane_env"], outs["four_lanes"]
    for key in a[0]:
        wa, wb = a[1][key + "_w"].astype(np.float64), b[1][key + "_w"].astype(np.float64)
        d = np.abs(wa - wb) / np.maximum(np.max(np.abs(wa), axis=1, keepdims=True), 1e-30)
        env_bad = np.max(d, axis=1) > 1e-4
        acc = np.abs(a[1][key + "_accum"] - b[1][key + "_accum"]) / np.maximum(np.abs(a[1][key + "_accum"]), 1e-30)
        print(f"{key:28s} fit {a[0][key]['fit_us_mean']:7.1f} -> {b[0][key]['fit_us_mean']:7.1f} us ({b[0][key]['kernel']['kernel']}, variant "
              f"{b[0][key]['kernel']['variant']});  weights: median rel diff {np.median(d):.2e}, envs beyond 1e-4: {int(env_bad.sum())} of "
              f"{len(env_bad)};  accum after 40 ticks: max rel diff {acc.max():.2e}", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
