#!/usr/bin/env python3
"""Tick time of the on-device optimiser (rcg_control_tick_opt, 5 iterations) at the C2 shape for the modes and critic
structures whose LDS footprint differs: prints one line per case.  `python tools/opt_probe.py [B]`."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    import torch
    from valu_probe import states

    from rcognita_amd import Engine
    from rcognita_amd.pool import preset_engine_config

    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    if len(sys.argv) > 2:  # A/B against another build of the library
        from rcognita_amd import _native

        _native.use_library(sys.argv[2])
    rng = np.random.default_rng(7)
    st = torch.cuda.current_stream()
    cases = [("MPC", None, 0, "f32"), ("MPC", None, 4, "f32"), ("RQL", "quad-nomix", 4, "f32"), ("RQL", "quad-mix", 4, "f32"),
             ("RQL", "quadratic", 4, "f32"), ("RQL", "quad-lin", 4, "f32"), ("SQL", "quad-nomix", 4, "f32"),
             ("SQL", "quad-lin", 4, "f32"), ("RQL", "quad-nomix", 4, "f64")]
    only = os.environ.get("OPT_PROBE_ONLY")  # e.g. "RQL:quad-mix,RQL:quad-nomix"
    if only:
        keep = {tuple(k.split(":")) for k in only.split(",")}
        cases = [c for c in cases if (c[0], str(c[1])) in keep and c[3] == "f32" and c[2] == 4]
    for mode, cs, mem, dt in cases:
        kw = dict(Nactor=10, dtype=dt)
        if mode != "MPC":
            kw.update(mode=mode, critic_struct=cs, buffer_size=10)
        e = Engine(preset_engine_config("3wrobot", B, **kw))
        e.set_stream(st.cuda_stream)
        e.set_state(states(rng, "3wrobot", B))
        e.set_optimizer(mem)
        for _ in range(5):
            e.control_tick_opt(iters=5)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        a.record(st)
        for _ in range(n):
            e.control_tick_opt(iters=5)
        b.record(st)
        torch.cuda.synchronize()
        from rcognita_amd import _native as Nn

        e.profile((Nn.KERNEL_CRITIC, Nn.KERNEL_ACTOR), stride=1)  # where the tick goes: the fit's launch against the optimiser's
        for _ in range(n):
            e.control_tick_opt(iters=5)
        e.synchronize()
        (cm, cn), (am, an) = e.profile_read(Nn.KERNEL_CRITIC), e.profile_read(Nn.KERNEL_ACTOR)
        split = f"k_actor_opt {am / max(an, 1) * 1e3:.0f} us" + (f", env step + push + fit {cm / max(cn, 1) * 1e3:.0f} us ({e.last_launch(Nn.KERNEL_CRITIC)['variant']})" if cn else "")
        print(f"{mode} {cs} memory {mem} {dt}: {a.elapsed_time(b) / n:.4f} ms per tick ({split}), launch {e.last_launch()}", flush=True)
        e.close()


if __name__ == "__main__":
    main()
