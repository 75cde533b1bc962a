"""Helper of tests/test_hip_knobs.py (not a test module): a short streamed closed loop on a ragged batch, printed as
one hash over every per-env field.  The development knobs of the launcher (RCG_*) are read once per process, so each
variant runs in its own interpreter."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from rcognita_amd import _native as N

    if "--lib" in sys.argv:  # the -DRCG_DEV twin, selected by the caller in code (the binding reads no environment)
        N.use_library(sys.argv[sys.argv.index("--lib") + 1])
    from rcognita_amd import Engine
    from rcognita_amd.pool import preset_engine_config

    rng = np.random.default_rng(5)
    B, K, Nh = 32773, 128, 5  # several envs per wave, a ragged last wave, two tiles per env
    eng = Engine(preset_engine_config("3wrobot", B, Nactor=Nh))
    x0 = np.stack([rng.uniform(-10, 10, B), rng.uniform(-10, 10, B), rng.uniform(-3, 3, B), rng.uniform(-1, 1, B),
                   rng.uniform(-1, 1, B)], -1).astype(np.float32)
    eng.set_state(x0)
    cand = eng.to_device((rng.random((B, K, Nh, 2), dtype=np.float32) - 0.5) * np.array([600, 200], dtype=np.float32))
    for _ in range(5):
        eng.control_tick(cand, K=K)
    ll = eng.last_launch(N.KERNEL_ACTOR)
    print("LAUNCH", ll["kernel"], ll["variant"], ll["envs_per_wave"])
    h = hashlib.sha256()
    for f in (N.FIELD_STATE, N.FIELD_STATE_PREV, N.FIELD_ACTION, N.FIELD_ACCUM, N.FIELD_STEP_IDX, N.FIELD_BEST_IDX,
              N.FIELD_BEST_J, N.FIELD_STATUS):
        h.update(np.ascontiguousarray(eng.get_field(f)).tobytes())
    # generated level grid on both robots, K = 256 (four tiles per lane: rolled out together with the shared heading
    # sub-trajectory unless RCG_NO_GEN_MULTI is set), single ticks and T ticks per launch
    # (3wrobot and the second 3wrobotNI run, gamma = 1 with the preset's R1: the hand-packed instances unless RCG_NO_PK)
    for name, gamma in (("3wrobot", 1.0), ("3wrobotNI", 0.97), ("3wrobotNI", 1.0)):
        B3 = 777
        e3 = Engine(preset_engine_config(name, B3, Nactor=7, gamma=gamma))
        ds = 5 if name == "3wrobot" else 3
        e3.set_state(rng.uniform(-3, 3, (B3, ds)).astype(np.float32))
        for _ in range(3):
            e3.control_tick(None, K=256)
        e3.control_ticks(4, 256)
        for f in (N.FIELD_STATE, N.FIELD_ACTION, N.FIELD_ACCUM, N.FIELD_BEST_IDX, N.FIELD_BEST_J):
            h.update(np.ascontiguousarray(e3.get_field(f)).tobytes())
        e3.close()
    # RQL / SQL closed loops past the point where the buffers have filled (env step + push + critic fit in one launch,
    # generated candidates), a ragged last block
    for name, cs, mode, Bc in (("2tank", "quadratic", "RQL", 2000), ("2tank", "quad-mix", "SQL", 700),
                               ("2tank", "quad-nomix", "RQL", 300), ("2tank", "quad-lin", "RQL", 300),
                               ("3wrobot", "quad-nomix", "RQL", 900), ("3wrobotNI", "quad-nomix", "SQL", 333)):
        ec = Engine(preset_engine_config(name, Bc, Nactor=6, mode=mode, critic_struct=cs, Ncritic=4, buffer_size=6))
        ds = {"2tank": 2, "3wrobot": 5, "3wrobotNI": 3}[name]
        ec.set_state(rng.uniform(0.1, 2.0, (Bc, ds)).astype(np.float32))
        for _ in range(14):
            ec.control_tick(None, K=64)
        for f in (N.FIELD_STATE, N.FIELD_ACTION, N.FIELD_ACCUM, N.FIELD_BEST_IDX, N.FIELD_W_CRITIC, N.FIELD_W_PREV,
                  N.FIELD_OBS_BUF, N.FIELD_ACT_BUF):
            h.update(np.ascontiguousarray(ec.get_field(f)).tobytes())
        ec.close()
    print("HASH", h.hexdigest())


if __name__ == "__main__":
    main()
