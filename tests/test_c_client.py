"""The drop-in boundary used from plain C: examples/c_tick.c is compiled with gcc against include/rcg.h + librcg.so and
run as its own process (no Python, no torch in it); its result must equal the same run driven through the ctypes
binding.  ``gpu`` marked (the CPU half - that the header is valid C and the symbols resolve - is tests/test_abi.py)."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_client_matches_python_binding(tmp_path):
    from rcognita_amd import Engine, EngineConfig, _native as N

    exe = tmp_path / "c_tick"
    libdir = os.path.join(ROOT, "rcognita_amd", "lib")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_tick.c"), "-L", libdir, "-lrcg", f"-Wl,-rpath,{libdir}",
                           "-lm", "-o", str(exe)])
    B, ticks = 4096, 40
    out = subprocess.run([str(exe), str(B), str(ticks)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    m = re.search(r"mean_dist0=(\S+) mean_dist1=(\S+) accum_sum=(\S+) count=(\S+) n_failed=(\S+) steps_ok=(\d)", out.stdout)
    assert m, out.stdout
    d0, d1, acc, count, n_failed, steps_ok = (float(v) for v in m.groups())
    assert count == B and n_failed == 0 and steps_ok == 1
    assert d1 < d0  # the controller drives the robots towards the origin

    # the same run through the Python binding
    b = np.arange(B)
    a, r = 2 * np.pi * b / B, 3.0 + 5.0 * (b % 7) / 7.0
    x0 = np.stack([r * np.cos(a), r * np.sin(a), a + np.pi / 2, 0 * a, 0 * a], axis=-1).astype(np.float32)
    eng = Engine(EngineConfig(sys_id=N.SYS_3WROBOT, batch=B, dtype="f32", Nactor=10, pars=[10, 1],
                              ctrl_bnds=np.array([[-300, 300], [-100, 100]], dtype=float), R1=[1, 10, 1, 0, 0, 0, 0],
                              dt_sim=0.01, sampling_time=0.01, pred_step_size=0.02))
    eng.set_state(x0)
    for _ in range(ticks):
        eng.control_tick(None, K=256)
    st = eng.get_state()
    summ, _ = eng.episode_stats(from_accum=True)
    np.testing.assert_allclose(d1, float(np.mean(np.hypot(st[:, 0], st[:, 1]))), rtol=1e-6)
    np.testing.assert_allclose(acc, summ["sum"], rtol=1e-9)  # same kernels, same inputs: same deterministic sums
