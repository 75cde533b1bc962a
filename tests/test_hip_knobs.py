"""The launcher's opt-in kernel variants (development knobs, rcg_sysops.hpp::DevKnobs) must reproduce the default
launch BIT FOR BIT: same arithmetic, different scheduling.  Each variant is a separate process (the knobs are read once
per process).  ``gpu`` marked."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(env_extra):
    env = dict(os.environ)
    for k in list(env):
        if k.startswith("RCG_") and k != "RCG_LIB":
            del env[k]
    env.update(env_extra)
    out = subprocess.run([sys.executable, os.path.join(HERE, "knob_probe.py")], capture_output=True, text=True, env=env,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("HASH ")]
    assert line, out.stdout
    return line[-1]


def test_scheduling_variants_are_bit_identical():
    base = _run({})
    for knobs in ({"RCG_GPW": "1", "RCG_LDS_PAD": "-1"},     # one env per wave, no residency cap (the first geometry)
                  {"RCG_GPW": "16", "RCG_PER_CU": "4"},
                  {"RCG_GPW": "3", "RCG_PER_CU": "8"},       # envs per wave not a power of two
                  {"RCG_NO_GEN_MULTI": "1"},                 # generated tiles one at a time: each candidate's cost the same bits
                  {"RCG_DBG": "7"}):                         # timing-only switches: compiled OUT of the production library
        assert _run(knobs) == base, knobs
