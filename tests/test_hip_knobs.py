"""Scheduling variants of the launcher (development knobs, rcg_sysops.hpp::DevKnobs) must reproduce the default launch
BIT FOR BIT: same arithmetic, different scheduling.  The knobs exist only in the -DRCG_DEV build (`make dev`,
librcg_dev.so); the production library never reads the environment, which is checked here too.  Each variant is a
separate process (the knobs are read once per process).  ``gpu`` marked."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
DEV_LIB = os.path.join(ROOT, "rcognita_amd", "lib", "librcg_dev.so")


def _run(env_extra, dev):
    env = {k: v for k, v in os.environ.items() if not k.startswith("RCG_")}
    env.update(env_extra)
    out = subprocess.run([sys.executable, os.path.join(HERE, "knob_probe.py")] + (["--lib", DEV_LIB] if dev else []),
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("HASH ")]
    launch = [l for l in out.stdout.splitlines() if l.startswith("LAUNCH ")]
    assert line and launch, out.stdout
    return line[-1], launch[-1]


def test_scheduling_variants_are_bit_identical():
    assert os.path.exists(DEV_LIB), f"{DEV_LIB} missing: `make dev` (built by __graft_entry__.build())"
    prod, prod_launch = _run({}, dev=False)
    assert prod_launch == "LAUNCH k_actor_dma 0 4", prod_launch  # B = 32773: 4 envs per wave (>= 8192 waves), MPC gamma = 1
    base, base_launch = _run({}, dev=True)
    assert (base, base_launch) == (prod, prod_launch), "the dev build without knobs is the production schedule"
    for knobs, launch in (({"RCG_GPW": "1", "RCG_LDS_PAD": "-1"}, "LAUNCH k_actor_dma 0 1"),  # the first geometry
                          ({"RCG_GPW": "16", "RCG_PER_CU": "4"}, "LAUNCH k_actor_dma 0 16"),
                          ({"RCG_GPW": "3", "RCG_PER_CU": "8"}, "LAUNCH k_actor_dma 0 3"),  # not a power of two
                          ({"RCG_NO_G1": "1"}, None),             # the discounted instance with gamma = 1 (rounding differs)
                          ({"RCG_NO_GEN_MULTI": "1"}, prod_launch),  # generated tiles one at a time: same bits per candidate
                          ({"RCG_NO_PK": "1"}, prod_launch)):  # generated grid / k_ticks without the hand-packed instances
        got, got_launch = _run(knobs, dev=True)
        if launch is None:
            assert got_launch == "LAUNCH k_actor_dma 1 4", (knobs, got_launch)
            continue
        assert got == base, knobs
        assert got_launch == launch, (knobs, got_launch)


def test_the_production_library_ignores_every_knob():
    """librcg.so as shipped has one schedule: RCG_* variables in the environment change neither the kernel that runs
    (rcg_last_launch) nor a single output bit."""
    prod = _run({}, dev=False)
    loud = _run({"RCG_ACTOR_KERNEL": "plain", "RCG_GPW": "3", "RCG_PER_CU": "8", "RCG_NO_G1": "1", "RCG_DBG": "7",
                 "RCG_LDS_PAD": "-1", "RCG_DMA_MPC_ONLY": "1", "RCG_NO_GEN_MULTI": "1", "RCG_PLAIN_LDS": "65536", "RCG_NO_PK": "1",
                 "RCG_NO_TICK_FUSE": "1", "RCG_NO_PACK": "1", "RCG_DMA_MINK": "64", "RCG_FIT_LANES": "4"},
                dev=False)
    assert loud == prod
