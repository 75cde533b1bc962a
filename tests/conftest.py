import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_sessionstart(session):
    """Some GPU tests use torch for device memory, streams and RCCL.  PyTorch-ROCm ships its own HIP runtime and must
    bring the GPU up BEFORE librcg loads the system one (rcognita_amd/_native.py::lib: the other order leaves torch without
    a device), and earlier tests of the session create handles - so initialise torch's side first, once."""
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    # experiments only: the suite against another build of the library, by an explicit option that marks the session as a
    # non-release one (e.g. the dev build with one of its knobs set:
    #   RCG_FIT_LANES=4 python -m pytest tests/test_hip_critic.py -m gpu --rcg-lib rcognita_amd/lib/librcg_dev.so)
    alt = session.config.getoption("--rcg-lib")
    if alt:
        from rcognita_amd import _native as N

        N.use_library(os.path.join(ROOT, alt))
        print(f"\n*** NON-RELEASE SESSION: GPU tests bound to {alt}, RCG_* knobs allowed - not evidence about librcg.so ***")


def pytest_addoption(parser):
    parser.addoption("--rcg-lib", action="store", default=None,
                     help="experiments only: bind the GPU tests to another build of librcg (relative to the repo root); "
                          "without it every GPU test asserts that the library loaded is the shipped rcognita_amd/lib/librcg.so")


def load_golden(name):
    import json

    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    return meta, z


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(autouse=True)
def _gpu_tests_run_the_library_as_shipped(request):
    """The `-m gpu` parity claims are made for librcg.so as built and shipped: no RCG_* variable (the dev build's A/B
    knobs) may be set in the environment of a GPU test, and the library the binding loads is rcognita_amd/lib/librcg.so
    (tests that probe the dev build do so in a child process or bind it explicitly, tests/test_hip_knobs.py)."""
    if request.node.get_closest_marker("gpu") is not None and not request.config.getoption("--rcg-lib"):
        bad = sorted(k for k in os.environ if k.startswith("RCG_"))
        assert not bad, f"GPU tests must run with a clean environment; unset {', '.join(bad)}"
        from rcognita_amd import _native as N

        assert os.path.realpath(N.LIB_PATH) == os.path.realpath(os.path.join(ROOT, "rcognita_amd", "lib", "librcg.so")), N.LIB_PATH
    yield
