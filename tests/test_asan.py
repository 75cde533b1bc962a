"""CPU sanitizer run ('race detection / sanitizers'): `make asan` builds the C oracle and the HOST side of
librcg (every .hip unit compiled --offload-host-only: C ABI, argument checks, launch-geometry arithmetic; no device
code) with clang -fsanitize=address,undefined, linked with tests/asan_driver.c.  Never touches a GPU."""
import os
import subprocess

from tests.conftest import ROOT


def test_oracle_and_abi_host_side_under_asan_ubsan():
    subprocess.check_call(["make", "-C", ROOT, "-j4", "asan"], stdout=subprocess.DEVNULL)
    exe = os.path.join(ROOT, "build", "asan", "abi_asan")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    out = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "asan_driver ok" in out.stdout
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr
