"""Host-side tooling that guards measurements (CPU only)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_bench_configs_compare_flags_a_regression(capsys):
    """tools/bench_configs.py --compare: rates (higher is better) and kernel times (lower is better) of a run against a
    stored one; more than 10 % worse anywhere -> exit code 3 (the guard that would have caught the 20 % slowdown of
    configs[2]'s generated-candidate kernels in round 2)."""
    bc = _load("bench_configs", "tools/bench_configs.py")
    old = {"C3_MPC": {"env_control_steps_per_s": 7.0e8, "actor_ms": 0.170, "n_failed": 0.0, "candidates": "grid"},
           "sim": {"kernel_ms": 0.21, "kernel_GBps": 5700.0}, "only_old": {"kernel_ms": 1.0}}
    same = {"C3_MPC": {"env_control_steps_per_s": 6.9e8, "actor_ms": 0.172, "n_failed": 0.0, "candidates": "grid"},
            "sim": {"kernel_ms": 0.205, "kernel_GBps": 5800.0}, "only_new": {"kernel_ms": 1.0}}
    assert bc.compare(same, old) == 0
    slow = {"C3_MPC": {"env_control_steps_per_s": 5.6e8, "actor_ms": 0.2156}, "sim": {"kernel_ms": 0.21}}
    assert bc.compare(slow, old) == 3
    err = capsys.readouterr().err
    assert "WORSE" in err and "actor_ms" in err
