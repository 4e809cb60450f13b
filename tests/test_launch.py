"""polymath_amd/launch.py -- the supervisor behind `python bench.py --gpus N` -- on a box WITHOUT GPUs (CPU test): every rank of
every attempt dies at its first GPU call, the supervisor walks the whole fallback chain, prints no JSON line and exits non-zero
-- loudly and quickly, never a hang.  (The success paths -- self-launch, external launcher, fallback chain, a rank dying mid-run
-- are GPU tests in tests/test_gpu_parity.py.)"""
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_attempt_list_and_pinning():
    from polymath_amd import launch
    assert launch.attempts_from_env({}) == launch.ATTEMPTS and launch.ATTEMPTS[0] == ("nccl", "rccl")
    assert launch.attempts_from_env({"BENCH_DIST_BACKEND": "gloo"}) == [("gloo", "callbacks")]
    assert launch.attempts_from_env({"BENCH_NO_RCCL": "1"}) == [("nccl", "callbacks")]
    e = launch._child_env({"TORCHELASTIC_USE_AGENT_STORE": "True", "X": "1"}, 3, 3, 8, 1234, 2, "gloo", "callbacks", True)
    assert (e["RANK"], e["WORLD_SIZE"], e["MASTER_PORT"], e["BENCH_ATTEMPT"], e["BENCH_CHILD"]) == ("3", "8", "1234", "2", "1")
    assert e["BENCH_NO_RCCL"] == "1" and "TORCHELASTIC_USE_AGENT_STORE" not in e and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_launcher_module_touches_neither_torch_nor_the_gpu():
    """The parent of the ranks must never initialise the GPU (a process that has may not start another program on this pool)."""
    code = "import sys; from polymath_amd import launch; assert 'torch' not in sys.modules and 'polymath_amd.api' not in sys.modules; print('clean')"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=120)
    assert out.returncode == 0 and "clean" in out.stdout, out.stderr


def test_bench_gpus_2_without_gpus_fails_loudly_and_quickly():
    from polymath_amd import api
    if api.load_library().pm_device_count() > 0:
        pytest.skip("GPU present: the success paths are covered by the GPU tests")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BENCH_DIST_BACKEND", "BENCH_NO_RCCL")}
    t0 = time.time()
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--log-constraints", "10",
                          "--no-cpu-baseline", "--msm-micro", "", "--no-live-traffic"], capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode != 0
    assert not [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert run.stderr.count("[launch] attempt") >= 8 and "all attempts failed" in run.stderr      # 4 attempts announced, 4 failures reported
    assert time.time() - t0 < 300
