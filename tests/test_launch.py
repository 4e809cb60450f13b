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
    assert e["PM_NTT_OVERLAP"] == "0"                          # host-staged attempts also drop the two-stream transforms
    assert "PM_NTT_OVERLAP" not in launch._child_env({}, 0, 0, 8, 1234, 0, "nccl", "rccl", True)


def test_budget_two_hung_rccl_attempts_still_leave_the_host_staged_attempt_400_s():
    """one global budget (1 500 s, inside the driver's 1 800 s).  Worst case: the two RCCL attempts hang until
    their deadlines -- the gloo + callbacks attempt must still get >= 400 s, and the chain must end inside the budget."""
    from polymath_amd import launch
    A = launch.ATTEMPTS
    assert A[2] == ("gloo", "callbacks")
    stage_sum = 600 + 3 * 25 + 120                       # bench.py's stage sum at --steps 20 --warmup 5
    left = float(launch.TOTAL_BUDGET_S)
    d0 = launch.attempt_deadline(A, 0, left, stage_sum)
    left -= d0
    d1 = launch.attempt_deadline(A, 1, left, stage_sum)
    left -= d1
    d2 = launch.attempt_deadline(A, 2, left, stage_sum)
    assert d0 >= 300 and d1 >= 300, (d0, d1)              # a healthy 8-GPU attempt needs ~200 s
    assert d2 >= 400, (d0, d1, d2)
    assert d0 + d1 + d2 <= launch.TOTAL_BUDGET_S - launch.EXIT_MARGIN_S
    left -= d2
    assert launch.attempt_deadline(A, 3, left, stage_sum) == 0        # nothing worth starting is left: skipped, not overrun
    # a pinned single configuration (tests, BENCH_DIST_BACKEND) may use the whole budget up to its stage sum
    assert launch.attempt_deadline([("gloo", "callbacks")], 0, 1500.0, stage_sum) == stage_sum
    # the arithmetic never hands out more than is left
    for rem in (50, 100, 430, 500, 900, 1500, 3000):
        for k in range(4):
            assert launch.attempt_deadline(A, k, float(rem), stage_sum) <= max(0, rem - launch.EXIT_MARGIN_S)


def test_supervisor_walks_the_chain_inside_its_budget_with_hanging_children(tmp_path):
    """The same, live and small: children that hang in the RCCL attempts and succeed in the host-staged one; a 14 s budget with a
    5 s reserve.  The supervisor kills the hung attempts on time, the third attempt runs and wins, the budget lines are printed."""
    child = tmp_path / "child.py"
    child.write_text("import os, sys, time\n"
                     "if os.environ.get('BENCH_NO_RCCL') != '1':\n"
                     "    time.sleep(600)\n"
                     "print('{\"ok\": %s}' % os.environ['BENCH_ATTEMPT'] if os.environ['RANK'] == '0' else '', flush=True)\n")
    code = ("import sys; sys.path.insert(0, %r); from polymath_amd import launch; launch.MIN_ATTEMPT_S = 2; launch.EXIT_MARGIN_S = 1; "
            "raise SystemExit(launch.supervise_all(%r, [], 2, 100))" % (ROOT, str(child)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BENCH_DIST_BACKEND", "BENCH_NO_RCCL")}
    env.update(BENCH_TOTAL_BUDGET_S="14", BENCH_FALLBACK_RESERVE_S="5")
    t0 = time.time()
    run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    dt = time.time() - t0
    assert run.returncode == 0, run.stderr
    assert run.stdout.strip() == '{"ok": 2}', run.stdout
    assert run.stderr.count("budget:") == 3 and "attempt 0 failed: deadline" in run.stderr and "attempt 1 failed: deadline" in run.stderr
    assert dt < 14 + 25, dt                                # the two hung attempts end at ~4.8 s and ~8 s (+ up to 10 s of SIGTERM grace each)


def test_retry_port_is_agreed_between_the_supervisors(tmp_path):
    """Under an external launcher a retry meets on a port rank 0 found FREE and published (MASTER_PORT + 37 k was a
    guess), and only once every supervisor has arrived at the retry."""
    import threading
    from polymath_amd import launch
    env = {"TORCHELASTIC_RUN_ID": "t%d" % os.getpid(), "MASTER_PORT": "65530", "PM_LAUNCHER_PID": "1"}
    assert launch._rendezvous_dir(env) != launch._rendezvous_dir(dict(env, PM_LAUNCHER_PID="2"))      # another launcher, another directory
    got = [None] * 3

    def sup(r, delay):
        time.sleep(delay)
        got[r] = launch.agree_on_retry(env, r, 3, 1, wait_s=20.0)
    th = [threading.Thread(target=sup, args=(r, 0.3 * r)) for r in range(3)]
    t0 = time.time()
    for t in th:
        t.start()
    for t in th:
        t.join(30)
    assert got[0] == got[1] == got[2] and 1024 < got[0] < 65536
    assert time.time() - t0 >= 0.55                        # nobody left before the slowest supervisor had arrived
    # the supervisors notice a failed attempt at very different times (one child crashes at once, the others sit out
    # their collective deadline).  A rank that arrives 3 s after rank 0 published must still take rank 0's port.
    env_skew = dict(env, TORCHELASTIC_RUN_ID="k%d" % os.getpid())
    late = [None] * 2

    def sup_skew(r, delay):
        time.sleep(delay)
        late[r] = launch.agree_on_retry(env_skew, r, 2, 1, wait_s=20.0)
    th = [threading.Thread(target=sup_skew, args=(r, 3.0 * r)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(40)
    assert late[0] == late[1] and 1024 < late[0] < 65536, late
    for r in range(2):
        launch._leave_rendezvous(env_skew, r)
    # a second incarnation under the same launcher (torchrun --max-restarts: same parent, run id and port) meets in its own
    # directory; and inside ONE directory a port file that does not carry this incarnation's token is not taken for the agreement
    assert launch._rendezvous_dir(env) != launch._rendezvous_dir(dict(env, TORCHELASTIC_RESTART_COUNT="1"))
    env3 = dict(env, TORCHELASTIC_RUN_ID="s%d" % os.getpid())
    d3 = launch._rendezvous_dir(env3)
    stale = os.path.join(d3, "attempt1.port")
    open(stale, "w").write("4711")
    old_t = time.time() - 600
    os.utime(stale, (old_t, old_t))
    open(os.path.join(d3, "attempt1.rank0"), "w").close()                # rank 0 "arrived" long ago as well, and published nothing new
    assert launch.agree_on_retry(env3, 1, 2, 1, wait_s=0.6) != 4711      # falls back to the validated guess instead of the stale port
    launch._leave_rendezvous(env3, 1)
    launch._leave_rendezvous(env3, 0)
    assert not os.path.exists(d3)                                        # the last supervisor out removes the directory
    for r in range(3):
        launch._leave_rendezvous(env, r)
    # nobody publishes (rank 0 is gone): the bounded wait ends and the guess is validated into the port range
    env2 = dict(env, TORCHELASTIC_RUN_ID="u%d" % os.getpid())
    assert 1024 < launch.agree_on_retry(env2, 1, 2, 2, wait_s=0.5) < 65536


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
