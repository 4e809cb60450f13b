"""Process supervision for multi-GPU runs of bench.py (one process per GPU).

The reference is single-process CPU code (prover.rs:66-237), so nothing here has a counterpart in it.  Rules this file
exists to keep (they come from the GPU pool, not from taste):
  * a process that has touched the GPU never exec()s another program -- ranks are CHILD processes of a parent that has made
    no GPU call (this module imports neither torch nor the HIP library);
  * a dead or stalled rank must end the job with a non-zero exit, not a hang: children run in their own process groups,
    the supervisor kills all of them when one fails or when the attempt's deadline passes;
  * the first contact with N real GPUs should produce a number: the exchange layer is tried in order of preference
    (ATTEMPTS) and the first configuration in which every rank finishes wins.  Rank 0 prints its JSON line only at the very
    end of a successful attempt, so a failed attempt leaves nothing on stdout.

One clock for the whole job (round 4).  The driver gives a multi-GPU bench 1 800 s; a chain of four attempts
with a deadline each could outlive that before the fallback that works is ever tried.  BENCH_TOTAL_BUDGET_S (default 1 500 s)
is divided by `attempt_deadline`: an attempt that still has a host-staged attempt (gloo + callbacks: nothing depends on
RCCL any more) AFTER it may use at most its own stage sum AND at most what is left minus FALLBACK_RESERVE_S (420 s) for that
attempt; the first of two such attempts gets 60 % of that.  Worst case at the defaults: attempt 0 killed at 648 s, attempt 1 at
1 080 s, the host-staged attempt starts with 420 s -- inside 1 500 s; every decision is printed to stderr.

Two ways in:
  python bench.py --gpus N            -> supervise_all(): this process spawns the N ranks itself;
  torchrun ... bench.py --gpus N      -> supervise_one(): every torchrun worker supervises ONE child (its rank); retries are
                                         agreed without talking: a failing child takes its peers down (fail-fast
                                         collectives, process-group timeouts), so every supervisor sees a failure and moves
                                         to the same next attempt, which meets on its own TCP store port.
"""
import os
import signal
import socket
import subprocess
import sys
import time

# (torch.distributed backend, exchange layer inside the library)
ATTEMPTS = [
    ("nccl", "rccl"),        # everything over RCCL/xGMI: the production form
    ("gloo", "rccl"),        # barrier/rendezvous over gloo, the data path still RCCL inside the library
    ("gloo", "callbacks"),   # nothing depends on RCCL any more: exchanges staged through the host over gloo
    ("nccl", "callbacks"),   # (RCCL for torch only; tried last: two RCCL-dependent attempts have failed by now)
]


TOTAL_BUDGET_S = 1500          # BENCH_TOTAL_BUDGET_S
FALLBACK_RESERVE_S = 420       # BENCH_FALLBACK_RESERVE_S: kept for the first attempt that does not depend on RCCL at all
EXIT_MARGIN_S = 10             # reporting, killing the children
MIN_ATTEMPT_S = 45             # an attempt with less than this left is not started


def attempt_deadline(attempts, k, remaining_s, stage_sum_s, reserve_s=FALLBACK_RESERVE_S):
    """Seconds attempt k of `attempts` may run when `remaining_s` of the total budget are left (pure arithmetic: tests/test_launch.py).
    0: do not start it."""
    usable = remaining_s - EXIT_MARGIN_S
    later_fallback = [j for j in range(k + 1, len(attempts)) if attempts[j] == ("gloo", "callbacks")]
    if later_fallback:
        usable -= reserve_s
        rccl_before_fallback = later_fallback[0] - k          # attempts (this one included) ahead of the host-staged one
        if rccl_before_fallback > 1:
            usable *= 0.6                                      # the first of several gets the larger share, not everything
    d = int(min(stage_sum_s, usable))
    return d if d >= MIN_ATTEMPT_S else 0


def attempts_from_env(env):
    """A caller that pins the backend (tests: several ranks on one GPU over gloo) gets exactly that configuration."""
    if env.get("BENCH_DIST_BACKEND") or env.get("BENCH_NO_RCCL"):
        backend = env.get("BENCH_DIST_BACKEND", "nccl")
        return [(backend, "callbacks" if env.get("BENCH_NO_RCCL") or backend != "nccl" else "rccl")]
    return list(ATTEMPTS)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _log(msg):
    print("[launch] " + msg, file=sys.stderr, flush=True)


def _kill(procs):
    for p in procs:
        if p.poll() is None:
            try:
                os.killpg(p.pid, signal.SIGTERM)
            except (ProcessLookupError, PermissionError):
                pass
    t_end = time.time() + 10
    for p in procs:
        while p.poll() is None and time.time() < t_end:
            time.sleep(0.05)
        if p.poll() is None:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                pass
            p.wait()


def _run_attempt(cmd, envs, deadline_s):
    """Start one child per env; -> (ok, description).  All children are dead on return."""
    procs = [subprocess.Popen(cmd, env=e, start_new_session=True) for e in envs]

    def on_term(signum, _frame):
        _kill(procs)
        os._exit(128 + signum)
    old = {s: signal.signal(s, on_term) for s in (signal.SIGTERM, signal.SIGINT)}
    t0 = time.time()
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(i, c) for i, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                _kill(procs)
                return False, "child %d exited with code %d" % bad[0]
            if all(c == 0 for c in codes):
                return True, "ok"
            if time.time() - t0 > deadline_s:
                _kill(procs)
                return False, "deadline of %d s passed" % deadline_s
            time.sleep(0.05)
    finally:
        _kill(procs)
        for s, h in old.items():
            signal.signal(s, h)


def _child_env(base, rank, local, world, port, attempt, backend, exchange, own_store):
    e = dict(base)
    e.update(RANK=str(rank), LOCAL_RANK=str(local), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
             BENCH_CHILD="1", BENCH_ATTEMPT=str(attempt), BENCH_DIST_BACKEND=backend)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: without it RCCL's ipc handles fail on this driver
    if exchange == "callbacks":
        e["BENCH_NO_RCCL"] = "1"
    else:
        e.pop("BENCH_NO_RCCL", None)
    # Only the FIRST attempt runs the two-stream transforms (the default since round 4: w's exchange from a second stream of the
    # same RCCL communicator -- a configuration that has not met a fabric yet).  Every later attempt, the second RCCL attempt
    # included, is the plainest configuration: one stream per rank, the exchange order round 3 validated.  A
    # context's defaults come from here (pm_ctx_create reads PM_NTT_OVERLAP once).
    if attempt > 0:
        e.setdefault("PM_NTT_OVERLAP", "0")
    if own_store:                                        # rank 0's child hosts the TCP store itself
        for k in ("TORCHELASTIC_USE_AGENT_STORE", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS"):
            e.pop(k, None)
    return e


class Budget:
    """The job's one clock.  stage_sum_s: what a healthy attempt may need at most (bench.py: import, key setup, proofs)."""

    def __init__(self, stage_sum_s, env=os.environ):
        self.t0 = time.time()
        self.total = float(env.get("BENCH_TOTAL_BUDGET_S", TOTAL_BUDGET_S))
        self.reserve = float(env.get("BENCH_FALLBACK_RESERVE_S", FALLBACK_RESERVE_S))
        self.stage_sum = float(stage_sum_s)

    def remaining(self):
        return self.total - (time.time() - self.t0)

    def deadline(self, attempts, k, who=""):
        rem = self.remaining()
        d = attempt_deadline(attempts, k, rem, self.stage_sum, self.reserve)
        _log("%sbudget: %.0f s of %.0f s left; attempt %d %s gets %d s (stage sum %.0f s, %.0f s reserved for a host-staged attempt)"
             % (who, rem, self.total, k, attempts[k], d, self.stage_sum, self.reserve if ("gloo", "callbacks") in attempts[k + 1:] else 0.0))
        return d


def supervise_all(script, argv, world, deadline_s=900):
    """`python bench.py --gpus N` without a launcher: spawn the N ranks, fall back through ATTEMPTS.  -> exit code.
    deadline_s: the stage sum of ONE healthy attempt; the job as a whole lives on Budget."""
    cmd = [sys.executable, script] + list(argv)
    why = "no attempt made"
    attempts = attempts_from_env(os.environ)
    budget = Budget(deadline_s)
    for k, (backend, exchange) in enumerate(attempts):
        d = budget.deadline(attempts, k)
        if not d:
            _log("attempt %d skipped: not enough of the budget left" % k)
            continue
        port = free_port()
        _log("attempt %d: %d ranks, torch.distributed=%s, exchange=%s, port %d" % (k, world, backend, exchange, port))
        envs = [_child_env(dict(os.environ, BENCH_ATTEMPT_LIMIT_S=str(d)), r, r, world, port, k, backend, exchange, True) for r in range(world)]
        ok, why = _run_attempt(cmd, envs, d)
        if ok:
            return 0
        _log("attempt %d failed: %s" % (k, why))
    _log("all attempts failed (last: %s)" % why)
    return 1


def _rendezvous_dir(env):
    """Where the supervisors of ONE externally launched job (one node) meet between attempts: a directory named by the launcher's
    run id, master port and process id (the workers of one launcher share their parent; back-to-back jobs that reuse a port and a
    run id -- the driver's N = 2, 4, 8 series -- must not meet each other's files)."""
    import tempfile
    # ... and a second incarnation under the same launcher (torchrun --max-restarts: same parent, run id and port) must not meet
    # the first one's markers either: the restart count is part of the name
    tag = "%s_%s_%s_%s" % (env.get("TORCHELASTIC_RUN_ID", "norun"), env.get("MASTER_PORT", "29500"), env.get("PM_LAUNCHER_PID", os.getppid()),
                           env.get("TORCHELASTIC_RESTART_COUNT", "0"))
    d = os.path.join(tempfile.gettempdir(), "pm_bench_" + "".join(c if c.isalnum() or c in "_-" else "_" for c in tag))
    os.makedirs(d, exist_ok=True)
    return d


def agree_on_retry(env, rank, world, k, wait_s=90.0):
    """Before attempt k > 0 under an external launcher: (a) every supervisor marks its arrival and waits (bounded) for the others --
    the ranks notice a failed attempt at different times, and a retry's rendezvous should not burn on that skew; (b) rank 0
    picks a FREE port for the retry's own TCP store and publishes it (the round-3 guess MASTER_PORT + 37 k could be taken or
    above 65535).  Files in one directory: the job is one node.  -> port (the validated guess if the agreement times out)."""
    d = _rendezvous_dir(env)
    port0 = int(env.get("MASTER_PORT", "29500"))
    guess = port0 + 37 * k
    if guess > 65000:
        guess = 20000 + (guess % 40000)
    port_file = os.path.join(d, "attempt%d.port" % k)
    if rank == 0:
        try:                                    # rank 0 publishes afresh on every arrival
            os.unlink(port_file)
        except OSError:
            pass
    mine = os.path.join(d, "attempt%d.rank%d" % (k, rank))
    open(mine, "w").close()
    # Which incarnation of the job a port file belongs to is said IN the file (launcher pid, restart count, attempt), not guessed
    # from its age: the supervisors notice a failed attempt seconds to minutes apart -- one child crashes at once, its peers sit
    # out their collective deadline -- so "rank 0 published long before I arrived" is the normal case, not a stale file.
    token = "%s:%s:%d" % (env.get("PM_LAUNCHER_PID", os.getppid()), env.get("TORCHELASTIC_RESTART_COUNT", "0"), k)
    if rank == 0:
        port = free_port()
        tmp = port_file + ".tmp"
        with open(tmp, "w") as f:
            f.write("%d %s" % (port, token))
        os.replace(tmp, port_file)
    t_end = time.time() + wait_s
    port = None
    while time.time() < t_end:
        if port is None and os.path.exists(port_file):
            try:
                fields = open(port_file).read().split()
                if len(fields) == 2 and fields[1] == token:
                    port = int(fields[0])
            except (ValueError, OSError):
                port = None
        if port is not None and all(os.path.exists(os.path.join(d, "attempt%d.rank%d" % (k, r))) for r in range(world)):
            return port
        time.sleep(0.1)
    _log("rank %d: retry %d not confirmed by every supervisor within %.0f s; going on with port %d" % (rank, k, wait_s, port or guess))
    return port or guess


def _leave_rendezvous(env, rank):
    """A supervisor's exit: its own markers go, rank 0's port files go, and whoever leaves last removes the directory."""
    d = _rendezvous_dir(env)
    try:
        for f in os.listdir(d):
            if f.endswith(".rank%d" % rank) or (rank == 0 and ".port" in f):
                try:
                    os.unlink(os.path.join(d, f))
                except OSError:
                    pass
        os.rmdir(d)                # fails while another rank's markers are still there: the last one out succeeds
    except OSError:
        pass


def supervise_one(script, argv, deadline_s=900):
    """A torchrun worker: supervise ONE child that does this rank's work.  -> exit code."""
    cmd = [sys.executable, script] + list(argv)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    port0 = int(os.environ.get("MASTER_PORT", "29500"))
    why = "no attempt made"
    attempts = attempts_from_env(os.environ)
    budget = Budget(deadline_s)
    who = "rank %d: " % rank
    try:
        for k, (backend, exchange) in enumerate(attempts):
            # attempt 0 meets on the launcher's own store; a retry needs a store no earlier attempt has written to
            port = port0 if k == 0 else agree_on_retry(os.environ, rank, world, k, wait_s=min(90.0, max(5.0, budget.remaining() / 10)))
            d = budget.deadline(attempts, k, who if rank else "")
            if not d:
                _log("%sattempt %d skipped: not enough of the budget left" % (who, k))
                continue
            if rank == 0:
                _log("attempt %d: %d ranks under the external launcher, torch.distributed=%s, exchange=%s, port %d" % (k, world, backend, exchange, port))
            env = dict(os.environ, BENCH_ATTEMPT_LIMIT_S=str(d))
            ok, why = _run_attempt(cmd, [_child_env(env, rank, local, world, port, k, backend, exchange, k > 0)], d)
            if ok:
                return 0
            _log("rank %d, attempt %d failed: %s" % (rank, k, why))
        return 1
    finally:
        _leave_rendezvous(os.environ, rank)


class Watchdog:
    """In a rank: exits the process (status 124) when a stage outlives its limit -- the backstop behind the library's own
    collective deadlines (a kernel that can not be aborted, a rendezvous that never completes)."""

    def __init__(self, rank):
        import threading
        self.rank, self.lock = rank, threading.Lock()
        self.stage, self.deadline = "start", None
        threading.Thread(target=self._run, daemon=True).start()

    def stage_begin(self, name, limit_s):
        with self.lock:
            self.stage, self.deadline = name, time.time() + limit_s

    def stage_end(self):
        with self.lock:
            self.deadline = None

    def _run(self):
        while True:
            time.sleep(0.5)
            with self.lock:
                late = self.deadline is not None and time.time() > self.deadline
                stage = self.stage
            if late:
                print("[bench rank %d] stage '%s' exceeded its limit -- exiting 124" % (self.rank, stage), file=sys.stderr, flush=True)
                os._exit(124)
