"""The RCCL branch of polymath_amd/csrc/comm.hip (RcclComm) with REAL PEER PROCESSES on one GPU, through a test-only stand-in
for librccl.so.1 (tests/native/fake_rccl.hip -- read its header: it moves bytes through host shared memory and says nothing
about xGMI or RCCL's kernels).  What runs for the first time with a peer: the unique-id hand-off, ncclCommInitRank on
N = 2, 4 processes (a GPU box admits 6 processes on its card: the test runner and 4 ranks), per-peer byte counts of ncclAllToAll, the host all-gather's staging, two user streams on one
communicator (PM_OPT_NTT_OVERLAP), the watchdog's ncclCommAbort on a stalled peer, ncclCommGetAsyncError on a dead one.
Rank processes are fresh children that never import torch (tests/rccl_standin_rank.py); the proofs they return are compared
with the CPU ORACLE's bytes (oracle/cpp, its own setup) on the same circuit, trapdoors and r_a.

CPU part: the stand-in builds and exports exactly the symbols comm.hip resolves."""
import json
import os
import re
import signal
import subprocess
import sys
import time

import pytest

import standin_rccl as SR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RANK_PROGRAM = os.path.join(ROOT, "tests", "rccl_standin_rank.py")
SEED = 0x5CC1


def test_stand_in_exports_what_comm_hip_resolves():
    """comm.hip:198-205 dlsym()s eight names; the stand-in exports those eight and nothing else of ncclXxx, and says in its
    header and in every error string that it is a stand-in."""
    lib = SR.build()
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l)
    assert exported == sorted(SR.SYMBOLS), exported
    comm_src = open(os.path.join(ROOT, "polymath_amd", "csrc", "comm.hip")).read()
    assert sorted(set(re.findall(r'sym\("(nccl[A-Za-z]+)"', comm_src))) == sorted(SR.SYMBOLS)
    head = open(SR.SRC).read()[:1200]
    assert "TEST INFRASTRUCTURE ONLY" in head and "never loaded by the product" in head
    # the product never names the stand-in
    for dirpath, _, files in os.walk(os.path.join(ROOT, "polymath_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h", ".hpp")):
                assert "fake_rccl" not in open(os.path.join(dirpath, f), errors="ignore").read(), f


# ----------------------------------------------------------------------------------------------------------- GPU
def _spawn(tmp, world, extra_args=(), env_extra=None, per_rank_args=None):
    SR.build()
    env = SR.rank_env(log_prefix=os.path.join(tmp, "standin"), extra=env_extra)
    procs = []
    for r in range(world):
        cmd = [sys.executable, RANK_PROGRAM, "--dir", tmp, "--rank", str(r), "--world", str(world)] + list(extra_args) + list((per_rank_args or {}).get(r, []))
        log = open(os.path.join(tmp, "rank_%d.log" % r), "w")
        procs.append(subprocess.Popen(cmd, env=env, stdout=log, stderr=subprocess.STDOUT, cwd=ROOT))
    return procs


def _wait_all(procs, seconds, tmp):
    t_end = time.time() + seconds
    codes = [None] * len(procs)
    try:
        for i, p in enumerate(procs):
            try:
                codes[i] = p.wait(max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                pass
    finally:
        for p in procs:           # never leave a rank (or a stopped rank) behind
            if p.poll() is None:
                try:
                    p.send_signal(signal.SIGCONT)
                    p.kill()
                except OSError:
                    pass
                p.wait(30)
    if any(c is None for c in codes):
        raise TimeoutError("ranks still running after %d s: %s\n%s" % (seconds, codes, _logs(tmp, len(procs))))
    return codes


def _logs(tmp, world):
    out = []
    for r in range(world):
        path = os.path.join(tmp, "rank_%d.log" % r)
        if os.path.exists(path):
            out.append("--- rank %d ---\n%s" % (r, open(path).read()[-3000:]))
    return "\n".join(out)


def _results(tmp, world):
    return [json.load(open(os.path.join(tmp, "result_%d.json" % r))) for r in range(world)]


def _standin_logs(tmp, world):
    return [json.load(open(os.path.join(tmp, "standin.rank%d.json" % r))) for r in range(world)]


_ORACLE_KEYS = {}


def _oracle_proofs(oracle, curve, log_nr, reps, transcript="merlin"):
    """The CPU restatement's proofs for the rank program's draws: x, z, then (r_a0, r_a1) per proof from SplitMix64(SEED)."""
    from oracle import driver as DR
    from oracle.pyref import serialize as SE, transcripts as T
    from oracle.pyref.fields import CURVES
    from polymath_amd import circuits as PC
    c = CURVES[curve]
    lc = PC.synthetic_r1cs_native(curve, (1 << log_nr) - 100)
    g = PC.SplitMix64(SEED)
    x, z = g.fr(c.r), g.fr(c.r)
    key = (curve, log_nr)
    if key not in _ORACLE_KEYS:
        class Shape:
            pass
        q = Shape()
        q.m0, q.mw, q.nr = lc.m0, lc.mw, lc.nr
        q.csr_arrays = [(a.rowptr, a.col, a.val) for a in lc.csrs]
        _ORACLE_KEYS.clear()                       # one CPU key at a time (2^16: ~1 GB)
        _ORACLE_KEYS[key] = oracle.OraclePk(curve, q, x, z, os.cpu_count() or 4)
    opk = _ORACLE_KEYS[key]
    omega = oracle.fr_from_mont_limbs(curve, opk.omega_limbs)[0]
    out = []
    for _ in range(reps):
        r_a = [g.fr(c.r), g.fr(c.r)]
        po = DR.prove(opk, opk.n, opk.sigma, omega, lc.instance, None, r_a, T.make_transcripts(c)[transcript], w_limbs=lc.wit_limbs)
        out.append(SE.ser_proof(c, po).hex())
    return out


def _prove_case(oracle, tmp, world, curve, log_nr, overlap, reps=2, seconds=240):
    args = ["--curve", curve, "--log-nr", str(log_nr), "--mode", "prove", "--reps", str(reps), "--seed", str(SEED)]
    if overlap is not None:
        args += ["--ntt-overlap", str(overlap)]
    procs = _spawn(tmp, world, args)
    want = _oracle_proofs(oracle, curve, log_nr, reps)            # on the CPU while the ranks run
    codes = _wait_all(procs, seconds, tmp)
    assert codes == [0] * world, (codes, _logs(tmp, world))
    res = _results(tmp, world)
    for r, out in enumerate(res):
        assert out["kind"] == "rccl", out["kind"]                                      # comm.hip's RcclComm, not the local / callback kinds
        assert any("tests/native/_build/fake_rccl/librccl.so.1" in p for p in out["librccl_mapped"]), out["librccl_mapped"]
        assert not any("/opt/rocm" in p or "torch" in p for p in out["librccl_mapped"]), out["librccl_mapped"]
        assert out["proofs"] == want, (r, out["proofs"], want)
    logs = _standin_logs(tmp, world)
    for lg in logs:
        assert lg["world"] == world and lg["async_error"] == 0 and lg["device_wait_timed_out"] == 0 and lg["end"] == "destroy"
        assert lg["all_to_all"] >= 1 + 4 * reps and lg["all_gather"] >= 2 + 3 * reps, lg     # fabric check + 4 transforms, 3 host exchanges per proof
    return res, logs


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])          # the GPU boxes allow 6 processes on the card at once: pytest itself + 4 ranks
@pytest.mark.parametrize("overlap", [1, 0])
def test_sharded_proofs_over_rccl_branch_with_peer_processes(oracle, tmp_path, world, overlap):
    """2^12 - 100 gates (n = 8192), N = 2 / 4 rank PROCESSES on one GPU over comm.hip's RcclComm: id hand-off through a
    file, ncclCommInitRank, the fabric check, pm_pk_generate_sharded (PM_SHARD_VECTOR) and two pm_host_prove_sharded proofs per
    rank, all byte-equal to the CPU oracle's.  PM_OPT_NTT_OVERLAP on: w's all-to-all is issued on a SECOND stream of the same
    communicator (the stand-in counts the distinct streams and runs collectives in issue order, like NCCL); off: one stream."""
    res, logs = _prove_case(oracle, str(tmp_path), world, "bls12_381", 12, overlap)
    assert all(out["ntt_overlap"] == overlap for out in res)
    streams = [lg["distinct_user_streams"] for lg in logs]
    # the null stream (fabric check), the context's stream, the communicator's side stream; + the helper stream with overlap
    assert all(s >= (4 if overlap else 3) for s in streams), streams
    if not overlap:
        assert all(s == 3 for s in streams), streams


@pytest.mark.gpu
@pytest.mark.parametrize("world,curve", [(2, "bls12_381"), (4, "bls12_381"), (4, "bn254")])
def test_sharded_proofs_over_rccl_branch_at_2p16(oracle, tmp_path, world, curve):
    """2^16 - 100 gates (n = 2^17: three-pass local transforms, window tables on the shards, 2 MB per rank per all-to-all),
    default options (NTT overlap on), against the CPU oracle's bytes."""
    _prove_case(oracle, str(tmp_path), world, curve, 16, None, reps=2, seconds=420)


@pytest.mark.gpu
def test_order_mismatch_fails_the_communicator(tmp_path):
    """NCCL's rule -- every rank issues the same collectives in the same order on a communicator -- is ENFORCED by the stand-in:
    one rank issues an all-gather where its peers issue an all-to-all; the stand-in reports ncclInvalidUsage as the
    asynchronous error, comm.hip's watchdog (ncclCommGetAsyncError) aborts, every rank's call returns PM_ERR_COMM."""
    tmp, world = str(tmp_path), 2
    procs = _spawn(tmp, world, ["--swap-order-on-rank", "1", "--timeout-ms", "20000"])
    codes = _wait_all(procs, 120, tmp)
    assert codes == [9] * world, (codes, _logs(tmp, world))
    for out in _results(tmp, world):
        assert out["status"] == 9 and out["failed"] and "asynchronous RCCL error" in out["last_error"] and "invalid usage" in out["last_error"], out
    assert "ORDER MISMATCH" in _logs(tmp, world)


def _await_progress(tmp, world, at_least, seconds, procs):
    t_end = time.time() + seconds
    while time.time() < t_end:
        done = []
        for r in range(world):
            try:
                done.append(int(open(os.path.join(tmp, "progress_%d" % r)).read() or 0))
            except (OSError, ValueError):
                done.append(0)
        if min(done) >= at_least:
            return done
        if any(p.poll() is not None for p in procs):
            raise AssertionError("a rank exited before the fault was injected:\n" + _logs(tmp, world))
        time.sleep(0.01)
    raise TimeoutError("ranks did not reach %d proofs: %s\n%s" % (at_least, done, _logs(tmp, world)))


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_killed_peer_ends_every_survivor_with_comm_error(tmp_path, world):
    """One rank is SIGKILLed while the group proves in a loop (mid-proof: the kill lands at an arbitrary point of some proof).
    The stand-in reports the dead peer the way RCCL does (ncclRemoteError as the asynchronous error); comm.hip's watchdog picks
    it up, calls ncclCommAbort (which ends the device-side wait), and EVERY survivor's pm_host_prove_sharded returns
    PM_ERR_COMM well inside the 20 s deadline, says why, fails the next call at once, and exits non-zero."""
    tmp = str(tmp_path)
    procs = _spawn(tmp, world, ["--mode", "loop", "--log-nr", "12", "--timeout-ms", "20000"])
    try:
        _await_progress(tmp, world, 3, 180, procs)
        victim = world - 1
        t_kill = time.time()
        procs[victim].kill()
        codes = _wait_all(procs, 60, tmp)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    elapsed = time.time() - t_kill
    assert codes[victim] == -signal.SIGKILL
    assert [c for r, c in enumerate(codes) if r != victim] == [9] * (world - 1), (codes, _logs(tmp, world))
    assert elapsed < 15, elapsed
    for r in range(world):
        if r == victim:
            continue
        out = json.load(open(os.path.join(tmp, "result_%d.json" % r)))
        assert out["status"] == 9 and out["failed"] and out["proofs_done"] >= 3, out
        assert "RCCL" in out["last_error"] or "collective" in out["last_error"] or "aborted" in out["last_error"], out["last_error"]
        assert out["second_call_status"] == 9 and out["second_call_s"] < 1.0, out          # sticky, and immediate
        assert out["failing_call_s"] < 15, out


@pytest.mark.gpu
def test_stalled_peer_trips_the_watchdog_deadline(tmp_path):
    """A peer that is alive but does not arrive (SIGSTOP; the stand-in's liveness check is off, as for a peer stuck in a driver
    call): nothing reports an error, the survivor's collective sits in its device-side wait until comm.hip's watchdog passes
    the communicator's deadline (3 s here), calls ncclCommAbort and the proof returns PM_ERR_COMM -- not a hang."""
    tmp, world = str(tmp_path), 2
    procs = _spawn(tmp, world, ["--mode", "loop", "--log-nr", "12", "--timeout-ms", "3000"], env_extra={"PM_FAKE_RCCL_NO_LIVENESS": "1"})
    try:
        _await_progress(tmp, world, 3, 180, procs)
        t_stop = time.time()
        procs[1].send_signal(signal.SIGSTOP)
        code0 = procs[0].wait(60)
        elapsed = time.time() - t_stop
    finally:
        for p in procs:
            if p.poll() is None:
                p.send_signal(signal.SIGCONT)
                p.kill()
                p.wait(30)
    assert code0 == 9, (code0, _logs(tmp, world))
    out = json.load(open(os.path.join(tmp, "result_0.json")))
    assert out["status"] == 9 and out["failed"], out
    assert "did not complete within 3000 ms" in out["last_error"] or "was not reached by its stream" in out["last_error"], out["last_error"]
    assert 2.5 < out["failing_call_s"] < 30 and elapsed < 40, (out, elapsed)
    lg = json.load(open(os.path.join(tmp, "standin.rank0.json")))
    assert lg["end"] == "abort" and lg["device_wait_timed_out"] == 0        # ended by ncclCommAbort, not by the stand-in's own backstop
