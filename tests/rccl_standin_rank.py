"""One RANK PROCESS of the librccl stand-in tests (tests/test_rccl_standin.py; TEST INFRASTRUCTURE).

    python tests/rccl_standin_rank.py --dir D --rank r --world N --curve C --log-nr K [--mode prove|loop] ...

A fresh child of the test (never a torch process: comm.hip's dlopen("librccl.so.1") must find the stand-in through
LD_LIBRARY_PATH, not a librccl torch already holds).  It does what a Rust / C++ host of INTEGRATION.md §4 does:
  rank 0 makes the 128-byte id (pm_comm_rccl_unique_id) and publishes it as a file; every rank calls pm_comm_rccl_create
  (ncclCommInitRank inside the library), checks the fabric once (all-gather of the ranks, all-to-all of tagged 64-byte
  blocks), joins its context, makes ITS share of a PM_SHARD_VECTOR key (pm_pk_generate_sharded) and proves with
  pm_host_prove_sharded.
mode prove: `--reps` proofs, written to D/result_<rank>.json.
mode loop : proofs until a collective fails (the test kills or stops a peer); the rank then reports how long the failing call
            took, what pm_comm_last_error says, and exits with status 9 (PM_ERR_COMM) -- a host's job after a failed communicator
            is to exit non-zero (include/polymath_hip.h, pm_comm contract)."""
import argparse
import ctypes as ct
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def _publish(path, data):
    tmp = path + ".tmp.%d" % os.getpid()
    with open(tmp, "wb" if isinstance(data, bytes) else "w") as f:
        f.write(data)
    os.replace(tmp, path)


def _await_file(path, seconds):
    t_end = time.time() + seconds
    while not os.path.exists(path):
        if time.time() > t_end:
            raise TimeoutError("no " + path)
        time.sleep(0.005)
    return open(path, "rb").read()


def fabric_check(comm, rank, world):
    """The check polymath_amd/distributed.py: make_comm runs before a prover depends on the block order, without torch."""
    probe = comm.all_gather(np.array([rank], dtype=np.int64)).reshape(-1).tolist()
    assert probe == list(range(world)), probe
    hip = ct.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ct.POINTER(ct.c_void_p), ct.c_size_t]
    hip.hipMemcpy.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_size_t, ct.c_int]
    hip.hipFree.argtypes = [ct.c_void_p]
    nbytes = 64 * world
    d_send, d_recv = ct.c_void_p(), ct.c_void_p()
    assert hip.hipMalloc(ct.byref(d_send), nbytes) == 0 and hip.hipMalloc(ct.byref(d_recv), nbytes) == 0
    send = np.repeat(np.arange(world, dtype=np.int64) + rank * world, 8)        # block p: 8 copies of (sender * world + receiver)
    recv = np.full(8 * world, -1, dtype=np.int64)
    assert hip.hipMemcpy(d_send, send.ctypes.data_as(ct.c_void_p), nbytes, 1) == 0
    assert hip.hipMemcpy(d_recv, recv.ctypes.data_as(ct.c_void_p), nbytes, 1) == 0
    comm.all_to_all_device(d_send.value, d_recv.value, 64)                      # the null stream
    assert hip.hipDeviceSynchronize() == 0
    assert hip.hipMemcpy(recv.ctypes.data_as(ct.c_void_p), d_recv, nbytes, 2) == 0
    assert recv.reshape(world, 8)[:, 0].tolist() == [r * world + rank for r in range(world)], recv.tolist()
    # device all-gather (the witness broadcast of phase 1)
    assert hip.hipMemcpy(d_send, send.ctypes.data_as(ct.c_void_p), 64, 1) == 0
    comm.all_gather_device(d_send.value, d_recv.value, 64)
    assert hip.hipDeviceSynchronize() == 0
    assert hip.hipMemcpy(recv.ctypes.data_as(ct.c_void_p), d_recv, nbytes, 2) == 0
    assert recv.reshape(world, 8)[:, 0].tolist() == [r * world for r in range(world)], recv.tolist()
    hip.hipFree(d_send)
    hip.hipFree(d_recv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", required=True)
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--curve", default="bls12_381")
    ap.add_argument("--log-nr", type=int, default=12)
    ap.add_argument("--mode", default="prove", choices=["prove", "loop"])
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--transcript", default="merlin")
    ap.add_argument("--ntt-overlap", type=int, default=-1)
    ap.add_argument("--timeout-ms", type=int, default=60000)
    ap.add_argument("--seed", type=int, default=0x5CC1)
    ap.add_argument("--swap-order-on-rank", type=int, default=-1,
                    help="this rank issues an all-gather where its peers issue an all-to-all (the stand-in must FAIL the communicator)")
    a = ap.parse_args()
    assert "torch" not in sys.modules
    from polymath_amd import api, circuits as PC
    from polymath_amd.polymath import Polymath, PolymathProverError
    from oracle.pyref.fields import CURVES          # test infrastructure: the modulus for the seeded draws only
    rank, world = a.rank, a.world
    uid_path = os.path.join(a.dir, "uid")
    if rank == 0:
        _publish(uid_path, api.Comm.rccl_unique_id())
    uid = _await_file(uid_path, 120)
    assert len(uid) == 128
    comm = api.Comm.rccl(uid, rank, world, 0)
    comm.set_timeout_ms(a.timeout_ms)
    kind = comm.kind
    with open("/proc/self/maps") as f:
        loaded = sorted({line.split()[-1] for line in f if "librccl" in line})
    fabric_check(comm, rank, world)

    if a.swap_order_on_rank >= 0:
        # NCCL's same-order rule, violated on purpose: one rank's next collective is of another kind than its peers'
        try:
            if rank == a.swap_order_on_rank:
                comm.all_gather(np.zeros(8, dtype=np.int64))
            else:
                hip = ct.CDLL("libamdhip64.so")
                hip.hipMalloc.argtypes = [ct.POINTER(ct.c_void_p), ct.c_size_t]
                d = ct.c_void_p()
                assert hip.hipMalloc(ct.byref(d), 128 * world) == 0
                comm.all_to_all_device(d.value, d.value + 64 * world, 64)
                assert hip.hipDeviceSynchronize() == 0
                comm.all_gather(np.zeros(8, dtype=np.int64))       # the next call sees the failed communicator
            status = 0
        except api.PolymathError as e:
            status = e.status
        _publish(os.path.join(a.dir, "result_%d.json" % rank), json.dumps({"status": status, "failed": comm.failed, "last_error": comm.last_error()}))
        sys.exit(9 if status == 9 else 1)

    c = CURVES[a.curve]
    lc = PC.synthetic_r1cs_native(a.curve, (1 << a.log_nr) - 100)
    g = PC.SplitMix64(a.seed)
    x, z = g.fr(c.r), g.fr(c.r)
    pm = Polymath(a.curve, a.transcript, device=0)
    pm.ctx.set_comm(comm)
    if a.ntt_overlap >= 0:
        pm.ctx.set_option("ntt_overlap", a.ntt_overlap)
    pk = pm.setup(lc, x, z, shard_rank=rank, shard_count=world, layout="vector")
    out = {"rank": rank, "world": world, "kind": kind, "librccl_mapped": loaded, "pid": os.getpid(), "n": pk.n,
           "ntt_overlap": pm.ctx.get_option("ntt_overlap"), "proofs": []}
    if a.mode == "prove":
        for rep in range(a.reps):
            r_a = [g.fr(c.r), g.fr(c.r)]
            out["proofs"].append(pm.prove_native(pk, lc.inst_limbs, lc.wit_limbs, r_a).hex())
        _publish(os.path.join(a.dir, "result_%d.json" % rank), json.dumps(out))
        pk.free()
        pm.ctx.set_comm(None)
        comm.close()
        return 0
    # loop: prove until the communicator fails
    done = 0
    while done < 100000:
        r_a = [g.fr(c.r), g.fr(c.r)]
        t0 = time.time()
        try:
            pm.prove_native(pk, lc.inst_limbs, lc.wit_limbs, r_a)
        except (PolymathProverError, api.PolymathError) as e:
            out.update(status=e.status, failing_call_s=time.time() - t0, failed=comm.failed, last_error=comm.last_error(), proofs_done=done,
                       failed_at=time.time())
            _publish(os.path.join(a.dir, "result_%d.json" % rank), json.dumps(out))
            # a failed communicator stays failed: the next proof returns at once
            t1 = time.time()
            try:
                pm.prove_native(pk, lc.inst_limbs, lc.wit_limbs, r_a)
                again = 0
            except (PolymathProverError, api.PolymathError) as e2:
                again = e2.status
            out.update(second_call_status=again, second_call_s=time.time() - t1)
            _publish(os.path.join(a.dir, "result_%d.json" % rank), json.dumps(out))
            os._exit(9 if e.status == 9 else 1)      # no teardown through a dead fabric: exit non-zero, the launcher's job from here
        done += 1
        _publish(os.path.join(a.dir, "progress_%d" % rank), str(done))
    return 1


if __name__ == "__main__":
    sys.exit(main())
