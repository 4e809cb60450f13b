#!/usr/bin/env python3
"""Per-rank cost of an N-GPU PM_SHARD_VECTOR proof, emulated on ONE GPU (no 8-GPU node needed): the N ranks run as N
threads over pm_comm_local_create with pm_comm_local_set_serialize -- between collectives only one rank runs at a time, so
every rank's kernels take what they would take alone; the exchanges are device-to-device copies (their xGMI cost is NOT
in the number).  per-rank ms = wall time of K proofs / (K N).
  python tools/shard_emulation.py --ranks 8 --log-constraints 20 --steps 3 [--layout pairs]"""
import argparse, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# PM_OPT_MSM_OVERLAP (default on): a rank enqueues its [a] and [c] MSM pipelines on two streams from ONE host thread, with no collective
# in between, so the overlap happens entirely inside the rank's turn -- as it does on a GPU of its own.  (Round 2's helper THREAD
# escaped the turnstile; the emulation then ran with the overlap off.)
# The N emulated ranks share ONE process, hence one set of hardware queues (ROCm default: 4); a real rank has its process's queues
# to itself.  With 2 N streams on 4 queues a rank's two MSM streams often land on the same queue and serialise (per-rank busy time
# bimodal: 12.0 / 13.3 ms at N = 8); 8 queues restore what a rank sees on a GPU of its own (profiles/r03_m_*).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from polymath_amd import api, circuits as PC   # noqa: E402
from polymath_amd.polymath import Polymath, FIELDS   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--log-constraints", type=int, default=20)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--curve", default="bls12_381")
ap.add_argument("--layout", default="vector")
ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="pm_ctx_set_option on every rank's context, e.g. --opt msm_overlap=0 --opt tables=wide")
a = ap.parse_args()
opts = {kv.split("=", 1)[0]: (kv.split("=", 1)[1] if not kv.split("=", 1)[1].lstrip("-").isdigit() else int(kv.split("=", 1)[1])) for kv in a.opt}
N, curve = a.ranks, a.curve
r = FIELDS[curve]["r"]
lc = PC.synthetic_r1cs_native(curve, (1 << a.log_constraints) - 100)
g = PC.SplitMix64(0xBE7C4)
x, z, r_a = g.fr(r), g.fr(r), [g.fr(r), g.fr(r)]
# the assignment in PINNED host memory, as bench.py hands it over (a pageable buffer costs the runtime's bounce copy on the host thread)
import ctypes as ct, numpy as np   # noqa: E402
_hip = ct.CDLL("libamdhip64.so")
_hip.hipHostMalloc.argtypes = [ct.POINTER(ct.c_void_p), ct.c_size_t, ct.c_uint]


def pinned_copy(arr):
    arr = np.ascontiguousarray(arr)
    p = ct.c_void_p()
    if _hip.hipHostMalloc(ct.byref(p), arr.nbytes, 0) != 0:
        return arr
    out = np.ctypeslib.as_array(ct.cast(p, ct.POINTER(ct.c_uint64)), shape=(arr.nbytes // 8,)).reshape(arr.shape)
    out[...] = arr
    return out


comms = api.Comm.local_group(N, serialize=True)
pms = [Polymath(curve, "merlin", device=0) for _ in range(N)]
for k in range(N):
    pms[k].ctx.set_comm(comms[k])
    for name, value in opts.items():
        pms[k].ctx.set_option(name, value)
t0 = time.time()
pks = [pms[k].setup(lc, x, z, shard_rank=k, shard_count=N, layout=a.layout) for k in range(N)]
setup_s = time.time() - t0
proofs, timings = [None] * N, [None] * N
x_host, w_host = pinned_copy(lc.inst_limbs), pinned_copy(lc.wit_limbs)


def body(k, steps):
    for _ in range(steps):
        proofs[k] = pms[k].prove_native(pks[k], x_host, w_host, r_a)
    timings[k] = pms[k].ctx.timings()


def run(steps):
    th = [threading.Thread(target=body, args=(k, steps)) for k in range(N)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    return time.perf_counter() - t0


run(1)
for c in comms:
    c.busy_ms()
dt = run(a.steps)
busy = [c.busy_ms() / a.steps for c in comms]
assert all(p == proofs[0] for p in proofs)
print(json.dumps({"ranks": N, "layout": a.layout, "curve": curve, "log_constraints": a.log_constraints, "steps": a.steps,
                  "emulated_ms_per_rank": dt / a.steps / N * 1e3,
                  "busy_ms_per_rank": [round(b, 3) for b in busy], "busy_ms_max_rank": max(busy), "wall_ms_per_proof_all_ranks_serialised": dt / a.steps * 1e3,
                  "options": {name: pms[0].ctx.get_option(name) for name in api.OPTIONS}, "setup_s_all_ranks": setup_s, "stage_ms_rank0": {k: round(v, 3) for k, v in timings[0].items()},
                  "stage_ms_last_rank": {k: round(v, 3) for k, v in timings[N - 1].items()},
                  "note": "exchanges are local device-to-device copies: xGMI latency / bandwidth not included", "proof": proofs[0].hex()}))
