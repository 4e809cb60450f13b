#!/usr/bin/env python3
"""Soak: many proofs on the same contexts -- does anything grow?  (A prover that serves requests for days must not leak
device memory, pinned memory, events or host heap per proof.)

  python tools/soak.py --log-constraints 18 --proofs 300 --ranks 4 --sharded-proofs 100

Proves the same synthetic circuit `--proofs` times on one context (alternating r_a, host-input entry point), then
`--sharded-proofs` times as ONE proof over `--ranks` rank threads (pm_comm_local_create), and prints one JSON line with the
free HBM (hipMemGetInfo) and the process RSS after a warm-up and at the end of each leg.  Every proof's bytes are compared
with the first one's of its r_a.  Exit status 1 when a leg's STEADY growth -- samples every 25 proofs, the largest single step left
out (the runtime's one-off hardware-queue creation: ~190 MB of RSS at a random proof) -- exceeds --tolerance-mb (default 16)."""
import argparse
import ctypes as ct
import json
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rss_mb():
    with open("/proc/self/status") as f:
        for line in f:
            if line.startswith("VmRSS:"):
                return int(line.split()[1]) / 1024.0
    return 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-constraints", type=int, default=18)
    ap.add_argument("--proofs", type=int, default=300)
    ap.add_argument("--ranks", type=int, default=4)
    ap.add_argument("--sharded-proofs", type=int, default=100)
    ap.add_argument("--tolerance-mb", type=float, default=16.0)
    a = ap.parse_args()
    from polymath_amd import api, circuits as PC
    from polymath_amd.polymath import FIELDS, Polymath
    hip = ct.CDLL("libamdhip64.so")

    def free_hbm_mb():
        fr, tot = ct.c_size_t(), ct.c_size_t()
        hip.hipMemGetInfo(ct.byref(fr), ct.byref(tot))
        return fr.value / 2.0**20

    curve = "bls12_381"
    r = FIELDS[curve]["r"]
    lc = PC.synthetic_r1cs_native(curve, (1 << a.log_constraints) - 100)
    g = PC.SplitMix64(0x50A4)
    x, z = g.fr(r), g.fr(r)
    ras = [[g.fr(r), g.fr(r)] for _ in range(2)]
    out = {"log_constraints": a.log_constraints, "legs": []}
    bad = False

    def steady_growth(trace):
        """growth of (free HBM down, RSS up) over samples taken every 25 proofs, WITHOUT the largest single step: the ROCm runtime
        creates a hardware queue (~190 MB of RSS, 2 MB of HBM) whenever it first maps a stream onto one -- once, hundreds of proofs
        into a run if it pleases -- while a leak grows at every step"""
        dh = [trace[i][0] - trace[i + 1][0] for i in range(len(trace) - 1)]
        dr = [trace[i + 1][1] - trace[i][1] for i in range(len(trace) - 1)]
        return (sum(dh) - max(dh + [0.0]), sum(dr) - max(dr + [0.0]))

    def leg(name, count, prove, warm=5):
        nonlocal bad
        want = [prove(k) for k in range(2)]
        for k in range(warm):
            assert prove(k % 2) == want[k % 2]
        h0, r0 = free_hbm_mb(), rss_mb()
        trace = [(h0, r0)]
        for k in range(count):
            if prove(k % 2) != want[k % 2]:
                raise SystemExit("%s: proof %d differs from the first proof of its r_a" % (name, k))
            if (k + 1) % 25 == 0:
                trace.append((free_hbm_mb(), rss_mb()))
        h1, r1 = free_hbm_mb(), rss_mb()
        trace.append((h1, r1))
        hg, rg = steady_growth(trace)
        rec = {"leg": name, "proofs": count, "free_hbm_mb_before": round(h0, 1), "free_hbm_mb_after": round(h1, 1),
               "hbm_growth_mb": round(h0 - h1, 1), "rss_mb_before": round(r0, 1), "rss_mb_after": round(r1, 1), "rss_growth_mb": round(r1 - r0, 1),
               "steady_hbm_growth_mb": round(hg, 1), "steady_rss_growth_mb": round(rg, 1)}
        out["legs"].append(rec)
        if hg > a.tolerance_mb or rg > a.tolerance_mb:
            bad = True

    pm = Polymath(curve, "merlin", device=0)
    pk = pm.setup(lc, x, z)
    leg("one context, host inputs", a.proofs, lambda k: pm.prove_native(pk, lc.inst_limbs, lc.wit_limbs, ras[k]))
    pk.free()

    N = a.ranks
    comms = api.Comm.local_group(N)
    pms = [Polymath(curve, "merlin", device=0) for _ in range(N)]
    for q in range(N):
        pms[q].ctx.set_comm(comms[q])
    pks = [pms[q].setup(lc, x, z, shard_rank=q, shard_count=N, layout="vector") for q in range(N)]

    # persistent rank threads (what a serving process has): thread q proves `count` times in a row, the collectives keep the
    # ranks in step.  (A fresh thread per proof would measure glibc's per-thread malloc arenas, not the library.)
    # The verdict is the growth over the SECOND HALF of the leg: the ROCm runtime creates hardware queues (and their scratch:
    # ~190 MB of RSS, 2 MB of HBM each) when it first maps a stream onto one, which can happen tens of proofs into a run.
    def sharded_leg(name, count, warm=5):
        nonlocal bad
        want = [None, None]
        wrong = [0] * N
        marks = {}

        def body(q):
            for k in range(count + warm + 2):
                done = k - warm - 2
                if q == 0 and done >= 0 and done % 25 == 0:
                    marks.setdefault("trace", []).append([done, round(free_hbm_mb(), 1), round(rss_mb(), 1)])
                if q == 0 and done == count // 2:
                    marks["half"] = (free_hbm_mb(), rss_mb())
                p = pms[q].prove_native(pks[q], lc.inst_limbs, lc.wit_limbs, ras[k % 2])
                if q == 0 and k < 2:
                    want[k] = p
                elif k >= 2 and p != want[k % 2]:      # (the collectives order rank 0's first two proofs before anybody's third)
                    wrong[q] += 1
        th = [threading.Thread(target=body, args=(q,)) for q in range(N)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        h1, r1 = free_hbm_mb(), rss_mb()
        h0, r0 = marks["half"]
        if any(wrong):
            raise SystemExit("%s: proofs differing from the first of their r_a, per rank: %s" % (name, wrong))
        tr = [(t[1], t[2]) for t in marks.get("trace", [])] + [(h1, r1)]
        hg, rg = steady_growth(tr)
        rec = {"leg": name, "proofs": count, "second_half_hbm_growth_mb": round(h0 - h1, 1), "second_half_rss_growth_mb": round(r1 - r0, 1),
               "steady_hbm_growth_mb": round(hg, 1), "steady_rss_growth_mb": round(rg, 1), "trace_proofs_freehbm_rss": marks.get("trace", [])}
        out["legs"].append(rec)
        if hg > a.tolerance_mb or rg > a.tolerance_mb:
            bad = True
    sharded_leg("one proof over %d persistent rank threads" % N, a.sharded_proofs)
    out["ok"] = not bad
    print(json.dumps(out))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
