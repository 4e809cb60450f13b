#!/usr/bin/env python3
"""Throughput with several proofs in flight on ONE GPU (not the bench.py metric, which proves one at a time).

A pm_pk is immutable and shareable; a pm_ctx owns a stream, workspaces and the state of one proof
(include/polymath_hip.h, threading note).  K host threads, each with its own context, prove against the same
resident key: the latency-bound stretches of one proof (bucket reductions, sort tails, host finishes) fill with
another proof's accumulation.

  python tools/throughput_inflight.py --inflight 2 --proofs 12
"""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--inflight", type=int, default=2)
    ap.add_argument("--proofs", type=int, default=12)
    ap.add_argument("--log-constraints", type=int, default=20)
    args = ap.parse_args()
    from polymath_amd import api, circuits as PC
    from polymath_amd.polymath import FIELDS, Polymath
    curve = "bls12_381"
    r = FIELDS[curve]["r"]
    nr = (1 << args.log_constraints) - 100
    r1cs, inst, wit = PC.synthetic_r1cs(r, nr)
    pm0 = Polymath(curve, "merlin", device=0)
    g = PC.SplitMix64(0xBE7C4)
    x_trap, z_trap, r_a = g.fr(r), g.fr(r), [g.fr(r), g.fr(r)]
    pk = pm0.setup((r1cs, inst, wit), x_trap, z_trap)
    x_l, w_l = pm0.field.fr_limbs(inst), pm0.field.fr_limbs(wit)
    workers = [pm0] + [Polymath(curve, "merlin", device=0) for _ in range(args.inflight - 1)]
    views = [pk] + [pk.view(w.ctx) for w in workers[1:]]       # same pm_pk handle, another context
    ref = pm0.prove_limbs(pk, inst, x_l, w_l, r_a).to_bytes()
    for w, v in zip(workers, views):
        assert w.prove_limbs(v, inst, x_l, w_l, r_a).to_bytes() == ref      # warm-up + same proof from every context
    per = args.proofs // args.inflight
    outs = [None] * args.inflight

    def run(i):
        for _ in range(per):
            outs[i] = workers[i].prove_limbs(views[i], inst, x_l, w_l, r_a).to_bytes()

    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(i,)) for i in range(args.inflight)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    assert all(o == ref for o in outs)
    total = per * args.inflight
    print("inflight=%d proofs=%d  %.2f ms per proof  %.2f M constraints/s" % (args.inflight, total, dt / total * 1e3, nr * total / dt / 1e6))


if __name__ == "__main__":
    main()
