#!/usr/bin/env python3
"""Headline benchmark: Polymath prove on the synthetic 2^20-constraint R1CS (BASELINE.json
configs[1]: random A*B=C gates, BLS12-381), one process per GPU.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one complete create_proof_with_assignment (prover.rs:66-237): scalar H2D, witness map,
NTTs, the three merged MSMs, both Fiat-Shamir hashes; the proving key (bases + matrices) is
resident in HBM before the timed region.  With N > 1 ONE proof is spread over the N GPUs (MSM pair
ranges sharded, partial points all-gathered over RCCL), so scaling is "strong".

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel: the MSM bucket accumulation,
HIP-event timed on the library's stream) and `cpu_baseline` (the CPU restatement oracle/cpp timed on
this box's host cores over a bounded sample; "CPU restatement -- not arkworks").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MSM_BYTES_PER_PAIR = {"bls12_381": 128, "bn254": 96}   # 32 B scalar + affine base
# SURVEY.md §8d also asks for achieved MAD/s against a measured v_mad_u64_u32 peak: tools/microbench_valu.hip on
# MI355X, >= 2 waves/SIMD (profiles/r01_microbench_valu.txt): 442.75 G wave-instr/s = 28.3 T lane-mads/s.
VALU_MAD_PEAK = 28.34e12
# v_mad_u64_u32 per XYZZ mixed add on 28-bit limbs (8M + 2S, 9 Montgomery reductions): N = 14 / 10 limbs
MADS_PER_MIXED_ADD = {"bls12_381": 3542, "bn254": 1810}


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def cpu_baseline(curve, log_nr, seconds_hint=20):
    """ORACLE leg (the only place bench.py touches oracle/): time the CPU restatement's whole prove
    on a smaller circuit of the same family and report constraints/s."""
    from oracle import cpp_oracle as CO, driver as DR
    from oracle.pyref import circuits as OC, protocol as PR, transcripts as OT
    from oracle.pyref.fields import CURVES
    c = CURVES[curve]
    cores = os.cpu_count() or 1
    nr = (1 << log_nr) - 100
    q, inst, wit = OC.synthetic_r1cs(c, nr)
    t0 = time.time()
    opk = CO.OraclePk(curve, q, 0x1234567, 0x7654321, cores)
    t_setup = time.time() - t0
    omega = CO.fr_from_mont_limbs(curve, opk.omega_limbs)[0]
    t0 = time.time()
    DR.prove(opk, opk.n, opk.sigma, omega, inst, wit, [3, 5], OT.make_transcripts(c)["merlin"])
    dt = time.time() - t0
    pairs = 14 * opk.n + 30
    return {"value": nr / dt, "unit": "constraints/s", "cores": cores, "kind": "port",
            "sample": "oracle/cpp CPU restatement (not arkworks): whole prove of the 2^%d-100-gate synthetic R1CS "
                      "(n=%d, %d MSM pairs) in %.2f s on %d threads; setup %.1f s untimed" % (log_nr, opk.n, pairs, dt, cores, t_setup),
            "msm_pairs_per_sec": pairs / dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-constraints", type=int, default=20)
    ap.add_argument("--curve", default="bls12_381")
    ap.add_argument("--transcript", default="merlin")
    ap.add_argument("--cpu-baseline-log", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
    import torch
    import torch.distributed as dist
    # test hooks: several ranks sharing one GPU over gloo (tests/test_gpu_parity.py::test_bench_two_ranks_one_gpu)
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if os.environ.get("BENCH_FORCE_DEVICE") is not None:
        local = int(os.environ["BENCH_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    from polymath_amd import circuits as PC
    from polymath_amd.distributed import PointCombiner
    from polymath_amd.polymath import FIELDS, Polymath

    curve = args.curve
    r = FIELDS[curve]["r"]
    nr = (1 << args.log_constraints) - 100          # benches/bench.rs:16 convention: n = 2^(k+1)
    t0 = time.time()
    r1cs, inst, wit = PC.synthetic_r1cs(r, nr)
    log(rank, "synthetic R1CS: nr=%d m0=%d mw=%d (%.1f s)" % (nr, r1cs.m0, r1cs.mw, time.time() - t0))
    pm = Polymath(curve, args.transcript, device=local)
    g = PC.SplitMix64(0xBE7C4)
    x_trap, z_trap, r_a = g.fr(r), g.fr(r), [g.fr(r), g.fr(r)]
    t0 = time.time()
    # dev hook: BENCH_FAKE_SHARD="r/N" times ONE rank's share of an N-GPU proof on a single GPU (no exchange;
    # the proof bytes are then not a valid proof) -- used to size the fixed per-rank costs without an 8-GPU node
    shard_rank, shard_count = rank, world
    if world == 1 and os.environ.get("BENCH_FAKE_SHARD"):
        shard_rank, shard_count = (int(v) for v in os.environ["BENCH_FAKE_SHARD"].split("/"))
    pk = pm.setup((r1cs, inst, wit), x_trap, z_trap, shard_rank=shard_rank, shard_count=shard_count)
    log(rank, "setup on device: n=%d, %d resident points (%.1f s)" % (pk.n, sum(pk.base_lens), time.time() - t0))
    x_l, w_l = pm.field.fr_limbs(inst), pm.field.fr_limbs(wit)
    combine = PointCombiner(pm.ctx, curve, pm.field.nq, rank, world, device=local, backend_gloo=(backend != "nccl")) if world > 1 else None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the assignment is resident in HBM before the timed region (contract: `value` excludes PCIe)
    d_x = torch.from_numpy(x_l.view("int64")).cuda()
    d_w = torch.from_numpy(w_l.view("int64")).cuda()
    dev_ptrs = (d_x.data_ptr(), d_w.data_ptr())
    # The whole create_proof_with_assignment is ONE native call (pm_host_prove[_sharded]: the library's C++ host mirror
    # runs the transcript and challenge arithmetic between the phases); with several ranks it calls back into
    # PointCombiner.many between the phases to exchange the partial points (RCCL all-gather + pm_g1_sum).
    native = (world == shard_count) and not os.environ.get("BENCH_PYTHON_GLUE")

    def prove_once():
        if native:
            return pm.prove_native(pk, x_l, w_l, r_a, dev_ptrs, combine)
        return pm.prove_limbs(pk, inst, x_l, w_l, r_a, combine, dev_ptrs).to_bytes()

    proof_b = None
    for _ in range(args.warmup):
        proof_b = prove_once()
    acc_all_ms, sort_all_ms, red_all_ms, ntt_all_ms, poly_all_ms = [], [], [], [], []
    acc_ms, msm_ms, sort_ms, red_ms, phase_ms, acc1_ms = [], [], [], [], [], []
    pm.collect_timings = not native                 # phase-by-phase path: keep the stage slots of every phase
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        proof_b = prove_once()
        tm = pm.ctx.timings()
        if native:                                  # slots accumulated over the three phases of the proof
            acc_all_ms.append(tm["msm_accumulate"]); sort_all_ms.append(tm["msm_sort"]); red_all_ms.append(tm["msm_reduce"])
            ntt_all_ms.append(tm["ntt"]); poly_all_ms.append(tm["poly"])
        else:                                       # phase-3 slots here, phase-1 slots from collect_timings
            acc_ms.append(tm["msm_accumulate"]); msm_ms.append(tm["msm_total"]); sort_ms.append(tm["msm_sort"])
            red_ms.append(tm["msm_reduce"]); phase_ms.append(tm["phase"])
            acc1_ms.append(pm.phase_timings[0]["msm_accumulate"])
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3
    pm.collect_timings = False
    # PCIe-inclusive variant (host x, w buffers through pm_prove_phase1): reported, never `value`
    barrier()
    t1 = time.perf_counter()
    proof_host_b = pm.prove_native(pk, x_l, w_l, r_a, None, combine) if native else pm.prove_limbs(pk, inst, x_l, w_l, r_a, combine).to_bytes()
    barrier()
    ms_host_inputs = (time.perf_counter() - t1) * 1e3
    assert proof_host_b == proof_b
    # one phase-by-phase proof outside the timed region: per-phase stage breakdown, and the two host paths agree
    pm.collect_timings = True
    t1 = time.perf_counter()
    proof_py = pm.prove_limbs(pk, inst, x_l, w_l, r_a, combine, dev_ptrs).to_bytes()
    ms_python_glue = (time.perf_counter() - t1) * 1e3
    pm.collect_timings = False
    assert proof_py == proof_b
    tm3, tm1 = pm.phase_timings[2], pm.phase_timings[0]
    if native:
        acc_ms, msm_ms, sort_ms, red_ms, phase_ms = [tm3["msm_accumulate"]], [tm3["msm_total"]], [tm3["msm_sort"]], [tm3["msm_reduce"]], [tm3["phase"]]
        acc1_ms = [tm1["msm_accumulate"]]
    if os.environ.get("BENCH_PHASES"):             # dev hook: stage timings of all three phases (stderr)
        log(rank, "phase-by-phase proof %.2f ms; phases:" % ms_python_glue)
        for i, tm in enumerate(pm.phase_timings):
            log(rank, " phase %d:" % (i + 1), {k: round(v, 2) for k, v in tm.items() if v})
    n = pk.n
    d_pairs_total = 10 * n + 22                     # quotient MSM M8 (prover.rs:229)
    d_pairs_rank = d_pairs_total * (shard_rank + 1) // shard_count - d_pairs_total * shard_rank // shard_count
    pairs_per_proof = (n + 3) + (2 * r1cs.m0 + r1cs.mw + nr + (n - 1) + (n + 1) + 5) + d_pairs_total
    avg = lambda v: sum(v) / max(len(v), 1)
    if rank == 0:
        bpp = MSM_BYTES_PER_PAIR[curve]
        # Dominant kernel: k_accumulate, launched once per merged MSM = 3 launches per proof ([a], [c], [d]).  Roofline
        # over ALL of them: algorithmic bytes per launch = 128 B x (this rank's pairs of the three MSMs) / 3, average
        # launch duration = (HIP-event time of the three launches) / 3 -- the same average rocprofv3 --stats prints.
        plans = [pk.msm_plan(k) for k in range(3)]              # (resident pairs, windows, bits, tables)
        launches = 3
        acc_s = (avg(acc_all_ms) if native else avg(acc_ms) + avg(acc1_ms)) * 1e-3   # per proof, all three launches
        pairs_rank = sum(p[0] for p in plans)
        achieved = (bpp * pairs_rank / acc_s / 1e9) if acc_s > 0 else 0.0
        mads_rank = float(MADS_PER_MIXED_ADD[curve]) * sum(p[0] * p[1] for p in plans)
        msm_windows = plans[2][1]
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                key = "k_accumulate/%s/2^%d/%dgpu/avg_launch" % (curve, args.log_constraints, shard_count)
                traffic = tj.get(key)
            except Exception:
                traffic = None
        out = {
            "metric": "prove_constraints_per_sec", "value": nr / (dt / args.steps), "unit": "constraints/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "2^%d-100-constraint synthetic R1CS (random A*B=C gates), %s, n=2^%d, transcript=%s" %
                       (args.log_constraints, curve, n.bit_length() - 1, args.transcript),
                       "msm_pairs_per_proof": pairs_per_proof, "parallelism": "msm-pairs-sharded x%d" % world},
            "msm_pairs_per_sec": pairs_per_proof / (dt / args.steps),
            "ms_per_step_pcie_inclusive": ms_host_inputs,
            "host_glue": "native (pm_host_prove%s: C++ transcript + challenge arithmetic inside the library)" % ("_sharded" if world > 1 else "")
                         if native else "python (phases driven from bench.py)",
            "ms_per_step_python_glue": ms_python_glue,
            "arithmetic": "integer, 28/32-bit limbs in u32 registers (255-bit Fr, 381-bit Fq Montgomery)",
            "msm_d_pairs_per_sec_kernel_time": d_pairs_rank / (avg(msm_ms) * 1e-3) if avg(msm_ms) > 0 else None,
            "stage_ms_phase3": {"msm_sort": avg(sort_ms), "msm_accumulate": avg(acc_ms), "msm_reduce": avg(red_ms),
                                "msm_total": avg(msm_ms), "phase3_total": avg(phase_ms)},
            "roofline": {"bound": "hbm", "kernel": "k_accumulate (MSM bucket accumulation; %d launches per proof: [a], [c], [d])" % launches,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "launches_per_step": launches, "avg_launch_ms": acc_s * 1e3 / launches,
                         "algorithmic_bytes_per_launch": bpp * pairs_rank / launches,
                         "largest_launch": {"pairs": d_pairs_rank, "ms": avg(acc_ms),
                                            "achieved": (bpp * d_pairs_rank / (avg(acc_ms) * 1e-3) / 1e9) if avg(acc_ms) > 0 else None},
                         "note": "algorithmic bytes = %d B/pair x %d pairs over %d launches; the kernel is integer-ALU-bound "
                                 "(%d mixed adds x %d v_mad_u64_u32 per pair: see `valu` and DESIGN.md §4.2)" %
                                 (bpp, pairs_rank, launches, msm_windows, MADS_PER_MIXED_ADD[curve])},
            "valu": {"kernel": "k_accumulate", "unit": "lane-mads/s (v_mad_u64_u32)", "mads_per_mixed_add": MADS_PER_MIXED_ADD[curve],
                     "mixed_adds_per_pair": [p[1] for p in plans], "achieved": mads_rank / acc_s if acc_s > 0 else None, "peak": VALU_MAD_PEAK,
                     "frac": (mads_rank / acc_s / VALU_MAD_PEAK) if acc_s > 0 else None,
                     "note": "the real bound of this kernel: mads are 76 % of its instruction stream (3 542 of 4 635 per mixed add)"},
            "proof_bytes": proof_b.hex(),
        }
        if not args.no_cpu_baseline:
            log(rank, "timing the CPU restatement (bounded sample) ...")
            out["cpu_baseline"] = cpu_baseline(curve, args.cpu_baseline_log)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
