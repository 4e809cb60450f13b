#!/usr/bin/env python3
"""Headline benchmark: Polymath prove on the synthetic 2^20-constraint R1CS (BASELINE.json
configs[1]: random A*B=C gates, BLS12-381), one process per GPU.

  python bench.py --gpus N --steps K --warmup W          (N > 1: this process spawns the N ranks itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one complete create_proof_with_assignment (prover.rs:66-237) FROM HOST INPUTS, as benches/bench.rs:79 times
it and SURVEY.md §8d defines the metric: scalar H2D (x, w from pinned host memory), witness map, NTTs, the three merged
MSMs, both Fiat-Shamir hashes; the proving key (bases + matrices + window tables) is resident in HBM before the timed
region.  `ms_per_step_hbm_resident` / `value_hbm_resident` in the same line are the variant with the assignment already in
HBM (pm_prove_phase1_device).  With N > 1 ONE proof is spread over the N GPUs (layout "vector": witness map, four-step
NTT with one all-to-all per transform, scans and MSM pairs sharded; the ranks joined by a pm_comm = RCCL inside the
library), so scaling is "strong".

Multi-GPU launches are supervised (polymath_amd/launch.py): ranks are child processes of a parent that never touches the
GPU, a failing or stalled rank ends the attempt (fail-fast collectives in the library, deadlines here), and the exchange
layer falls back RCCL -> host-staged torch.distributed until one configuration completes; the JSON line says which ran
(`config.exchange`, `n_ranks_seen`).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel: the MSM bucket accumulation,
HIP-event timed on the library's stream), `cpu_baseline` (the CPU restatement oracle/cpp proving the SAME
2^20-100-gate circuit on this box's host cores, once; "CPU restatement -- not arkworks") and
`msm_micro` (the metric's second half: standalone resident G1 MSM pairs/s at 2^20 ... 2^26 pairs), `ntt_micro` (standalone
resident transforms at 2^21, 2^22, 2^24) and `stages` (per stage ms, algorithmic GB/s / 8 TB/s, lane-mads/s / peak).
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


def cpu_baseline(curve, log_nr, lc, gpk, r_a, gpu_proof, transcript):
    """ORACLE leg (the only place bench.py touches oracle/): time the CPU restatement's whole prove of THE SAME
    circuit, key and r_a the GPU just proved (the headline workload when log_nr == --log-constraints) on this box's host
    cores.  The oracle's key is seeded with the bases exported from HBM (its own CPU setup would take many minutes and
    is not the metric); the two proofs must be byte-identical."""
    from oracle import cpp_oracle as CO, driver as DR
    from oracle.pyref import serialize as SE, transcripts as OT
    from oracle.pyref.fields import CURVES
    c = CURVES[curve]
    cores = os.cpu_count() or 1

    class Shape:
        pass
    q = Shape()
    q.m0, q.mw, q.nr = lc.m0, lc.mw, lc.nr
    q.csr_arrays = [(a.rowptr, a.col, a.val) for a in lc.csrs]
    t0 = time.time()
    opk = CO.OraclePk(curve, q, None, None, cores)
    for i in range(6):
        opk.import_bases(i, gpk.export_bases(i))
    t_setup = time.time() - t0
    # The metric's second half on the CPU FIRST: the restatement's standalone Pippenger on a bounded sample of the
    # quotient MSM's own bases at cores, cores / 2, / 4, / 8 threads (os.cpu_count() counts SMT siblings and ignores the container's
    # CPU quota: where does it peak?) and on ONE thread (the per-thread rate a reader can hold against a tuned library: arkworks'
    # msm_unchecked does ~0.2-0.3 M pairs/s per core on BLS12-381).  The whole prove below then runs at the peak's thread count.
    import numpy as np
    msm, threads = {}, cores
    quota_cores = None
    try:
        quota = open("/sys/fs/cgroup/cpu.max").read().split()
        msm["cgroup_cpu_max"] = " ".join(quota)
        if quota[0] != "max":
            quota_cores = max(1, -(-int(quota[0]) // int(quota[1])))          # the CPUs this container may really use
            msm["cpu_quota_cores"] = quota_cores
    except Exception:         # noqa: BLE001
        pass
    try:
        ln = min(1 << 20, opk.base_lens[3])
        bases = opk.export_bases(3, 0, ln)
        rng = np.random.default_rng(5)
        sc = rng.integers(0, 1 << 64, size=(ln, 4), dtype=np.uint64)
        sc[:, 3] &= np.uint64((1 << (c.r.bit_length() - 193)) - 1)            # below r: one bit under its top limb
        CO.msm(curve, bases[:4096], sc[:4096], 2)                             # warm-up: tables, page faults
        sweep = {cores, max(1, cores // 2), max(1, cores // 4), max(1, cores // 8)}
        if quota_cores:
            sweep |= {min(cores, quota_cores), min(cores, 2 * quota_cores)}
        sweep = sorted(sweep, reverse=True)
        rates = {}
        for nt in sweep:
            t0 = time.time()
            CO.msm(curve, bases, sc, nt)
            rates[nt] = ln / (time.time() - t0)
        l1 = min(ln, 1 << 15)
        t0 = time.time()
        CO.msm(curve, bases[:l1], sc[:l1], 1)
        one = l1 / (time.time() - t0)
        threads = max(rates, key=rates.get)
        msm.update({"msm_sample": "standalone oracle/cpp Pippenger, 2^%d pairs of the quotient MSM's bases, uniform scalars" % (ln.bit_length() - 1),
                    "msm_pairs_per_sec_by_threads": {str(k): v for k, v in rates.items()}, "msm_pairs_per_sec": rates[threads], "msm_threads_at_peak": threads,
                    "msm_pairs_per_sec_one_thread": one, "msm_one_thread_sample_pairs": l1})
    except Exception as e:    # noqa: BLE001 -- the baseline leg must not cost the run its line
        msm["msm_error"] = repr(e)
    opk.nthreads = threads
    omega = CO.fr_from_mont_limbs(curve, opk.omega_limbs)[0]
    inst = lc.instance
    os.environ.setdefault("PO_PROFILE", "1")      # the restatement's stage clock on stderr: where its seconds go
    t0 = time.time()
    ref = DR.prove(opk, opk.n, opk.sigma, omega, inst, None, r_a, OT.make_transcripts(c)[transcript], w_limbs=lc.wit_limbs)
    dt = time.time() - t0
    same = gpu_proof is None or SE.ser_proof(c, ref) == gpu_proof
    pairs = 14 * opk.n + 30
    cores = threads           # what the timed prove really used
    out = {"value": lc.nr / dt, "unit": "constraints/s", "cores": cores, "kind": "port",
           "sample": "oracle/cpp CPU restatement (not arkworks; 64-bit-limb C++, %d threads): ONE whole prove of the 2^%d-100-gate "
                     "synthetic R1CS (n=%d, %d MSM pairs) in %.2f s, on the key exported from HBM (import %.1f s untimed); "
                     "proof bytes %s the GPU's" % (cores, log_nr, opk.n, pairs, dt, t_setup, "==" if same else "!="),
           "seconds": dt, "proof_identical_to_gpu": same, "msm_pairs_per_sec_whole_prove": pairs / dt}
    out.update(msm)
    out["host_threads_online"] = os.cpu_count()
    return out


LIVE_EXTRAS = {}      # what live_traffic measured besides the bytes: the accumulation's effective clock


def live_traffic(args, timeout_s=None):
    """HBM traffic of k_accumulate measured NOW, by this run: two child processes of this script under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, as MI355X_MICROARCH.md's HBM section
    prescribes), --opt msm_overlap=0 so that every launch runs alone; per-launch bytes = 2 x FETCH_SIZE (gfx950 tallies 128-B
    requests at 64 B; calibrated on this access pattern by tools/pmc_gather_calib.hip) + WRITE_SIZE, both in KB.
    -> (bytes per launch, description) or (None, reason): the caller then falls back to the committed profile."""
    import csv, shutil, signal, subprocess, tempfile
    if timeout_s is None:
        timeout_s = 200 * (1 << max(0, args.log_constraints - 20))
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself being profiled"
    totals = {}
    launches = None
    LIVE_EXTRAS.clear()
    for ctr in ("FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE"):
        d = tempfile.mkdtemp(prefix="pm_pmc_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "run", "--", sys.executable,
               os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--msm-micro", "", "--ntt-micro", "", "--no-live-traffic", "--inflight", "0", "--other-configs", "",
               "--log-constraints", str(args.log_constraints), "--curve", args.curve, "--transcript", args.transcript, "--opt", "msm_overlap=0"]
        env = dict(os.environ, TMPDIR="/tmp")
        try:
            # own session: on a timeout the whole tree (rocprofv3 AND the profiled python) is killed, not just the profiler
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = proc.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)
                proc.wait()
                if ctr == "GRBM_GUI_ACTIVE":          # the clock pass is an extra: its failure must not cost the traffic figure
                    continue
                return None, "%s pass exceeded %d s and was killed" % (ctr, timeout_s)
            if rc != 0:
                if ctr == "GRBM_GUI_ACTIVE":
                    continue
                return None, "%s pass exited with code %d" % (ctr, rc)
            path = None
            for root_, _dirs, files in os.walk(d):
                if "run_counter_collection.csv" in files:
                    path = os.path.join(root_, "run_counter_collection.csv")
            rows = [r for r in csv.DictReader(open(path)) if "k_accumulate" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
            vals = [float(r["Counter_Value"]) for r in rows]
            if ctr == "GRBM_GUI_ACTIVE":
                # the clock the kernel really ran at: busy cycles, summed over the 8 XCDs, over the dispatches'
                # own durations.  A third pass of its own; its failure costs the clock, not the traffic figure.
                ns = sum(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in rows)
                if vals and ns > 0:
                    LIVE_EXTRAS["effective_clock_GHz"] = sum(vals) / 8.0 / ns
                    LIVE_EXTRAS["effective_clock_source"] = ("rocprofv3 --pmc GRBM_GUI_ACTIVE child pass of this run: busy cycles / 8 XCDs / dispatch duration over %d "
                                                             "k_accumulate launches (msm_overlap = 0; nominal 2.4 GHz)" % len(vals))
                continue
            if not vals:
                return None, "no k_accumulate rows in the %s pass" % ctr
            totals[ctr] = sum(vals) / len(vals)
            launches = len(vals)
        except Exception as e:       # noqa: BLE001 -- a missing profiler must not cost the bench line
            if ctr == "GRBM_GUI_ACTIVE":
                continue
            return None, "%s pass failed: %s" % (ctr, type(e).__name__)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return (2.0 * totals["FETCH_SIZE"] + totals["WRITE_SIZE"]) * 1024.0, \
        "measured by this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes (msm_overlap = 0, %d launches each); bytes = 2 x FETCH_SIZE + WRITE_SIZE" % launches


def msm_micro(ctx, curve, logs, reps=3):
    """The second half of the metric ("G1 MSM pairs/s = L / time of pm_msm_g1, bases resident";
    benches/bench.rs:82-91 is the reference's only analogue): standalone resident MSM, bases P_i = (i+1) G built on the
    device with their window tables, uniform scalars already in HBM."""
    import torch
    from polymath_amd import api
    out = []
    for lg in logs:
        n = 1 << lg
        bases = api.Bases.multiples(ctx, curve, n).precompute()
        g = torch.Generator(device="cuda").manual_seed(1234 + lg)
        sc = torch.randint(0, 2**62, (n, 4), dtype=torch.int64, device="cuda", generator=g) * 4 + torch.randint(0, 4, (n, 4), dtype=torch.int64, device="cuda", generator=g)
        sc[:, 3] &= (1 << 61) - 1      # < 2^253: canonical residues of both scalar fields, taken as Montgomery forms of uniform elements
        torch.cuda.synchronize()
        best, tm = None, None
        for rep in range(reps + 1):
            t0 = time.perf_counter()
            bases.msm(None, 0, n, device_ptr=sc.data_ptr())
            dt = time.perf_counter() - t0
            if rep and (best is None or dt < best):
                best, tm = dt, ctx.timings()
        out.append({"len": n, "ms": best * 1e3, "pairs_per_sec": n / best,
                    "hbm_frac_algorithmic": MSM_BYTES_PER_PAIR[curve] * n / best / 1e9 / HBM_PEAK_GBS,
                    "stage_ms": {k: round(v, 3) for k, v in tm.items() if k.startswith("msm")}})
        bases.free()
        del sc
        torch.cuda.empty_cache()
    return out


NTT_MADS_PER_BUTTERFLY = 162      # one carry-free product on 9 limbs of 29 bits (81 + 81 multiplier operations)


def ntt_micro(ctx, curve, logs, reps=5):
    """SURVEY.md §8d per-stage report for the transforms (prover.rs:239-243, 315-328): standalone resident NTT (pm_ntt_device,
    canonical inputs already in HBM), forward direction, at the domain sizes of the BASELINE configurations.  Algorithmic bytes =
    64 B / element / transform (one read + one write of 32 B); the kernels are bound by the 64-bit multiplier, so the achieved
    lane-mads/s against the measured v_mad_u64_u32 peak is reported beside the HBM fraction."""
    import torch
    from polymath_amd.polymath import FIELDS
    top = FIELDS[curve]["r"] >> 192
    out = []
    for lg in logs:
        n = 1 << lg
        g = torch.Generator(device="cuda").manual_seed(lg)
        x = torch.randint(0, 2**62, (n, 4), dtype=torch.int64, device="cuda", generator=g) * 4 + torch.randint(0, 4, (n, 4), dtype=torch.int64, device="cuda", generator=g)
        x[:, 3] = torch.randint(0, top, (n,), dtype=torch.int64, device="cuda", generator=g)
        torch.cuda.synchronize()
        best_wall, best_gpu = None, None
        for rep in range(reps + 1):
            t0 = time.perf_counter()
            ctx.ntt_device(curve, x.data_ptr(), lg, False)
            dt = time.perf_counter() - t0
            gpu_ms = ctx.timings()["ntt"]
            if rep and (best_wall is None or dt < best_wall):
                best_wall, best_gpu = dt, gpu_ms
        mads = NTT_MADS_PER_BUTTERFLY * (n // 2) * lg
        sec = best_gpu * 1e-3
        out.append({"log_n": lg, "ms": best_gpu, "ms_wall_incl_launch_and_sync": best_wall * 1e3,
                    "algorithmic_GBps": 64.0 * n / sec / 1e9, "hbm_frac_algorithmic": 64.0 * n / sec / 1e9 / HBM_PEAK_GBS,
                    "lane_mads_per_sec": mads / sec, "valu_frac": mads / sec / VALU_MAD_PEAK})
        del x
        torch.cuda.empty_cache()
    return out


def stage_report(curve, r1cs, pk, tm1, tm3, plans):
    """"Report, per stage: achieved GB/s / 8 TB/s and achieved MAD/s / measured peak".  Stage times are the
    HIP-event slots of ONE proof with msm_overlap = 0 (every stage alone on the chip), algorithmic bytes as §8d defines them;
    where §8d gives no figure the definition is in the entry's `bytes_are`."""
    n, m0, mw, nr = pk.n, r1cs.m0, r1cs.mw, r1cs.nr
    nnz = sum(int(a.rowptr[-1]) for a in r1cs.csrs)
    pairs = sum(p[0] for p in plans)
    entries = sum(p[0] * p[1] for p in plans)
    buckets = sum((1 << (p[2] - 1)) * (1 if p[3] else p[1]) for p in plans)       # one shared set with tables, one per window without
    log_n = n.bit_length() - 1

    def entry(ms, nbytes, bytes_are, **extra):
        e = {"ms": ms, "algorithmic_bytes": nbytes, "bytes_are": bytes_are}
        if ms and ms > 0:
            e["algorithmic_GBps"] = nbytes / (ms * 1e-3) / 1e9
            e["hbm_frac_algorithmic"] = e["algorithmic_GBps"] / HBM_PEAK_GBS
        e.update(extra)
        return e
    ntt_mads = 4 * NTT_MADS_PER_BUTTERFLY * (n // 2) * log_n
    acc_ms = tm1["msm_accumulate"] + tm3["msm_accumulate"]
    acc_mads = float(MADS_PER_MIXED_ADD[curve]) * entries
    return {
        "source": "HIP-event stage slots of one proof with msm_overlap = 0, outside the timed region",
        "witness_map": entry(tm1["witness_map"], 36 * nnz + 32 * (m0 + mw) + 64 * n, "36 nnz(A,B,C) + 32 (m0 + mw) + 64 n (SURVEY.md §8d)"),
        "ntt": entry(tm1["ntt"], 4 * 64 * n, "64 B / element / transform x the 4 size-n transforms this prover runs (u, w, and the forward / inverse pair of "
                     "the negacyclic square; the reference's 3 size-n + 2 size-2n of prover.rs:94-96,319-325 would be 7 n elements)",
                     lane_mads_per_sec=ntt_mads / (tm1["ntt"] * 1e-3) if tm1["ntt"] else None,
                     valu_frac=ntt_mads / (tm1["ntt"] * 1e-3) / VALU_MAD_PEAK if tm1["ntt"] else None),
        "pointwise_phase1": entry(tm1["poly"], 64 * 5 * n, "64 B / element over twist, square, untwist + h, and the two scalar vectors (5 n elements)"),
        "division_scan": entry(tm3["poly"], 64 * (10 * n + 23), "64 B / element of the numerator / quotient index space (10 n + 23; SURVEY.md §8d)"),
        "msm_sort": entry(tm1["msm_sort"] + tm3["msm_sort"], 32 * pairs + 4 * entries,
                          "32 B scalar read per pair + 4 B sorted table index written per (pair, window) entry: the least a bucket sort moves"),
        "msm_accumulate": entry(acc_ms, MSM_BYTES_PER_PAIR[curve] * pairs, "%d B / pair (SURVEY.md §8d)" % MSM_BYTES_PER_PAIR[curve],
                                lane_mads_per_sec=acc_mads / (acc_ms * 1e-3) if acc_ms else None,
                                valu_frac=acc_mads / (acc_ms * 1e-3) / VALU_MAD_PEAK if acc_ms else None),
        "msm_reduce": entry(tm1["msm_reduce"] + tm3["msm_reduce"], 4 * 48 * buckets if curve == "bls12_381" else 4 * 32 * buckets,
                            "one XYZZ partial per bucket read (%d buckets): chains of dependent point additions, latency-bound" % buckets),
    }


def other_config(ctx, curve, log_nr, transcript, proofs=3):
    """One more BASELINE.json configuration on this GPU, after the headline: `proofs` whole proofs through
    pm_host_prove from pinned host buffers (the headline's entry point and clock), one msm_overlap = 0 proof for the stage
    block, Polymath::verify on the bytes.  Reported under `other_configs`; never part of `value`."""
    import numpy as np
    import torch
    from polymath_amd import circuits as PC
    from polymath_amd.polymath import FIELDS, Polymath
    r = FIELDS[curve]["r"]
    nr = (1 << log_nr) - 100
    t0 = time.time()
    lc = PC.synthetic_r1cs_native(curve, nr)
    t_synth = time.time() - t0
    pm = Polymath(curve, transcript, ctx=ctx)
    g = PC.SplitMix64(0xBE7C4 + log_nr)
    x_trap, z_trap, r_a = g.fr(r), g.fr(r), [g.fr(r), g.fr(r)]
    t0 = time.time()
    pk = pm.setup(lc, x_trap, z_trap)
    torch.cuda.synchronize()
    t_setup = time.time() - t0
    x_pin = torch.from_numpy(np.ascontiguousarray(lc.inst_limbs).view("int64")).pin_memory()
    w_pin = torch.from_numpy(np.ascontiguousarray(lc.wit_limbs).view("int64")).pin_memory()
    x_l, w_l = x_pin.numpy().view("uint64"), w_pin.numpy().view("uint64")
    proof = pm.prove_native(pk, x_l, w_l, r_a)                       # warm-up: workspaces, twiddles
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(proofs):
        again = pm.prove_native(pk, x_l, w_l, r_a)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / proofs * 1e3
    overlap_was = ctx.get_option("msm_overlap")
    ctx.set_option("msm_overlap", 0)
    pm.collect_timings = True
    serial = pm.prove_limbs(pk, lc.instance, x_l, w_l, r_a).to_bytes()
    pm.collect_timings = False
    ctx.set_option("msm_overlap", overlap_was)
    plans = [pk.msm_plan(k) for k in range(3)]
    n = pk.n
    pairs = (n + 3) + (2 * lc.m0 + lc.mw + nr + (n - 1) + (n + 1) + 5) + 10 * n + 22
    out = {"workload": "2^%d-100-constraint synthetic R1CS (random A*B=C gates), %s, n=2^%d, transcript=%s, one GPU" % (log_nr, curve, n.bit_length() - 1, transcript),
           "proofs": proofs, "ms_per_proof": ms, "constraints_per_sec": nr / (ms * 1e-3), "msm_pairs_per_sec": pairs / (ms * 1e-3),
           "timed_entry_point": "pm_host_prove from pinned HOST buffers (%.1f MB H2D inside every proof)" % ((x_l.nbytes + w_l.nbytes) / 1e6),
           "proofs_identical": again == proof and serial == proof,
           "proof_verified": bool(pm.verify(pm.make_vk(pk, x_trap, z_trap), lc.instance[1:], proof)),
           "msm_plan": [{"msm": "acd"[k], "pairs": p[0], "windows": p[1], "window_bits": p[2], "tables": p[3]} for k, p in enumerate(plans)],
           "stages": stage_report(curve, lc, pk, dict(pm.phase_timings[0]), dict(pm.phase_timings[2]), plans),
           "setup_s": t_setup, "synthesis_s": t_synth, "proof_bytes": proof.hex()}
    pk.free()
    del x_pin, w_pin
    torch.cuda.empty_cache()
    return out


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-constraints", type=int, default=20)
    ap.add_argument("--curve", default="bls12_381")
    ap.add_argument("--transcript", default="merlin")
    ap.add_argument("--cpu-baseline-log", type=int, default=0, help="0 = the headline workload itself when this box has >= 32 host threads, else 2^16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--msm-micro", default="20,22,24,26", help="log2 lengths of the standalone resident MSM legs ('' = none)")
    ap.add_argument("--ntt-micro", default="21,22,24", help="log2 sizes of the standalone resident NTT legs ('' = none)")
    ap.add_argument("--other-configs", default="bn254:20,bls12_381:22,bls12_381:24",
                    help="curve:log2(constraints) legs run after the headline on the same GPU (BASELINE.json configs[4], configs[2], configs[3] on one GPU; '' = none)")
    ap.add_argument("--inflight", type=int, default=2, help="proofs in flight of the serving-throughput leg beside `value` (0 = skip; single GPU only)")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not spawn the two rocprofv3 --pmc child passes that measure `roofline.traffic`")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="pm_ctx_set_option on the proving context before the key is generated, e.g. --opt msm_overlap=0 --opt tables=wide")
    return ap.parse_args()


def main():
    """Dispatch.  N = 1: the bench runs in this process.  N > 1: this process only SUPERVISES (it never touches the GPU):
    without a launcher it spawns the N ranks, under torch.distributed.run it supervises its own rank's child
    (polymath_amd/launch.py); the children come back here with BENCH_CHILD=1 and run worker()."""
    args = parse_args()
    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not os.environ.get("BENCH_CHILD"):
        world_env = int(os.environ.get("WORLD_SIZE", "1")) if under_launcher else 1
        # what ONE healthy attempt may need at most: import + synthesis + key setup + K proofs, generous, scaled with the circuit.
        # The job as a whole lives on one clock (launch.Budget: BENCH_TOTAL_BUDGET_S, default 1 500 s, inside the driver's 1 800 s).
        scale = 1 << max(0, args.log_constraints - 20)
        deadline = int(os.environ.get("BENCH_ATTEMPT_DEADLINE_S", str(600 * scale + 3 * (args.steps + args.warmup) * scale + 120)))
        if args.log_constraints > 20 and "BENCH_TOTAL_BUDGET_S" not in os.environ:
            os.environ["BENCH_TOTAL_BUDGET_S"] = str(1500 * scale)       # larger circuits are run by hand, not by the driver
        if under_launcher and (world_env > 1 or os.environ.get("BENCH_FORCE_VECTOR")):
            if world_env != args.gpus and world_env > 1:
                raise SystemExit("--gpus %d but the launcher started %d ranks" % (args.gpus, world_env))
            from polymath_amd import launch
            raise SystemExit(launch.supervise_one(os.path.abspath(__file__), sys.argv[1:], deadline))
        if not under_launcher and args.gpus > 1:
            from polymath_amd import launch
            raise SystemExit(launch.supervise_all(os.path.abspath(__file__), sys.argv[1:], args.gpus, deadline))
        return worker(args)
    # a rank of a supervised job: any failure is a non-zero exit NOW (interpreter teardown with a dead communicator may hang)
    try:
        worker(args)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
    except BaseException:       # noqa: BLE001
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)


def worker(args):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not (world == 1 and os.environ.get("BENCH_FORCE_VECTOR")):
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    supervised = bool(os.environ.get("BENCH_CHILD"))
    from polymath_amd.launch import Watchdog
    wd = Watchdog(rank) if supervised else None
    comm_timeout_s = int(os.environ.get("BENCH_COMM_TIMEOUT_S", "60"))

    def stage(name, limit_s):
        if wd:
            wd.stage_begin(name, limit_s)
    attempt_limit = float(os.environ.get("BENCH_ATTEMPT_LIMIT_S", "1e9"))   # the supervisor kills the attempt then anyway
    stage("import torch + rendezvous", min(600.0, attempt_limit))
    import datetime
    import numpy as np
    import torch
    import torch.distributed as dist
    # test hooks: several ranks sharing one GPU over gloo (tests/test_gpu_parity.py::test_bench_two_ranks_one_gpu)
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if os.environ.get("BENCH_FORCE_DEVICE") is not None:
        local = int(os.environ["BENCH_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    # test hook: BENCH_FORCE_VECTOR=1 under `torch.distributed.run --nproc-per-node 1` takes the whole N > 1 code path with
    # a world of ONE (process group, RCCL communicator inside the library next to torch's own, vector-sharded key): what
    # a single-GPU box can check of the multi-GPU launch (tests/test_sharded_vector.py)
    force_vec = bool(os.environ.get("BENCH_FORCE_VECTOR")) and "MASTER_ADDR" in os.environ
    multi = world > 1 or force_vec
    if multi:
        pg_timeout = datetime.timedelta(seconds=max(2 * comm_timeout_s, 120))
        kw = {"device_id": torch.device("cuda", local)} if backend == "nccl" else {}
        attempt = int(os.environ.get("BENCH_ATTEMPT", "0"))
        if supervised and attempt > 0:
            # A retry meets on a store of its own (launch.py gave it a fresh port).  Under an external launcher the ranks'
            # supervisors notice the previous attempt's failure at different times -- the rank that failed at once, its peers when
            # their collective deadline passed -- so the rendezvous waits much longer than any collective does.
            store = dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]), world, rank == 0,
                                  timeout=datetime.timedelta(seconds=max(30.0, min(300.0, attempt_limit / 3))), wait_for_workers=True)
            dist.init_process_group(backend, store=store, rank=rank, world_size=world, timeout=pg_timeout, **kw)
        else:
            dist.init_process_group(backend, timeout=pg_timeout, **kw)
        dist.barrier()          # every rank is up before anybody starts the expensive part

    from polymath_amd import circuits as PC
    from polymath_amd.distributed import PointCombiner
    from polymath_amd.polymath import FIELDS, Polymath

    curve = args.curve
    r = FIELDS[curve]["r"]
    nr = (1 << args.log_constraints) - 100          # benches/bench.rs:16 convention: n = 2^(k+1)
    t0 = time.time()
    r1cs = PC.synthetic_r1cs_native(curve, nr)      # pm_synth_r1cs: the same draws as circuits.synthetic_r1cs, no Python big ints
    inst = r1cs.instance
    log(rank, "synthetic R1CS: nr=%d m0=%d mw=%d (%.1f s)" % (nr, r1cs.m0, r1cs.mw, time.time() - t0))
    pm = Polymath(curve, args.transcript, device=local)
    for kv in args.opt:                              # modes are per-context options of the library, not environment variables
        name, value = kv.split("=", 1)
        pm.ctx.set_option(name, int(value) if value.lstrip("-").isdigit() else value)
    g = PC.SplitMix64(0xBE7C4)
    x_trap, z_trap, r_a = g.fr(r), g.fr(r), [g.fr(r), g.fr(r)]
    t0 = time.time()
    # dev hook: BENCH_FAKE_SHARD="r/N" times ONE rank's share of an N-GPU proof on a single GPU (no exchange;
    # the proof bytes are then not a valid proof) -- used to size the fixed per-rank costs without an 8-GPU node
    shard_rank, shard_count = rank, world
    if world == 1 and os.environ.get("BENCH_FAKE_SHARD"):
        shard_rank, shard_count = (int(v) for v in os.environ["BENCH_FAKE_SHARD"].split("/"))
    # N > 1: ONE proof over the N GPUs.  layout "vector" (default): witness map, NTTs, scans AND MSM pairs sharded, the
    # ranks joined by a pm_comm (RCCL inside the library); "pairs": only the MSM pair ranges
    # sharded, the partial points combined through torch.distributed (row 1 only; BENCH_SHARD_LAYOUT=pairs).
    layout = os.environ.get("BENCH_SHARD_LAYOUT", "vector") if multi else "pairs"
    comm_desc, comm_meta = None, None
    if multi and layout == "vector":
        from polymath_amd.distributed import make_comm
        stage("communicator setup", 60 + 3 * comm_timeout_s)
        want_native = not os.environ.get("BENCH_NO_RCCL") and (supervised or backend == "nccl")
        comm, comm_desc, comm_meta = make_comm(rank, world, local, native=want_native, timeout_s=comm_timeout_s)
        pm.ctx.set_comm(comm)
        log(rank, "pm_comm:", comm_desc, comm_meta)
    stage("key setup", 300 * (1 << max(0, args.log_constraints - 20)))
    pk = pm.setup(r1cs, x_trap, z_trap, shard_rank=shard_rank, shard_count=shard_count, layout=layout)
    log(rank, "setup on device: n=%d, layout=%s (%.1f s)" % (pk.n, layout, time.time() - t0))
    if multi:
        dist.barrier()          # the ranks' setups differ by seconds: do not let that skew eat into the first proof's collective deadline
    x_l, w_l = r1cs.inst_limbs, r1cs.wit_limbs
    combine = PointCombiner(pm.ctx, curve, pm.field.nq, rank, world, device=local, backend_gloo=(backend != "nccl")) if world > 1 and layout == "pairs" else None
    py_combine = combine            # for the phase-by-phase (Python glue) proofs
    # (a PM_SHARD_VECTOR key needs no combiner at all: its phases return points already summed over the ranks)

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # The metric (benches/bench.rs:79): prove FROM HOST INPUTS -- the scalar H2D is inside the step.  The
    # host buffers are pinned (what a host that cares about the 33 MB copy hands over; a pageable buffer costs ~0.4 ms
    # more through the runtime's bounce buffers).  The HBM-resident variant (pm_prove_phase1_device) is timed beside it.
    x_pin = torch.from_numpy(np.ascontiguousarray(x_l).view("int64")).pin_memory()
    w_pin = torch.from_numpy(np.ascontiguousarray(w_l).view("int64")).pin_memory()
    x_l, w_l = x_pin.numpy().view("uint64"), w_pin.numpy().view("uint64")
    d_x, d_w = x_pin.cuda(), w_pin.cuda()
    dev_ptrs = (d_x.data_ptr(), d_w.data_ptr())
    # The whole create_proof_with_assignment is ONE native call (pm_host_prove[_sharded]: the library's C++ host mirror
    # runs the transcript and challenge arithmetic between the phases); with several ranks it calls back into
    # PointCombiner.many between the phases to exchange the partial points (RCCL all-gather + pm_g1_sum).
    native = (world == shard_count) and not os.environ.get("BENCH_PYTHON_GLUE")
    comm_info = comm_desc

    def prove_once(resident=False):
        ptrs = dev_ptrs if resident else None
        if native:
            return pm.prove_native(pk, x_l, w_l, r_a, ptrs, combine)
        return pm.prove_limbs(pk, inst, x_l, w_l, r_a, py_combine, ptrs).to_bytes()

    stage("warmup proofs", 120 + 20 * args.warmup * (1 << max(0, args.log_constraints - 20)))
    proof_b = None
    for _ in range(args.warmup):
        proof_b = prove_once()
    stage("timed proofs", 120 + 10 * (args.steps + 4) * (1 << max(0, args.log_constraints - 20)))
    if os.environ.get("BENCH_TEST_DIE"):            # test hook "rank,seconds": that rank dies (exit 17) mid-run, the job must fail fast
        die_rank, die_after = os.environ["BENCH_TEST_DIE"].split(",")
        if int(die_rank) == rank:
            import threading
            threading.Timer(float(die_after), lambda: os._exit(17)).start()
    acc_all_ms, sort_all_ms, red_all_ms, ntt_all_ms, poly_all_ms = [], [], [], [], []
    acc_ms, msm_ms, sort_ms, red_ms, phase_ms, acc1_ms = [], [], [], [], [], []
    pm.collect_timings = not native                 # phase-by-phase path: keep the stage slots of every phase
    barrier()
    t0 = time.perf_counter()
    for step_i in range(args.steps):
        proof_b = prove_once()
        if native:                                  # slots accumulated over the three phases of the proof
            # pm_host_prove leaves its stage timers unread (pm_last_timings reads ~50 event pairs, ~0.1 ms of host time): the LAST timed
            # proof is the sample -- a prover does not query its timers between proofs
            if step_i + 1 < args.steps:
                continue
            tm = pm.ctx.timings()
            acc_all_ms.append(tm["msm_accumulate"]); sort_all_ms.append(tm["msm_sort"]); red_all_ms.append(tm["msm_reduce"])
            ntt_all_ms.append(tm["ntt"]); poly_all_ms.append(tm["poly"])
            continue
        tm = pm.ctx.timings()
        if True:                                    # phase-3 slots here, phase-1 slots from collect_timings
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
    # the same proofs with the assignment already resident in HBM (pm_prove_phase1_device): reported beside `value`
    barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        proof_res_b = prove_once(resident=True)
    barrier()
    dt_res = time.perf_counter() - t1
    if world > 1:
        tt = torch.tensor([dt_res], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt_res = float(tt.item())
    ms_resident = dt_res / args.steps * 1e3
    assert proof_res_b == proof_b
    # one phase-by-phase proof outside the timed region: per-phase stage breakdown, and the two host paths agree
    pm.collect_timings = True
    t1 = time.perf_counter()
    proof_py = pm.prove_limbs(pk, inst, x_l, w_l, r_a, py_combine, dev_ptrs).to_bytes()
    ms_python_glue = (time.perf_counter() - t1) * 1e3
    pm.collect_timings = False
    assert proof_py == proof_b
    tm3, tm1 = pm.phase_timings[2], pm.phase_timings[0]
    if native:
        acc_ms, msm_ms, sort_ms, red_ms, phase_ms = [tm3["msm_accumulate"]], [tm3["msm_total"]], [tm3["msm_sort"]], [tm3["msm_reduce"]], [tm3["phase"]]
        acc1_ms = [tm1["msm_accumulate"]]
    # Roofline sample, outside the timed region: ONE proof with PM_OPT_MSM_OVERLAP = 0, so that each k_accumulate launch runs
    # alone on the chip (in the timed proofs the [a] launch shares it with the transforms and the [c] MSM, which
    # inflates its HIP-event duration); these are the durations rocprofv3 --stats of the same configuration prints.
    overlap_was = pm.ctx.get_option("msm_overlap")
    pm.ctx.set_option("msm_overlap", 0)
    pm.collect_timings = True
    proof_serial = pm.prove_limbs(pk, inst, x_l, w_l, r_a, py_combine, dev_ptrs).to_bytes()
    pm.collect_timings = False
    pm.ctx.set_option("msm_overlap", overlap_was)
    assert proof_serial == proof_b
    acc_serial_ms = pm.phase_timings[0]["msm_accumulate"] + pm.phase_timings[2]["msm_accumulate"]
    tm1_serial, tm3_serial = dict(pm.phase_timings[0]), dict(pm.phase_timings[2])
    if os.environ.get("BENCH_PHASES"):             # dev hook: stage timings of all three phases (stderr)
        log(rank, "phase-by-phase proof %.2f ms; phases:" % ms_python_glue)
        for i, tm in enumerate(pm.phase_timings):
            log(rank, " phase %d:" % (i + 1), {k: round(v, 2) for k, v in tm.items() if v})
    n = pk.n
    d_pairs_total = 10 * n + 22                     # quotient MSM M8 (prover.rs:229)
    d_pairs_rank = pk.msm_plan(2)[0]
    pairs_per_proof = (n + 3) + (2 * r1cs.m0 + r1cs.mw + nr + (n - 1) + (n + 1) + 5) + d_pairs_total
    avg = lambda v: sum(v) / max(len(v), 1)
    if rank == 0:
        bpp = MSM_BYTES_PER_PAIR[curve]
        # Dominant kernel: k_accumulate, launched once per merged MSM = 3 launches per proof ([a], [c], [d]).  Roofline
        # over ALL of them: algorithmic bytes per launch = 128 B x (this rank's pairs of the three MSMs) / 3, average
        # launch duration = (HIP-event time of the three launches) / 3 -- the same average rocprofv3 --stats prints.
        plans = [pk.msm_plan(k) for k in range(3)]              # (resident pairs, windows, bits, tables)
        launches = 3
        acc_s = acc_serial_ms * 1e-3                             # per proof, all three launches, each running alone
        acc_overlapped_s = (avg(acc_all_ms) if native else avg(acc_ms) + avg(acc1_ms)) * 1e-3   # as in the timed proofs
        pairs_rank = sum(p[0] for p in plans)
        achieved = (bpp * pairs_rank / acc_s / 1e9) if acc_s > 0 else 0.0
        mads_rank = float(MADS_PER_MIXED_ADD[curve]) * sum(p[0] * p[1] for p in plans)
        msm_windows = plans[2][1]
        traffic, traffic_source = None, None
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
                       "msm_pairs_per_proof": pairs_per_proof,
                       "parallelism": ("one proof over %d GPUs: witness map, four-step NTT (one all-to-all per transform), scans and MSM pairs sharded; %s"
                                       % (world, comm_desc)) if layout == "vector" and multi else ("msm-pairs-sharded x%d" % world if world > 1 else "one GPU")},
            "msm_pairs_per_sec": pairs_per_proof / (dt / args.steps),
            "timed_entry_point": "pm_host_prove%s from pinned HOST buffers (x, w: %.1f MB H2D inside every step; SURVEY.md §8d, benches/bench.rs:79)"
                                 % ("_sharded" if world > 1 else "", (x_l.nbytes + w_l.nbytes) / 1e6),
            "ms_per_step_hbm_resident": ms_resident, "value_hbm_resident": nr / (dt_res / args.steps),
            "n_ranks_seen": comm_meta["ranks_seen"] if comm_meta else world,
            "exchange": comm_meta if comm_meta else {"kind": "none (single GPU)" if not multi else "torch.distributed point all-gather (pairs layout)", "ranks_seen": world},
            "launch": {"supervised": supervised, "attempt": int(os.environ.get("BENCH_ATTEMPT", "0")), "torch_distributed_backend": backend if multi else None},
            "host_glue": "native (pm_host_prove%s: C++ transcript + challenge arithmetic inside the library)" % ("_sharded" if world > 1 else "")
                         if native else "python (phases driven from bench.py)",
            "ms_per_step_python_glue": ms_python_glue,
            "arithmetic": "integer, 28/32-bit limbs in u32 registers (255-bit Fr, 381-bit Fq Montgomery)",
            "msm_d_pairs_per_sec_kernel_time": d_pairs_rank / (avg(msm_ms) * 1e-3) if avg(msm_ms) > 0 else None,
            "stage_ms_phase3": {"msm_sort": avg(sort_ms), "msm_accumulate": avg(acc_ms), "msm_reduce": avg(red_ms),
                                "msm_total": avg(msm_ms), "phase3_total": avg(phase_ms)},
            "roofline": {"bound": "hbm", "kernel": "k_accumulate (MSM bucket accumulation; %d launches per proof: [a], [c], [d])" % launches,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": "profiles/pmc_traffic.json (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)",
                         "launches_per_step": launches, "avg_launch_ms": acc_s * 1e3 / launches,
                         "timing": "HIP events on the library's stream around each launch of one msm_overlap = 0 proof outside the timed region "
                                   "(launches run alone); in the timed proofs [a] overlaps other kernels: %.3f ms per proof there" % (acc_overlapped_s * 1e3),
                         "algorithmic_bytes_per_launch": bpp * pairs_rank / launches,
                         "largest_launch": {"pairs": d_pairs_rank, "ms": avg(acc_ms),
                                            "achieved": (bpp * d_pairs_rank / (avg(acc_ms) * 1e-3) / 1e9) if avg(acc_ms) > 0 else None},
                         "note": "algorithmic bytes = %d B/pair x %d pairs over %d launches; the kernel is integer-ALU-bound "
                                 "(%d mixed adds x %d v_mad_u64_u32 per pair: see `valu` and DESIGN.md §4.2)" %
                                 (bpp, pairs_rank, launches, msm_windows, MADS_PER_MIXED_ADD[curve])},
            "valu": {"kernel": "k_accumulate", "unit": "lane-mads/s (v_mad_u64_u32)", "mads_per_mixed_add": MADS_PER_MIXED_ADD[curve],
                     "mixed_adds_per_pair": [p[1] for p in plans], "achieved": mads_rank / acc_s if acc_s > 0 else None, "peak": VALU_MAD_PEAK,
                     "frac": (mads_rank / acc_s / VALU_MAD_PEAK) if acc_s > 0 else None,
                     "note": "the real bound of this kernel: mads are 79 % of its instruction stream (3 542 of 4 468 per mixed add, point load included); "
                             "SQ counters (profiles/r04_pmc_sort_summary.txt): 0.94 of the VALU issue slots at the 2.0-2.1 GHz the package power limit leaves (profiles/r04_power_probe.txt)"},
            "stages": stage_report(curve, r1cs, pk, tm1_serial, tm3_serial, plans),
            "proof_bytes": proof_b.hex(),
            # Polymath::verify (verifier.rs:19-62) on the timed proof, by the library's own CPU verifier (pm_host_verify)
            "proof_verified": bool(pm.verify(pm.make_vk(pk, x_trap, z_trap), inst[1:], proof_b)),
        }
        if world == 1 and shard_count == 1 and not multi:
            if not args.no_live_traffic:
                pk_bytes_note = "(the parent's key stays resident: the children build their own)"
                log(rank, "measuring k_accumulate's HBM traffic and clock: three rocprofv3 --pmc child passes %s ..." % pk_bytes_note)
                live, how = live_traffic(args)
                if live is not None:
                    out["roofline"]["traffic_committed_profile"] = out["roofline"]["traffic"]
                    out["roofline"]["traffic"], out["roofline"]["traffic_source"] = live, how
                else:
                    out["roofline"]["traffic_source"] += "; live measurement unavailable (%s)" % how
                if "effective_clock_GHz" in LIVE_EXTRAS:
                    clk = LIVE_EXTRAS["effective_clock_GHz"]
                    out["valu"]["effective_clock_GHz"] = clk
                    out["valu"]["effective_clock_source"] = LIVE_EXTRAS["effective_clock_source"]
                    out["valu"]["clock_limiter_note"] = ("NOT measured by this run: see profiles/r05_clock_limiter.txt (tools/clock_limiter.py, an earlier box) for which "
                                                         "limiter held the clock there")
            if args.inflight > 1 and native:
                # A prover that SERVES keeps more than one proof in flight: K host threads, one context each (own stream, workspaces and
                # proof state), the same resident key (pm_pk is immutable: include/polymath_hip.h, "Threading").  One proof's latency-bound
                # stretches -- bucket reductions, sort tails, the host's transcript between the phases -- fill with another's accumulation.
                # Reported BESIDE `value` (which stays one proof at a time, as benches/bench.rs:79 times it), never instead of it.
                try:
                    import threading
                    K, per = args.inflight, max(3, args.steps // 2)
                    pms = [pm] + [Polymath(curve, args.transcript, device=local) for _ in range(K - 1)]
                    for extra in pms[1:]:
                        for kv in args.opt:
                            name, value = kv.split("=", 1)
                            extra.ctx.set_option(name, int(value) if value.lstrip("-").isdigit() else value)
                    views = [pk] + [pk.view(e.ctx) for e in pms[1:]]
                    outs = [None] * K
                    for e, v in zip(pms, views):                       # warm-up: every context allocates its workspaces
                        assert e.prove_native(v, x_l, w_l, r_a) == proof_b

                    def serve(i):
                        for _ in range(per):
                            outs[i] = pms[i].prove_native(views[i], x_l, w_l, r_a)
                    th = [threading.Thread(target=serve, args=(i,)) for i in range(K)]
                    torch.cuda.synchronize()
                    t_s = time.perf_counter()
                    for t in th:
                        t.start()
                    for t in th:
                        t.join()
                    torch.cuda.synchronize()
                    dt_s = time.perf_counter() - t_s
                    out["throughput_in_flight"] = {"proofs_in_flight": K, "proofs": K * per, "ms_per_proof": dt_s / (K * per) * 1e3,
                                                   "constraints_per_sec": nr * K * per / dt_s, "proofs_identical": all(o == proof_b for o in outs),
                                                   "note": "serving throughput: %d host threads, one pm_ctx each, one resident pm_pk; `value` above is one proof at a time" % K}
                    for e in pms[1:]:
                        e.ctx.close()
                except Exception as e:      # noqa: BLE001 -- an extra leg must not cost the run its line
                    out["throughput_in_flight"] = {"error": repr(e)}
            if not args.no_cpu_baseline:
                cores = os.cpu_count() or 1
                cb_log = args.cpu_baseline_log or (args.log_constraints if cores >= 32 and args.log_constraints <= 20 else 16)
                log(rank, "timing the CPU restatement on the 2^%d-100-gate circuit (%d host threads) ..." % (cb_log, cores))
                if cb_log == args.log_constraints:
                    out["cpu_baseline"] = cpu_baseline(curve, cb_log, r1cs, pk, r_a, proof_b, args.transcript)
                else:
                    lc_s = PC.synthetic_r1cs_native(curve, (1 << cb_log) - 100)
                    pk_s = pm.setup(lc_s, x_trap, z_trap)
                    gp = pm.prove_native(pk_s, lc_s.inst_limbs, lc_s.wit_limbs, r_a)
                    out["cpu_baseline"] = cpu_baseline(curve, cb_log, lc_s, pk_s, r_a, gp, args.transcript)
                    pk_s.free()
            if args.msm_micro or args.ntt_micro:
                pk.free()                                        # give the HBM back: the 2^26 leg holds 100 GB of window tables
            if args.ntt_micro:
                log(rank, "standalone resident NTT legs: 2^{%s} points ..." % args.ntt_micro)
                out["ntt_micro"] = ntt_micro(pm.ctx, curve, [int(v) for v in args.ntt_micro.split(",")])
            if args.msm_micro:
                log(rank, "standalone resident MSM legs: 2^{%s} pairs ..." % args.msm_micro)
                out["msm_micro"] = msm_micro(pm.ctx, curve, [int(v) for v in args.msm_micro.split(",")])
            if args.other_configs:
                out["other_configs"] = []
                for spec in args.other_configs.split(","):
                    cv, lg = spec.split(":")
                    log(rank, "other BASELINE configuration on this GPU: %s at 2^%s - 100 gates ..." % (cv, lg))
                    try:
                        out["other_configs"].append(other_config(pm.ctx, cv, int(lg), args.transcript))
                    except Exception as e:      # noqa: BLE001 -- an extra leg must not cost the run its line
                        out["other_configs"].append({"workload": spec, "error": repr(e)})
        stage("final agreement", 120)
        line = json.dumps(out)
    if multi:
        # the line is printed only when EVERY rank got here: a failed attempt must leave nothing on stdout (launch.py)
        dist.barrier()
        if rank == 0:
            print(line, flush=True)
        dist.barrier()
        if not supervised:                  # supervised ranks leave through os._exit (main): teardown of a communicator
            dist.destroy_process_group()    # that a peer has already left is where multi-process jobs like to hang
    elif rank == 0:
        print(line, flush=True)
    if wd:
        wd.stage_end()


if __name__ == "__main__":
    main()
