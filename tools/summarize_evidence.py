#!/usr/bin/env python3
"""The numbers of an evidence set (tools/collect_profiles.sh <tag> bench|tests -> gpurun_out/<tag>/) in the order the READMEs quote them.
    python tools/summarize_evidence.py [gpurun_out/r06_final]"""
import csv, glob, json, os, re, sys
O = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r06_final"


def last(name):
    p = os.path.join(O, name)
    if not os.path.exists(p):
        return None
    lines = [l for l in open(p).read().strip().split("\n") if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None


d = last("bench.json")
if d:
    r, v = d["roofline"], d["valu"]
    print("2^20: %.2f ms per proof = %.2f M constraints/s (resident %.2f); k_accumulate %.2f ms per launch, roofline.frac %.4f, traffic %.2f GB, valu.frac %.3f at %.2f GHz"
          % (d["ms_per_step"], d["value"] / 1e6, d.get("ms_per_step_hbm_resident", 0), r["avg_launch_ms"], r["frac"], (r["traffic"] or 0) / 1e9, v["frac"], v.get("effective_clock_GHz") or 0))
    print("stages:", {k: round(x["ms"], 2) for k, x in d["stages"].items() if isinstance(x, dict)})
    print("ntt_micro:", [(x["log_n"], round(x["ms"], 4), round(x["algorithmic_GBps"]), round(x["valu_frac"], 3)) for x in d.get("ntt_micro", [])])
    print("msm_micro M pairs/s:", [(x["len"], round(x["pairs_per_sec"] / 1e6)) for x in d.get("msm_micro", [])])
    t = d.get("throughput_in_flight")
    if t:
        print("in flight: %.2f ms per proof" % t["ms_per_proof"])
    c = d.get("cpu_baseline")
    if c:
        print("cpu_baseline: %.2f s = %.1f K constraints/s on %s threads, MSM %.2f M pairs/s, identical %s" % (c["seconds"], c["value"] / 1e3, c["cores"], c["msm_pairs_per_sec"] / 1e6, c["proof_identical_to_gpu"]))
    for o in d.get("other_configs", []):
        print("other:", o.get("workload", "")[:48], "%.2f ms" % o["ms_per_proof"] if "ms_per_proof" in o else o.get("error"), o.get("proof_verified"))
for f in ("bench_bn254.json", "bench_2p22.json", "bench_2p24.json", "bench_under_rocprof.json"):
    x = last(f)
    if x:
        print(f, "%.2f ms, %.2f M constraints/s, verified %s, k_accumulate %.2f ms" % (x["ms_per_step"], x["value"] / 1e6, x.get("proof_verified"), x["roofline"]["avg_launch_ms"]))
for f in sorted(glob.glob(os.path.join(O, "shard_emulation*.json"))):
    x = last(os.path.basename(f))
    if x:
        print(os.path.basename(f), "%.2f ms per rank" % x["emulated_ms_per_rank"], [round(b, 1) for b in x["busy_ms_per_rank"]])
ks = os.path.join(O, "stats", "run_kernel_stats.csv")
if os.path.exists(ks):
    for row in csv.DictReader(open(ks)):
        if "k_accumulate" in row["Name"]:
            print("rocprofv3 --stats k_accumulate: %s calls, %.3f ms average" % (row["Calls"], float(row["AverageNs"]) / 1e6))
tot = {}
for c in ("fetch", "write"):
    p = os.path.join(O, "pmc_" + c, "run_counter_collection.csv")
    if os.path.exists(p):
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(p)) if "k_accumulate" in r["Kernel_Name"]]
        tot[c] = (sum(vals), len(vals))
if len(tot) == 2:
    print("2 x FETCH_SIZE + WRITE_SIZE = %.2f GB per launch over %d launches" % ((2 * tot["fetch"][0] + tot["write"][0]) * 1024 / tot["fetch"][1] / 1e9, tot["fetch"][1]))
p = os.path.join(O, "pytest_gpu.log")
if os.path.exists(p):
    m = re.findall(r"^\d+ passed.*$", open(p).read(), re.M)
    print("pytest -m gpu:", m[-1] if m else "no summary line")
p = os.path.join(O, "smoke.log")
if os.path.exists(p):
    print("smoke:", open(p).read().strip().split("\n")[-1])
