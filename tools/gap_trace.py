#!/usr/bin/env python3
"""GPU idle gaps inside ONE proof and the HIP calls the host made during each: from
  rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d OUT -o t -- python3 bench.py --steps 2 --warmup 1 ...
  python tools/gap_trace.py OUT/t_kernel_trace.csv OUT/t_hip_api_trace.csv [min_gap_us] [proof]
proof = -1: the last one in the trace (bench.py: the msm_overlap = 0 proof of the stage report), -2 (default): the last TIMED proof."""
import csv, sys, collections
k = list(csv.DictReader(open(sys.argv[1])))
a = list(csv.DictReader(open(sys.argv[2])))
floor = float(sys.argv[3]) if len(sys.argv) > 3 else 8.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pm::", "")[:44]) for r in k)
wit = [i for i, e in enumerate(ev) if "k_witness_rows" in e[2]]
which = int(sys.argv[4]) if len(sys.argv) > 4 else -2
seg = ev[wit[which]:wit[which + 1]] if which < -1 else ev[wit[-1]:]
api = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r["Thread_Id"]) for r in a)
t0, ce, last = seg[0][0], seg[0][1], seg[0][2]
tot = 0
for s, e, n in seg[1:]:
    if s > ce:
        g = (s - ce) / 1e3
        tot += g
        if g >= floor:
            calls = collections.Counter()
            spent = collections.Counter()
            for cs, c_e, f, tid in api:
                if c_e > ce and cs < s:
                    calls[f] += 1
                    spent[f] += (min(c_e, s) - max(cs, ce)) / 1e3
            inside = sum(spent.values())
            print("%7.1f us at %7.3f ms  after %-28s before %-28s  HIP calls %.0f us: %s" % (g, (ce - t0) / 1e6, last[:28], n[:28], inside,
                  ", ".join("%s x%d %.0f" % (f.replace("hip", ""), calls[f], spent[f]) for f in sorted(spent, key=lambda f: -spent[f])[:5])))
    if e > ce:
        ce, last = e, n
print("proof span %.3f ms, idle %.3f ms" % ((ce - t0) / 1e6, tot / 1e3))
