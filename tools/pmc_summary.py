#!/usr/bin/env python3
"""Per-kernel totals of rocprofv3 --pmc csv output (one row per dispatch and counter).
  python tools/pmc_summary.py <run_counter_collection.csv> [kernel substring]"""
import csv, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
dur = collections.defaultdict(float)
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[k].add(r["Dispatch_Id"])
    if (k, r["Dispatch_Id"]) not in seen:
        seen.add((k, r["Dispatch_Id"]))
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    rows[k]["_vgpr"] = float(r["VGPR_Count"]); rows[k]["_scratch"] = float(r["Scratch_Size"]); rows[k]["_lds"] = float(r["LDS_Block_Size"])
for k in sorted(rows, key=lambda k: -dur[k]):
    c = rows[k]
    print("%-70s calls %3d  ms %9.3f  %s" % (k[:70], len(calls[k]), dur[k], "  ".join("%s=%.4g" % (n, v) for n, v in sorted(c.items()))))
