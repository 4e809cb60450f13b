#!/usr/bin/env python3
"""MSM micro-benchmark (SURVEY.md §8d MSM micro-inputs): bases P_i = (i+1)G generated on the device,
uniform scalars, both resident in HBM; prints pairs/s and the HIP-event stage times.
  python tools/msm_bench.py --log-len 22 --reps 3 [--curve bn254]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from polymath_amd import api

ap = argparse.ArgumentParser()
ap.add_argument("--log-len", type=int, default=22)
ap.add_argument("--len", type=int, default=0)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--curve", default="bls12_381")
ap.add_argument("--tables", action="store_true", help="pm_bases_precompute: window tables")
ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="pm_ctx_set_option, e.g. --opt table_window_bits=24 --opt msm_task_len=256")
a = ap.parse_args()
n = a.len or (1 << a.log_len)
ctx = api.Context(0)
for kv in a.opt:
    ctx.set_option(kv.split("=", 1)[0], int(kv.split("=", 1)[1]))
t0 = time.time()
bases = api.Bases.multiples(ctx, a.curve, n)
gen_s = time.time() - t0
t0 = time.time()
if a.tables:
    bases.precompute()
tbl_s = time.time() - t0
g = torch.Generator(device="cuda").manual_seed(1234)
sc = torch.randint(0, 2**62, (n, 4), dtype=torch.int64, device="cuda", generator=g) * 4 + torch.randint(0, 4, (n, 4), dtype=torch.int64, device="cuda", generator=g)
sc[:, 3] &= (1 << 61) - 1          # < 2^253: a valid residue for both scalar fields
torch.cuda.synchronize()
out = None
res = []
for rep in range(a.reps + 1):
    t0 = time.perf_counter()
    out, inf = bases.msm(None, 0, n, device_ptr=sc.data_ptr())
    dt = time.perf_counter() - t0
    tm = ctx.timings()
    if rep:
        res.append((dt, tm))
best = min(r[0] for r in res)
tm = res[-1][1]
print(json.dumps({"curve": a.curve, "len": n, "best_ms": best * 1e3, "pairs_per_sec": n / best, "gen_s": gen_s, "tables": a.tables, "tables_s": tbl_s,
                  "options": {k: ctx.get_option(k) for k in api.OPTIONS},
                  "stage_ms": {k: round(v, 3) for k, v in tm.items() if k.startswith("msm")}}))
