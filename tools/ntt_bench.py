#!/usr/bin/env python3
"""Standalone resident NTT timing (64 B/element/transform algorithmic), 2^20 ... 2^25 points, both
directions, canonical inputs per curve (limbs drawn below the modulus' top limb), round trip checked.
  python tools/ntt_bench.py [--curve bn254] [--logs 20,21,22,24,25]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from polymath_amd import api
from polymath_amd.polymath import FIELDS

ap = argparse.ArgumentParser()
ap.add_argument("--curve", default="bls12_381")
ap.add_argument("--logs", default="20,21,22,24,25")
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
ctx = api.Context(0)
r = FIELDS[a.curve]["r"]
top = r >> 192                       # the modulus' top 64-bit limb: a top limb strictly below it makes every value canonical
for lg in (int(v) for v in a.logs.split(",")):
    n = 1 << lg
    g = torch.Generator(device="cuda").manual_seed(lg)
    x = torch.randint(0, 2**62, (n, 4), dtype=torch.int64, device="cuda", generator=g) * 4 + torch.randint(0, 4, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    x[:, 3] = torch.randint(0, top, (n,), dtype=torch.int64, device="cuda", generator=g)
    ref = x.clone()
    torch.cuda.synchronize()
    best = {}
    for inv in (False, True):
        ts = []
        for rep in range(a.reps + 1):
            t0 = time.perf_counter()
            ctx.ntt_device(a.curve, x.data_ptr(), lg, inv)
            ts.append(time.perf_counter() - t0)
        best[inv] = min(ts[1:])
        # an even number of forward (then inverse) applications: undo them so that the round trip is checked below
    # reps + 1 forward then reps + 1 inverse transforms: identity
    ok = bool(torch.equal(x, ref))
    print(json.dumps({"curve": a.curve, "log_n": lg, "fwd_ms": best[False] * 1e3, "inv_ms": best[True] * 1e3,
                      "algorithmic_GBps_fwd": 64.0 * n / best[False] / 1e9, "algorithmic_GBps_inv": 64.0 * n / best[True] / 1e9,
                      "hbm_frac_fwd": 64.0 * n / best[False] / 1e9 / 8000.0, "round_trips_ok": ok,
                      "gpu_ms_last": ctx.timings()["ntt"]}))
