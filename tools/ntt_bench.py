#!/usr/bin/env python3
"""Standalone NTT timing (pm_ntt_device, data resident in HBM): ms and algorithmic GB/s (64 B per element per
transform) for several sizes.   python tools/ntt_bench.py [--curve bls12_381] [--logs 20 21 22 24 25]"""
import argparse
import ctypes as ct
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--curve", default="bls12_381")
    ap.add_argument("--logs", type=int, nargs="+", default=[20, 21, 22, 24, 25])
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    from polymath_amd import api
    ctx = api.Context(0)
    hip = ct.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ct.POINTER(ct.c_void_p), ct.c_size_t]
    hip.hipMemcpy.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_size_t, ct.c_int]
    hip.hipFree.argtypes = [ct.c_void_p]
    hip.hipDeviceSynchronize.argtypes = []
    rng = np.random.default_rng(1)
    for lg in args.logs:
        n = 1 << lg
        a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)   # < r: top limb below 2^62
        p = ct.c_void_p()
        assert hip.hipMalloc(ct.byref(p), a.nbytes) == 0
        assert hip.hipMemcpy(p, a.ctypes.data_as(ct.c_void_p), a.nbytes, 1) == 0
        ctx.ntt_device(args.curve, p.value, lg, False)               # warm-up (twiddle tables)
        ctx.ntt_device(args.curve, p.value, lg, True)
        hip.hipDeviceSynchronize()
        best = {}
        for inv in (False, True):
            ts = []
            for _ in range(args.reps):
                t0 = time.perf_counter()
                ctx.ntt_device(args.curve, p.value, lg, inv)
                hip.hipDeviceSynchronize()
                ts.append(time.perf_counter() - t0)
            best["inverse" if inv else "forward"] = min(ts) * 1e3
        back = np.empty_like(a)
        assert hip.hipMemcpy(back.ctypes.data_as(ct.c_void_p), p, a.nbytes, 2) == 0
        hip.hipFree(p)
        print(json.dumps({"curve": args.curve, "log_n": lg, "forward_ms": best["forward"], "inverse_ms": best["inverse"],
                          "algorithmic_GBps_forward": 64 * n / (best["forward"] * 1e-3) / 1e9,
                          "round_trips_ok": bool(np.array_equal(back, a))}), flush=True)


if __name__ == "__main__":
    main()
