#!/usr/bin/env python3
"""Host-side cost of the pm_comm collectives on the RCCL implementation with a WORLD OF ONE (what one GPU can measure: the
library's staging, RCCL's launch path, the stream synchronisation -- no fabric): microseconds per call.
  python tools/comm_latency.py"""
import ctypes as ct, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from polymath_amd import api

hip = ct.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ct.POINTER(ct.c_void_p), ct.c_size_t]
hip.hipStreamCreate.argtypes = [ct.POINTER(ct.c_void_p)]
hip.hipStreamSynchronize.argtypes = [ct.c_void_p]
ctx = api.Context(0)
comm = api.Comm.rccl(api.Comm.rccl_unique_id(), 0, 1, 0)
st = ct.c_void_p()
assert hip.hipStreamCreate(ct.byref(st)) == 0
out = {}
for nbytes in (64, 25000):
    a = np.zeros(nbytes // 8, dtype=np.int64)
    for _ in range(20):
        comm.all_gather(a)
    t0 = time.perf_counter()
    for _ in range(500):
        comm.all_gather(a)
    out["host_all_gather_%dB_us" % nbytes] = (time.perf_counter() - t0) / 500 * 1e6
for nbytes in (1 << 20, 4 << 20):
    s, r = ct.c_void_p(), ct.c_void_p()
    assert hip.hipMalloc(ct.byref(s), nbytes) == 0 and hip.hipMalloc(ct.byref(r), nbytes) == 0
    for kind, fn in (("all_to_all", comm.all_to_all_device), ("all_gather_device", comm.all_gather_device)):
        for _ in range(20):
            fn(s.value, r.value, nbytes, st.value)
        hip.hipStreamSynchronize(st)
        t0 = time.perf_counter()
        for _ in range(200):
            fn(s.value, r.value, nbytes, st.value)
            hip.hipStreamSynchronize(st)
        out["%s_%dMB_us" % (kind, nbytes >> 20)] = (time.perf_counter() - t0) / 200 * 1e6
print(json.dumps({"comm": comm.kind, "world": 1, **{k: round(v, 1) for k, v in out.items()}}))
