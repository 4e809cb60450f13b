#!/usr/bin/env python3
"""Where key generation spends its time: PM_PROFILE_HOST=1 python tools/setup_time.py --log-constraints 24
prints the host thread's split (stderr) and the wall time of pm_pk_generate."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PM_PROFILE_HOST", "1")
from polymath_amd import circuits as PC
from polymath_amd.polymath import Polymath, FIELDS
ap = argparse.ArgumentParser()
ap.add_argument("--log-constraints", type=int, default=20)
ap.add_argument("--curve", default="bls12_381")
ap.add_argument("--shards", type=int, default=1)
a = ap.parse_args()
r = FIELDS[a.curve]["r"]
t0 = time.time()
lc = PC.synthetic_r1cs_native(a.curve, (1 << a.log_constraints) - 100)
t_synth = time.time() - t0
g = PC.SplitMix64(0xBE7C4)
x, z = g.fr(r), g.fr(r)
pm = Polymath(a.curve, "merlin", device=0)
out = []
for rep in range(2):
    t0 = time.time()
    pk = pm.setup(lc, x, z, shard_rank=0, shard_count=a.shards, layout="vector" if a.shards > 1 else "pairs")
    out.append(time.time() - t0)
    pk.free()
print(json.dumps({"log_constraints": a.log_constraints, "shards": a.shards, "synth_s": t_synth, "setup_s": out}))
