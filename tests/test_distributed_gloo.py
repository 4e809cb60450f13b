"""CPU, world_size 2, gloo: the N > 1 path of bench.py / polymath_amd.distributed -- per-rank partial G1
points are all-gathered and summed with pm_g1_sum (host code of the product library; RCCL has no
elliptic-curve reduction).  Each rank holds the MSM of its contiguous pair range, computed here by the CPU
oracle; the combined point must equal the whole MSM on every rank, including a rank whose part is infinity."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch.distributed as dist
from oracle import cpp_oracle as CO
from polymath_amd.distributed import PointCombiner
sys.path.insert(0, os.path.join(%(root)r, "tests"))
from helpers import rand_fr_limbs
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
for curve, nq in (("bls12_381", 6), ("bn254", 4)):
    n = 257
    bases, sc = CO.g1_multiples(curve, n), rand_fr_limbs(curve, n, 5)
    whole, winf = CO.msm(curve, bases, sc, 1)
    lo, hi = n * rank // world, n * (rank + 1) // world          # the shard rule of pm_pk_* (res_lo / res_hi)
    part, pinf = CO.msm(curve, bases[lo:hi], sc[lo:hi], 1)
    comb = PointCombiner(None, curve, nq, rank, world, backend_gloo=True)
    got, ginf = comb(part, pinf)
    assert ginf == 0 and np.array_equal(got, whole), (curve, rank)
    # one rank contributes the point at infinity
    zero = np.zeros_like(part)
    got, ginf = comb(part if rank == 0 else zero, 0 if rank == 0 else 1)
    first, _ = CO.msm(curve, bases[:n // world], sc[:n // world], 1)
    assert np.array_equal(got, first), (curve, rank, "infinity part")
dist.barrier()
dist.destroy_process_group()
sys.stdout.write("rank" + str(rank) + "-ok\n"); sys.stdout.flush()
'''


def test_point_combiner_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("-ok") == 2 and "rank0" in out.stdout and "rank1" in out.stdout, out.stdout
