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


# The exchange of the four-step transform (PM_SHARD_VECTOR) between two REAL processes: every rank
# holds only the rows it owns (cyclic), transforms them locally (CPU oracle), sends block p of the result to rank p
# with ONE all_to_all over gloo -- the same block order pm_comm->all_to_all moves on the GPUs -- and finishes with the
# size-N butterfly on what it received; its coefficients (blocked layout, pm_layout_indices) must equal the direct
# size-n transform's.  The small host collectives are checked through the library's own pm_comm (TorchComm callbacks ->
# pm_comm_all_gather, pm_comm_combine_points).
WORKER_NTT = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import torch.distributed as dist
from oracle import cpp_oracle as CO
from oracle.pyref.fields import CURVES
from polymath_amd import api
from polymath_amd.distributed import TorchComm
rank, N = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
curve, log_n = "bls12_381", 8
c = CURVES[curve]
n = 1 << log_n
m, B = n // N, n // N // N
rng = np.random.default_rng(7)
evals = [int(v) for v in rng.integers(1, 1 << 62, size=n)]                      # the same vector on both ranks ...
direct = CO.fr_from_mont_limbs(curve, CO.ntt(curve, CO.fr_to_mont_limbs(curve, evals), log_n, True))
rows, coef = api.layout_indices(n, N, rank, coefficients=False), api.layout_indices(n, N, rank, coefficients=True)
mine = CO.fr_to_mont_limbs(curve, [evals[int(i)] for i in rows])                # ... of which a rank touches only its rows
local = CO.ntt(curve, mine, log_n - (N.bit_length() - 1), True)                  # [m, 4] limbs, natural order k2
send = torch.from_numpy(local.view(np.int64).reshape(-1).copy())                # block p = k2 in [pB, (p+1)B)
recv = torch.empty_like(send)
dist.all_to_all_single(recv, send)
got = CO.fr_from_mont_limbs(curve, recv.numpy().view(np.uint64).reshape(m, 4))  # [r][b]: rank r's values at k2 = rank B + b
omega = pow(c.two_adic_root, 1 << (c.two_adicity - log_n), c.r)
winv, ninv = pow(omega, -1, c.r), pow(N, -1, c.r)
for b in range(B):
    k2 = rank * B + b
    col = [got[r * B + b] * pow(winv, r * k2, c.r) %% c.r for r in range(N)]
    for k1 in range(N):
        v = sum(col[r] * pow(winv, m * r * k1, c.r) for r in range(N)) %% c.r * ninv %% c.r
        assert v == direct[int(coef[k1 * B + b])], (rank, k1, b)
# the library's pm_comm over this process group: host all-gather and the native point combine
tc = TorchComm(rank, N)
allv = tc.comm.all_gather(np.array([rank * 10 + 1, 7], dtype=np.int64))
assert allv.tolist() == [[1, 7], [11, 7]]
import ctypes as ct
bases, sc = CO.g1_multiples(curve, 64), CO.fr_to_mont_limbs(curve, list(range(1, 65)))
whole, _ = CO.msm(curve, bases, sc, 1)
lo, hi = 64 * rank // N, 64 * (rank + 1) // N
part, pinf = CO.msm(curve, bases[lo:hi], sc[lo:hi], 1)
xy = np.ascontiguousarray(part, dtype=np.uint64).copy()
inf = (ct.c_int * 1)(pinf)
L = api.load_library()
assert L.pm_comm_combine_points(tc.comm.h, 0, 1, xy.ctypes.data_as(ct.POINTER(ct.c_uint64)), inf) == 0
assert inf[0] == 0 and np.array_equal(xy, whole)
dist.barrier()
dist.destroy_process_group()
sys.stdout.write("rank" + str(rank) + "-ok\n"); sys.stdout.flush()
'''


def test_four_step_exchange_and_pm_comm_world2_gloo(tmp_path):
    script = tmp_path / "worker_ntt.py"
    script.write_text(WORKER_NTT % {"root": ROOT})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29534", str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("-ok") == 2 and "rank0" in out.stdout and "rank1" in out.stdout, out.stdout
