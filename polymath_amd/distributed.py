"""Multi-GPU glue: one process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI).

The MSM pair ranges of a proving key are sharded contiguously over ranks (pm_pk_* shard_rank /
shard_count); each rank's phase outputs are PARTIAL G1 sums.  RCCL has no
elliptic-curve reduction op (ncclSum over limbs is wrong for points and for Montgomery
residues), so partial points are all-gathered as bytes (104 B each) and summed locally with
pm_g1_sum -- latency-bound, bandwidth irrelevant.  Challenges are then identical on all ranks
because every rank hashes the same combined points.
"""
import numpy as np


class PointCombiner:
    def __init__(self, ctx, curve, nq, rank, world, device=None, backend_gloo=False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.ctx, self.curve, self.nq, self.rank, self.world = ctx, curve, nq, rank, world
        self.dev = torch.device("cpu") if backend_gloo or device is None else torch.device("cuda", device)
        self.words = 2 * nq + 1

    def __call__(self, xy, inf):
        return self.many([(xy, inf)])[0]

    def many(self, points):
        """[(xy, inf), ...] partial points of this rank -> the same list summed over all ranks, with ONE all-gather
        (phase 1 returns two points: [a]_1 and [c]_1)."""
        if self.world == 1:
            return list(points)
        torch, dist = self.torch, self.dist
        k, words = len(points), self.words
        mine = np.zeros(k * words, dtype=np.int64)
        for j, (xy, inf) in enumerate(points):
            mine[j * words:j * words + 2 * self.nq] = np.asarray(xy, dtype=np.uint64).view(np.int64)
            mine[(j + 1) * words - 1] = int(inf)
        t = torch.from_numpy(mine).to(self.dev)
        out = torch.empty(self.world * k * words, dtype=torch.int64, device=self.dev)
        dist.all_gather_into_tensor(out, t)
        allp = out.cpu().numpy().reshape(self.world, k, words)
        from . import api
        res = []
        for j in range(k):
            pts = np.ascontiguousarray(allp[:, j, :2 * self.nq]).view(np.uint64)
            infs = np.ascontiguousarray(allp[:, j, -1]).astype(np.int32)
            res.append(api.g1_sum(self.curve, pts, infs))
        return res
