"""Multi-GPU glue: one process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI).

The MSM pair ranges of a proving key are sharded contiguously over ranks (pm_pk_* shard_rank /
shard_count); each rank's phase outputs are PARTIAL G1 sums.  RCCL has no
elliptic-curve reduction op (ncclSum over limbs is wrong for points and for Montgomery
residues), so partial points are all-gathered as bytes (104 B each) and summed locally with
pm_g1_sum -- latency-bound, bandwidth irrelevant.  Challenges are then identical on all ranks
because every rank hashes the same combined points.
"""
import ctypes as ct

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


class CommPointCombiner:
    """PointCombiner's interface on top of a pm_comm (pm_comm_combine_points: all-gather + pm_g1_sum inside the library),
    for hosts that drive the three phases themselves (Polymath.prove_limbs) on a communicator-joined context."""

    def __init__(self, comm, curve, nq):
        self.comm, self.curve, self.nq = comm, curve, nq

    def __call__(self, xy, inf):
        return self.many([(xy, inf)])[0]

    def many(self, points):
        from . import api
        k, words = len(points), 2 * self.nq
        xy = np.zeros(k * words, dtype=np.uint64)
        inf = (ct.c_int * k)()
        for j, (p, i) in enumerate(points):
            xy[j * words:(j + 1) * words] = np.asarray(p, dtype=np.uint64)
            inf[j] = int(i)
        st = self.comm.L.pm_comm_combine_points(self.comm.h, api.CURVE_IDS[self.curve], k, xy.ctypes.data_as(ct.POINTER(ct.c_uint64)), inf)
        if st:
            raise api.PolymathError(st, "pm_comm_combine_points")
        return [(xy[j * words:(j + 1) * words].copy(), int(inf[j])) for j in range(k)]


# ---------------------------------------------------------------------------------------- pm_comm construction
class _CommOps(ct.Structure):
    _fields_ = [("user", ct.c_void_p),
                ("all_to_all", ct.CFUNCTYPE(ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_size_t, ct.c_void_p)),
                ("all_gather", ct.CFUNCTYPE(ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_size_t))]


class TorchComm:
    """A pm_comm whose two collectives run over torch.distributed with ANY backend, staging through host memory
    (pm_comm_from_callbacks).  For tests (gloo, several ranks sharing one GPU) and hosts without a usable RCCL; the
    production form is api.Comm.rccl, where the all-to-all never leaves the GPUs."""

    def __init__(self, rank, world):
        import torch
        import torch.distributed as dist
        from . import api
        self.torch, self.dist, self.rank_, self.world_ = torch, dist, rank, world
        self.dev = "cuda" if dist.get_backend() == "nccl" else "cpu"     # NCCL/RCCL process groups move device tensors only
        self.hip = ct.CDLL("libamdhip64.so")
        self.hip.hipMemcpy.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_size_t, ct.c_int]
        self.hip.hipStreamSynchronize.argtypes = [ct.c_void_p]
        self.failure = None
        self._ops = _CommOps(None, _CommOps._fields_[1][1](self._all_to_all), _CommOps._fields_[2][1](self._all_gather))
        L = api.load_library()
        h = ct.c_void_p()
        st = L.pm_comm_from_callbacks(ct.byref(self._ops), rank, world, ct.byref(h))
        if st:
            raise api.PolymathError(st, "pm_comm_from_callbacks")
        self.comm = api.Comm(h, keep=self)

    def _all_to_all(self, _user, d_send, d_recv, nbytes, stream):
        try:
            total = nbytes * self.world_
            if self.hip.hipStreamSynchronize(stream):
                return 6
            src = np.empty(total, dtype=np.uint8)
            if self.hip.hipMemcpy(src.ctypes.data_as(ct.c_void_p), d_send, total, 2):        # device -> host
                return 6
            t_in = self.torch.from_numpy(src).to(self.dev)
            t_out = self.torch.empty_like(t_in)
            self.dist.all_to_all_single(t_out, t_in)
            out = t_out.cpu().numpy()
            if self.hip.hipMemcpy(d_recv, out.ctypes.data_as(ct.c_void_p), total, 1):         # host -> device
                return 6
            return 0
        except Exception as e:   # noqa: BLE001 -- never unwind through the C frames
            self.failure = e
            return 8

    def _all_gather(self, _user, send, recv, nbytes):
        try:
            mine = np.ctypeslib.as_array(ct.cast(send, ct.POINTER(ct.c_uint8)), shape=(nbytes,)).copy()
            t_in = self.torch.from_numpy(mine).to(self.dev)
            outs = [self.torch.empty_like(t_in) for _ in range(self.world_)]
            self.dist.all_gather(outs, t_in)
            dst = np.ctypeslib.as_array(ct.cast(recv, ct.POINTER(ct.c_uint8)), shape=(nbytes * self.world_,))
            for r, t in enumerate(outs):
                dst[r * nbytes:(r + 1) * nbytes] = t.cpu().numpy()
            return 0
        except Exception as e:   # noqa: BLE001
            self.failure = e
            return 8


def make_comm(rank, world, device, native=True, timeout_s=60):
    """This rank's pm_comm for a torch.distributed job.  native: RCCL inside the library (rank 0's unique id travels
    over torch.distributed, whatever its backend); every rank learns whether ALL ranks succeeded, otherwise all fall back to
    TorchComm.  -> (api.Comm, description, info) with info = {"kind", "ranks_seen", "fallback_reason"}: `ranks_seen` is
    the number of distinct ranks whose tag came back from an all-gather over the communicator that will carry the proofs."""
    import torch
    import torch.distributed as dist
    from . import api
    tdev = "cuda" if dist.get_backend() == "nccl" else "cpu"

    def all_ok(ok):
        flag = torch.tensor([int(ok)], dtype=torch.int32, device=tdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item()) == 1
    reason = None if native else "not requested"
    if native:
        uid = None
        try:
            uid = api.Comm.rccl_unique_id()              # every rank: proves librccl loads here (rank 0's id is the one used)
        except Exception as e:       # noqa: BLE001 -- the fallback below is the handling
            uid, reason = None, "librccl not loadable: %s" % e
        if all_ok(uid is not None):                       # nobody enters ncclCommInitRank unless everybody can
            box = [uid if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            comm, ok = None, False
            try:
                comm = api.Comm.rccl(box[0], rank, world, device)
                comm.set_timeout_ms(timeout_s * 1000)
                probe = comm.all_gather(np.array([rank], dtype=np.int64))
                seen = len(set(probe.reshape(-1).tolist()) & set(range(world)))
                ok = probe.reshape(-1).tolist() == list(range(world))
                if ok:
                    # the transforms rely on the block order of the all-to-all (block p -> rank p, block r <- rank r): check it
                    # on this fabric once, with 64-byte blocks tagged (sender, receiver), before a prover depends on it
                    dev = torch.device("cuda", device)
                    send = torch.tensor([[rank * world + p] * 8 for p in range(world)], dtype=torch.int64, device=dev).reshape(-1)
                    recv = torch.full_like(send, -1)
                    torch.cuda.synchronize(dev)
                    comm.all_to_all_device(send.data_ptr(), recv.data_ptr(), 64, torch.cuda.current_stream(dev).cuda_stream)
                    torch.cuda.synchronize(dev)
                    ok = recv.cpu().reshape(world, 8)[:, 0].tolist() == [r * world + rank for r in range(world)]
                    if not ok:
                        reason = "all-to-all block order check failed"
                else:
                    reason = "all-gather probe returned %s" % probe.reshape(-1).tolist()
            except Exception as e:   # noqa: BLE001
                ok, reason = False, "RCCL communicator: %s" % e
            if all_ok(ok):
                return comm, "rccl (native: ncclAllToAll / ncclAllGather inside the library)", {"kind": comm.kind, "ranks_seen": seen, "fallback_reason": None}
            reason = reason or "a peer could not use RCCL"
            if comm is not None:
                # falling back: do not leave the RCCL communicator (watchdog thread, 64 events, pinned / device staging, perhaps a
                # half-finished collective on its stream) alive beside the transport that replaces it.  abort first: the destructor
                # skips ncclCommDestroy on an aborted communicator, so a peer that never arrived cannot block the teardown.
                try:
                    comm.abort("falling back to host-staged exchanges: " + reason)
                    comm.close()
                except Exception:    # noqa: BLE001 -- a failing teardown must not cost the fallback
                    pass
        else:
            reason = reason or "a peer could not load librccl"
    tc = TorchComm(rank, world)
    probe = tc.comm.all_gather(np.array([rank], dtype=np.int64))
    seen = len(set(probe.reshape(-1).tolist()) & set(range(world)))
    return tc.comm, "torch.distributed callbacks (%s), host-staged" % dist.get_backend(), \
        {"kind": "%s over torch.distributed/%s" % (tc.comm.kind, dist.get_backend()), "ranks_seen": seen, "fallback_reason": reason}
