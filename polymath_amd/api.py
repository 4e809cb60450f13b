"""ctypes binding of libpolymath_hip.so (include/polymath_hip.h).

This is the Python face of the C ABI: thin, typed wrappers on numpy uint64 limb arrays
(Montgomery form, arkworks' in-memory layout).  There is NO CPU fallback: if the library is
missing or no GPU is present, calls raise.
"""
import ctypes as ct
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# POLYMATH_HIP_LIB: another BUILD of the same library (same-box A/B runs load variants from ab/ through this instead of
# copying them over the in-tree file; tools/README.md).  Still the HIP library: there is no CPU fallback behind it.
LIB_PATH = os.environ.get("POLYMATH_HIP_LIB") or os.path.join(HERE, "libpolymath_hip.so")

PM_BLS12_381, PM_BN254 = 0, 1
CURVE_IDS = {"bls12_381": PM_BLS12_381, "bn254": PM_BN254}
FQ_LIMBS64 = {PM_BLS12_381: 6, PM_BN254: 4}

STATUS = {0: "PM_OK", 1: "PM_ERR_INVALID_ARG", 2: "PM_ERR_LEN_MISMATCH", 3: "PM_ERR_DOMAIN_TOO_LARGE",
          4: "PM_ERR_REMAINDER_NONZERO", 5: "PM_ERR_DEGREE_BOUND", 6: "PM_ERR_HIP", 7: "PM_ERR_NO_DEVICE",
          8: "PM_ERR_STATE", 9: "PM_ERR_COMM"}
(X_POWERS, X_POWERS_Y_ALPHA, X_POWERS_Y_GAMMA, X_POWERS_Y_GAMMA_Z, X_POWERS_ZH_BY_Y_ALPHA,
 UJ_WJ_LCS_BY_Y_ALPHA) = range(6)
TIMING_SLOTS = ["witness_map", "ntt", "poly", "msm_sort", "msm_accumulate", "msm_reduce", "msm_total", "phase"]

u64p = ct.POINTER(ct.c_uint64)
u32p = ct.POINTER(ct.c_uint32)
intp = ct.POINTER(ct.c_int)


# pm_combine_fn: int (*)(void *user, int count, uint64_t *xy, int *inf)
COMBINE_FN = ct.CFUNCTYPE(ct.c_int, ct.c_void_p, ct.c_int, ct.POINTER(ct.c_uint64), ct.POINTER(ct.c_int))


class PmCsr(ct.Structure):
    _fields_ = [("nrows", ct.c_uint64), ("rowptr", u64p), ("col", u32p), ("val", u64p)]


class PmBaseArray(ct.Structure):
    _fields_ = [("points", ct.c_void_p), ("len", ct.c_size_t), ("stride", ct.c_size_t)]


class PolymathError(RuntimeError):
    def __init__(self, status, detail=""):
        super().__init__("%s (%d) %s" % (STATUS.get(status, "?"), status, detail))
        self.status = status


EXPORTS = [
    "pm_device_count", "pm_ctx_create", "pm_ctx_destroy", "pm_last_error", "pm_last_timings", "pm_ntt",
    "pm_ntt_device", "pm_msm_g1", "pm_bases_upload", "pm_bases_generate_multiples", "pm_bases_download",
    "pm_bases_precompute", "pm_bases_len", "pm_bases_free", "pm_msm_g1_resident", "pm_g1_sum", "pm_pk_load", "pm_pk_generate",
    "pm_pk_info", "pm_pk_msm_plan", "pm_pk_export_bases", "pm_pk_free", "pm_prove_phase1", "pm_prove_phase1_device", "pm_prove_phase2", "pm_prove_phase3", "pm_host_prove", "pm_host_prove_sharded",
    "pm_prove_tap", "pm_host_keccak_f1600", "pm_synth_r1cs", "pm_selftest_field",
    "pm_pk_load_sharded", "pm_pk_generate_sharded", "pm_layout_indices", "pm_pk_msm_pieces",
    "pm_comm_rccl_unique_id", "pm_comm_rccl_create", "pm_comm_local_create", "pm_comm_from_callbacks", "pm_comm_destroy", "pm_comm_rank",
    "pm_comm_world", "pm_comm_last_error", "pm_comm_kind", "pm_comm_set_timeout_ms", "pm_comm_abort", "pm_comm_failed", "pm_comm_busy_ms", "pm_host_make_vk", "pm_host_verify", "pm_comm_all_gather", "pm_comm_all_to_all", "pm_comm_all_gather_device", "pm_comm_combine_points", "pm_ctx_set_comm",
    "pm_ctx_set_option", "pm_ctx_get_option", "pm_comm_local_set_serialize",
]
# pm_option / pm_tables_mode (include/polymath_hip.h)
OPTIONS = {"msm_overlap": 0, "ntt_overlap": 1, "tables": 2, "msm_max_piece_log": 3, "max_seg_log": 4, "inflight_contexts": 5,
           "msm_task_len": 6, "table_window_bits": 7}
TABLES_MODES = {"off": 0, "auto": 1, "wide": 2, "no_wide": 3}
SHARD_PAIRS, SHARD_VECTOR = 0, 1
LAYOUTS = {"pairs": SHARD_PAIRS, "vector": SHARD_VECTOR, 0: 0, 1: 1}

_lib = None


def load_library():
    """Loads the HIP library or raises -- the product never falls back to a CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libpolymath_hip.so not built: run `python -m polymath_amd.build` "
                          "(__graft_entry__.build()).  There is no CPU fallback.")
    L = ct.CDLL(LIB_PATH)
    vp, sz, u64, i = ct.c_void_p, ct.c_size_t, ct.c_uint64, ct.c_int
    L.pm_device_count.restype = i
    L.pm_ctx_create.argtypes = [i, ct.POINTER(vp)]
    L.pm_ctx_destroy.argtypes = [vp]
    L.pm_ctx_destroy.restype = None
    L.pm_last_error.argtypes = [vp]
    L.pm_last_error.restype = ct.c_char_p
    L.pm_last_timings.argtypes = [vp, ct.POINTER(ct.c_double), i]
    L.pm_ntt.argtypes = [vp, i, u64p, ct.c_uint, i]
    L.pm_ntt_device.argtypes = [vp, i, vp, ct.c_uint, i]
    L.pm_msm_g1.argtypes = [vp, i, vp, sz, u64p, sz, u64p, intp]
    L.pm_bases_upload.argtypes = [vp, i, vp, sz, sz, ct.POINTER(vp)]
    L.pm_bases_generate_multiples.argtypes = [vp, i, sz, ct.POINTER(vp)]
    L.pm_bases_download.argtypes = [vp, vp, sz, sz, u64p]
    L.pm_bases_precompute.argtypes = [vp, vp]
    L.pm_bases_len.argtypes = [vp]
    L.pm_bases_len.restype = sz
    L.pm_bases_free.argtypes = [vp]
    L.pm_bases_free.restype = None
    L.pm_msm_g1_resident.argtypes = [vp, vp, sz, vp, i, sz, u64p, intp]
    L.pm_g1_sum.argtypes = [i, u64p, intp, sz, u64p, intp]
    L.pm_pk_load.argtypes = [vp, i, u64, u64, u64, u64, u64, ct.POINTER(PmCsr), ct.POINTER(PmCsr), ct.POINTER(PmCsr),
                             ct.POINTER(PmBaseArray), i, i, ct.POINTER(vp)]
    L.pm_pk_generate.argtypes = [vp, i, u64, u64, u64, ct.POINTER(PmCsr), ct.POINTER(PmCsr), ct.POINTER(PmCsr), u64p,
                                 u64p, i, i, ct.POINTER(vp)]
    L.pm_pk_info.argtypes = [vp, u64p, u64p, u64p, u64p, u64p]
    L.pm_host_prove.argtypes = [vp, vp, ct.c_int, u64p, ct.c_void_p, ct.c_void_p, ct.c_int, u64p, ct.c_char_p, ct.c_size_t, ct.POINTER(ct.c_size_t)]
    L.pm_host_prove_sharded.argtypes = [vp, vp, ct.c_int, u64p, ct.c_void_p, ct.c_void_p, ct.c_int, u64p, COMBINE_FN, ct.c_void_p, ct.c_char_p,
                                        ct.c_size_t, ct.POINTER(ct.c_size_t)]
    L.pm_pk_msm_plan.argtypes = [vp, ct.c_int, u64p, ct.POINTER(ct.c_uint), ct.POINTER(ct.c_uint), intp]
    L.pm_pk_export_bases.argtypes = [vp, vp, i, sz, sz, u64p]
    L.pm_pk_free.argtypes = [vp]
    L.pm_pk_free.restype = None
    L.pm_prove_phase1.argtypes = [vp, vp, u64p, u64p, u64p, u64p, intp, u64p, intp]
    L.pm_prove_phase1_device.argtypes = [vp, vp, vp, vp, u64p, u64p, intp, u64p, intp]
    L.pm_prove_phase2.argtypes = [vp, u64p, u64p]
    L.pm_prove_phase3.argtypes = [vp, u64p, u64p, u64p, u64p, u64p, intp]
    L.pm_prove_tap.argtypes = [vp, i, u64p, sz, ct.POINTER(sz)]
    L.pm_host_keccak_f1600.argtypes = [u64p]
    L.pm_host_keccak_f1600.restype = None
    L.pm_synth_r1cs.argtypes = [i, u64, u64, u64p, u32p, u64p, u32p, u64p, u32p, u64p, u64p]
    L.pm_selftest_field.argtypes = [vp, sz, u64, u64p]
    L.pm_pk_load_sharded.argtypes = [vp, i, u64, u64, u64, u64, u64, ct.POINTER(PmCsr), ct.POINTER(PmCsr), ct.POINTER(PmCsr),
                                     ct.POINTER(PmBaseArray), i, i, i, ct.POINTER(vp)]
    L.pm_pk_generate_sharded.argtypes = [vp, i, u64, u64, u64, ct.POINTER(PmCsr), ct.POINTER(PmCsr), ct.POINTER(PmCsr), u64p,
                                         u64p, i, i, i, ct.POINTER(vp)]
    L.pm_layout_indices.argtypes = [u64, i, i, i, u64p]
    L.pm_pk_msm_pieces.argtypes = [vp, i, u64p, u64p, sz, ct.POINTER(sz)]
    L.pm_comm_rccl_unique_id.argtypes = [vp]
    L.pm_comm_rccl_create.argtypes = [vp, i, i, i, ct.POINTER(vp)]
    L.pm_comm_local_create.argtypes = [i, ct.POINTER(vp)]
    L.pm_comm_from_callbacks.argtypes = [vp, i, i, ct.POINTER(vp)]
    L.pm_comm_destroy.argtypes = [vp]
    L.pm_comm_destroy.restype = None
    L.pm_comm_rank.argtypes = [vp]
    L.pm_comm_world.argtypes = [vp]
    L.pm_comm_last_error.argtypes = [vp]
    L.pm_comm_last_error.restype = ct.c_char_p
    L.pm_comm_kind.argtypes = [vp]
    L.pm_comm_kind.restype = ct.c_char_p
    L.pm_comm_set_timeout_ms.argtypes = [vp, ct.c_long]
    L.pm_comm_abort.argtypes = [vp, ct.c_char_p]
    L.pm_comm_failed.argtypes = [vp]
    L.pm_host_make_vk.argtypes = [i, u64, u64, u64, u64p, u64p, u64p, ct.c_char_p, sz, ct.POINTER(sz)]
    L.pm_host_verify.argtypes = [i, i, ct.c_char_p, sz, u64p, sz, ct.c_char_p, sz, intp]
    L.pm_comm_busy_ms.argtypes = [vp, i]
    L.pm_comm_busy_ms.restype = ct.c_double
    L.pm_comm_all_gather.argtypes = [vp, vp, vp, sz]
    L.pm_comm_all_to_all.argtypes = [vp, vp, vp, sz, vp]
    L.pm_comm_all_gather_device.argtypes = [vp, vp, vp, sz, vp]
    L.pm_comm_combine_points.argtypes = [vp, i, i, u64p, intp]
    L.pm_ctx_set_comm.argtypes = [vp, vp]
    L.pm_ctx_set_option.argtypes = [vp, i, ct.c_longlong]
    L.pm_ctx_get_option.argtypes = [vp, i, ct.POINTER(ct.c_longlong)]
    L.pm_comm_local_set_serialize.argtypes = [vp, i]
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(u64p)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def g1_sum(curve, pts, infs=None):
    """pm_g1_sum: host-side sum of G1 points (combining per-GPU partial MSM results); needs no GPU."""
    L = load_library()
    cid = CURVE_IDS[curve]
    pts = _c(pts).reshape(-1, 2 * FQ_LIMBS64[cid])
    out = np.zeros(2 * FQ_LIMBS64[cid], dtype=np.uint64)
    inf = ct.c_int(0)
    ia = np.ascontiguousarray(infs, dtype=np.int32) if infs is not None else None
    st = L.pm_g1_sum(cid, _p(pts), ia.ctypes.data_as(intp) if ia is not None else None, len(pts), _p(out), ct.byref(inf))
    if st:
        raise PolymathError(st, "pm_g1_sum")
    return out, inf.value


TRANSCRIPT_IDS = {"merlin": 0, "keccak256": 1, "blake3": 2}


def make_vk(curve, n, m0, sigma, omega_limbs, x_trapdoor_limbs, z_trapdoor_limbs):
    """pm_host_make_vk: VerifyingKey::serialize_compressed bytes of the key the two trapdoors define (host code, no GPU)."""
    L = load_library()
    buf = ct.create_string_buffer(512)
    n_out = ct.c_size_t(0)
    st = L.pm_host_make_vk(CURVE_IDS[curve], n, m0, sigma, _p(_c(omega_limbs)), _p(_c(x_trapdoor_limbs)), _p(_c(z_trapdoor_limbs)), buf, len(buf),
                           ct.byref(n_out))
    if st:
        raise PolymathError(st, "pm_host_make_vk")
    return buf.raw[:n_out.value]


def verify(curve, transcript, vk_bytes, public_input_limbs, proof_bytes):
    """pm_host_verify: Polymath::verify (verifier.rs:19-62) on the CPU with the library's own pairing.  public inputs WITHOUT the
    leading one, Montgomery limbs [k, 4].  Malformed bytes raise; -> bool."""
    L = load_library()
    pub = _c(public_input_limbs).reshape(-1, 4) if len(public_input_limbs) else np.zeros((0, 4), dtype=np.uint64)
    acc = ct.c_int(0)
    st = L.pm_host_verify(CURVE_IDS[curve], TRANSCRIPT_IDS[transcript], bytes(vk_bytes), len(vk_bytes), _p(pub) if len(pub) else None, len(pub),
                          bytes(proof_bytes), len(proof_bytes), ct.byref(acc))
    if st:
        raise PolymathError(st, "pm_host_verify")
    return bool(acc.value)


def synth_r1cs(curve, nr, seed):
    """pm_synth_r1cs: the SURVEY §8d synthetic R1CS as limb arrays -> (CsrArrays A, B, C, instance [2,4], witness [nr+1,4])."""
    L = load_library()
    vals = [np.zeros((nr, 4), dtype=np.uint64) for _ in range(3)]
    cols = [np.zeros(nr, dtype=np.uint32) for _ in range(3)]
    inst, wit = np.zeros((2, 4), dtype=np.uint64), np.zeros((nr + 1, 4), dtype=np.uint64)
    st = L.pm_synth_r1cs(CURVE_IDS[curve], nr, seed, _p(vals[0]), cols[0].ctypes.data_as(u32p), _p(vals[1]), cols[1].ctypes.data_as(u32p),
                         _p(vals[2]), cols[2].ctypes.data_as(u32p), _p(inst), _p(wit))
    if st:
        raise PolymathError(st, "pm_synth_r1cs")
    rowptr = np.arange(nr + 1, dtype=np.uint64)
    return [CsrArrays(rowptr, cols[k], vals[k]) for k in range(3)], inst, wit


def layout_indices(n, shard_count, shard_rank, coefficients=True):
    """pm_layout_indices: global coefficient indices (blocked layout) or evaluation rows (cyclic) of a rank's local positions."""
    L = load_library()
    out = np.zeros(n // shard_count, dtype=np.uint64)
    st = L.pm_layout_indices(n, shard_count, shard_rank, int(coefficients), _p(out))
    if st:
        raise PolymathError(st, "pm_layout_indices")
    return out


class Comm:
    """pm_comm: one rank's end of the exchange layer."""

    def __init__(self, handle, keep=None):
        self.L, self.h, self._keep = load_library(), handle, keep

    @classmethod
    def local_group(cls, world, serialize=False):
        """`world` ranks as threads of this process (tests, emulation, single-process multi-GPU).  serialize: the ranks take turns
        between collectives (pm_comm_local_set_serialize: the per-rank emulation of N GPUs on one)."""
        L = load_library()
        arr = (ct.c_void_p * world)()
        st = L.pm_comm_local_create(world, arr)
        if st:
            raise PolymathError(st, "pm_comm_local_create")
        if serialize:
            st = L.pm_comm_local_set_serialize(ct.c_void_p(arr[0]), 1)
            if st:
                raise PolymathError(st, "pm_comm_local_set_serialize")
        return [cls(ct.c_void_p(arr[r])) for r in range(world)]

    @staticmethod
    def rccl_unique_id():
        L = load_library()
        buf = ct.create_string_buffer(128)
        st = L.pm_comm_rccl_unique_id(buf)
        if st:
            raise PolymathError(st, "pm_comm_rccl_unique_id (librccl not loadable?)")
        return buf.raw

    @classmethod
    def rccl(cls, unique_id, rank, world, device):
        """ncclCommInitRank over the 128-byte id rank 0 made with rccl_unique_id(); collective."""
        L = load_library()
        h = ct.c_void_p()
        st = L.pm_comm_rccl_create(ct.c_char_p(unique_id), rank, world, device, ct.byref(h))
        if st:
            raise PolymathError(st, "pm_comm_rccl_create")
        return cls(h)

    @property
    def rank(self):
        return self.L.pm_comm_rank(self.h)

    @property
    def world(self):
        return self.L.pm_comm_world(self.h)

    def busy_ms(self, reset=True):
        return float(self.L.pm_comm_busy_ms(self.h, int(reset)))

    @property
    def kind(self):
        return self.L.pm_comm_kind(self.h).decode()

    @property
    def failed(self):
        return bool(self.L.pm_comm_failed(self.h))

    def last_error(self):
        return self.L.pm_comm_last_error(self.h).decode()

    def set_timeout_ms(self, ms):
        st = self.L.pm_comm_set_timeout_ms(self.h, int(ms))
        if st:
            raise PolymathError(st, "pm_comm_set_timeout_ms")

    def abort(self, why="aborted by the host"):
        self.L.pm_comm_abort(self.h, why.encode())

    def all_gather(self, arr):
        arr = np.ascontiguousarray(arr)
        out = np.zeros((self.world,) + arr.shape, dtype=arr.dtype)
        st = self.L.pm_comm_all_gather(self.h, arr.ctypes.data_as(ct.c_void_p), out.ctypes.data_as(ct.c_void_p), arr.nbytes)
        if st:
            raise PolymathError(st, self.L.pm_comm_last_error(self.h).decode())
        return out

    def all_to_all_device(self, d_send, d_recv, bytes_per_peer, stream=None):
        """pm_comm_all_to_all on device pointers: block p of d_send goes to rank p, block r of d_recv comes from rank r
        (enqueued on `stream`; the caller synchronises)."""
        st = self.L.pm_comm_all_to_all(self.h, ct.c_void_p(d_send), ct.c_void_p(d_recv), bytes_per_peer, ct.c_void_p(stream or 0))
        if st:
            raise PolymathError(st, self.L.pm_comm_last_error(self.h).decode())

    def all_gather_device(self, d_send, d_recv, nbytes, stream=None):
        """pm_comm_all_gather_device on device pointers: rank r's `nbytes` land at d_recv + r * nbytes (the caller synchronises)."""
        st = self.L.pm_comm_all_gather_device(self.h, ct.c_void_p(d_send), ct.c_void_p(d_recv), nbytes, ct.c_void_p(stream or 0))
        if st:
            raise PolymathError(st, self.L.pm_comm_last_error(self.h).decode())

    def close(self):
        """pm_comm_destroy.  Detach it from its contexts first (Context.set_comm(None)): a context does not own its comm."""
        if getattr(self, "h", None):
            self.L.pm_comm_destroy(self.h)
            self.h = None


class Context:
    """pm_ctx: one HIP stream + workspaces on one GPU; one proof in flight."""

    def __init__(self, device=0):
        self.L = load_library()
        h = ct.c_void_p()
        st = self.L.pm_ctx_create(device, ct.byref(h))
        if st:
            raise PolymathError(st, "pm_ctx_create(device=%d)" % device)
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.L.pm_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, st):
        if st:
            raise PolymathError(st, self.L.pm_last_error(self.h).decode())

    def set_comm(self, comm):
        """Join this context to its rank's communicator (needed by PM_SHARD_VECTOR keys)."""
        self.comm = comm
        self.check(self.L.pm_ctx_set_comm(self.h, comm.h if comm is not None else None))

    def set_option(self, name, value):
        """pm_ctx_set_option: name in OPTIONS (or the pm_option number); "tables" also takes a TABLES_MODES name."""
        key = OPTIONS[name] if isinstance(name, str) else int(name)
        if key == OPTIONS["tables"] and isinstance(value, str):
            value = TABLES_MODES[value]
        self.check(self.L.pm_ctx_set_option(self.h, key, int(value)))

    def get_option(self, name):
        key = OPTIONS[name] if isinstance(name, str) else int(name)
        v = ct.c_longlong(0)
        self.check(self.L.pm_ctx_get_option(self.h, key, ct.byref(v)))
        return int(v.value)

    def selftest_field(self, products_per_field=4096, seed=1):
        """pm_selftest_field: the device's field products vs the host's CIOS -> mismatch counts {field: n} (all zero when healthy)."""
        bad = (ct.c_uint64 * 4)()
        self.check(self.L.pm_selftest_field(self.h, products_per_field, seed, bad))
        return dict(zip(("bls12_381_fr", "bn254_fr", "bls12_381_fq", "bn254_fq"), (int(v) for v in bad)))

    def timings(self):
        arr = (ct.c_double * len(TIMING_SLOTS))()
        self.L.pm_last_timings(self.h, arr, len(TIMING_SLOTS))
        return dict(zip(TIMING_SLOTS, list(arr)))

    # ---- standalone kernels
    def ntt(self, curve, data, log_n, inverse=False):
        data = _c(data).copy()
        self.check(self.L.pm_ntt(self.h, CURVE_IDS[curve], _p(data), log_n, int(inverse)))
        return data

    def ntt_device(self, curve, dptr, log_n, inverse=False):
        self.check(self.L.pm_ntt_device(self.h, CURVE_IDS[curve], ct.c_void_p(dptr), log_n, int(inverse)))

    def msm(self, curve, bases, scalars):
        cid = CURVE_IDS[curve]
        bases, scalars = _c(bases), _c(scalars)
        out = np.zeros(2 * FQ_LIMBS64[cid], dtype=np.uint64)
        inf = ct.c_int(0)
        self.check(self.L.pm_msm_g1(self.h, cid, bases.ctypes.data_as(ct.c_void_p), max(bases.strides[0], 16 * FQ_LIMBS64[cid]) if bases.ndim > 1 else 16 * FQ_LIMBS64[cid],
                                    _p(scalars), scalars.shape[0], _p(out), ct.byref(inf)))
        return out, inf.value

    def g1_sum(self, curve, pts, infs=None):
        return g1_sum(curve, pts, infs)


class Bases:
    """pm_bases: a G1 base vector resident in HBM."""

    def __init__(self, ctx, curve, handle):
        self.ctx, self.curve, self.cid, self.h = ctx, curve, CURVE_IDS[curve], handle

    @classmethod
    def upload(cls, ctx, curve, bases):
        bases = _c(bases)
        h = ct.c_void_p()
        ctx.check(ctx.L.pm_bases_upload(ctx.h, CURVE_IDS[curve], bases.ctypes.data_as(ct.c_void_p), max(bases.strides[0], 16 * FQ_LIMBS64[CURVE_IDS[curve]]),
                                        bases.shape[0], ct.byref(h)))
        return cls(ctx, curve, h)

    @classmethod
    def multiples(cls, ctx, curve, length):
        h = ct.c_void_p()
        ctx.check(ctx.L.pm_bases_generate_multiples(ctx.h, CURVE_IDS[curve], length, ct.byref(h)))
        return cls(ctx, curve, h)

    def precompute(self):
        """pm_bases_precompute: build the window tables (W x memory) for faster resident MSMs."""
        self.ctx.check(self.ctx.L.pm_bases_precompute(self.ctx.h, self.h))
        return self

    def __len__(self):
        return int(self.ctx.L.pm_bases_len(self.h))

    def download(self, offset=0, length=None):
        if length is None:
            length = len(self) - offset
        out = np.zeros((length, 2 * FQ_LIMBS64[self.cid]), dtype=np.uint64)
        self.ctx.check(self.ctx.L.pm_bases_download(self.ctx.h, self.h, offset, length, _p(out)))
        return out

    def msm(self, scalars, offset=0, length=None, device_ptr=None):
        """scalars: host np.uint64 [len,4], or device_ptr (int) + length for scalars already in HBM."""
        out = np.zeros(2 * FQ_LIMBS64[self.cid], dtype=np.uint64)
        inf = ct.c_int(0)
        if device_ptr is not None:
            st = self.ctx.L.pm_msm_g1_resident(self.ctx.h, self.h, offset, ct.c_void_p(device_ptr), 1, length, _p(out), ct.byref(inf))
        else:
            scalars = _c(scalars)
            length = scalars.shape[0] if length is None else length
            st = self.ctx.L.pm_msm_g1_resident(self.ctx.h, self.h, offset, scalars.ctypes.data_as(ct.c_void_p), 0, length,
                                               _p(out), ct.byref(inf))
        self.ctx.check(st)
        return out, inf.value

    def free(self):
        if self.h:
            self.ctx.L.pm_bases_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class CsrArrays:
    """numpy-backed pm_csr.  rows: list of rows of (value_montgomery_limbs_or_index...)."""

    def __init__(self, rowptr, col, val):
        self.rowptr = np.ascontiguousarray(rowptr, dtype=np.uint64)
        self.col = np.ascontiguousarray(col if len(col) else [0], dtype=np.uint32)
        self.val = _c(val) if len(val) else np.zeros((1, 4), dtype=np.uint64)
        self.struct = PmCsr(len(self.rowptr) - 1, _p(self.rowptr), self.col.ctypes.data_as(u32p), _p(self.val))


class ProvingKey:
    """pm_pk: R1CS matrices + the six base vectors resident on one GPU (optionally one shard)."""

    def __init__(self, ctx, curve, handle, csrs):
        self.ctx, self.curve, self.cid, self.h, self._csrs = ctx, curve, CURVE_IDS[curve], handle, csrs
        n, m0, sigma = ct.c_uint64(), ct.c_uint64(), ct.c_uint64()
        omega = np.zeros(4, dtype=np.uint64)
        lens = np.zeros(6, dtype=np.uint64)
        ctx.check(ctx.L.pm_pk_info(self.h, ct.byref(n), ct.byref(m0), ct.byref(sigma), _p(omega), _p(lens)))
        self.n, self.m0, self.sigma, self.omega_limbs = n.value, m0.value, sigma.value, omega
        self.base_lens = [int(v) for v in lens]
        self.nq = FQ_LIMBS64[self.cid]

    @classmethod
    def generate(cls, ctx, curve, m0, mw, nr, a, b, c, x_trapdoor, z_trapdoor, shard_rank=0, shard_count=1, layout="pairs"):
        """a, b, c: CsrArrays; trapdoors: np.uint64[4] Montgomery (generator.rs:72,77 draws).  layout: "pairs" (MSM pair
        ranges sharded, vector phases replicated) or "vector" (everything sharded; the context needs set_comm)."""
        h = ct.c_void_p()
        ctx.check(ctx.L.pm_pk_generate_sharded(ctx.h, CURVE_IDS[curve], m0, mw, nr, ct.byref(a.struct), ct.byref(b.struct),
                                               ct.byref(c.struct), _p(_c(x_trapdoor)), _p(_c(z_trapdoor)), shard_rank, shard_count,
                                               LAYOUTS[layout], ct.byref(h)))
        pk = cls(ctx, curve, h, (a, b, c))
        pk.shard_rank, pk.shard_count, pk.layout = shard_rank, shard_count, LAYOUTS[layout]
        return pk

    @classmethod
    def load(cls, ctx, curve, n, m0, mw, nr, sigma, a, b, c, base_arrays, shard_rank=0, shard_count=1, layout="pairs"):
        """base_arrays: six np.uint64 2-D arrays in pm_base_vec order."""
        arrs = [_c(x) for x in base_arrays]
        BA = (PmBaseArray * 6)()
        for k, x in enumerate(arrs):
            BA[k] = PmBaseArray(x.ctypes.data_as(ct.c_void_p), x.shape[0], x.strides[0])
        h = ct.c_void_p()
        ctx.check(ctx.L.pm_pk_load_sharded(ctx.h, CURVE_IDS[curve], n, m0, mw, nr, sigma, ct.byref(a.struct), ct.byref(b.struct),
                                           ct.byref(c.struct), BA, shard_rank, shard_count, LAYOUTS[layout], ct.byref(h)))
        pk = cls(ctx, curve, h, (a, b, c))
        pk.shard_rank, pk.shard_count, pk.layout = shard_rank, shard_count, LAYOUTS[layout]
        return pk

    def view(self, ctx):
        """The same resident key used from another context (a pm_pk is immutable and shareable; each context runs
        one proof at a time).  The view does not own the handle: free the original."""
        v = ProvingKey.__new__(ProvingKey)
        v.__dict__.update(self.__dict__)
        v.ctx, v._view = ctx, True
        return v

    def msm_plan(self, which):
        """(pairs, windows = bucket additions per pair, window bits, has tables) of merged MSM 0 = a, 1 = c, 2 = d."""
        pairs, win, bits, tb = ct.c_uint64(), ct.c_uint(), ct.c_uint(), ct.c_int()
        self.ctx.check(self.ctx.L.pm_pk_msm_plan(self.h, which, ct.byref(pairs), ct.byref(win), ct.byref(bits), ct.byref(tb)))
        return pairs.value, win.value, bits.value, bool(tb.value)

    def msm_pieces(self, which):
        """[(cat_lo, count), ...]: the resident pairs of merged MSM `which` as ranges of the logical base concatenation."""
        n = ct.c_size_t(0)
        self.ctx.check(self.ctx.L.pm_pk_msm_pieces(self.h, which, None, None, 0, ct.byref(n)))
        lo, cnt = np.zeros(n.value, dtype=np.uint64), np.zeros(n.value, dtype=np.uint64)
        self.ctx.check(self.ctx.L.pm_pk_msm_pieces(self.h, which, _p(lo), _p(cnt), n.value, ct.byref(n)))
        return [(int(a), int(b)) for a, b in zip(lo, cnt)]

    def msm_windows(self, which):
        return self.msm_plan(which)[1]

    def export_bases(self, which, offset=0, length=None):
        if length is None:
            length = self.base_lens[which] - offset
        out = np.zeros((length, 2 * self.nq), dtype=np.uint64)
        self.ctx.check(self.ctx.L.pm_pk_export_bases(self.ctx.h, self.h, which, offset, length, _p(out)))
        return out

    # --- the three prover phases: (status, outputs...) tuples, status codes of pm_status
    def phase1(self, x, w, r_a):
        a = np.zeros(2 * self.nq, dtype=np.uint64)
        c = np.zeros(2 * self.nq, dtype=np.uint64)
        ai, ci = ct.c_int(0), ct.c_int(0)
        w = _c(w) if len(w) else np.zeros((1, 4), dtype=np.uint64)
        rc = self.ctx.L.pm_prove_phase1(self.ctx.h, self.h, _p(_c(x)), _p(w), _p(_c(r_a)), _p(a), ct.byref(ai), _p(c), ct.byref(ci))
        return rc, a, ai.value, c, ci.value

    def phase1_device(self, d_x, d_w, r_a):
        """assignment already in HBM: d_x, d_w are device pointers (ints) to m0 / mw Montgomery Fr."""
        a = np.zeros(2 * self.nq, dtype=np.uint64)
        c = np.zeros(2 * self.nq, dtype=np.uint64)
        ai, ci = ct.c_int(0), ct.c_int(0)
        rc = self.ctx.L.pm_prove_phase1_device(self.ctx.h, self.h, ct.c_void_p(d_x), ct.c_void_p(d_w), _p(_c(r_a)), _p(a), ct.byref(ai),
                                               _p(c), ct.byref(ci))
        return rc, a, ai.value, c, ci.value

    TRANSCRIPT_IDS = {"merlin": 0, "keccak256": 1, "blake3": 2}

    def host_prove(self, transcript, instance_limbs, x, w, r_a, on_device=False, combine_many=None):
        """pm_host_prove / pm_host_prove_sharded: all three phases and the Fiat-Shamir glue in one native call.
        x, w: numpy limb arrays, or device pointers (ints) with on_device=True.  combine_many (sharded keys):
        [(xy, inf), ...] -> the same list summed over all ranks (polymath_amd.distributed.PointCombiner.many).
        -> (status, proof bytes)."""
        buf = ct.create_string_buffer(256)
        n = ct.c_size_t(0)
        words = 2 * self.nq
        failure = []

        def _cb(_user, count, xy, inf):
            try:
                pts = [(np.ctypeslib.as_array(xy, shape=(count * words,))[j * words:(j + 1) * words].copy(), int(inf[j])) for j in range(count)]
                for j, (sxy, sinf) in enumerate(combine_many(pts)):
                    for k, v in enumerate(np.asarray(sxy, dtype=np.uint64).tolist()):
                        xy[j * words + k] = v
                    inf[j] = int(sinf)
                return 0
            except Exception as e:          # never unwind through the C frames
                failure.append(e)
                return 8                    # PM_ERR_STATE
        cb = COMBINE_FN(_cb) if combine_many is not None else ct.cast(None, COMBINE_FN)
        if on_device:
            px, pw = ct.c_void_p(x), ct.c_void_p(w)
        else:
            x = _c(x)
            w = _c(w) if len(w) else np.zeros((1, 4), dtype=np.uint64)
            px, pw = x.ctypes.data_as(ct.c_void_p), w.ctypes.data_as(ct.c_void_p)
        rc = self.ctx.L.pm_host_prove_sharded(self.ctx.h, self.h, self.TRANSCRIPT_IDS[transcript], _p(_c(instance_limbs)), px, pw, int(on_device),
                                              _p(_c(r_a)), cb, None, buf, len(buf), ct.byref(n))
        if failure:
            raise failure[0]
        return rc, buf.raw[:n.value]

    def phase2(self, x1):
        out = np.zeros(4, dtype=np.uint64)
        rc = self.ctx.L.pm_prove_phase2(self.ctx.h, _p(_c(x1)), _p(out))
        return rc, out

    def phase3(self, x1, x2, a_at_x1, c_at_x1):
        d = np.zeros(2 * self.nq, dtype=np.uint64)
        di = ct.c_int(0)
        rc = self.ctx.L.pm_prove_phase3(self.ctx.h, _p(_c(x1)), _p(_c(x2)), _p(_c(a_at_x1)), _p(_c(c_at_x1)), _p(d), ct.byref(di))
        return rc, d, di.value

    def tap(self, which, max_elems):
        out = np.zeros((max_elems, 4), dtype=np.uint64)
        n = ct.c_size_t(0)
        self.ctx.check(self.ctx.L.pm_prove_tap(self.ctx.h, which, _p(out), max_elems, ct.byref(n)))
        return out[:min(n.value, max_elems)]

    def free(self):
        if self.h and not getattr(self, "_view", False):
            self.ctx.L.pm_pk_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
