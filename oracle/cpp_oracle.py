"""ORACLE (test infrastructure only) -- ctypes loader for oracle/_build/libpolymath_oracle.so
(the C++ CPU restatement, oracle/cpp/oracle.cpp) plus int <-> Montgomery-limb helpers.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as ct
import os
import subprocess

import numpy as np

from .pyref.fields import CURVES, CURVE_IDS

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpolymath_oracle.so")
_lib = None

u64p = ct.POINTER(ct.c_uint64)
u32p = ct.POINTER(ct.c_uint32)
intp = ct.POINTER(ct.c_int)


class PoCsr(ct.Structure):
    _fields_ = [("nrows", ct.c_uint64), ("rowptr", u64p), ("col", u32p), ("val", u64p)]


def build(force=False):
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ct.CDLL(_SO)
        L.po_pk_generate.restype = ct.c_void_p
        L.po_pk_generate.argtypes = [ct.c_int, ct.c_uint64, ct.c_uint64, ct.c_uint64, ct.POINTER(PoCsr),
                                     ct.POINTER(PoCsr), ct.POINTER(PoCsr), u64p, u64p, ct.c_int, intp]
        for name in ("po_pk_free", "po_pk_info", "po_pk_export_bases", "po_pk_import_bases", "po_prove_phase1",
                     "po_prove_phase2", "po_prove_phase3", "po_prove_tap"):
            getattr(L, name).argtypes = None
        L.po_init()
        _lib = L
    return _lib


# ------------------------------------------------------------------ limb helpers
def ints_to_limbs(vals, nlimbs):
    """list of python ints -> np.uint64 array [len, nlimbs] little-endian limbs."""
    out = np.zeros((len(vals), nlimbs), dtype=np.uint64)
    mask = (1 << 64) - 1
    for i, v in enumerate(vals):
        for k in range(nlimbs):
            out[i, k] = (v >> (64 * k)) & mask
    return out


def limbs_to_ints(arr):
    arr = np.asarray(arr, dtype=np.uint64)
    arr = arr.reshape(-1, arr.shape[-1])
    return [sum(int(arr[i, k]) << (64 * k) for k in range(arr.shape[1])) for i in range(arr.shape[0])]


def fr_to_mont_limbs(curve, vals):
    c = CURVES[curve]
    return ints_to_limbs([c.fr_to_mont(v % c.r) for v in vals], c.fr_limbs64)


def fr_from_mont_limbs(curve, arr):
    c = CURVES[curve]
    return [c.fr_from_mont(v) for v in limbs_to_ints(arr)]


def g1_to_mont_limbs(curve, pts):
    """list of affine (x, y) or None -> np.uint64 [len, 2*fq_limbs] (infinity = all zero)."""
    c = CURVES[curve]
    nq = c.fq_limbs64
    flat = []
    for P in pts:
        if P is None:
            flat += [0, 0]
        else:
            flat += [c.fq_to_mont(P[0]), c.fq_to_mont(P[1])]
    return ints_to_limbs(flat, nq).reshape(len(pts), 2 * nq)


def g1_from_mont_limbs(curve, arr, infs=None):
    c = CURVES[curve]
    nq = c.fq_limbs64
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 2 * nq)
    out = []
    for i in range(arr.shape[0]):
        x, y = limbs_to_ints(arr[i].reshape(2, nq))
        if (infs is not None and infs[i]) or (x == 0 and y == 0):
            out.append(None)
        else:
            out.append((c.fq_from_mont(x), c.fq_from_mont(y)))
    return out


def _p(arr):
    return arr.ctypes.data_as(u64p)


class CsrBuf:
    """Keeps numpy buffers alive behind a PoCsr / pm_csr struct."""

    @classmethod
    def from_arrays(cls, rowptr, col, val, struct_cls=PoCsr):
        """Ready CSR arrays (rowptr u64, col u32, val [nnz,4] Montgomery limbs) -- no Python integers involved."""
        self = cls.__new__(cls)
        self.rowptr = np.ascontiguousarray(rowptr, dtype=np.uint64)
        self.col = np.ascontiguousarray(col, dtype=np.uint32)
        self.val = np.ascontiguousarray(val, dtype=np.uint64)
        self.struct = struct_cls(len(self.rowptr) - 1, _p(self.rowptr), self.col.ctypes.data_as(u32p), _p(self.val))
        return self

    def __init__(self, curve, rows, struct_cls=PoCsr):
        rowptr = [0]
        cols, vals = [], []
        for row in rows:
            for (v, j) in row:
                cols.append(j)
                vals.append(v)
            rowptr.append(len(cols))
        self.rowptr = np.array(rowptr, dtype=np.uint64)
        self.col = np.array(cols if cols else [0], dtype=np.uint32)
        self.val = fr_to_mont_limbs(curve, vals) if vals else np.zeros((1, 4), dtype=np.uint64)
        self.struct = struct_cls(len(rows), _p(self.rowptr), self.col.ctypes.data_as(u32p), _p(self.val))


# ------------------------------------------------------------------- thin API
def fr_op(curve, op, a, b=None):
    out = np.zeros(4, dtype=np.uint64)
    rc = lib().po_fr_op(CURVE_IDS[curve], op, _p(a), _p(b) if b is not None else None, _p(out))
    assert rc == 0
    return out


def fq_op(curve, op, a, b=None):
    n = CURVES[curve].fq_limbs64
    out = np.zeros(n, dtype=np.uint64)
    rc = lib().po_fq_op(CURVE_IDS[curve], op, _p(a), _p(b) if b is not None else None, _p(out))
    assert rc == 0
    return out


def ntt(curve, data, log_n, inverse, nthreads=1):
    data = np.ascontiguousarray(data, dtype=np.uint64).copy()
    rc = lib().po_ntt(CURVE_IDS[curve], _p(data), ct.c_uint(log_n), int(inverse), nthreads)
    assert rc == 0, rc
    return data


def msm(curve, bases, scalars, nthreads=1):
    """bases np.uint64 [len, 2*nq] (or wider rows with a stride), scalars [len,4] Montgomery."""
    nq = CURVES[curve].fq_limbs64
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    out = np.zeros(2 * nq, dtype=np.uint64)
    inf = ct.c_int(0)
    rc = lib().po_msm_g1(CURVE_IDS[curve], bases.ctypes.data_as(ct.c_void_p), ct.c_size_t(bases.strides[0]),
                         _p(scalars), ct.c_size_t(len(scalars)), nthreads, _p(out), ct.byref(inf))
    assert rc == 0
    return out, inf.value


def g1_multiples(curve, length):
    nq = CURVES[curve].fq_limbs64
    out = np.zeros((length, 2 * nq), dtype=np.uint64)
    assert lib().po_g1_multiples(CURVE_IDS[curve], ct.c_size_t(length), _p(out)) == 0
    return out


def g1_mul(curve, base_xy, scalar):
    nq = CURVES[curve].fq_limbs64
    out = np.zeros(2 * nq, dtype=np.uint64)
    inf = ct.c_int(0)
    assert lib().po_g1_mul(CURVE_IDS[curve], _p(np.ascontiguousarray(base_xy)), _p(np.ascontiguousarray(scalar)),
                           _p(out), ct.byref(inf)) == 0
    return out, inf.value


def g1_sum(curve, pts, infs=None):
    nq = CURVES[curve].fq_limbs64
    pts = np.ascontiguousarray(pts, dtype=np.uint64).reshape(-1, 2 * nq)
    out = np.zeros(2 * nq, dtype=np.uint64)
    inf = ct.c_int(0)
    infs_arr = None
    if infs is not None:
        infs_arr = np.ascontiguousarray(infs, dtype=np.int32)
    assert lib().po_g1_sum(CURVE_IDS[curve], _p(pts), infs_arr.ctypes.data_as(intp) if infs_arr is not None else None,
                           ct.c_size_t(len(pts)), _p(out), ct.byref(inf)) == 0
    return out, inf.value


def g1_is_on_curve(curve, xy):
    return bool(lib().po_g1_is_on_curve(CURVE_IDS[curve], _p(np.ascontiguousarray(xy, dtype=np.uint64))))


class OraclePk:
    """po_pk handle: setup (generator.rs) + the three prove phases (prover.rs)."""
    BASE_NAMES = ["x_powers_g1", "x_powers_y_alpha_g1", "x_powers_y_gamma_g1", "x_powers_y_gamma_z_g1",
                  "x_powers_zh_by_y_alpha_g1", "uj_wj_lcs_by_y_alpha_g1"]

    def __init__(self, curve, r1cs, x_trapdoor=None, z_trapdoor=None, nthreads=1):
        self.curve, self.cid, self.nthreads = curve, CURVE_IDS[curve], nthreads
        self.nq = CURVES[curve].fq_limbs64
        if hasattr(r1cs, "csr_arrays"):       # (rowptr, col, val limbs) x 3, e.g. from the library's native circuit generator
            self._bufs = [CsrBuf.from_arrays(*t) for t in r1cs.csr_arrays]
        else:
            self._bufs = [CsrBuf(curve, r1cs.a), CsrBuf(curve, r1cs.b), CsrBuf(curve, r1cs.c)]
        st = ct.c_int(0)
        xt = fr_to_mont_limbs(curve, [x_trapdoor]) if x_trapdoor is not None else None
        zt = fr_to_mont_limbs(curve, [z_trapdoor]) if z_trapdoor is not None else None
        self.h = ct.c_void_p(lib().po_pk_generate(
            self.cid, r1cs.m0, r1cs.mw, r1cs.nr, ct.byref(self._bufs[0].struct), ct.byref(self._bufs[1].struct),
            ct.byref(self._bufs[2].struct), _p(xt) if xt is not None else None, _p(zt) if zt is not None else None,
            nthreads, ct.byref(st)))
        if not self.h:
            raise RuntimeError("po_pk_generate failed: status %d" % st.value)
        n, m0, sigma = ct.c_uint64(), ct.c_uint64(), ct.c_uint64()
        omega = np.zeros(4, dtype=np.uint64)
        lens = np.zeros(6, dtype=np.uint64)
        lib().po_pk_info(self.h, ct.byref(n), ct.byref(m0), ct.byref(sigma), _p(omega), _p(lens))
        self.n, self.m0, self.sigma, self.omega_limbs = n.value, m0.value, sigma.value, omega
        self.base_lens = [int(v) for v in lens]

    def __del__(self):
        if getattr(self, "h", None):
            lib().po_pk_free(self.h)
            self.h = None

    def export_bases(self, which, offset=0, length=None):
        if length is None:
            length = self.base_lens[which] - offset
        out = np.zeros((length, 2 * self.nq), dtype=np.uint64)
        rc = lib().po_pk_export_bases(self.h, which, ct.c_size_t(offset), ct.c_size_t(length), _p(out))
        assert rc == 0
        return out

    def import_bases(self, which, arr):
        arr = np.ascontiguousarray(arr, dtype=np.uint64)
        rc = lib().po_pk_import_bases(self.h, which, arr.ctypes.data_as(ct.c_void_p), ct.c_size_t(arr.strides[0]),
                                      ct.c_size_t(len(arr)))
        assert rc == 0
        self.base_lens[which] = len(arr)

    def phase1(self, x, w, r_a):
        """x, w, r_a: np.uint64 Montgomery limb arrays.  Returns (rc, a_xy, a_inf, c_xy, c_inf)."""
        a = np.zeros(2 * self.nq, dtype=np.uint64)
        c = np.zeros(2 * self.nq, dtype=np.uint64)
        ai, ci = ct.c_int(0), ct.c_int(0)
        rc = lib().po_prove_phase1(self.h, _p(x), _p(w), _p(r_a), self.nthreads, _p(a), ct.byref(ai), _p(c), ct.byref(ci))
        return rc, a, ai.value, c, ci.value

    def phase2(self, x1):
        out = np.zeros(4, dtype=np.uint64)
        rc = lib().po_prove_phase2(self.h, _p(x1), _p(out))
        return rc, out

    def phase3(self, x1, x2, a_at_x1, c_at_x1):
        d = np.zeros(2 * self.nq, dtype=np.uint64)
        di = ct.c_int(0)
        rc = lib().po_prove_phase3(self.h, _p(x1), _p(x2), _p(a_at_x1), _p(c_at_x1), self.nthreads, _p(d), ct.byref(di))
        return rc, d, di.value

    def tap(self, which, max_elems):
        out = np.zeros((max_elems, 4), dtype=np.uint64)
        n = ct.c_size_t(0)
        rc = lib().po_prove_tap(self.h, which, _p(out), ct.c_size_t(max_elems), ct.byref(n))
        assert rc == 0
        return out[:min(n.value, max_elems)]
