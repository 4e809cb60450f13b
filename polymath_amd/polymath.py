"""Host-side mirror of the reference's SNARK facade for the prover path
(/root/reference/src/lib.rs:44-98):

    Polymath(curve, transcript).setup(circuit, x, z) -> ProvingKey      (generator.rs:24-167)
    Polymath(...).prove(pk, circuit/assignment, r_a) -> Proof           (prover.rs:27-237)

Circuits are anything with `.generate_constraints(cs)` filling a ConstraintSystem (the
ark-relations ConstraintSynthesizer shape, tests/dummy.rs:25-35), or a ready (R1CS, instance,
witness) triple.  Everything O(n) or larger runs on the GPU behind the C ABI; this file only does
the O(m0) scalar glue of common.rs:21-98 and the two Fiat-Shamir calls.  RNG stays with the
caller: trapdoors and r_a are arguments (generator.rs:72,77; prover.rs:110).
"""
import struct

import numpy as np

from . import api
from .transcript import TRANSCRIPTS

MINUS_ALPHA, MINUS_GAMMA = 3, 5        # common.rs:11,14
B_POLYMATH = b"polymath"               # common.rs:8

FIELDS = {
    "bls12_381": dict(r=0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
                      p=0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB,
                      fq_limbs=6),
    "bn254": dict(r=21888242871839275222246405745257275088548364400416034343698204186575808495617,
                  p=21888242871839275222246405745257275088696311157297823662689037894645226208583, fq_limbs=4),
}
_M64 = (1 << 64) - 1


class PolymathProverError(RuntimeError):
    """The reference panics on these (assert!/unwrap, prover.rs:107,108,221,222,381); the C ABI
    returns a status and this mirror raises."""

    def __init__(self, phase, status):
        super().__init__("prove phase %d: %s" % (phase, api.STATUS.get(status, status)))
        self.phase, self.status = phase, status


# ------------------------------------------------------------------ limb <-> int (Montgomery)
def _to_limbs(vals, nl):
    if not len(vals):
        return np.zeros((0, nl), dtype=np.uint64)
    nb = 8 * nl
    return np.frombuffer(b"".join(int(v).to_bytes(nb, "little") for v in vals), dtype=np.uint64).reshape(-1, nl).copy()


def _from_limbs(row):
    return sum(int(row[k]) << (64 * k) for k in range(len(row)))


class Field:
    def __init__(self, curve):
        f = FIELDS[curve]
        self.curve, self.r, self.p, self.nq = curve, f["r"], f["p"], f["fq_limbs"]
        self.Rr, self.Rq = pow(2, 256, self.r), pow(2, 64 * self.nq, self.p)
        self.Rr_inv, self.Rq_inv = pow(self.Rr, -1, self.r), pow(self.Rq, -1, self.p)

    def fr_limbs(self, vals):
        return _to_limbs([(v % self.r) * self.Rr % self.r for v in vals], 4)

    def fr_int(self, limbs):
        return _from_limbs(limbs) * self.Rr_inv % self.r

    def g1_affine(self, xy, inf):
        """x||y Montgomery limbs -> (x, y) canonical ints or None."""
        if inf:
            return None
        return (_from_limbs(xy[:self.nq]) * self.Rq_inv % self.p, _from_limbs(xy[self.nq:]) * self.Rq_inv % self.p)


# ------------------------------------------------------------------------- constraint system
class ConstraintSystem:
    """Minimal ark-relations ConstraintSystem: variables, assignments and A/B/C rows of
    (coefficient, column).  Column 0 = One, then instance variables, then witness variables
    (ConstraintMatrices layout, generator.rs:46-54)."""
    ONE = ("one", 0)

    def __init__(self, r):
        self.r = r
        self.instance, self.witness = [1], []
        self.rows = []   # (a_lc, b_lc, c_lc) with lc = [(coeff, var)]

    def new_input_variable(self, value):
        self.instance.append(value % self.r)
        return ("inst", len(self.instance) - 1)

    def new_witness_variable(self, value):
        self.witness.append(value % self.r)
        return ("wit", len(self.witness) - 1)

    def enforce_constraint(self, a, b, c):
        self.rows.append((list(a), list(b), list(c)))

    def _col(self, var):
        kind, idx = var
        return idx if kind in ("one", "inst") else len(self.instance) + idx

    def to_r1cs(self):
        conv = lambda lc: [(coef % self.r, self._col(v)) for coef, v in lc]
        return R1CS(len(self.instance), len(self.witness), [conv(a) for a, _, _ in self.rows],
                    [conv(b) for _, b, _ in self.rows], [conv(c) for _, _, c in self.rows])


class R1CS:
    def __init__(self, m0, mw, a, b, c):
        assert len(a) == len(b) == len(c)
        self.m0, self.mw, self.nr, self.a, self.b, self.c = m0, mw, len(a), a, b, c


class LimbCircuit:
    """An R1CS plus assignment held as limb arrays (api.CsrArrays + Montgomery numpy arrays) instead of Python
    integers: what circuits.synthetic_r1cs_native returns -- the 2^22 / 2^24-gate configurations never pass through
    Python big integers.  Accepted by Polymath.setup / prove_native / prove_limbs wherever a circuit is."""

    def __init__(self, field, m0, mw, nr, csrs, inst_limbs, wit_limbs):
        self.field, self.m0, self.mw, self.nr, self.csrs = field, m0, mw, nr, csrs
        self.inst_limbs, self.wit_limbs = inst_limbs, wit_limbs

    @property
    def instance(self):
        return [self.field.fr_int(row) for row in self.inst_limbs]


def _csr(field, rows):
    rowptr, cols, vals = [0], [], []
    for row in rows:
        for v, j in row:
            cols.append(j)
            vals.append(v)
        rowptr.append(len(cols))
    return api.CsrArrays(rowptr, cols, field.fr_limbs(vals) if vals else [])


# ------------------------------------------------------------------------------ wire format
def ser_fr(field, v):
    return int(v % field.r).to_bytes(32, "little")


def ser_fr_slice(field, vs):
    return struct.pack("<Q", len(vs)) + b"".join(ser_fr(field, v) for v in vs)


def ser_g1(field, P):
    """ark-serialize compressed G1 (macro.rs:7-12 -> serialize_compressed)."""
    if field.curve == "bls12_381":      # 48 B big-endian x, flags: 0x80 compressed, 0x40 infinity, 0x20 y > -y
        if P is None:
            return bytes([0xC0]) + bytes(47)
        b = bytearray(P[0].to_bytes(48, "big"))
        b[0] |= 0x80 | (0x20 if P[1] > (field.p - 1) // 2 else 0)
        return bytes(b)
    if P is None:                        # BN254: 32 B little-endian x, top byte 0x80 y > -y, 0x40 infinity
        return bytes(31) + bytes([0x40])
    b = bytearray(P[0].to_bytes(32, "little"))
    if P[1] > field.p - P[1]:
        b[31] |= 0x80
    return bytes(b)


def ser_g1_slice(field, Ps):
    return struct.pack("<Q", len(Ps)) + b"".join(ser_g1(field, P) for P in Ps)


# ProvingKey / VerifyingKey wire format (data_structures.rs:25-73, common.rs:112-127): the
# derive(CanonicalSerialize) field order, compressed mode.  The key's base vectors come back from HBM through
# pm_pk_export_bases; loading decompresses the points on the host and goes through pm_pk_load.
PK_WIRE_VECTORS = (0, 1, 4, 2, 3, 5)   # pm_base_vec ids in declaration order: x_powers, y_alpha, zh_by_y_alpha, y_gamma, y_gamma_z, lcs
FQ2_B = (4, 4)                          # BLS12-381 twist: y^2 = x^3 + 4(1 + u)


def _u64(v):
    return struct.pack("<Q", v)


def _fq2_gt(a, b):
    return (a[1], a[0]) > (b[1], b[0])


def ser_g2(field, Q):
    """BLS12-381 G2, zcash format: x.c1 || x.c0 big-endian, flags as for G1, sign = y > -y in Fq2 order."""
    assert field.curve == "bls12_381"
    if Q is None:
        return bytes([0xC0]) + bytes(95)
    (x0, x1), (y0, y1) = Q
    b = bytearray(x1.to_bytes(48, "big") + x0.to_bytes(48, "big"))
    b[0] |= 0x80 | (0x20 if _fq2_gt((y0, y1), ((-y0) % field.p, (-y1) % field.p)) else 0)
    return bytes(b)


def _fq_sqrt(p, a):
    s = pow(a, (p + 1) // 4, p)            # p = 3 mod 4 (BLS12-381 and BN254 base fields)
    return s if s * s % p == a % p else None


def _fq2_sqrt(p, a):
    a0, a1 = a
    if a1 == 0:
        s = _fq_sqrt(p, a0)
        if s is not None:
            return (s, 0)
        s = _fq_sqrt(p, (-a0) % p)
        return None if s is None else (0, s)
    alpha = _fq_sqrt(p, (a0 * a0 + a1 * a1) % p)
    if alpha is None:
        return None
    inv2 = pow(2, -1, p)
    x0 = _fq_sqrt(p, (a0 + alpha) * inv2 % p)
    if x0 is None:
        x0 = _fq_sqrt(p, (a0 - alpha) * inv2 % p)
        if x0 is None:
            return None
    return (x0, a1 * pow(2 * x0, -1, p) % p)


def deser_g1(field, b):
    p = field.p
    if field.curve == "bls12_381":
        if len(b) != 48 or not b[0] & 0x80:
            raise ValueError("G1: not a compressed BLS12-381 point")
        if b[0] & 0x40:
            return None
        x, larger, curve_b = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:], "big"), bool(b[0] & 0x20), 4
    else:
        if len(b) != 32:
            raise ValueError("G1: wrong length")
        if b[31] & 0x40:
            return None
        x, larger, curve_b = int.from_bytes(b[:31] + bytes([b[31] & 0x3F]), "little"), bool(b[31] & 0x80), 3
    y = _fq_sqrt(p, (x * x * x + curve_b) % p) if x < p else None
    if y is None:
        raise ValueError("G1: not on the curve")
    return (x, p - y if (y > (p - 1) // 2) != larger else y)


def deser_g2(field, b):
    p = field.p
    if field.curve != "bls12_381" or len(b) != 96 or not b[0] & 0x80:
        raise ValueError("G2: not a compressed BLS12-381 point")
    if b[0] & 0x40:
        return None
    x1, x0 = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:48], "big"), int.from_bytes(b[48:], "big")
    xx = ((x0 * x0 - x1 * x1) % p, 2 * x0 * x1 % p)
    x3 = ((xx[0] * x0 - xx[1] * x1) % p, (xx[0] * x1 + xx[1] * x0) % p)
    y = _fq2_sqrt(p, ((x3[0] + FQ2_B[0]) % p, (x3[1] + FQ2_B[1]) % p))
    if y is None:
        raise ValueError("G2: not on the twist")
    neg = ((-y[0]) % p, (-y[1]) % p)
    return ((x0, x1), neg if _fq2_gt(y, neg) != bool(b[0] & 0x20) else y)


def ser_matrix(field, rows):
    out = [_u64(len(rows))]
    for row in rows:
        out.append(_u64(len(row)))
        out.extend(ser_fr(field, v) + _u64(j) for v, j in row)
    return b"".join(out)


class VerifyingKey:
    """data_structures.rs:38-52 with PairingVK (:25-35); G2 points as ((x0, x1), (y0, y1)) canonical ints."""

    def __init__(self, field, one_g1, one_g2, x_g2, z_g2, n, m0, sigma, omega):
        self.field, self.one_g1, self.one_g2, self.x_g2, self.z_g2 = field, one_g1, one_g2, x_g2, z_g2
        self.n, self.m0, self.sigma, self.omega = n, m0, sigma, omega

    def to_bytes(self):
        f = self.field
        return (ser_g1(f, self.one_g1) + ser_g2(f, self.one_g2) + ser_g2(f, self.x_g2) + ser_g2(f, self.z_g2) +
                _u64(self.n) + _u64(self.m0) + _u64(self.sigma) + ser_fr(f, self.omega))

    @classmethod
    def read(cls, field, rd):
        g1n = 48 if field.curve == "bls12_381" else 32
        one_g1, one_g2, x_g2, z_g2 = deser_g1(field, rd.take(g1n)), deser_g2(field, rd.take(96)), deser_g2(field, rd.take(96)), deser_g2(field, rd.take(96))
        n, m0, sigma = rd.u64(), rd.u64(), rd.u64()
        return cls(field, one_g1, one_g2, x_g2, z_g2, n, m0, sigma, int.from_bytes(rd.take(32), "little"))


class _Reader:
    def __init__(self, b):
        self.b, self.o = bytes(b), 0

    def take(self, n):
        if self.o + n > len(self.b):
            raise ValueError("truncated key")
        v = self.b[self.o:self.o + n]
        self.o += n
        return v

    def u64(self):
        return struct.unpack("<Q", self.take(8))[0]


class Proof:
    """data_structures.rs:10-19."""

    def __init__(self, field, a_g1, c_g1, a_at_x1, d_g1):
        self.field, self.a_g1, self.c_g1, self.a_at_x1, self.d_g1 = field, a_g1, c_g1, a_at_x1, d_g1

    def to_bytes(self):
        f = self.field
        return ser_g1(f, self.a_g1) + ser_g1(f, self.c_g1) + ser_fr(f, self.a_at_x1) + ser_g1(f, self.d_g1)

    def as_dict(self):
        return dict(a_g1=self.a_g1, c_g1=self.c_g1, a_at_x1=self.a_at_x1, d_g1=self.d_g1)


# ------------------------------------------------------------------------------------ facade
class Polymath:
    """Polymath<E, T> (lib.rs:44-50): curve = pairing engine, transcript = Fiat-Shamir choice."""

    def __init__(self, curve="bls12_381", transcript="merlin", device=0, ctx=None):
        self.curve, self.field = curve, Field(curve)
        self.transcript_cls = TRANSCRIPTS[transcript] if isinstance(transcript, str) else transcript
        self.transcript_name = transcript if isinstance(transcript, str) else None
        self.ctx = ctx if ctx is not None else api.Context(device)
        self.collect_timings, self.phase_timings = False, []    # per-phase stage timings (pm_last_timings), opt-in

    # circuit_specific_setup (lib.rs:63-70) -> generate_proving_key (generator.rs:24-167)
    def setup(self, circuit, x_trapdoor, z_trapdoor=None, shard_rank=0, shard_count=1, layout="pairs"):
        """setup(circuit, rng): the reference's signature (lib.rs:63-70) -- the two trapdoors are drawn from `rng` exactly as
        generate_proving_key does (generator.rs:72,77: sample_element_outside_domain twice, x then z; polymath_amd.rng).
        setup(circuit, x, z): the draws supplied by the caller."""
        f = self.field
        if z_trapdoor is None and hasattr(x_trapdoor, "next_u64"):
            from . import rng as RNG
            rng = x_trapdoor
            shape = circuit if isinstance(circuit, LimbCircuit) else self._synthesize(circuit)[0]
            n = 1
            while n < 2 * (shape.m0 + shape.nr):
                n <<= 1
            x_trapdoor = RNG.sample_element_outside_domain(rng, f.r, n)
            z_trapdoor = RNG.sample_element_outside_domain(rng, f.r, n)
            self.last_trapdoors = (x_trapdoor, z_trapdoor)
        if isinstance(circuit, LimbCircuit):
            r1cs, (A, B, C) = circuit, circuit.csrs
        else:
            r1cs = self._synthesize(circuit)[0]
            A, B, C = _csr(f, r1cs.a), _csr(f, r1cs.b), _csr(f, r1cs.c)
        pk = api.ProvingKey.generate(self.ctx, self.curve, r1cs.m0, r1cs.mw, r1cs.nr, A, B, C,
                                     f.fr_limbs([x_trapdoor])[0], f.fr_limbs([z_trapdoor])[0], shard_rank, shard_count, layout)
        pk.omega = f.fr_int(pk.omega_limbs)
        return pk

    def _synthesize(self, circuit):
        if isinstance(circuit, tuple):
            return circuit
        cs = ConstraintSystem(self.field.r)
        circuit.generate_constraints(cs)                  # prover.rs:44 / generator.rs:37
        return cs.to_r1cs(), cs.instance, cs.witness

    # prove (lib.rs:72-78) -> create_proof (prover.rs:27-64) -> create_proof_with_assignment (:66-237)
    def prove(self, pk, circuit, r_a, combine=None):
        """prove(pk, circuit, rng): the reference's signature (lib.rs:72-78): r_a = two F::rand(rng) draws, constant term first
        (prover.rs:110).  prove(pk, circuit, [r0, r1]): the draws supplied by the caller.
        `combine(xy, inf) -> (xy, inf)` merges per-shard partial points across ranks (RCCL
        all-gather + pm_g1_sum, polymath_amd.distributed); None for a whole key."""
        if hasattr(r_a, "next_u64"):
            from . import rng as RNG
            r_a = [RNG.fr_rand(r_a, self.field.r), RNG.fr_rand(r_a, self.field.r)]
        if isinstance(circuit, LimbCircuit):
            return self.prove_limbs(pk, circuit.instance, circuit.inst_limbs, circuit.wit_limbs, r_a, combine)
        _, instance, witness = self._synthesize(circuit)
        f = self.field
        x = f.fr_limbs(instance)
        w = f.fr_limbs(witness)
        return self.prove_limbs(pk, instance, x, w, r_a, combine)

    def prove_native(self, pk, x_limbs, w_limbs, r_a, device_ptrs=None, combine=None):
        """create_proof_with_assignment in ONE native call (pm_host_prove: the C++ host mirror inside the library
        runs the transcript and the challenge arithmetic between the phases), the three transcripts of the
        reference.  A sharded key needs `combine` with a .many([(xy, inf), ...]) method (PointCombiner): it is called
        back between the phases with this rank's partial points.  -> Proof::serialize_compressed bytes."""
        if self.transcript_name is None:
            raise ValueError("prove_native needs one of the reference's transcripts: " + ", ".join(TRANSCRIPTS))
        f = self.field
        many = combine.many if combine is not None else None
        if device_ptrs is not None:
            rc, data = pk.host_prove(self.transcript_name, x_limbs, device_ptrs[0], device_ptrs[1], f.fr_limbs(r_a), on_device=True, combine_many=many)
        else:
            rc, data = pk.host_prove(self.transcript_name, x_limbs, x_limbs, w_limbs, f.fr_limbs(r_a), combine_many=many)
        if rc:
            raise PolymathProverError(0, rc)
        return data

    def prove_limbs(self, pk, instance, x_limbs, w_limbs, r_a, combine=None, device_ptrs=None):
        """device_ptrs = (d_x, d_w): the assignment is already resident in HBM (pm_prove_phase1_device)."""
        f, r = self.field, self.field.r
        if device_ptrs is not None:
            rc, a_xy, a_inf, c_xy, c_inf = pk.phase1_device(device_ptrs[0], device_ptrs[1], f.fr_limbs(r_a))
        else:
            rc, a_xy, a_inf, c_xy, c_inf = pk.phase1(x_limbs, w_limbs, f.fr_limbs(r_a))
        if rc:
            raise PolymathProverError(1, rc)
        if self.collect_timings:
            self.phase_timings = [self.ctx.timings()]
        if combine:
            if hasattr(combine, "many"):                       # one exchange for both phase-1 points
                (a_xy, a_inf), (c_xy, c_inf) = combine.many([(a_xy, a_inf), (c_xy, c_inf)])
            else:
                a_xy, a_inf = combine(a_xy, a_inf)
                c_xy, c_inf = combine(c_xy, c_inf)
        a_g1, c_g1 = f.g1_affine(a_xy, a_inf), f.g1_affine(c_xy, c_inf)
        t = self.transcript_cls(B_POLYMATH, r)                                   # prover.rs:125
        x1 = self.compute_x1(t, instance, [a_g1, c_g1])                          # :126
        y1 = pow(x1, pk.sigma, r)                                                # :128
        y1_inv = pow(y1, -1, r)
        y1_alpha = pow(y1_inv, MINUS_ALPHA, r)                                   # :130
        rc, u_at = pk.phase2(f.fr_limbs([x1])[0])
        if rc:
            raise PolymathProverError(2, rc)
        if self.collect_timings:
            self.phase_timings.append(self.ctx.timings())
        a_at_x1 = (f.fr_int(u_at) + (r_a[0] + r_a[1] * x1) * y1_alpha) % r       # :132
        y1_gamma = pow(y1_inv, MINUS_GAMMA, r)                                   # :134
        pi_at_x1 = self.compute_pi_at_x1(pk.n, pk.omega, instance, x1, y1_gamma)  # :135
        c_at_x1 = ((a_at_x1 + y1_gamma) * a_at_x1 - pi_at_x1) % r * pow(y1_alpha, -1, r) % r   # :138, common.rs:73-75
        x2 = self.compute_x2(t, x1, [a_at_x1, c_at_x1])                          # :189
        L = lambda v: f.fr_limbs([v])[0]
        rc, d_xy, d_inf = pk.phase3(L(x1), L(x2), L(a_at_x1), L(c_at_x1))
        if rc:
            raise PolymathProverError(3, rc)
        if self.collect_timings:
            self.phase_timings.append(self.ctx.timings())
        if combine:
            d_xy, d_inf = combine(d_xy, d_inf)
        return Proof(f, a_g1, c_g1, a_at_x1, f.g1_affine(d_xy, d_inf))           # :231-236

    # ---- verify (lib.rs:80-90) and the verifying key (generator.rs:139-157): host code of the library, both pairing engines
    def make_vk(self, pk, x_trapdoor, z_trapdoor):
        """VerifyingKey::serialize_compressed bytes of the key made from the two trapdoors."""
        f = self.field
        return api.make_vk(self.curve, pk.n, pk.m0, pk.sigma, pk.omega_limbs, f.fr_limbs([x_trapdoor])[0], f.fr_limbs([z_trapdoor])[0])

    def verify(self, vk_bytes, public_inputs, proof):
        """Polymath::verify(vk, public_inputs, proof): public_inputs WITHOUT the leading one (verifier.rs:26); proof: a Proof or its
        serialize_compressed bytes.  Runs the library's own CPU verifier (pairing.hpp)."""
        if self.transcript_name is None:
            raise ValueError("verify needs one of the reference's transcripts: " + ", ".join(TRANSCRIPTS))
        pb = proof.to_bytes() if isinstance(proof, Proof) else bytes(proof)
        return api.verify(self.curve, self.transcript_name, vk_bytes, self.field.fr_limbs(list(public_inputs)), pb)

    # ---- ProvingKey wire format (§8 f-4)
    def pk_to_bytes(self, pk, r1cs, vk):
        """ProvingKey { vk, sap_matrices, six Vec<G1Affine> } (data_structures.rs:56-73).  `r1cs` carries the
        matrices as synthesised (the device copy has row-duplicate columns dropped, common.rs:100-105)."""
        f = self.field
        out = [vk.to_bytes(), _u64(r1cs.m0), _u64(r1cs.mw), _u64(r1cs.nr), ser_matrix(f, r1cs.a), ser_matrix(f, r1cs.b),
               ser_matrix(f, r1cs.c)]
        for which in PK_WIRE_VECTORS:
            xy = pk.export_bases(which)
            inf = ~xy.any(axis=1)
            out.append(ser_g1_slice(f, [f.g1_affine(xy[i], bool(inf[i])) for i in range(xy.shape[0])]))
        return b"".join(out)

    def pk_from_bytes(self, data, shard_rank=0, shard_count=1):
        """-> (device-resident ProvingKey through pm_pk_load, VerifyingKey, R1CS)."""
        f, rd = self.field, _Reader(data)
        vk = VerifyingKey.read(f, rd)
        m0, mw, nr = rd.u64(), rd.u64(), rd.u64()
        mats = []
        for _ in range(3):
            rows = []
            for _ in range(rd.u64()):
                row = []
                for _ in range(rd.u64()):
                    v = int.from_bytes(rd.take(32), "little")
                    row.append((v, rd.u64()))
                rows.append(row)
            mats.append(rows)
        g1n = 48 if f.curve == "bls12_381" else 32
        arrays = [None] * 6
        for which in PK_WIRE_VECTORS:
            pts = [deser_g1(f, rd.take(g1n)) for _ in range(rd.u64())]
            arr = np.zeros((len(pts), 2 * f.nq), dtype=np.uint64)
            live = [i for i, P in enumerate(pts) if P is not None]
            if live:
                arr[live, :f.nq] = _to_limbs([pts[i][0] * f.Rq % f.p for i in live], f.nq)
                arr[live, f.nq:] = _to_limbs([pts[i][1] * f.Rq % f.p for i in live], f.nq)
            arrays[which] = arr
        if rd.o != len(rd.b):
            raise ValueError("trailing bytes after the key")
        r1cs = R1CS(m0, mw, mats[0], mats[1], mats[2])
        A, B, C = _csr(f, r1cs.a), _csr(f, r1cs.b), _csr(f, r1cs.c)
        pk = api.ProvingKey.load(self.ctx, self.curve, vk.n, m0, mw, nr, vk.sigma, A, B, C, arrays, shard_rank, shard_count)
        pk.omega = vk.omega
        return pk, vk, r1cs

    # ---- common.rs:21-71
    def compute_x1(self, t, public_inputs, commitments):
        t.append_message(b"public_inputs", ser_fr_slice(self.field, public_inputs))
        t.append_message(b"commitments", ser_g1_slice(self.field, commitments))
        return t.challenge(b"x1")

    def compute_x2(self, t, x1, values):
        t.append_message(b"x1", ser_fr(self.field, x1))
        t.append_message(b"values", ser_fr_slice(self.field, values))
        return t.challenge(b"x2")

    def compute_pi_at_x1(self, n, omega, public_inputs, x1, y1_gamma):
        r = self.field.r
        m0 = len(public_inputs)
        num = (pow(x1, n, r) - 1) * pow(n, -1, r) % r
        w_i, s = 1, 0
        for i in range(2 * m0):
            if i == 0:
                zt = 2
            elif i < m0:
                zt = 1 + public_inputs[i]
            elif i == m0:
                zt = 0
            else:
                zt = 1 - public_inputs[i - m0]
            s = (s + zt * num * pow((x1 - w_i) % r, -1, r)) % r
            num = num * omega % r
            w_i = w_i * omega % r
        return s * y1_gamma % r
