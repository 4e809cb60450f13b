"""ORACLE (test infrastructure only) -- ark-serialize `serialize_compressed`
wire format as used by the reference's `to_bytes!` (src/macro.rs:7-12).
[ark, from memory -- SURVEY.md App. C; unverifiable in this image]

  Fr            : 32 B little-endian canonical integer
  &[T] / Vec<T> : u64-LE length prefix, then the elements
  BLS12-381 G1  : 48 B big-endian x; bit7 = compressed(1), bit6 = infinity,
                  bit5 = y is the lexicographically larger root   (zcash format)
  BN254 G1      : 32 B little-endian x; top byte bit7 = y > -y, bit6 = infinity
"""
import struct


def ser_fr(c, v):
    return int(v % c.r).to_bytes(32, "little")


def ser_fr_slice(c, vs):
    return struct.pack("<Q", len(vs)) + b"".join(ser_fr(c, v) for v in vs)


def ser_g1(c, P):
    if c.name == "bls12_381":
        if P is None:
            return bytes([0xC0]) + bytes(47)
        x, y = P
        b = bytearray(x.to_bytes(48, "big"))
        b[0] |= 0x80
        if y > (c.p - 1) // 2:
            b[0] |= 0x20
        return bytes(b)
    # generic short-Weierstrass flags (BN254)
    if P is None:
        b = bytearray(32)
        b[31] |= 0x40
        return bytes(b)
    x, y = P
    b = bytearray(x.to_bytes(32, "little"))
    if y > (c.p - y) % c.p:
        b[31] |= 0x80
    return bytes(b)


def ser_g1_slice(c, Ps):
    return struct.pack("<Q", len(Ps)) + b"".join(ser_g1(c, P) for P in Ps)


def ser_proof(c, proof):
    """Proof { a_g1, c_g1, a_at_x1, d_g1 } field order (src/data_structures.rs:10-19)."""
    return ser_g1(c, proof["a_g1"]) + ser_g1(c, proof["c_g1"]) + ser_fr(c, proof["a_at_x1"]) + ser_g1(c, proof["d_g1"])


# ---------------------------------------------------------------------------------------------
# ProvingKey / VerifyingKey wire format: the `#[derive(CanonicalSerialize)]` field order
# of src/data_structures.rs:25-73 and src/common.rs:112-127, compressed mode.  [ark, from memory]
#   usize / u64   : 8 B little-endian
#   (F, usize)    : Fr then u64
#   BLS12-381 G2  : 96 B = x.c1 (48 B big-endian) || x.c0, flags in the first byte as for G1; the sign bit
#                   compares y with -y in Fq2 order (c1 first, then c0)                      (zcash format)
# ---------------------------------------------------------------------------------------------
def _u64(v):
    return struct.pack("<Q", v)


def _fq2_gt(a, b):
    return (a[1], a[0]) > (b[1], b[0])


def ser_g2(c, Q):
    if c.name == "bn254":
        # ark-serialize's default short-Weierstrass compressed form: x.c0 || x.c1 little-endian, flags in the LAST byte
        # (0x80: y > -y in the Fq2 order, 0x40: infinity)                                     [ark, from memory]
        if Q is None:
            return bytes(63) + bytes([0x40])
        (x0, x1), (y0, y1) = Q
        b = bytearray(x0.to_bytes(32, "little") + x1.to_bytes(32, "little"))
        if _fq2_gt((y0, y1), ((-y0) % c.p, (-y1) % c.p)):
            b[63] |= 0x80
        return bytes(b)
    assert c.name == "bls12_381"
    if Q is None:
        return bytes([0xC0]) + bytes(95)
    (x0, x1), (y0, y1) = Q
    b = bytearray(x1.to_bytes(48, "big") + x0.to_bytes(48, "big"))
    b[0] |= 0x80
    if _fq2_gt((y0, y1), ((-y0) % c.p, (-y1) % c.p)):
        b[0] |= 0x20
    return bytes(b)


def _fq_sqrt(c, a):
    """p = 3 mod 4 on both curves' base fields."""
    assert c.p % 4 == 3
    s = pow(a, (c.p + 1) // 4, c.p)
    return s if s * s % c.p == a % c.p else None


def _fq2_sqrt(c, a):
    """Complex method for Fq2 = Fq[u]/(u^2+1), p = 3 mod 4."""
    p = c.p
    a0, a1 = a
    if a1 == 0:
        s = _fq_sqrt(c, a0)
        if s is not None:
            return (s, 0)
        s = _fq_sqrt(c, (-a0) % p)
        return None if s is None else (0, s)
    alpha = _fq_sqrt(c, (a0 * a0 + a1 * a1) % p)
    if alpha is None:
        return None
    inv2 = pow(2, -1, p)
    delta = (a0 + alpha) * inv2 % p
    x0 = _fq_sqrt(c, delta)
    if x0 is None:
        delta = (a0 - alpha) * inv2 % p
        x0 = _fq_sqrt(c, delta)
        if x0 is None:
            return None
    x1 = a1 * pow(2 * x0, -1, p) % p
    return (x0, x1)


def deser_g1(c, b):
    """Inverse of ser_g1 (decompression: y = sqrt(x^3 + b), root chosen by the sign flag)."""
    if c.name == "bls12_381":
        assert len(b) == 48 and b[0] & 0x80, "compressed flag"
        if b[0] & 0x40:
            return None
        x = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:], "big")
        larger = bool(b[0] & 0x20)
    else:
        assert len(b) == 32
        if b[31] & 0x40:
            return None
        x = int.from_bytes(b[:31] + bytes([b[31] & 0x3F]), "little")
        larger = bool(b[31] & 0x80)
    assert x < c.p
    y = _fq_sqrt(c, (x * x * x + c.b) % c.p)
    assert y is not None, "not on the curve"
    if (y > (c.p - 1) // 2) != larger:
        y = c.p - y
    return (x, y)


def deser_g2(c, b):
    assert c.name == "bls12_381" and len(b) == 96 and b[0] & 0x80
    if b[0] & 0x40:
        return None
    x1 = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:48], "big")
    x0 = int.from_bytes(b[48:], "big")
    larger = bool(b[0] & 0x20)
    p = c.p
    x = (x0, x1)
    xx = ((x0 * x0 - x1 * x1) % p, 2 * x0 * x1 % p)
    x3 = ((xx[0] * x0 - xx[1] * x1) % p, (xx[0] * x1 + xx[1] * x0) % p)
    y = _fq2_sqrt(c, ((x3[0] + 4) % p, (x3[1] + 4) % p))      # twist b' = 4(1 + u)
    assert y is not None, "not on the twist"
    if _fq2_gt(y, ((-y[0]) % p, (-y[1]) % p)) != larger:
        y = ((-y[0]) % p, (-y[1]) % p)
    return (x, y)


def ser_vk(c, vk):
    """VerifyingKey { e: PairingVK { one_g1, one_g2, x_g2, z_g2 }, n, m0, sigma, omega } (data_structures.rs:25-52)."""
    return (ser_g1(c, vk["one_g1"]) + ser_g2(c, vk["one_g2"]) + ser_g2(c, vk["x_g2"]) + ser_g2(c, vk["z_g2"]) +
            _u64(vk["n"]) + _u64(vk["m0"]) + _u64(vk["sigma"]) + ser_fr(c, vk["omega"]))


def ser_matrix(c, rows):
    out = [_u64(len(rows))]
    for row in rows:
        out.append(_u64(len(row)))
        for v, j in row:
            out.append(ser_fr(c, v) + _u64(j))
    return b"".join(out)


def ser_sap_matrices(c, q):
    """SAPMatrices { num_instance_variables, num_r1cs_witness_variables, num_r1cs_constraints, a, b, c } (common.rs:112-127)."""
    return _u64(q.m0) + _u64(q.mw) + _u64(q.nr) + ser_matrix(c, q.a) + ser_matrix(c, q.b) + ser_matrix(c, q.c)


PK_VECTORS = ("x_powers_g1", "x_powers_y_alpha_g1", "x_powers_zh_by_y_alpha_g1", "x_powers_y_gamma_g1",
              "x_powers_y_gamma_z_g1", "uj_wj_lcs_by_y_alpha_g1")      # data_structures.rs:60-72, declaration order


def ser_pk(c, pk, vk):
    """ProvingKey { vk, sap_matrices, six Vec<G1Affine> } (data_structures.rs:56-73)."""
    return ser_vk(c, vk) + ser_sap_matrices(c, pk.r1cs) + b"".join(ser_g1_slice(c, getattr(pk, name)) for name in PK_VECTORS)


class _Reader:
    def __init__(self, b):
        self.b, self.o = b, 0

    def take(self, n):
        assert self.o + n <= len(self.b), "truncated"
        v = self.b[self.o:self.o + n]
        self.o += n
        return v

    def u64(self):
        return struct.unpack("<Q", self.take(8))[0]


def deser_pk(c, data):
    """-> (vk dict, (m0, mw, nr, a, b, c), {vector name: [points]})"""
    rd = _Reader(data)
    g1n = 48 if c.name == "bls12_381" else 32
    vk = dict(one_g1=deser_g1(c, rd.take(g1n)), one_g2=deser_g2(c, rd.take(96)), x_g2=deser_g2(c, rd.take(96)),
              z_g2=deser_g2(c, rd.take(96)))
    vk["n"], vk["m0"], vk["sigma"] = rd.u64(), rd.u64(), rd.u64()
    vk["omega"] = int.from_bytes(rd.take(32), "little")
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
    vecs = {}
    for name in PK_VECTORS:
        vecs[name] = [deser_g1(c, rd.take(g1n)) for _ in range(rd.u64())]
    assert rd.o == len(data), "trailing bytes"
    return vk, (m0, mw, nr, mats[0], mats[1], mats[2]), vecs
