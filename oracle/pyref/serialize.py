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
