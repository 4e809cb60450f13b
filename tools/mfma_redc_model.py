#!/usr/bin/env python3
"""Big-integer model of the matrix-core Montgomery reduction priced in DESIGN.md §7.3 (round 4).

The reduction of f28_mul (fq28.cuh) multiplies by CONSTANTS (p and -p^-1): 196 + 14 of the 406 multiplier
operations of a BLS12-381 Fq product.  This model restates it as ONE constant matrix product that a wave's
matrix core can take for its 64 lanes at once:

    T = sum_k col_k 2^(28 k)            27 column sums of the 14 x 14 limb product, 64-bit each
    T 2^-392 = T_hi + sum_{k<14} sum_{j<8} byte_j(col_k) * C[k][j]      (mod p)
    C[k][j]  = 2^(28 k + 8 j - 392) mod p  (symmetric residue; a plain power of two once the exponent is >= 0)

The bytes of the UN-NORMALISED low columns are the B operand (signed: the columns start at the bias
0x0080808080808080 and are XOR-ed with it, so that byte - 128 is what the i8 core sees), the 7-bit signed
digits of C at positions 28 L + 7 d (limb L, d < 4) are the rows of the A operand, and three more rows hold
C / p on 21 fractional bits: their sum estimates the quotient, so that one multiple of p comes off and the
result is < 2p with tight limbs (the bound f28_mul's callers rely on).  No carry chain over the low half, no
second product.

Run: python tools/mfma_redc_model.py [--emit tools/mfma_redc_tables.h]
"""
import argparse
import random

P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
W, N = 28, 14
MASK = (1 << W) - 1
RBITS = W * N                     # 392
BIAS = 0x0080808080808080         # low seven bytes biased; the top byte stays below 128 (column < 2^63 - 2^55)
XORW = (0x80808080, 0x00808080)   # what turns the biased column words into signed bytes
NDIG = 4 * N                      # 56 digit rows
QFRAC = 21


def limbs(x, n=N):
    return [(x >> (W * i)) & MASK for i in range(n)]


def sym(x):
    x %= P
    return x - P if x > P // 2 else x


def const_of(k, j):
    e = W * k + 8 * j - RBITS
    return (1 << e) if e >= 0 else sym(pow(2, e, P))


def digits7(v, n=NDIG):
    """signed digits in [-64, 63] at 7-bit spacing: sum d_i 128^i == v"""
    out = []
    for _ in range(n):
        d = ((v + 64) % 128) - 64
        out.append(d)
        v = (v - d) // 128
    assert v == 0, "constant does not fit the digit rows"
    return out


def build_matrix():
    """rows[(k, j)] -> (56 digits, 3 quotient digits)"""
    rows = {}
    for k in range(N):
        for j in range(8):
            c = const_of(k, j)
            dg = digits7(c)
            phi = round(c * (1 << QFRAC) / P) if W * k + 8 * j < RBITS else round(c * (1 << QFRAC) / P)
            phi = max(-(1 << 20), min((1 << 20) - 1, phi))
            qd = digits7(phi, 3)
            rows[(k, j)] = (dg, qd)
    return rows


ROWS = build_matrix()


def sbyte(word64, j):
    u = (word64 >> (8 * j)) & 0xff
    return u - 256 if u >= 128 else u


def reduce_columns(cols):
    """cols: 27 non-negative column sums (low 14 WITHOUT the bias) -> 14 tight limbs, value < 2p, == T 2^-392 mod p"""
    out = [0] * NDIG
    qrow = [0, 0, 0]
    for k in range(N):
        assert cols[k] + BIAS < (1 << 63), "low column too large for the signed-byte form"
        w = (cols[k] + BIAS) ^ (XORW[0] | (XORW[1] << 32))
        rec = 0
        for j in range(8):
            s = sbyte(w, j)
            rec += s << (8 * j)
            dg, qd = ROWS[(k, j)]
            for m in range(NDIG):
                out[m] += s * dg[m]
            for m in range(3):
                qrow[m] += s * qd[m]
        assert rec == cols[k]
    assert all(abs(o) < (1 << 31) for o in out + qrow)
    t1 = qrow[2] + (qrow[1] >> 7) + (qrow[0] >> 14)     # ~ 128 r_lo / p, floor errors < 2
    q = (t1 - 1) >> 7
    c = 0
    res = []
    for k in range(N):
        lo = out[4 * k] + (out[4 * k + 1] << 7)
        hi = out[4 * k + 2] + (out[4 * k + 3] << 7)
        assert abs(lo) < (1 << 31) and abs(hi) < (1 << 31)
        f = (cols[N + k] if N + k < 2 * N - 1 else 0) + lo + (hi << 14) - q * ((P >> (W * k)) & MASK)
        assert abs(f) < (1 << 63)
        c += f
        res.append(c & MASK)
        c >>= W
    assert c == 0, "value outside [0, 2^392)"
    return res, q, max(abs(o) for o in out)


def product_columns(a, b):
    cols = [0] * (2 * N - 1)
    for i in range(N):
        for j in range(N):
            cols[i + j] += a[i] * b[j]
    return cols


def value(l):
    return sum(x << (W * i) for i, x in enumerate(l))


def check(a, b):
    cols = product_columns(a, b)
    res, q, mx = reduce_columns(cols)
    v = value(res)
    want = value(a) * value(b) * pow(2, -RBITS, P) % P
    assert v % P == want, "wrong residue"
    assert v < 2 * P, "not tight"
    return v / P, q, mx


def selftest(n=2000, seed=1):
    rnd = random.Random(seed)
    worst = 0.0
    mxo = 0
    qs = []
    # lazy operands: limbs up to 2^28 + K16-limb (the P and R of the mixed add), values up to 18p
    top = (1 << 28) + 0x1ffaaab0
    for it in range(n):
        kind = it % 5
        if kind == 0:
            a = limbs(rnd.randrange(2 * P)); b = limbs(rnd.randrange(2 * P))
        elif kind == 1:
            a = [rnd.randrange(top) for _ in range(N - 1)] + [rnd.randrange(18 * 0x1a012)]
            b = list(a)
        elif kind == 2:
            a = [top - 1] * (N - 1) + [18 * 0x1a011]; b = list(a)
        elif kind == 3:
            a = [0] * N; b = limbs(rnd.randrange(P))
        else:
            a = limbs(rnd.randrange(14 * P)); b = limbs(rnd.randrange(2 * P))
        vp, q, mx = check(a, b)
        worst = max(worst, vp)
        mxo = max(mxo, mx)
        qs.append(q)
    return worst, mxo, min(qs), max(qs)


def emit(path):
    """A operand image: [mb][t][lane][16 bytes], the layout tools/mfma_redc.hip reads with one ds_read_b128 per tile.
    lane = 32 h' + rho holds row 32 mb + rho, K slots (t, h', 0..15); row rho <-> (h = (rho >> 2) & 1, r = 4 (rho >> 3) + (rho & 3)),
    slot s = 16 mb + r: digit 28 h + s for s < 28, quotient rows for s = 28, 29, 30 (2^14, 2^7, 1 weights)."""
    img = []
    for mb in range(2):
        for t in range(4):
            for lane in range(64):
                hp, rho = lane >> 5, lane & 31
                h, r = (rho >> 2) & 1, 4 * (rho >> 3) + (rho & 3)
                s = 16 * mb + r
                for i in range(16):
                    k = 7 * hp + 2 * t + (i >> 3)
                    j = i & 7
                    v = 0
                    if k < 7 * hp + 7:
                        dg, qd = ROWS[(k, j)]
                        if s < 28:
                            v = dg[28 * h + s]
                        elif s < 31:
                            v = qd[2 - (s - 28)]
                    img.append(v & 0xff)
    with open(path, "w") as f:
        f.write("// generated by tools/mfma_redc_model.py --emit: A operand of the matrix-core reduction, [mb][t][lane][16]\n")
        f.write("static const unsigned char MFMA_REDC_A[%d] = {\n" % len(img))
        for i in range(0, len(img), 32):
            f.write("    " + ",".join("%d" % b for b in img[i:i + 32]) + ",\n")
        f.write("};\n")
        f.write("static const unsigned int MFMA_REDC_P[14] = {" + ",".join("0x%08xu" % l for l in limbs(P)) + "};\n")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--emit")
    ap.add_argument("--n", type=int, default=2000)
    a = ap.parse_args()
    worst, mxo, qmin, qmax = selftest(a.n)
    print("model ok: max value/p %.4f, max |row sum| 2^%.2f, q in [%d, %d]" % (worst, __import__("math").log2(mxo), qmin, qmax))
    if a.emit:
        emit(a.emit)
        print("wrote", a.emit)
