"""ORACLE (test infrastructure only) -- big-integer field / curve constants.

PARITY UNPINNED: the reference (sigma0-dev/polymath) ships no golden vectors and
cannot be built in this image (no cargo/rustc, arkworks not vendored).  These
constants are therefore pinned by mathematics only: every value below is
re-derived / self-checked in tests/test_oracle_pyref.py (primality-free checks:
generator on curve, r*G == O, root-of-unity order, Montgomery constants).

Nothing under oracle/ may be imported by the product (polymath_amd/).

Curves: BLS12-381 (the only curve the reference instantiates,
/root/reference/Cargo.toml:35) and BN254 (BASELINE.json configs[4]).
"""

# ---------------------------------------------------------------- BLS12-381
BLS12_381_P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
BLS12_381_R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
BLS12_381_B = 4
BLS12_381_G1 = (
    0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
    0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
)
# G2 generator (x = x0 + x1*u, y = y0 + y1*u), Fq2 = Fq[u]/(u^2+1); twist b' = 4(1+u)
BLS12_381_G2 = (
    (0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
     0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
    (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
     0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE),
)
BLS12_381_X = -0xD201000000010000  # curve parameter (negative)
BLS12_381_FR_GENERATOR = 7        # multiplicative generator of Fr (ark-bls12-381 FrConfig)
BLS12_381_FR_TWO_ADICITY = 32

# -------------------------------------------------------------------- BN254
BN254_P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
BN254_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
BN254_B = 3
BN254_G1 = (1, 2)
# G2 generator of ark-bn254 / EIP-197 (x = x0 + x1*i, y = y0 + y1*i), Fq2 = Fq[i]/(i^2+1); twist b' = 3/(9+i).
# Checked in tests/test_oracle_pyref.py: on the twist, r * G2 == O.
BN254_G2 = (
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
)
BN254_X = 4965661367192848881     # curve parameter: p = 36x^4 + 36x^3 + 24x^2 + 6x + 1, r = 36x^4 + 36x^3 + 18x^2 + 6x + 1
BN254_FR_GENERATOR = 5
BN254_FR_TWO_ADICITY = 28


class Curve:
    """Parameter bundle.  limbs64 = number of u64 Montgomery limbs arkworks uses."""

    def __init__(self, name, p, r, b, g1, fr_gen, two_adicity, fq_limbs64, fr_limbs64):
        self.name, self.p, self.r, self.b, self.g1 = name, p, r, b, g1
        self.fr_gen, self.two_adicity = fr_gen, two_adicity
        self.fq_limbs64, self.fr_limbs64 = fq_limbs64, fr_limbs64
        self.fq_R = pow(2, 64 * fq_limbs64, p)   # Montgomery radix mod p
        self.fr_R = pow(2, 64 * fr_limbs64, r)
        # 2^s-th primitive root of unity = g^((r-1)/2^s)   [ark-ff MontConfig derive]
        self.two_adic_root = pow(fr_gen, (r - 1) >> two_adicity, r)

    # --- Montgomery in/out (arkworks keeps field elements in Montgomery form in memory)
    def fr_to_mont(self, a):
        return a * self.fr_R % self.r

    def fr_from_mont(self, a):
        return a * pow(self.fr_R, -1, self.r) % self.r

    def fq_to_mont(self, a):
        return a * self.fq_R % self.p

    def fq_from_mont(self, a):
        return a * pow(self.fq_R, -1, self.p) % self.p

    def root_of_unity(self, n):
        """ark-poly Radix2EvaluationDomain::new: group_gen = two_adic_root^(2^(s-log n))."""
        log_n = n.bit_length() - 1
        assert 1 << log_n == n and log_n <= self.two_adicity
        return pow(self.two_adic_root, 1 << (self.two_adicity - log_n), self.r)


BLS12_381 = Curve("bls12_381", BLS12_381_P, BLS12_381_R, BLS12_381_B, BLS12_381_G1,
                  BLS12_381_FR_GENERATOR, BLS12_381_FR_TWO_ADICITY, 6, 4)
BN254 = Curve("bn254", BN254_P, BN254_R, BN254_B, BN254_G1,
              BN254_FR_GENERATOR, BN254_FR_TWO_ADICITY, 4, 4)

CURVES = {"bls12_381": BLS12_381, "bn254": BN254}
CURVE_IDS = {"bls12_381": 0, "bn254": 1}


# ------------------------------------------------------------- G1 arithmetic
# Affine points are (x, y) tuples of ints, None is the point at infinity.

def g1_is_on_curve(c, P):
    if P is None:
        return True
    x, y = P
    return (y * y - x * x * x - c.b) % c.p == 0


def g1_neg(c, P):
    if P is None:
        return None
    return (P[0], (-P[1]) % c.p)


def g1_add(c, P, Q):
    if P is None:
        return Q
    if Q is None:
        return P
    p = c.p
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % p == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, p) % p
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
    x3 = (lam * lam - x1 - x2) % p
    y3 = (lam * (x1 - x3) - y1) % p
    return (x3, y3)


def _jac_dbl(p, X, Y, Z):
    if Y == 0:
        return (1, 1, 0)
    A = X * X % p
    B = Y * Y % p
    C = B * B % p
    D = 2 * ((X + B) * (X + B) - A - C) % p
    E = 3 * A % p
    X3 = (E * E - 2 * D) % p
    Y3 = (E * (D - X3) - 8 * C) % p
    Z3 = 2 * Y * Z % p
    return (X3, Y3, Z3)


def _jac_add_affine(p, X1, Y1, Z1, x2, y2):
    if Z1 == 0:
        return (x2, y2, 1)
    Z1Z1 = Z1 * Z1 % p
    U2 = x2 * Z1Z1 % p
    S2 = y2 * Z1 * Z1Z1 % p
    H = (U2 - X1) % p
    r = (S2 - Y1) % p
    if H == 0:
        if r == 0:
            return _jac_dbl(p, X1, Y1, Z1)
        return (1, 1, 0)
    HH = H * H % p
    HHH = H * HH % p
    V = X1 * HH % p
    X3 = (r * r - HHH - 2 * V) % p
    Y3 = (r * (V - X3) - Y1 * HHH) % p
    Z3 = Z1 * H % p
    return (X3, Y3, Z3)


def g1_mul(c, P, k):
    """Scalar multiplication k*P (k reduced mod r), Jacobian double-and-add."""
    if P is None:
        return None
    k %= c.r
    if k == 0:
        return None
    p = c.p
    X, Y, Z = 1, 1, 0
    for bit in bin(k)[2:]:
        X, Y, Z = _jac_dbl(p, X, Y, Z)
        if bit == "1":
            X, Y, Z = _jac_add_affine(p, X, Y, Z, P[0], P[1])
    if Z == 0:
        return None
    zi = pow(Z, -1, p)
    zi2 = zi * zi % p
    return (X * zi2 % p, Y * zi2 * zi % p)


def g1_msm_naive(c, bases, scalars):
    """Definition of E::G1::msm_unchecked (prover.rs:380-384): zip to the shorter
    length, sum of scalar_i * base_i.  The result is a canonical group element."""
    acc = None
    for P, s in zip(bases, scalars):
        acc = g1_add(c, acc, g1_mul(c, P, s))
    return acc
