"""ORACLE (test infrastructure only) -- ate pairings over big integers for BLS12-381 and BN254,
used solely as the accept/reject check of /root/reference/src/verifier.rs:50-61
(`E::multi_pairing(..).is_one()`).

Any non-degenerate bilinear map gives the same accept/reject answer for a
product-equals-one check, so this uses the simplest correct construction per curve:
Fq12 = Fq[w]/(w^12 - A w^6 + B), G2 points untwisted into E(Fq12), generic affine line
functions, final exponentiation by plain powering with (p^12 - 1)/r.
  BLS12-381: w^12 = 2 w^6 - 2  (u = w^6 - 1),  M-type twist (x / w^2, y / w^3),  Miller loop over |x|.
  BN254    : w^12 = 18 w^6 - 82 (i = w^6 - 9), D-type twist (x w^2, y w^3), optimal ate: Miller loop
             over 6x + 2 followed by the two Frobenius line steps (Q1 = pi(Q), -Q2 = -pi^2(Q)).
Bilinearity and non-degeneracy are unit-tested (tests/test_oracle_pyref.py).
Slow (seconds per check) -- fine for O(1) verifier work.

The module-level names (FQ12, twist, miller_loop, g2_add, g2_mul, pairing_check, make_vk, ...) are the
BLS12-381 engine's, as before; ENGINES["bn254"] carries the same API for BN254
(BASELINE.json configs[4]; the reference itself instantiates Bls12_381 only, Cargo.toml:35).
"""
from . import fields as _F


def _deg(p):
    d = len(p) - 1
    while d and p[d] == 0:
        d -= 1
    return d


class Engine:
    """One pairing-friendly curve: field tower, twist, G2 group law, Miller loop, verifier check."""

    def __init__(self, curve, g2_gen, mod_a, mod_b, shift, twist_mul, twist_b, loop_count, frobenius_steps):
        self.C, self.P, self.g2_gen = curve, curve.p, g2_gen
        self.mod_a, self.mod_b = mod_a, mod_b            # w^12 = mod_a w^6 - mod_b
        self.shift = shift                               # Fq2 unit u (u^2 = -1) = w^6 - shift
        self.twist_mul = twist_mul                       # D-type: multiply by w^2 / w^3; M-type: divide
        self.twist_b = twist_b                           # b' of the twist curve y^2 = x^3 + b' over Fq2
        self.loop_count, self.frobenius_steps = loop_count, frobenius_steps
        eng = self
        P = self.P

        class FQ12:
            __slots__ = ("c",)

            def __init__(self, coeffs):
                self.c = [x % P for x in coeffs]

            @staticmethod
            def one():
                return FQ12([1] + [0] * 11)

            @staticmethod
            def zero():
                return FQ12([0] * 12)

            @staticmethod
            def scalar(v):
                return FQ12([v] + [0] * 11)

            def __add__(self, o):
                return FQ12([a + b for a, b in zip(self.c, o.c)])

            def __sub__(self, o):
                return FQ12([a - b for a, b in zip(self.c, o.c)])

            def __neg__(self):
                return FQ12([-a for a in self.c])

            def __eq__(self, o):
                return self.c == o.c

            def is_zero(self):
                return all(a == 0 for a in self.c)

            def __mul__(self, o):
                if isinstance(o, int):
                    return FQ12([a * o for a in self.c])
                b = [0] * 23
                for i, x in enumerate(self.c):
                    if x:
                        for j, y in enumerate(o.c):
                            b[i + j] += x * y
                for k in range(22, 11, -1):  # reduce: w^k = A w^(k-6) - B w^(k-12)
                    top = b[k]
                    if top:
                        b[k - 6] += eng.mod_a * top
                        b[k - 12] -= eng.mod_b * top
                return FQ12(b[:12])

            def __pow__(self, e):
                out, base = FQ12.one(), self
                while e:
                    if e & 1:
                        out = out * base
                    base = base * base
                    e >>= 1
                return out

            def inv(self):
                # extended Euclid over Fq[w]
                lm, hm = [1] + [0] * 12, [0] * 13
                low, high = self.c + [0], [c % P for c in eng._mod_full()]
                while _deg(low):
                    r = eng._poly_rounded_div(high, low)
                    r += [0] * (13 - len(r))
                    nm, new = list(hm), list(high)
                    for i in range(13):
                        for j in range(13 - i):
                            nm[i + j] -= lm[i] * r[j]
                            new[i + j] -= low[i] * r[j]
                    nm = [x % P for x in nm]
                    new = [x % P for x in new]
                    lm, low, hm, high = nm, new, lm, low
                li = pow(low[0], -1, P)
                return FQ12([x * li for x in lm[:12]])

            def __truediv__(self, o):
                return self * o.inv()

        self.FQ12 = FQ12
        self.W = FQ12([0, 1] + [0] * 10)

    # ------------------------------------------------------------------ Fq12 helpers
    def _mod_full(self):
        return [self.mod_b, 0, 0, 0, 0, 0, -self.mod_a, 0, 0, 0, 0, 0, 1]     # w^12 - A w^6 + B

    def _poly_rounded_div(self, a, b):
        P = self.P
        dega, degb = _deg(a), _deg(b)
        temp = list(a)
        o = [0] * len(a)
        binv = pow(b[degb], -1, P)
        for i in range(dega - degb, -1, -1):
            q = temp[degb + i] * binv % P
            o[i] = (o[i] + q) % P
            for cidx in range(degb + 1):
                temp[cidx + i] = (temp[cidx + i] - q * b[cidx]) % P
        return o[:_deg(o) + 1]

    def twist(self, Q):
        """E'(Fq2) -> E(Fq12).  Q = ((x0,x1),(y0,y1)); the Fq2 unit is w^6 - shift."""
        (x0, x1), (y0, y1) = Q
        FQ12, W, s = self.FQ12, self.W, self.shift
        nx = FQ12([x0 - s * x1] + [0] * 5 + [x1] + [0] * 5)
        ny = FQ12([y0 - s * y1] + [0] * 5 + [y1] + [0] * 5)
        if self.twist_mul:
            return (nx * (W * W), ny * (W * W * W))
        return (nx / (W * W), ny / (W * W * W))

    def cast_g1(self, Pt):
        return (self.FQ12.scalar(Pt[0]), self.FQ12.scalar(Pt[1]))

    @staticmethod
    def _dbl(pt):
        x, y = pt
        m = (x * x * 3) / (y * 2)
        nx = m * m - x * 2
        return (nx, m * (x - nx) - y)

    def _add(self, p1, p2):
        if p1 is None:
            return p2
        if p2 is None:
            return p1
        x1, y1 = p1
        x2, y2 = p2
        if x1 == x2:
            if y1 == y2:
                return self._dbl(p1)
            return None
        m = (y2 - y1) / (x2 - x1)
        nx = m * m - x1 - x2
        return (nx, m * (x1 - nx) - y1)

    @staticmethod
    def _line(p1, p2, t):
        x1, y1 = p1
        x2, y2 = p2
        xt, yt = t
        if not (x1 == x2):
            m = (y2 - y1) / (x2 - x1)
            return m * (xt - x1) - (yt - y1)
        if y1 == y2:
            m = (x1 * x1 * 3) / (y1 * 2)
            return m * (xt - x1) - (yt - y1)
        return xt - x1

    def miller_loop(self, Q12, P12):
        """f_{T,Q}(P) without the final exponentiation (T = |x| on BLS12, 6x + 2 with the two Frobenius
        steps of the optimal ate pairing on BN)."""
        if Q12 is None or P12 is None:
            return self.FQ12.one()
        R, f = Q12, self.FQ12.one()
        T = self.loop_count
        for i in range(T.bit_length() - 2, -1, -1):
            f = f * f * self._line(R, R, P12)
            R = self._dbl(R)
            if T >> i & 1:
                f = f * self._line(R, Q12, P12)
                R = self._add(R, Q12)
        if self.frobenius_steps:
            p = self.P
            Q1 = (Q12[0] ** p, Q12[1] ** p)
            nQ2 = (Q1[0] ** p, -(Q1[1] ** p))
            f = f * self._line(R, Q1, P12)
            R = self._add(R, Q1)
            f = f * self._line(R, nQ2, P12)
        return f

    def final_exponentiation(self, f):
        return f ** ((self.P ** 12 - 1) // self.C.r)

    def pairing_product_is_one(self, pairs):
        """pairs = [(G1 affine or None, G2 affine or None)].  True iff prod e(P_i,Q_i) == 1."""
        f = self.FQ12.one()
        for (Pt, Q) in pairs:
            if Pt is None or Q is None:
                continue
            f = f * self.miller_loop(self.twist(Q), self.cast_g1(Pt))
        return self.final_exponentiation(f) == self.FQ12.one()

    # ------------------------------------------------------------- G2 (Fq2) ops
    def _f2mul(self, a, b):
        P = self.P
        return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)

    def _f2inv(self, a):
        P = self.P
        d = pow(a[0] * a[0] + a[1] * a[1], -1, P)
        return (a[0] * d % P, (-a[1]) * d % P)

    def _f2sub(self, a, b):
        return ((a[0] - b[0]) % self.P, (a[1] - b[1]) % self.P)

    def _f2add(self, a, b):
        return ((a[0] + b[0]) % self.P, (a[1] + b[1]) % self.P)

    def g2_add(self, A, B):
        if A is None:
            return B
        if B is None:
            return A
        (x1, y1), (x2, y2) = A, B
        m2, inv, sub, add = self._f2mul, self._f2inv, self._f2sub, self._f2add
        if x1 == x2:
            if add(y1, y2) == (0, 0):
                return None
            m = m2(m2((3, 0), m2(x1, x1)), inv(m2((2, 0), y1)))
        else:
            m = m2(sub(y2, y1), inv(sub(x2, x1)))
        x3 = sub(sub(m2(m, m), x1), x2)
        y3 = sub(m2(m, sub(x1, x3)), y1)
        return (x3, y3)

    def g2_neg(self, A):
        if A is None:
            return None
        return (A[0], ((-A[1][0]) % self.P, (-A[1][1]) % self.P))

    def g2_mul(self, A, k):
        k %= self.C.r
        out = None
        for bit in bin(k)[2:] if k else "":
            out = self.g2_add(out, out)
            if bit == "1":
                out = self.g2_add(out, A)
        return out

    def g2_is_on_curve(self, A):
        if A is None:
            return True
        x, y = A
        lhs = self._f2mul(y, y)
        rhs = self._f2add(self._f2mul(self._f2mul(x, x), x), self.twist_b)
        return lhs == rhs

    def make_vk_from_trapdoors(self, n, m0, sigma, omega, x, z):
        """PairingVK + VerifyingKey (src/generator.rs:139-157) from the two trapdoors."""
        return dict(n=n, m0=m0, sigma=sigma, omega=omega, one_g1=self.C.g1, one_g2=self.g2_gen,
                    x_g2=self.g2_mul(self.g2_gen, x), z_g2=self.g2_mul(self.g2_gen, z))

    def make_vk(self, pk):
        """... from a pyref ProvingKey (which remembers its trapdoors)."""
        x, z = pk.trapdoor
        return self.make_vk_from_trapdoors(pk.n, pk.m0, pk.sigma, pk.omega, x, z)

    def pairing_check(self, lhs_g1, neg_d_g1, x1, vk):
        """src/verifier.rs:48-61: e(lhs,[z]_2) * e(-d, [x]_2 - x1 [1]_2) == 1."""
        x_minus_x1 = self.g2_add(vk["x_g2"], self.g2_neg(self.g2_mul(vk["one_g2"], x1)))
        return self.pairing_product_is_one([(lhs_g1, vk["z_g2"]), (neg_d_g1, x_minus_x1)])


BLS12_381_ENGINE = Engine(_F.BLS12_381, _F.BLS12_381_G2, mod_a=2, mod_b=2, shift=1, twist_mul=False, twist_b=(4, 4),
                          loop_count=0xD201000000010000, frobenius_steps=False)        # |x|, b' = 4 (1 + u)
# BN254: x = 4965661367192848881, optimal-ate loop count 6x + 2; twist y^2 = x^3 + 3 / (9 + i)
_BN_XI_INV = (lambda p: (9 * pow(82, -1, p) % p, (-pow(82, -1, p)) % p))(_F.BN254_P)     # 1 / (9 + i) = (9 - i) / 82
BN254_ENGINE = Engine(_F.BN254, _F.BN254_G2, mod_a=18, mod_b=82, shift=9, twist_mul=True,
                      twist_b=(3 * _BN_XI_INV[0] % _F.BN254_P, 3 * _BN_XI_INV[1] % _F.BN254_P),
                      loop_count=6 * _F.BN254_X + 2, frobenius_steps=True)
ENGINES = {"bls12_381": BLS12_381_ENGINE, "bn254": BN254_ENGINE}

# ---- module-level BLS12-381 API (unchanged names)
C = _F.BLS12_381
P = C.p
ATE_LOOP_COUNT = BLS12_381_ENGINE.loop_count
FQ12 = BLS12_381_ENGINE.FQ12
W = BLS12_381_ENGINE.W
twist = BLS12_381_ENGINE.twist
cast_g1 = BLS12_381_ENGINE.cast_g1
miller_loop = BLS12_381_ENGINE.miller_loop
final_exponentiation = BLS12_381_ENGINE.final_exponentiation
pairing_product_is_one = BLS12_381_ENGINE.pairing_product_is_one
g2_add = BLS12_381_ENGINE.g2_add
g2_neg = BLS12_381_ENGINE.g2_neg
g2_mul = BLS12_381_ENGINE.g2_mul
g2_is_on_curve = BLS12_381_ENGINE.g2_is_on_curve
make_vk = BLS12_381_ENGINE.make_vk
pairing_check = BLS12_381_ENGINE.pairing_check
