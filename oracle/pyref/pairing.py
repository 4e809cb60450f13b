"""ORACLE (test infrastructure only) -- BLS12-381 ate pairing over big integers,
used solely as the accept/reject check of /root/reference/src/verifier.rs:50-61
(`E::multi_pairing(..).is_one()`).

Any non-degenerate bilinear map gives the same accept/reject answer for a
product-equals-one check, so this uses the simplest correct construction:
Fq12 = Fq[w]/(w^12 - 2 w^6 + 2), G2 points untwisted into E(Fq12), generic line
functions, final exponentiation by plain powering.  Bilinearity is unit-tested.
Slow (seconds per check) -- fine for O(1) verifier work.
"""
from .fields import BLS12_381 as C

P = C.p
_MOD_COEFFS = [2, 0, 0, 0, 0, 0, -2, 0, 0, 0, 0, 0]  # w^12 = 2 w^6 - 2
ATE_LOOP_COUNT = 0xD201000000010000                   # |x|


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
        for k in range(22, 11, -1):  # reduce: w^k = 2 w^(k-6) - 2 w^(k-12)
            top = b[k]
            if top:
                b[k - 6] += 2 * top
                b[k - 12] -= 2 * top
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
        def deg(p):
            d = len(p) - 1
            while d and p[d] == 0:
                d -= 1
            return d

        lm, hm = [1] + [0] * 12, [0] * 13
        low, high = self.c + [0], [c % P for c in _mod_full()]
        while deg(low):
            r = _poly_rounded_div(high, low)
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


def _mod_full():
    # w^12 - 2 w^6 + 2
    return [2, 0, 0, 0, 0, 0, -2, 0, 0, 0, 0, 0, 1]


def _poly_rounded_div(a, b):
    def deg(p):
        d = len(p) - 1
        while d and p[d] == 0:
            d -= 1
        return d

    dega, degb = deg(a), deg(b)
    temp = list(a)
    o = [0] * len(a)
    binv = pow(b[degb], -1, P)
    for i in range(dega - degb, -1, -1):
        q = temp[degb + i] * binv % P
        o[i] = (o[i] + q) % P
        for cidx in range(degb + 1):
            temp[cidx + i] = (temp[cidx + i] - q * b[cidx]) % P
    return o[:deg(o) + 1]


W = FQ12([0, 1] + [0] * 10)


def twist(Q):
    """E'(Fq2) -> E(Fq12).  Q = ((x0,x1),(y0,y1)), u = w^6 - 1 (since w^12-2w^6+2=0
    means (w^6-1)^2 = -1)."""
    (x0, x1), (y0, y1) = Q
    nx = FQ12([x0 - x1] + [0] * 5 + [x1] + [0] * 5)
    ny = FQ12([y0 - y1] + [0] * 5 + [y1] + [0] * 5)
    return (nx / (W * W), ny / (W * W * W))


def cast_g1(Pt):
    return (FQ12.scalar(Pt[0]), FQ12.scalar(Pt[1]))


def _dbl(pt):
    x, y = pt
    m = (x * x * 3) / (y * 2)
    nx = m * m - x * 2
    return (nx, m * (x - nx) - y)


def _add(p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if y1 == y2:
            return _dbl(p1)
        return None
    m = (y2 - y1) / (x2 - x1)
    nx = m * m - x1 - x2
    return (nx, m * (x1 - nx) - y1)


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


def miller_loop(Q12, P12):
    """f_{|x|,Q}(P) without final exponentiation."""
    if Q12 is None or P12 is None:
        return FQ12.one()
    R, f = Q12, FQ12.one()
    for i in range(ATE_LOOP_COUNT.bit_length() - 2, -1, -1):
        f = f * f * _line(R, R, P12)
        R = _dbl(R)
        if ATE_LOOP_COUNT >> i & 1:
            f = f * _line(R, Q12, P12)
            R = _add(R, Q12)
    return f


def final_exponentiation(f):
    return f ** ((P ** 12 - 1) // C.r)


def pairing_product_is_one(pairs):
    """pairs = [(G1 affine or None, G2 affine or None)].  True iff prod e(P_i,Q_i) == 1."""
    f = FQ12.one()
    for (Pt, Q) in pairs:
        if Pt is None or Q is None:
            continue
        f = f * miller_loop(twist(Q), cast_g1(Pt))
    return final_exponentiation(f) == FQ12.one()


# ----------------------------------------------------------------- G2 (Fq2) ops
def _f2mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def _f2inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], -1, P)
    return (a[0] * d % P, (-a[1]) * d % P)


def _f2sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def _f2add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def g2_add(A, B):
    if A is None:
        return B
    if B is None:
        return A
    (x1, y1), (x2, y2) = A, B
    if x1 == x2:
        if _f2add(y1, y2) == (0, 0):
            return None
        m = _f2mul(_f2mul((3, 0), _f2mul(x1, x1)), _f2inv(_f2mul((2, 0), y1)))
    else:
        m = _f2mul(_f2sub(y2, y1), _f2inv(_f2sub(x2, x1)))
    x3 = _f2sub(_f2sub(_f2mul(m, m), x1), x2)
    y3 = _f2sub(_f2mul(m, _f2sub(x1, x3)), y1)
    return (x3, y3)


def g2_neg(A):
    if A is None:
        return None
    return (A[0], ((-A[1][0]) % P, (-A[1][1]) % P))


def g2_mul(A, k):
    k %= C.r
    out = None
    for bit in bin(k)[2:] if k else "":
        out = g2_add(out, out)
        if bit == "1":
            out = g2_add(out, A)
    return out


def g2_is_on_curve(A):
    if A is None:
        return True
    x, y = A
    lhs = _f2mul(y, y)
    rhs = _f2add(_f2mul(_f2mul(x, x), x), (4, 4))  # b' = 4(1+u)
    return lhs == rhs


def make_vk(pk):
    """PairingVK + VerifyingKey (src/generator.rs:139-157) from a pyref ProvingKey."""
    from .fields import BLS12_381_G2
    x, z = pk.trapdoor
    return dict(n=pk.n, m0=pk.m0, sigma=pk.sigma, omega=pk.omega, one_g1=pk.curve.g1,
                one_g2=BLS12_381_G2, x_g2=g2_mul(BLS12_381_G2, x), z_g2=g2_mul(BLS12_381_G2, z))


def pairing_check(lhs_g1, neg_d_g1, x1, vk):
    """src/verifier.rs:48-61: e(lhs,[z]_2) * e(-d, [x]_2 - x1 [1]_2) == 1."""
    x_minus_x1 = g2_add(vk["x_g2"], g2_neg(g2_mul(vk["one_g2"], x1)))
    return pairing_product_is_one([(lhs_g1, vk["z_g2"]), (neg_d_g1, x_minus_x1)])
