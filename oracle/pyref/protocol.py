"""ORACLE (test infrastructure only) -- literal big-integer restatement of the
Polymath protocol as the reference implements it, DENSE like the reference, so it
is only usable on tiny circuits.  It exists to pin the *semantics* that the fast
sparse paths (oracle/cpp, polymath_amd/csrc) must reproduce.

PARITY UNPINNED (see fields.py header): no reference golden vectors exist; this
file is checked by (i) the SAP identity (Uz)^2 == Wz, (ii) the two `rem == 0`
asserts of prover.rs:108,221, (iii) pairing-verifier acceptance (pairing.py),
(iv) tamper rejection.

Every function cites the reference lines it follows (paths relative to
/root/reference/).
"""
from .fields import g1_add, g1_mul, g1_msm_naive, g1_neg

MINUS_ALPHA = 3   # src/common.rs:11
MINUS_GAMMA = 5   # src/common.rs:14
B_POLYMATH = b"polymath"  # src/common.rs:8


# ------------------------------------------------------------------ R1CS / SAP
class R1CS:
    """ark-relations ConstraintMatrices as the reference consumes them
    (src/generator.rs:46-54): column 0 = constant One, 1..m0 instance, then
    witness.  Rows are lists of (value, column)."""

    def __init__(self, m0, mw, a, b, c):
        assert len(a) == len(b) == len(c)
        self.m0, self.mw, self.nr = m0, mw, len(a)
        self.a, self.b, self.c = a, b, c


def m_at(m, i, j):
    """src/common.rs:100-105 -- first entry with that column, else 0."""
    for (v, idx) in m[i]:
        if idx == j:
            return v
    return 0


class SAPMatrices:
    """src/common.rs:113-230, transcribed arm by arm."""

    def __init__(self, r1cs, r):
        self.q = r1cs
        self.r = r

    def m0_m_n(self):  # common.rs:224-229
        m0 = self.q.m0
        return m0, m0 + self.q.mw, self.q.nr

    def size(self):  # common.rs:131-135
        m0, m, n = self.m0_m_n()
        return (m0 + n) * 2, m0 * 2 + m + n

    def u(self, i, j):  # common.rs:138-172
        m0, m, n = self.m0_m_n()
        dm0, dm0n, dm02n, m0m = m0 + m0, m0 + m0 + n, m0 + m0 + n + n, m0 + m
        r = self.r
        if (i, j) == (0, 0):
            return 2
        if i < m0 and j == 0:
            return 1
        if i < m0 and j == i:
            return 1
        if i < m0:
            return 0
        if i == m0 and j == 0:
            return 0
        if i < dm0 and j == 0:
            return 1
        if i < dm0 and j == i - m0:
            return r - 1
        if i < dm0:
            return 0
        if j < m0:
            return 0
        if i < dm0n and j < m0m:
            return (m_at(self.q.a, i - dm0, j - m0) + m_at(self.q.b, i - dm0, j - m0)) % r
        if i < dm02n and j < m0m:
            return (m_at(self.q.a, i - dm0n, j - m0) - m_at(self.q.b, i - dm0n, j - m0)) % r
        return 0

    def w(self, i, j):  # common.rs:175-207
        m0, m, n = self.m0_m_n()
        dm0, dm0n, dm02n, m0m = m0 + m0, m0 + m0 + n, m0 + m0 + n + n, m0 + m
        r = self.r
        if i < m0 and j == i + m0:
            return 4
        if i < m0 and j == i + m0m:
            return 1
        if i < m0:
            return 0
        if i < dm0 and j == i + m:
            return 1
        if i < dm0:
            return 0
        if j < m0:
            return 0
        if i < dm0n and j < m0m:
            return m_at(self.q.c, i - dm0, j - m0) * 4 % r
        if i < dm0n and j == i + m:
            return 1
        if i < dm0n:
            return 0
        if i < dm02n and j == i - n + m:
            return 1
        return 0


# ------------------------------------------------------------ domain / polys
def next_pow2(k):
    n = 1
    while n < k:
        n <<= 1
    return n


def ntt_naive(c, vals, n, inverse=False):
    """ark-poly Radix2EvaluationDomain::{fft, ifft}: natural order in and out,
    zero-padded to n, ifft scales by 1/n.  O(n^2) definition."""
    r = c.r
    w = c.root_of_unity(n)
    if inverse:
        w = pow(w, -1, r)
    v = list(vals) + [0] * (n - len(vals))
    out = []
    for k in range(n):
        wk = pow(w, k, r)
        acc, x = 0, 1
        for j in range(n):
            acc = (acc + v[j] * x) % r
            x = x * wk % r
        out.append(acc)
    if inverse:
        ninv = pow(n, -1, r)
        out = [o * ninv % r for o in out]
    return out


def ntt_fast(c, vals, n, inverse=False):
    """Same map as ntt_naive by recursive radix-2 (for mid-size fixtures)."""
    r = c.r
    w = c.root_of_unity(n)
    if inverse:
        w = pow(w, -1, r)
    v = list(vals) + [0] * (n - len(vals))

    def rec(a, w):
        m = len(a)
        if m == 1:
            return a
        e = rec(a[0::2], w * w % r)
        o = rec(a[1::2], w * w % r)
        out = [0] * m
        x = 1
        for k in range(m // 2):
            t = x * o[k] % r
            out[k] = (e[k] + t) % r
            out[k + m // 2] = (e[k] - t) % r
            x = x * w % r
        return out

    out = rec(v, w)
    if inverse:
        ninv = pow(n, -1, r)
        out = [o * ninv % r for o in out]
    return out


def strip(p):
    """DensePolynomial::from_coefficients_vec strips trailing zeros."""
    p = list(p)
    while p and p[-1] == 0:
        p.pop()
    return p


def poly_eval(p, x, r):
    acc = 0
    for coef in reversed(p):
        acc = (acc * x + coef) % r
    return acc


def poly_add(a, b, r):
    n = max(len(a), len(b))
    return strip([((a[i] if i < len(a) else 0) + (b[i] if i < len(b) else 0)) % r for i in range(n)])


def poly_scale(a, s, r):
    return strip([x * s % r for x in a])


def poly_mul_naive(a, b, r):
    if not a or not b:
        return []
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            out[i + j] = (out[i + j] + x * y) % r
    return strip(out)


def poly_shift(a, k):
    """mul_by_x_power, prover.rs:254-258."""
    return ([0] * k + list(a)) if a else []


# ------------------------------------------------------------------ generator
class ProvingKey:
    pass


def generate_proving_key(c, r1cs, x, z, g2_mul=None):
    """src/generator.rs:24-167 with the two rng draws (x then z, :72,:77) passed in.
    DENSE (uses SAPMatrices.u/w element by element) -- tiny circuits only."""
    r = c.r
    sap = SAPMatrices(r1cs, r)
    rows, cols = sap.size()                       # :59
    n = next_pow2(rows)                           # :60,:66
    m, m0, bnd_a, sigma = cols, r1cs.m0, 1, n + 3  # :67-70
    omega = c.root_of_unity(n)
    assert pow(x, n, r) != 1 and pow(z, n, r) != 1  # sample_element_outside_domain
    y = pow(x, sigma, r)                          # :73
    y_inv = pow(y, -1, r)
    y_alpha = pow(y_inv, MINUS_ALPHA, r)          # :74
    y_to_minus_alpha = pow(y, MINUS_ALPHA, r)     # :75
    y_gamma = pow(y_inv, MINUS_GAMMA, r)          # :76
    g1 = c.g1

    def generate(max_index, f):                   # :169-177
        return [g1_mul(c, g1, f(j)) for j in range(max_index + 1)]

    pk = ProvingKey()
    pk.curve, pk.r1cs, pk.sap = c, r1cs, sap
    pk.n, pk.m0, pk.sigma, pk.omega = n, m0, sigma, omega
    pk.x_powers_g1 = generate(n + bnd_a - 1, lambda j: pow(x, j, r))                     # :82
    pk.x_powers_y_alpha_g1 = generate(2 * bnd_a, lambda j: pow(x, j, r) * y_alpha % r)   # :86
    pk.x_powers_y_gamma_g1 = generate(bnd_a, lambda j: pow(x, j, r) * y_gamma % r)       # :90
    dmax = 2 * (n - 1) + sigma * (MINUS_ALPHA + MINUS_GAMMA)                             # :95-96
    pk.x_powers_y_gamma_z_g1 = generate(dmax, lambda j: pow(x, j, r) * y_gamma % r * z % r)  # :97-99
    zh_at_x = (pow(x, n, r) - 1) % r                                                    # :106
    pk.x_powers_zh_by_y_alpha_g1 = generate(
        n - 2, lambda j: pow(x, j, r) * zh_at_x % r * y_to_minus_alpha % r)              # :107
    # evaluate_all_lagrange_coefficients(x)  :113
    ninv = pow(n, -1, r)
    l_at_x = [zh_at_x * ninv % r * pow(omega, i, r) % r * pow((x - pow(omega, i, r)) % r, -1, r) % r
              for i in range(n)]

    def lc(j):                                                                          # :115-135
        uj = sum(l_at_x[i] * sap.u(i, j + m0) for i in range(n)) % r
        wj = sum(l_at_x[i] * sap.w(i, j + m0) for i in range(n)) % r
        return (uj * y_gamma + wj) % r * y_to_minus_alpha % r

    pk.lcs_scalars = [lc(j) for j in range(m - m0)]
    pk.uj_wj_lcs_by_y_alpha_g1 = [g1_mul(c, g1, s) for s in pk.lcs_scalars]
    pk.trapdoor = (x, z)  # kept only so tests can build [x]_2, [z]_2
    return pk


# --------------------------------------------------------------------- prover
def compute_y_vec(c, r1cs, x, w):
    """src/prover.rs:279-302."""
    r = c.r
    y_m0 = [pow((1 - x[j]) % r, 2, r) for j in range(1, r1cs.m0)]
    xw = list(x) + list(w)
    y_n = []
    for i in range(r1cs.nr):
        v = sum((m_at(r1cs.a, i, j) - m_at(r1cs.b, i, j)) * xw[j] for j in range(r1cs.m0 + r1cs.mw)) % r
        y_n.append(v * v % r)
    return [0] + y_m0 + y_n


def compute_pi_at_x1(c, n, omega, public_inputs, x1, y1_gamma):
    """src/common.rs:49-71 + z_tilde_i :77-97."""
    r = c.r
    m0 = len(public_inputs)

    def z_tilde(i):
        if i == 0:
            return 2
        if i < m0:
            return (1 + public_inputs[i]) % r
        if i == m0:
            return 0
        return (1 - public_inputs[i - m0]) % r

    num = (pow(x1, n, r) - 1) * pow(n, -1, r) % r
    w_i = 1
    s = 0
    for i in range(2 * m0):
        li = num * pow((x1 - w_i) % r, -1, r) % r
        s = (s + z_tilde(i) * li) % r
        num = num * omega % r
        w_i = w_i * omega % r
    return s * y1_gamma % r


def compute_c_at_x1(c, y1_gamma, y1_alpha, a_at_x1, pi_at_x1):
    """src/common.rs:73-75."""
    r = c.r
    return ((a_at_x1 + y1_gamma) * a_at_x1 - pi_at_x1) % r * pow(y1_alpha, -1, r) % r


def create_proof_with_assignment(c, pk, instance, witness, r_a, transcript_cls, trace=None):
    """src/prover.rs:66-237, line by line, dense.  `r_a` = the two F::rand draws of
    :110 (constant term first).  Returns dict(a_g1, c_g1, a_at_x1, d_g1)."""
    r = c.r
    r1cs, sap = pk.r1cs, pk.sap
    yv = compute_y_vec(c, r1cs, instance, witness)
    z = list(instance) + list(instance) + list(witness) + yv            # :75-80
    rows, cols = sap.size()                                             # :82
    n = next_pow2(rows)                                                 # :83-85
    assert len(z) == cols
    uj = [[sap.u(i, j) for i in range(n)] for j in range(cols)]         # :87
    wj = [[sap.w(i, j) for i in range(n)] for j in range(cols)]         # :88
    ujz = [[e * z[j] % r for e in col] for j, col in enumerate(uj)]     # :90
    wjz = [[e * z[j] % r for e in col] for j, col in enumerate(wj)]     # :91
    u_evals = [sum(col[i] for col in ujz) % r for i in range(n)]        # :93
    w_evals = [sum(col[i] for col in wjz) % r for i in range(n)]        # :95
    u_coeffs = ntt_fast(c, u_evals, n, inverse=True)                    # :94
    w_coeffs = ntt_fast(c, w_evals, n, inverse=True)                    # :96
    # square_polynomial :315-328
    n2 = next_pow2(2 * len(u_coeffs))
    ev = ntt_fast(c, u_coeffs, n2)
    u2_coeffs = ntt_fast(c, [e * e % r for e in ev], n2, inverse=True)
    u_poly, u2_poly, w_poly = strip(u_coeffs), strip(u2_coeffs), strip(w_coeffs)  # :100-102
    h_num = poly_add(u2_poly, [(-e) % r for e in w_poly], r)            # :104
    # divide_by_vanishing_poly(domain) :105   (X^n - 1)
    h_poly = strip(h_num[n:])
    q_ext = list(h_num) + [0] * (2 * n - len(h_num))
    rem = strip([(q_ext[i] + q_ext[i + n]) % r for i in range(n)])
    assert h_poly and len(h_poly) - 1 <= n - 2, "DEGREE_BOUND prover.rs:107"
    assert not rem, "REMAINDER_NONZERO prover.rs:108"
    r_a_poly = strip(list(r_a))                                         # :110
    assert len(u_poly) <= n                                             # :113
    msm = lambda sc, bs: (_assert(len(sc) <= len(bs)), g1_msm_naive(c, bs, sc))[1]  # :380-384
    # compute_a_g1 :330-338
    a_g1 = g1_add(c, msm(u_poly, pk.x_powers_g1), msm(r_a_poly, pk.x_powers_y_alpha_g1))
    # compute_r_g1 :340-357
    two_ra_u = poly_scale(poly_mul_naive(u_poly, r_a_poly, r), 2, r)
    ra_sq = poly_mul_naive(r_a_poly, r_a_poly, r)
    r_g1 = g1_add(c, g1_add(c, msm(two_ra_u, pk.x_powers_g1), msm(ra_sq, pk.x_powers_y_alpha_g1)),
                  msm(r_a_poly, pk.x_powers_y_gamma_g1))
    h_g1 = msm(h_poly, pk.x_powers_zh_by_y_alpha_g1)                    # :118
    m0 = len(instance)
    z_tail = z[m0:]                                                     # z[1..].concat() :121
    lcs_g1 = msm(z_tail, pk.uj_wj_lcs_by_y_alpha_g1)
    c_g1 = g1_add(c, g1_add(c, lcs_g1, h_g1), r_g1)                     # :123
    t = transcript_cls(B_POLYMATH)                                      # :125
    x1 = compute_x1(c, t, instance, [a_g1, c_g1])                       # :126
    sigma = pk.sigma
    y1 = pow(x1, sigma, r)                                              # :128
    y1_alpha = pow(pow(y1, -1, r), MINUS_ALPHA, r)                      # :130
    a_at_x1 = (poly_eval(u_poly, x1, r) + poly_eval(r_a_poly, x1, r) * y1_alpha) % r  # :132
    y1_gamma = pow(pow(y1, -1, r), MINUS_GAMMA, r)                      # :134
    pi_at_x1 = compute_pi_at_x1(c, pk.n, pk.omega, instance, x1, y1_gamma)  # :135
    c_at_x1 = compute_c_at_x1(c, y1_gamma, y1_alpha, a_at_x1, pi_at_x1)  # :138
    # batch commitment :142-185
    a_by = poly_add(poly_shift(u_poly, sigma * MINUS_GAMMA),
                    poly_shift(r_a_poly, sigma * (MINUS_GAMMA - MINUS_ALPHA)), r)       # :145-152
    # compute_r_x_by_y_gamma_poly :359-377
    r_by = poly_add(poly_add(poly_shift(two_ra_u, sigma * MINUS_GAMMA),
                             poly_shift(ra_sq, sigma * (MINUS_GAMMA - MINUS_ALPHA)), r), r_a_poly, r)
    wit_u_evals = [sum(col[i] for col in ujz[m0:]) % r for i in range(n)]  # :157,:160
    wit_w_evals = [sum(col[i] for col in wjz[m0:]) % r for i in range(n)]  # :158,:164
    wit_u = strip(ntt_fast(c, wit_u_evals, n, inverse=True))            # :161-162
    wit_w = strip(ntt_fast(c, wit_w_evals, n, inverse=True))            # :165-166
    c_by = poly_add(poly_add(poly_add(
        poly_shift(wit_u, sigma * MINUS_ALPHA),                          # :168-171
        poly_shift(wit_w, sigma * (MINUS_ALPHA + MINUS_GAMMA)), r),      # :172-175
        poly_shift(h_num, sigma * (MINUS_ALPHA + MINUS_GAMMA)), r),      # :177-180
        r_by, r)                                                         # :182-185
    x2 = compute_x2(c, t, x1, [a_at_x1, c_at_x1])                       # :189
    ytmg = poly_shift([1], sigma * MINUS_GAMMA)                         # :191-194
    num = poly_add(poly_add(poly_add(a_by, poly_scale(c_by, x2, r), r),
                            poly_scale(ytmg, (-a_at_x1) % r, r), r),
                   poly_scale(ytmg, (-(c_at_x1 * x2)) % r, r), r)        # :211-216
    # divide_with_q_and_r by (X - x1) :217-220  (synthetic division)
    q = [0] * (len(num) - 1)
    carry = 0
    for k in range(len(num) - 1, 0, -1):
        carry = (num[k] + x1 * carry) % r
        q[k - 1] = carry
    rem2 = (num[0] + x1 * carry) % r
    assert rem2 == 0, "REMAINDER_NONZERO prover.rs:221"
    q = strip(q)
    assert len(q) - 1 <= 2 * (n - 1) + sigma * (MINUS_ALPHA + MINUS_GAMMA), "DEGREE_BOUND prover.rs:222"
    d_g1 = msm(q, pk.x_powers_y_gamma_z_g1)                             # :229
    if trace is not None:
        trace.update(dict(z=z, u_evals=u_evals, w_evals=w_evals, u=u_coeffs, w=w_coeffs,
                          u2=u2_coeffs, h=h_poly, wit_u=wit_u, x1=x1, x2=x2, c_at_x1=c_at_x1,
                          pi_at_x1=pi_at_x1, numerator=num, quotient=q, z_tail=z_tail,
                          two_ra_u=two_ra_u))
    return dict(a_g1=a_g1, c_g1=c_g1, a_at_x1=a_at_x1, d_g1=d_g1)


def _assert(cond):
    assert cond, "LEN_MISMATCH prover.rs:381"


# ------------------------------------------------- transcript-facing helpers
def compute_x1(c, t, public_inputs, commitments):
    """src/common.rs:21-30."""
    from .serialize import ser_fr_slice, ser_g1_slice
    t.append_message(b"public_inputs", ser_fr_slice(c, public_inputs))
    t.append_message(b"commitments", ser_g1_slice(c, commitments))
    return t.challenge(b"x1")


def compute_x2(c, t, x1, values):
    """src/common.rs:32-37."""
    from .serialize import ser_fr, ser_fr_slice
    t.append_message(b"x1", ser_fr(c, x1))
    t.append_message(b"values", ser_fr_slice(c, values))
    return t.challenge(b"x2")


# ------------------------------------------------------------------- verifier
def verify_proof(c, vk, proof, public_inputs, transcript_cls, pairing_check):
    """src/verifier.rs:19-62.  vk = dict(n, m0, sigma, omega, one_g1, one_g2, x_g2, z_g2).
    `pairing_check(pairs)` returns True iff prod e(P_i, Q_i) == 1."""
    r = c.r
    t = transcript_cls(B_POLYMATH)                                       # :24
    pub = [1] + list(public_inputs)                                      # :26
    x1 = compute_x1(c, t, pub, [proof["a_g1"], proof["c_g1"]])           # :29
    y1 = pow(x1, vk["sigma"], r)                                         # :32
    y1_gamma = pow(pow(y1, -1, r), MINUS_GAMMA, r)                       # :34
    pi_at_x1 = compute_pi_at_x1(c, vk["n"], vk["omega"], pub, x1, y1_gamma)  # :35
    y1_alpha = pow(pow(y1, -1, r), MINUS_ALPHA, r)                       # :37
    c_at_x1 = compute_c_at_x1(c, y1_gamma, y1_alpha, proof["a_at_x1"], pi_at_x1)  # :40
    x2 = compute_x2(c, t, x1, [proof["a_at_x1"], c_at_x1])               # :42
    lhs = g1_msm_naive(c, [proof["a_g1"], proof["c_g1"], vk["one_g1"]],
                       [1, x2, (-(proof["a_at_x1"] + x2 * c_at_x1)) % r])  # :44-47
    return pairing_check(lhs, g1_neg(c, proof["d_g1"]), x1, vk)          # :48-61
