"""ORACLE (test infrastructure only) -- the reference's harness circuits restated as
(R1CS matrices, instance, witness) tuples (SURVEY.md §8 a-H), plus the synthetic
"random A*B=C gates" R1CS of SURVEY.md §8d used by bench.py.

Variable numbering follows ark-relations `to_matrices` [ark, from memory]:
column 0 = One, 1..m0-1 = instance variables in allocation order, then witness
variables in allocation order.
"""
from .protocol import R1CS

SPLITMIX_SEED = 0x706F6C796D617468  # "polymath"
_M64 = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed=SPLITMIX_SEED):
        self.s = seed & _M64

    def next_u64(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & _M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)

    def fr(self, r):
        """4 draws (limb 0 first) masked to the modulus bit length, rejection."""
        nb = r.bit_length()
        while True:
            v = 0
            for i in range(4):
                v |= self.next_u64() << (64 * i)
            v &= (1 << nb) - 1
            if v < r:
                return v

    def below(self, n):
        return self.next_u64() % n


def dummy_circuit(c, a, b):
    """tests/dummy.rs:20-35 -- witnesses a, b; public c = a*b; one constraint a*b=c."""
    r = c.r
    m0, mw = 2, 2
    A = [[(1, 2)]]      # a = witness 0 -> column m0+0
    B = [[(1, 3)]]
    Cm = [[(1, 1)]]     # c = instance 1
    return R1CS(m0, mw, A, B, Cm), [1, a * b % r], [a % r, b % r]


def bench_circuit(c, a, b, num_variables, num_constraints):
    """benches/bench.rs:38-61 -- witnesses a, b, public c=a*b, num_variables-3 padding
    witnesses all equal to a; num_constraints-1 copies of a*b=c plus one empty row."""
    r = c.r
    m0 = 2
    mw = 2 + (num_variables - 3)
    A = [[(1, 2)] for _ in range(num_constraints - 1)] + [[]]
    B = [[(1, 3)] for _ in range(num_constraints - 1)] + [[]]
    Cm = [[(1, 1)] for _ in range(num_constraints - 1)] + [[]]
    wit = [a % r, b % r] + [a % r] * (num_variables - 3)
    return R1CS(m0, mw, A, B, Cm), [1, a * b % r], wit


def mimc_circuit(c, xl, xr, constants):
    """tests/mimc.rs:46-61 (native) and :74-143 (constraints), MIMC_ROUNDS = len(constants).
    Per round i:  tmp = (xl + C_i)^2 ;  new_xl = xr + tmp*(xl + C_i); last new_xl is public."""
    r = c.r
    rounds = len(constants)
    m0 = 2
    wit = [xl % r, xr % r]
    col_xl, col_xr = m0 + 0, m0 + 1
    v_xl, v_xr = xl % r, xr % r
    A, B, Cm = [], [], []
    nwit = 2
    for i in range(rounds):
        ci = constants[i] % r
        tmp_v = pow((v_xl + ci) % r, 2, r)
        col_tmp = m0 + nwit
        nwit += 1
        wit.append(tmp_v)
        # (xl + Ci) * (xl + Ci) = tmp      (mimc.rs:98-102); LC order: variable term then One
        lc = [(1, col_xl), (ci, 0)]
        A.append(list(lc))
        B.append(list(lc))
        Cm.append([(1, col_tmp)])
        new_v = ((v_xl + ci) * tmp_v + v_xr) % r
        if i == rounds - 1:
            col_new = 1                      # public input (mimc.rs:115-118)
            pub = new_v
        else:
            col_new = m0 + nwit
            nwit += 1
            wit.append(new_v)
        # tmp * (xl + Ci) = new_xl - xr     (mimc.rs:123-127)
        A.append([(1, col_tmp)])
        B.append(list(lc))
        Cm.append([(1, col_new), (r - 1, col_xr)])
        col_xr, v_xr = col_xl, v_xl
        col_xl, v_xl = col_new, new_v
    return R1CS(m0, nwit, A, B, Cm), [1, pub], wit


def mimc_native(c, xl, xr, constants):
    r = c.r
    for ci in constants:
        t = (xl + ci) % r
        xl, xr = (pow(t, 3, r) + xr) % r, xl
    return xl


def synthetic_r1cs(c, nr, seed=SPLITMIX_SEED):
    """m0 = 2 (One, out); gate i: A_i={(alpha_i,p_i)}, B_i={(beta_i,q_i)},
    C_i={(1, t_i)}, t_i = (alpha_i z_{p_i})(beta_i z_{q_i}); p_i,q_i uniform over the
    already-defined variables (One, then the two seed witnesses, then earlier gate
    outputs); the last gate's output is the public input (column 1).
    Draw order per gate: alpha, beta, p, q.  Seed witnesses w0, w1 are drawn first."""
    r = c.r
    g = SplitMix64(seed)
    m0 = 2
    wit = [g.fr(r), g.fr(r)]
    # "defined" list holds (column, value); instance column 1 is defined only at the end
    defined_cols = [0, m0 + 0, m0 + 1]
    defined_vals = [1, wit[0], wit[1]]
    A, B, Cm = [], [], []
    pub = None
    for i in range(nr):
        alpha, beta = g.fr(r), g.fr(r)
        pi, qi = g.below(len(defined_cols)), g.below(len(defined_cols))
        t = (alpha * defined_vals[pi] % r) * (beta * defined_vals[qi] % r) % r
        if i == nr - 1:
            col = 1
            pub = t
        else:
            col = m0 + len(wit)
            wit.append(t)
            defined_cols.append(col)
            defined_vals.append(t)
        A.append([(alpha, defined_cols[pi])])
        B.append([(beta, defined_cols[qi])])
        Cm.append([(1, col)])
    return R1CS(m0, len(wit), A, B, Cm), [1, pub], wit


def r1cs_is_satisfied(c, q, inst, wit):
    r = c.r
    zz = list(inst) + list(wit)
    for ra, rb, rc in zip(q.a, q.b, q.c):
        az = sum(v * zz[j] for v, j in ra) % r
        bz = sum(v * zz[j] for v, j in rb) % r
        cz = sum(v * zz[j] for v, j in rc) % r
        if az * bz % r != cz:
            return False
    return True


def first_entry_dot(r, row, zz):
    """(M z)_row as the SAP matrices see it: m_at (common.rs:100-105) returns the FIRST entry of a column, so
    later entries of the same column inside a row do not count."""
    seen, s = set(), 0
    for v, j in row:
        if j in seen:
            continue
        seen.add(j)
        s += v * zz[j]
    return s % r


def random_r1cs(c, seed, m0, nr, max_entries=4, unused_every=5, extreme=False):
    """A seeded random R1CS of ARBITRARY shape for differential testing of the witness map
    (common.rs:77-97, 131-207; prover.rs:75-96, 156-166) -- everything the harness circuits never do:
    m0 - 1 free public inputs (m0 = 1: none), 0..max_entries entries per row in each of A, B, C, entries on
    column 0 and on instance columns, zero / one / minus-one coefficients, duplicate columns inside a row
    (only the first counts: m_at), empty rows, witness columns no row uses (points at infinity in
    uj_wj_lcs), a fresh witness per gate with a random non-zero coefficient somewhere inside its C row.
    Satisfied under m_at's first-entry semantics.  Draw order: instance values, seed witnesses, then per
    gate: the A row, the B row, the C row.  extreme: the free values (public inputs, seed and unused witnesses)
    come from {0, 1, r - 1, r - 2, 2} three times out of four -- zero scalars, 1 - x = 0, products that vanish."""
    r = c.r
    g = SplitMix64(seed)

    def free_value():
        if extreme and g.below(4):
            return (0, 1, r - 1, r - 2, 2)[g.below(5)]
        return g.fr(r)
    inst = [1] + [free_value() for _ in range(m0 - 1)]
    z = inst + [free_value() for _ in range(1 + g.below(3))]     # One, instance, then the witnesses as they are allocated

    def coef():
        k = g.below(8)
        return 0 if k == 0 else 1 if k == 1 else r - 1 if k == 2 else g.below(1 << 16) if k == 3 else g.fr(r)

    def column():
        k = g.below(8)
        if k == 0:
            return 0
        if k <= 2 and m0 > 1:
            return 1 + g.below(m0 - 1)
        return m0 + g.below(len(z) - m0)

    def row(k):
        out = []
        for _ in range(k):
            if out and g.below(5) == 0:
                out.append((coef(), out[g.below(len(out))][1]))      # a repeated column: ignored by m_at
            else:
                out.append((coef(), column()))
        return out

    A, B, Cm = [], [], []
    for i in range(nr):
        ra, rb = row(g.below(max_entries + 1)), row(g.below(max_entries + 1))
        prod = first_entry_dot(r, ra, z) * first_entry_dot(r, rb, z) % r
        rc = row(g.below(max_entries))
        rest = first_entry_dot(r, rc, z)
        if prod == rest and g.below(2) == 0:
            pass                                                     # already satisfied (e.g. an empty A row): no fresh witness
        else:
            k = 0
            while k == 0:
                k = coef()
            t = (prod - rest) * pow(k, -1, r) % r
            col = len(z)
            z.append(t)
            rc.insert(g.below(len(rc) + 1), (k, col))
            if g.below(6) == 0:
                rc.append((coef(), col))                             # the fresh column once more, later in the row
        A.append(ra)
        B.append(rb)
        Cm.append(rc)
        if unused_every and g.below(unused_every) == 0:
            z.append(free_value())                                   # a witness no row refers to
    wit = z[m0:]
    q = R1CS(m0, len(wit), A, B, Cm)
    assert all(first_entry_dot(r, a, z) * first_entry_dot(r, b, z) % r == first_entry_dot(r, cc, z) for a, b, cc in zip(A, B, Cm))
    return q, inst, wit
