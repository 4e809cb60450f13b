"""Workload generators: the reference's harness circuits as ConstraintSynthesizer-shaped classes
(tests/dummy.rs:20-35, tests/mimc.rs:74-143, benches/bench.rs:38-61) and the synthetic
"random A*B=C gates" R1CS BASELINE.json quotes the headline metric on."""
from .polymath import ConstraintSystem, R1CS

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
        nb = r.bit_length()
        while True:
            v = 0
            for i in range(4):
                v |= self.next_u64() << (64 * i)
            v &= (1 << nb) - 1
            if v < r:
                return v


class DummyCircuit:
    """tests/dummy.rs:20-35: witnesses a, b; public c = a*b; a * b = c."""

    def __init__(self, a, b):
        self.a, self.b = a, b

    def generate_constraints(self, cs):
        a = cs.new_witness_variable(self.a)
        b = cs.new_witness_variable(self.b)
        c = cs.new_input_variable(self.a * self.b)
        cs.enforce_constraint([(1, a)], [(1, b)], [(1, c)])


class BenchCircuit:
    """benches/bench.rs:38-61."""

    def __init__(self, a, b, num_variables, num_constraints):
        self.a, self.b, self.nv, self.nc = a, b, num_variables, num_constraints

    def generate_constraints(self, cs):
        a = cs.new_witness_variable(self.a)
        b = cs.new_witness_variable(self.b)
        c = cs.new_input_variable(self.a * self.b)
        for _ in range(self.nv - 3):
            cs.new_witness_variable(self.a)
        for _ in range(self.nc - 1):
            cs.enforce_constraint([(1, a)], [(1, b)], [(1, c)])
        cs.enforce_constraint([], [], [])


class MiMCDemo:
    """tests/mimc.rs:66-143 (LongsightF322p3 when len(constants) == 322)."""

    def __init__(self, xl, xr, constants):
        self.xl, self.xr, self.constants = xl, xr, constants

    def generate_constraints(self, cs):
        r = cs.r
        xl_v, xr_v = self.xl % r, self.xr % r
        xl, xr = cs.new_witness_variable(xl_v), cs.new_witness_variable(xr_v)
        n = len(self.constants)
        for i, ci in enumerate(self.constants):
            tmp_v = pow(xl_v + ci, 2, r)
            tmp = cs.new_witness_variable(tmp_v)
            lc = [(1, xl), (ci, ConstraintSystem.ONE)]
            cs.enforce_constraint(lc, lc, [(1, tmp)])
            new_v = ((xl_v + ci) * tmp_v + xr_v) % r
            new_xl = cs.new_input_variable(new_v) if i == n - 1 else cs.new_witness_variable(new_v)
            cs.enforce_constraint([(1, tmp)], lc, [(1, new_xl), (r - 1, xr)])
            xr, xr_v = xl, xl_v
            xl, xl_v = new_xl, new_v


def mimc_native(r, xl, xr, constants):
    for ci in constants:
        xl, xr = (pow(xl + ci, 3, r) + xr) % r, xl
    return xl


def synthetic_r1cs(r, nr, seed=SPLITMIX_SEED):
    """m0 = 2; gate i: A_i = {(alpha_i, p_i)}, B_i = {(beta_i, q_i)}, C_i = {(1, t_i)},
    t_i = (alpha_i z_p)(beta_i z_q); p, q uniform over already-defined variables; the last gate's
    output is the public input.  Draw order: w0, w1, then per gate alpha, beta, p, q."""
    g = SplitMix64(seed)
    m0 = 2
    wit = [g.fr(r), g.fr(r)]
    cols, vals = [0, m0, m0 + 1], [1, wit[0], wit[1]]
    A, B, C = [], [], []
    pub = None
    for i in range(nr):
        alpha, beta = g.fr(r), g.fr(r)
        pi, qi = g.next_u64() % len(cols), g.next_u64() % len(cols)
        t = (alpha * vals[pi] % r) * (beta * vals[qi] % r) % r
        if i == nr - 1:
            col, pub = 1, t
        else:
            col = m0 + len(wit)
            wit.append(t)
            cols.append(col)
            vals.append(t)
        A.append([(alpha, cols[pi])])
        B.append([(beta, cols[qi])])
        C.append([(1, col)])
    return R1CS(m0, len(wit), A, B, C), [1, pub], wit
