"""ORACLE (test infrastructure only) -- drives ANY three-phase prover backend (the C++ oracle's
OraclePk, or the product's polymath_amd.api.ProvingKey through the C ABI) through
create_proof_with_assignment (/root/reference/src/prover.rs:66-237), doing the host-side
Fiat-Shamir and O(m0) scalar glue (common.rs:21-98) with the big-integer restatement.

A backend exposes: curve (name), phase1(x, w, r_a) -> (rc, a_xy, a_inf, c_xy, c_inf),
phase2(x1) -> (rc, u_at_x1), phase3(x1, x2, a_at_x1, c_at_x1) -> (rc, d_xy, d_inf),
all on np.uint64 Montgomery limb arrays (include/polymath_hip.h conventions).
"""
from . import cpp_oracle as CO
from .pyref import protocol as PR
from .pyref.fields import CURVES


class ProverError(Exception):
    def __init__(self, phase, rc):
        super().__init__("phase %d failed with status %d" % (phase, rc))
        self.phase, self.rc = phase, rc


def prove(backend, n, sigma, omega, instance, witness, r_a, transcript_cls, trace=None, w_limbs=None):
    """witness: list of integers, or None with w_limbs = the same as Montgomery limbs [mw, 4] (large circuits)."""
    cname = backend.curve
    c = CURVES[cname]
    r = c.r
    x = CO.fr_to_mont_limbs(cname, instance)
    w = w_limbs if w_limbs is not None else CO.fr_to_mont_limbs(cname, witness)
    ra = CO.fr_to_mont_limbs(cname, r_a)
    rc, a_xy, a_inf, c_xy, c_inf = backend.phase1(x, w, ra)
    if rc:
        raise ProverError(1, rc)
    a_g1 = CO.g1_from_mont_limbs(cname, a_xy, [a_inf])[0]
    c_g1 = CO.g1_from_mont_limbs(cname, c_xy, [c_inf])[0]
    t = transcript_cls(PR.B_POLYMATH)                                    # prover.rs:125
    x1 = PR.compute_x1(c, t, instance, [a_g1, c_g1])                     # :126
    y1 = pow(x1, sigma, r)                                               # :128
    y1_alpha = pow(pow(y1, -1, r), PR.MINUS_ALPHA, r)                    # :130
    rc, u_at = backend.phase2(CO.fr_to_mont_limbs(cname, [x1]))
    if rc:
        raise ProverError(2, rc)
    u_at_x1 = CO.fr_from_mont_limbs(cname, u_at)[0]
    a_at_x1 = (u_at_x1 + (r_a[0] + r_a[1] * x1) * y1_alpha) % r          # :132
    y1_gamma = pow(pow(y1, -1, r), PR.MINUS_GAMMA, r)                    # :134
    pi_at_x1 = PR.compute_pi_at_x1(c, n, omega, instance, x1, y1_gamma)  # :135
    c_at_x1 = PR.compute_c_at_x1(c, y1_gamma, y1_alpha, a_at_x1, pi_at_x1)  # :138
    x2 = PR.compute_x2(c, t, x1, [a_at_x1, c_at_x1])                     # :189
    L = lambda v: CO.fr_to_mont_limbs(cname, [v])
    rc, d_xy, d_inf = backend.phase3(L(x1), L(x2), L(a_at_x1), L(c_at_x1))
    if rc:
        raise ProverError(3, rc)
    d_g1 = CO.g1_from_mont_limbs(cname, d_xy, [d_inf])[0]
    if trace is not None:
        trace.update(dict(x1=x1, x2=x2, c_at_x1=c_at_x1, u_at_x1=u_at_x1))
    return dict(a_g1=a_g1, c_g1=c_g1, a_at_x1=a_at_x1, d_g1=d_g1)
