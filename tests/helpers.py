"""Shared helpers for the test-suite (test infrastructure; may import oracle/)."""
import json
import os

import numpy as np

from oracle import cpp_oracle as CO
from oracle.pyref import protocol as PR
from oracle.pyref.fields import CURVES

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
I = lambda s: int(s, 16)
PT = lambda p: None if p is None else (int(p[0], 16), int(p[1], 16))
BASE_NAMES = CO.OraclePk.BASE_NAMES  # pm_base_vec order


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def r1cs_from_json(j):
    row = lambda rw: [(int(v, 16), c) for v, c in rw]
    return PR.R1CS(j["m0"], j["mw"], [row(r) for r in j["a"]], [row(r) for r in j["b"]], [row(r) for r in j["c"]])


def rand_fr_limbs(curve, count, seed):
    """count uniformly random Fr elements as Montgomery limbs [count,4] (numpy RNG; rejection)."""
    r = CURVES[curve].r
    rng = np.random.default_rng(seed)
    top_mask = (1 << (r.bit_length() - 192)) - 1
    out = rng.integers(0, 1 << 64, size=(count, 4), dtype=np.uint64)
    out[:, 3] &= np.uint64(top_mask)
    # reject >= r (compare top limb only is not exact; fix rare rows exactly)
    r_limbs = [(r >> (64 * k)) & ((1 << 64) - 1) for k in range(4)]
    bad = out[:, 3] >= np.uint64(r_limbs[3])
    for i in np.nonzero(bad)[0]:
        v = sum(int(out[i, k]) << (64 * k) for k in range(4))
        while v >= r:
            v >>= 1
        for k in range(4):
            out[i, k] = (v >> (64 * k)) & ((1 << 64) - 1)
    return out


def pm_csrs(curve, r1cs):
    """Three polymath_amd.api.CsrArrays (A, B, C) for an R1CS given with integer coefficients."""
    from polymath_amd import api
    out = []
    for rows in (r1cs.a, r1cs.b, r1cs.c):
        rowptr, cols, vals = [0], [], []
        for row in rows:
            for v, j in row:
                cols.append(j)
                vals.append(v)
            rowptr.append(len(cols))
        out.append(api.CsrArrays(rowptr, cols, CO.fr_to_mont_limbs(curve, vals) if vals else []))
    return out


TABLES_OPT = {"1": "auto", "0": "off", "wide": "wide"}     # the parametrisations' historical names (PM_TABLES values)
