"""CPU: the C++ restatement (oracle/cpp, 64-bit limbs, sparse algorithms) against the golden vectors
generated from the big-integer restatement (tools/gen_golden.py) and against its own algebra."""
import numpy as np
import pytest

from helpers import BASE_NAMES, I, PT, load_golden, r1cs_from_json, rand_fr_limbs
from oracle import driver as DR
from oracle.pyref import transcripts as T
from oracle.pyref.fields import CURVES


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_field_and_curve_kats(oracle, curve):
    CO, c = oracle, CURVES[curve]
    k = load_golden("field_curve_kats.json")[curve]
    assert I(k["fr_R"]) == c.fr_R and I(k["fq_R"]) == c.fq_R
    for a, b, ab, s, d, inv in k["fr"]:
        A, B = CO.fr_to_mont_limbs(curve, [I(a)])[0], CO.fr_to_mont_limbs(curve, [I(b)])[0]
        assert CO.fr_from_mont_limbs(curve, CO.fr_op(curve, 0, A, B))[0] == I(ab)
        assert CO.fr_from_mont_limbs(curve, CO.fr_op(curve, 1, A, B))[0] == I(s)
        assert CO.fr_from_mont_limbs(curve, CO.fr_op(curve, 2, A, B))[0] == I(d)
        assert CO.fr_from_mont_limbs(curve, CO.fr_op(curve, 3, A))[0] == I(inv)
    L = lambda v: CO.ints_to_limbs([c.fq_to_mont(v)], c.fq_limbs64)[0]
    U = lambda arr: c.fq_from_mont(CO.limbs_to_ints(arr)[0])
    for a, b, ab, s, d in k["fq"]:
        assert U(CO.fq_op(curve, 0, L(I(a)), L(I(b)))) == I(ab)
        assert U(CO.fq_op(curve, 1, L(I(a)), L(I(b)))) == I(s)
        assert U(CO.fq_op(curve, 2, L(I(a)), L(I(b)))) == I(d)
    g = CO.g1_to_mont_limbs(curve, [PT(k["g1"])])[0]
    for kk, pt in k["g1_muls"]:
        out, inf = CO.g1_mul(curve, g, CO.fr_to_mont_limbs(curve, [I(kk)])[0])
        assert CO.g1_from_mont_limbs(curve, out, [inf])[0] == PT(pt)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_ntt_msm_golden(oracle, curve):
    CO = oracle
    v = load_golden("ntt_msm.json")[curve]
    for t in v["ntt"]:
        inp = CO.fr_to_mont_limbs(curve, [I(x) for x in t["input"]])
        assert CO.fr_from_mont_limbs(curve, CO.ntt(curve, inp, t["log_n"], False)) == [I(x) for x in t["fwd"]]
        assert CO.fr_from_mont_limbs(curve, CO.ntt(curve, inp, t["log_n"], True, nthreads=2)) == [I(x) for x in t["inv"]]
    for t in v["msm"]:
        bases = CO.g1_to_mont_limbs(curve, [PT(p) for p in t["bases"]])
        sc = CO.fr_to_mont_limbs(curve, [I(s) for s in t["scalars"]])
        for nt in (1, 3):
            out, inf = CO.msm(curve, bases, sc, nt)
            assert CO.g1_from_mont_limbs(curve, out, [inf])[0] == PT(t["result"])


def test_ntt_roundtrip_and_linearity(oracle):
    CO, curve = oracle, "bls12_381"
    r = CURVES[curve].r
    a, b = rand_fr_limbs(curve, 1 << 12, 1), rand_fr_limbs(curve, 1 << 12, 2)
    fa = CO.ntt(curve, a, 12, False, 4)
    assert np.array_equal(CO.ntt(curve, fa, 12, True, 4), a)
    ai, bi = CO.fr_from_mont_limbs(curve, a[:64]), CO.fr_from_mont_limbs(curve, b[:64])
    s = CO.fr_to_mont_limbs(curve, [(x + y) % r for x, y in zip(ai, bi)])
    f = lambda arr: CO.fr_from_mont_limbs(curve, CO.ntt(curve, arr, 6, False))
    assert f(s) == [(x + y) % r for x, y in zip(f(a[:64]), f(b[:64]))]


def test_msm_matches_naive_and_thread_count_independent(oracle):
    CO, curve = oracle, "bls12_381"
    n = 600
    bases = CO.g1_multiples(curve, n)
    sc = rand_fr_limbs(curve, n, 3)
    one, _ = CO.msm(curve, bases, sc, 1)
    many, _ = CO.msm(curve, bases, sc, 5)
    assert np.array_equal(one, many)
    # definition: sum of individual scalar multiplications
    parts = np.stack([CO.g1_mul(curve, bases[i], sc[i])[0] for i in range(40)])
    ref, _ = CO.g1_sum(curve, parts)
    got, _ = CO.msm(curve, bases[:40], sc[:40], 2)
    assert np.array_equal(ref, got) and CO.g1_is_on_curve(curve, got)


def test_setup_and_proofs_match_pyref_golden(oracle):
    """Sparse setup + three-phase prove of the C++ restatement == dense literal transcription
    (bases, every intermediate vector, and the proof bytes for all three transcripts)."""
    CO = oracle
    for fx in load_golden("proofs.json") + load_golden("proofs_bn254.json"):
        curve = fx["curve"]
        c = CURVES[curve]
        TR = T.make_transcripts(c)
        q = r1cs_from_json(fx["r1cs"])
        pk = CO.OraclePk(curve, q, I(fx["x_trapdoor"]), I(fx["z_trapdoor"]), 2)
        assert (pk.n, pk.sigma) == (fx["n"], fx["sigma"])
        assert CO.fr_from_mont_limbs(curve, pk.omega_limbs)[0] == I(fx["omega"])
        for i, nm in enumerate(BASE_NAMES):
            assert CO.g1_from_mont_limbs(curve, pk.export_bases(i)) == [PT(p) for p in fx["bases"][nm]], (fx["name"], nm)
        inst, wit, r_a = [I(v) for v in fx["instance"]], [I(v) for v in fx["witness"]], [I(v) for v in fx["r_a"]]
        from oracle.pyref import serialize as SE
        for tname, ref in fx["proofs"].items():
            tr = {}
            proof = DR.prove(pk, pk.n, pk.sigma, I(fx["omega"]), inst, wit, r_a, TR[tname], tr)
            assert SE.ser_proof(c, proof).hex() == ref["bytes"], (fx["name"], tname)
            assert tr["x1"] == I(ref["x1"]) and tr["x2"] == I(ref["x2"])
            if tname != "keccak256":   # the fixture's trace (quotient depends on x1, x2) is the keccak run
                continue
            for which, key in [(0, "u_evals"), (1, "w_evals"), (2, "u"), (3, "w"), (4, "h"), (5, "wit_u"), (6, "z_tail"), (7, "quotient")]:
                got = CO.fr_from_mont_limbs(curve, pk.tap(which, 1 << 16))
                want = [I(v) for v in fx["trace"][key]]
                assert got[:len(want)] == want and not any(got[len(want):]), (fx["name"], key)


def test_unsatisfied_and_state_errors(oracle):
    CO = oracle
    fx = load_golden("proofs.json")[1]
    curve = fx["curve"]
    c = CURVES[curve]
    TR = T.make_transcripts(c)
    q = r1cs_from_json(fx["r1cs"])
    pk = CO.OraclePk(curve, q, 5, 7, 1)
    inst, wit = [I(v) for v in fx["instance"]], [I(v) for v in fx["witness"]]
    wit[1] = (wit[1] + 1) % c.r
    with pytest.raises(DR.ProverError) as e:
        DR.prove(pk, pk.n, pk.sigma, 1, inst, wit, [1, 2], TR["keccak256"])
    assert (e.value.phase, e.value.rc) == (1, 4)     # REMAINDER_NONZERO == prover.rs:108
    pk2 = CO.OraclePk(curve, q, 5, 7, 1)
    rc, _ = pk2.phase2(CO.fr_to_mont_limbs(curve, [3]))
    assert rc == 8                                    # phase 2 before phase 1


@pytest.mark.parametrize("curve,seed,m0,nr", [("bls12_381", 11, 1, 1), ("bls12_381", 12, 4, 4), ("bls12_381", 13, 9, 5),
                                              ("bn254", 14, 2, 7), ("bn254", 15, 17, 2)])
def test_random_r1cs_shapes_sparse_equals_dense_literal(oracle, curve, seed, m0, nr):
    """CI.random_r1cs (any m0, multi-entry rows on column 0 / instance columns, duplicate columns, zero coefficients, empty
    rows, unused witnesses; nr = 1): the sparse closed form of oracle/cpp against the arm-by-arm dense
    transcription of common.rs:138-207 in oracle/pyref -- bases, proof, all 8 intermediate vectors, challenges."""
    from oracle.pyref import circuits as CI, protocol as PR
    CO, c = oracle, CURVES[curve]
    q, inst, wit = CI.random_r1cs(c, seed, m0, nr)
    g = CI.SplitMix64(seed * 7919)
    x, z, r_a = g.fr(c.r), g.fr(c.r), [g.fr(c.r), g.fr(c.r)]
    dense = PR.generate_proving_key(c, q, x, z)
    pk = CO.OraclePk(curve, q, x, z, 2)
    assert (pk.n, pk.sigma) == (dense.n, dense.sigma)
    for i, nm in enumerate(BASE_NAMES):
        assert CO.g1_from_mont_limbs(curve, pk.export_bases(i)) == getattr(dense, nm), nm
    TR = T.make_transcripts(c)["blake3"]
    tr_d, tr_s = {}, {}
    want = PR.create_proof_with_assignment(c, dense, inst, wit, r_a, TR, tr_d)
    got = DR.prove(pk, pk.n, pk.sigma, dense.omega, inst, wit, r_a, TR, tr_s)
    assert got == want and (tr_s["x1"], tr_s["x2"]) == (tr_d["x1"], tr_d["x2"])
    for which, key in [(0, "u_evals"), (1, "w_evals"), (2, "u"), (3, "w"), (4, "h"), (5, "wit_u"), (6, "z_tail"), (7, "quotient")]:
        have = CO.fr_from_mont_limbs(curve, pk.tap(which, 1 << 16))
        assert have[:len(tr_d[key])] == tr_d[key] and not any(have[len(tr_d[key]):]), key
    # one witness value off by one: the remainder assert of prover.rs:108, if any row reads that column at all
    bad = list(wit)
    bad[0] = (bad[0] + 1) % c.r
    zz = inst + bad
    if not all(CI.first_entry_dot(c.r, a, zz) * CI.first_entry_dot(c.r, b, zz) % c.r == CI.first_entry_dot(c.r, cc, zz)
               for a, b, cc in zip(q.a, q.b, q.c)):
        with pytest.raises(DR.ProverError) as e:
            DR.prove(pk, pk.n, pk.sigma, dense.omega, inst, bad, r_a, TR)
        assert (e.value.phase, e.value.rc) == (1, 4)
