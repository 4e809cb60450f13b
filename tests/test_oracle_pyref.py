"""CPU: pins the big-integer restatement (oracle/pyref) by published known-answer vectors and by
the protocol's own algebra.  The reference ships no golden vectors (tests/dummy.rs:69-72 and
tests/mimc.rs:214 only check verify(prove(..)) == true), so this is what anchors everything else."""
import hashlib

from oracle.pyref import circuits as CI, pairing as PA, protocol as PR, transcripts as T
from oracle.pyref.fields import BLS12_381, BLS12_381_G2, BN254, CURVES, g1_is_on_curve, g1_mul, g1_neg


def test_curve_constants():
    for c in CURVES.values():
        assert g1_is_on_curve(c, c.g1)
        assert g1_mul(c, c.g1, c.r) is None and g1_mul(c, c.g1, c.r - 1) == g1_neg(c, c.g1)
        s = c.two_adicity
        assert (c.r - 1) % (1 << s) == 0 and ((c.r - 1) >> s) % 2 == 1
        assert pow(c.two_adic_root, 1 << s, c.r) == 1 and pow(c.two_adic_root, 1 << (s - 1), c.r) == c.r - 1
    # SURVEY.md App. B values recomputed independently there
    assert BLS12_381.two_adic_root == 10238227357739495823651030575849232062558860180284477541189508159991286009131
    assert BLS12_381.fr_R == 0x1824B159ACC5056F998C4FEFECBC4FF55884B7FA0003480200000001FFFFFFFE
    assert (-pow(BLS12_381.r, -1, 1 << 64)) % (1 << 64) == 0xFFFFFFFEFFFFFFFF
    assert (-pow(BLS12_381.p, -1, 1 << 64)) % (1 << 64) == 0x89F3FFFCFFFCFFFD
    assert PA.g2_is_on_curve(BLS12_381_G2) and PA.g2_mul(BLS12_381_G2, BLS12_381.r - 1) == PA.g2_neg(BLS12_381_G2)
    assert BN254.r.bit_length() == 254 and BN254.p.bit_length() == 254


def test_hash_known_answers():
    assert T.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    for msg in (b"", b"abc", b"x" * 135, b"y" * 136, b"z" * 1000):
        assert T.sha3_256(msg) == hashlib.sha3_256(msg).digest()      # same permutation + sponge, other padding
    assert T.blake3(b"").hex() == "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262"
    pat = lambda n: bytes(i % 251 for i in range(n))                  # BLAKE3 official test-vector input pattern
    assert T.blake3(pat(1025)).hex().startswith("d00278ae47eb27b34faecf67b4fe263f")
    assert T.blake3(pat(2049)).hex().startswith("5f4d72f40d7a5f82b15ca2b2e44b1de3")


def test_merlin_equivalence_vector():
    # merlin 3.0.0 src/transcript.rs `equivalence_simple`
    m = T.MerlinTranscriptRaw(b"test protocol")
    m.append_message(b"some label", b"some data")
    assert m.challenge_bytes(b"challenge", 32).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"


def test_pairing_bilinear():
    c = BLS12_381
    a, b = 0x1234567, 0x89ABCDEF01
    P, Q = g1_mul(c, c.g1, a), PA.g2_mul(BLS12_381_G2, b)
    # e(aG1, bG2) * e(-(ab)G1, G2) == 1
    assert PA.pairing_product_is_one([(P, Q), (g1_neg(c, g1_mul(c, c.g1, a * b)), BLS12_381_G2)])
    assert not PA.pairing_product_is_one([(P, Q), (g1_neg(c, g1_mul(c, c.g1, a * b + 1)), BLS12_381_G2)])


def test_pairing_bilinear_bn254():
    """BN254 optimal-ate pairing (BASELINE configs[4] acceptance oracle): curve parameters, G2 generator on the twist
    and of order r, bilinear, non-degenerate."""
    from oracle.pyref import fields as F
    c, E, G2 = BN254, PA.ENGINES["bn254"], F.BN254_G2
    x = F.BN254_X
    assert 36 * x**4 + 36 * x**3 + 24 * x**2 + 6 * x + 1 == c.p and 36 * x**4 + 36 * x**3 + 18 * x**2 + 6 * x + 1 == c.r
    assert E.g2_is_on_curve(G2) and E.g2_mul(G2, c.r - 1) == E.g2_neg(G2)
    a, b = 0x1234567, 0x89ABCDEF01
    P, Q = g1_mul(c, c.g1, a), E.g2_mul(G2, b)
    assert E.pairing_product_is_one([(P, Q), (g1_neg(c, g1_mul(c, c.g1, a * b)), G2)])
    assert not E.pairing_product_is_one([(P, Q), (g1_neg(c, g1_mul(c, c.g1, a * b + 1)), G2)])
    assert not E.pairing_product_is_one([(c.g1, G2)])                      # e(G1, G2) != 1
    assert E.pairing_product_is_one([(c.g1, G2), (g1_neg(c, c.g1), G2)])


def test_bn254_golden_proofs_pass_the_pairing_verifier():
    """Every committed BN254 fixture proof (tests/golden/proofs_bn254.json, 5 circuits x 3 transcripts; m0 = 1, 2 and 12) is accepted by
    verify_proof (verifier.rs:19-62) over the BN254 pairing; tampering and a wrong public input are rejected."""
    import json, os
    c, E = BN254, PA.ENGINES["bn254"]
    TR = T.make_transcripts(c)
    I = lambda s: int(s, 16)
    PT = lambda p: None if p is None else (I(p[0]), I(p[1]))
    for fx in json.load(open(os.path.join(os.path.dirname(__file__), "golden", "proofs_bn254.json"))):
        vk = E.make_vk_from_trapdoors(fx["n"], fx["r1cs"]["m0"], fx["sigma"], I(fx["omega"]), I(fx["x_trapdoor"]), I(fx["z_trapdoor"]))
        inst = [I(v) for v in fx["instance"]]
        for tname, ref in fx["proofs"].items():
            proof = dict(a_g1=PT(ref["a_g1"]), c_g1=PT(ref["c_g1"]), a_at_x1=I(ref["a_at_x1"]), d_g1=PT(ref["d_g1"]))
            assert PR.verify_proof(c, vk, proof, inst[1:], TR[tname], E.pairing_check), (fx["name"], tname)
        assert not PR.verify_proof(c, vk, dict(proof, a_at_x1=(proof["a_at_x1"] + 1) % c.r), inst[1:], TR[tname], E.pairing_check)
        if len(inst) > 1:                # m0 = 1 has no public input to get wrong
            assert not PR.verify_proof(c, vk, proof, [(v + 1) % c.r for v in inst[1:]], TR[tname], E.pairing_check)
        else:                            # ... but one too many is refused
            assert not PR.verify_proof(c, vk, proof, [5], TR[tname], E.pairing_check)


def test_dummy_prove_verify_all_transcripts():
    """tests/dummy.rs:75-80 restated: setup -> prove -> verify accepts, for the three transcripts;
    plus the negative cases the reference lacks."""
    c = BLS12_381
    TR = T.make_transcripts(c)
    g = CI.SplitMix64(42)
    q, inst, wit = CI.dummy_circuit(c, g.fr(c.r), g.fr(c.r))
    pk = PR.generate_proving_key(c, q, g.fr(c.r), g.fr(c.r))
    vk = PA.make_vk(pk)
    for name in ("merlin", "keccak256", "blake3"):
        tr = {}
        proof = PR.create_proof_with_assignment(c, pk, inst, wit, [g.fr(c.r), g.fr(c.r)], TR[name], tr)
        assert all(u * u % c.r == w for u, w in zip(tr["u_evals"], tr["w_evals"]))   # SAP identity (Uz)^2 = Wz
        assert PR.verify_proof(c, vk, proof, inst[1:], TR[name], PA.pairing_check)
    bad = dict(proof, a_at_x1=(proof["a_at_x1"] + 1) % c.r)
    assert not PR.verify_proof(c, vk, bad, inst[1:], TR[name], PA.pairing_check)
    assert not PR.verify_proof(c, vk, proof, [(inst[1] + 1) % c.r], TR[name], PA.pairing_check)


def test_mimc_circuit_shape_and_native():
    """tests/mimc.rs: shape m0=2, mw=645, nr=644 at 322 rounds; native hash agrees."""
    c = BLS12_381
    g = CI.SplitMix64(5)
    consts = [g.fr(c.r) for _ in range(322)]
    xl, xr = g.fr(c.r), g.fr(c.r)
    q, inst, wit = CI.mimc_circuit(c, xl, xr, consts)
    assert (q.m0, q.mw, q.nr) == (2, 645, 644)
    assert inst[1] == CI.mimc_native(c, xl, xr, consts)
    assert CI.r1cs_is_satisfied(c, q, inst, wit)


def test_unsatisfied_witness_trips_remainder_assert():
    c = BLS12_381
    q, inst, wit = CI.dummy_circuit(c, 3, 5)
    pk = PR.generate_proving_key(c, q, 11, 13)
    TR = T.make_transcripts(c)
    import pytest
    with pytest.raises(AssertionError, match="REMAINDER_NONZERO"):
        PR.create_proof_with_assignment(c, pk, inst, [3, 6], [1, 2], TR["keccak256"])
