"""The library's own verifier and verifying-key builder through the C ABI (pm_host_verify / pm_host_make_vk: Polymath::verify,
lib.rs:80-90 -> verifier.rs:19-62; generator.rs:139-157) -- host code, so these run WITHOUT a GPU.  Every committed fixture
proof (5 BLS12-381 + 3 BN254 circuits x 3 transcripts, tests/golden/) must be accepted by the product's C++ pairing code,
tampering / wrong inputs rejected, malformed bytes refused; the vk bytes must equal the oracle's independent serialisation."""
import numpy as np
import pytest

from helpers import I, load_golden
from oracle import cpp_oracle as CO
from oracle.pyref import pairing as PA, serialize as SE
from oracle.pyref.fields import CURVES


def _fixture_vk(api, fx):
    curve = fx["curve"]
    omega = CO.fr_to_mont_limbs(curve, [I(fx["omega"])])[0]
    x, z = CO.fr_to_mont_limbs(curve, [I(fx["x_trapdoor"])])[0], CO.fr_to_mont_limbs(curve, [I(fx["z_trapdoor"])])[0]
    return api.make_vk(curve, fx["n"], fx["r1cs"]["m0"], fx["sigma"], omega, x, z)


@pytest.mark.parametrize("name", ["proofs.json", "proofs_bn254.json"])
def test_product_verifier_accepts_every_golden_proof(name):
    from polymath_amd import api
    for fx in load_golden(name):
        curve = fx["curve"]
        c = CURVES[curve]
        vk = _fixture_vk(api, fx)
        assert len(vk) == (392 if curve == "bls12_381" else 280)
        # the same bytes from the oracle's G2 arithmetic and serialiser (independent of the C++ code)
        E = PA.ENGINES[curve]
        ovk = E.make_vk_from_trapdoors(fx["n"], fx["r1cs"]["m0"], fx["sigma"], I(fx["omega"]), I(fx["x_trapdoor"]), I(fx["z_trapdoor"]))
        assert vk == SE.ser_vk(c, ovk), fx["name"]
        pub = CO.fr_to_mont_limbs(curve, [I(v) for v in fx["instance"][1:]])
        first = fx is load_golden(name)[0] or fx["name"] == load_golden(name)[0]["name"]
        full = first or fx["r1cs"]["m0"] != 2         # the m0 = 1, 3, 12 shapes: pi(x1) over 2 m0 Lagrange terms (common.rs:49-71)
        for tname, ref in fx["proofs"].items():
            if not full and tname != "merlin":        # every fixture with Merlin, the first one and m0 != 2 with all three (a check is ~2 s of CPU)
                continue
            proof = bytes.fromhex(ref["bytes"])
            assert api.verify(curve, tname, vk, pub, proof), (fx["name"], tname)
        if not full:
            continue
        # tampering: a_at_x1 + 1 (bytes 2 x |G1| ..), a wrong public input, another transcript
        g1 = 48 if curve == "bls12_381" else 32
        bad = bytearray(proof)
        bad[2 * g1] ^= 1
        assert not api.verify(curve, tname, vk, pub, bytes(bad))
        if len(fx["instance"]) > 1:
            for k in {0, len(fx["instance"]) - 2}:          # the first and the last public input, one at a time
                vals = [I(v) for v in fx["instance"][1:]]
                vals[k] = (vals[k] + 1) % c.r
                assert not api.verify(curve, tname, vk, CO.fr_to_mont_limbs(curve, vals), proof)
        assert not api.verify(curve, "merlin" if tname != "merlin" else "keccak256", vk, pub, proof)


def test_product_verifier_refuses_malformed_bytes():
    from polymath_amd import api
    fx = load_golden("proofs.json")[0]
    vk = _fixture_vk(api, fx)
    pub = CO.fr_to_mont_limbs("bls12_381", [I(v) for v in fx["instance"][1:]])
    proof = bytes.fromhex(fx["proofs"]["merlin"]["bytes"])
    for bad_vk, bad_proof in [(vk[:-1], proof), (vk, proof[:-1]), (vk, proof + b"\0"), (vk, b"\xff" * len(proof)), (b"\0" * len(vk), proof)]:
        with pytest.raises(api.PolymathError):
            api.verify("bls12_381", "merlin", bad_vk, pub, bad_proof)
    off_curve = bytearray(proof)           # x of a_g1 moved off the curve (or onto another point): refused or rejected, never accepted
    off_curve[47] ^= 0x01
    try:
        assert not api.verify("bls12_381", "merlin", vk, pub, bytes(off_curve))
    except api.PolymathError:
        pass
    # ark's deserialize_compressed validates (Validate::Yes): points ON the curve but OUTSIDE the prime-order subgroup are
    # refused -- the pairing is undefined there -- and so are non-canonical encodings of the point at infinity
    from oracle.pyref import fields as F
    c = CURVES["bls12_381"]
    x = 1
    while True:                         # a random curve point lies in G1 with probability 1/h ~ 2^-126: take the first x that works
        y2 = (x * x * x + 4) % c.p
        y = pow(y2, (c.p + 1) // 4, c.p)
        if y * y % c.p == y2 and F.g1_add(c, F.g1_mul(c, (x, y), c.r - 1), (x, y)) is not None:     # [r] P != O
            break
        x += 1
    enc = bytearray(x.to_bytes(48, "big"))
    enc[0] |= 0x80 | (0x20 if y > c.p - y else 0)
    outside = bytes(enc) + proof[48:]
    with pytest.raises(api.PolymathError):
        api.verify("bls12_381", "merlin", vk, pub, outside)
    with pytest.raises(api.PolymathError):          # the same point as the vk's one_g1
        api.verify("bls12_381", "merlin", bytes(enc) + vk[48:], pub, proof)
    inf_ok = bytes([0xC0]) + bytes(47)
    for dirty in (bytes([0xE0]) + bytes(47), bytes([0xC0]) + bytes(46) + b"\x01", bytes([0xC0, 0x01]) + bytes(46)):
        with pytest.raises(api.PolymathError):
            api.verify("bls12_381", "merlin", vk, pub, dirty + proof[48:])
    assert api.verify("bls12_381", "merlin", vk, pub, inf_ok + proof[48:]) is False        # canonical infinity parses; the proof is wrong
    # a G2 point of the vk on the twist but outside G2 (x_g2 is bytes 144..240): refused, as is a dirty G2 infinity
    g2_inf_dirty = bytes([0xC0]) + bytes(94) + b"\x01"
    with pytest.raises(api.PolymathError):
        api.verify("bls12_381", "merlin", vk[:144] + g2_inf_dirty + vk[240:], pub, proof)
    twisted = bytearray(vk[144:240])
    for tweak in range(1, 13):          # walk x.c0 until the candidate is on the twist: it is then outside G2 (cofactor ~ 2^508)
        cand = bytearray(twisted)
        cand[95] = (cand[95] + tweak) & 0xFF
        try:
            api.verify("bls12_381", "merlin", vk[:144] + bytes(cand) + vk[240:], pub, proof)
        except api.PolymathError:
            continue
        raise AssertionError("a tampered G2 x-coordinate was parsed as a member of G2")
    # BN254 (ark-serialize's SWFlags in the top bits of the last byte): both flag bits set is not an encoding, a dirty
    # infinity is refused, the canonical one parses
    fxb = load_golden("proofs_bn254.json")[0]
    vkb = _fixture_vk(api, fxb)
    pubb = CO.fr_to_mont_limbs("bn254", [I(v) for v in fxb["instance"][1:]])
    pb = bytes.fromhex(fxb["proofs"]["merlin"]["bytes"])
    assert api.verify("bn254", "merlin", vkb, pubb, pb)
    both = bytearray(pb)
    both[31] |= 0xC0
    for bad_a in (bytes(both[:32]), b"\x01" + bytes(30) + b"\x40", bytes(31) + b"\xC0"):
        with pytest.raises(api.PolymathError):
            api.verify("bn254", "merlin", vkb, pubb, bad_a + pb[32:])
    assert api.verify("bn254", "merlin", vkb, pubb, bytes(31) + b"\x40" + pb[32:]) is False


@pytest.mark.gpu
def test_product_verifier_on_gpu_proofs_both_curves():
    """setup(circuit, rng) -> prove(pk, circuit, rng) -> Polymath.verify: the reference's three calls (tests/dummy.rs:52-72) end
    to end on the product alone -- GPU prover, C++ verifier -- on both pairing engines and all three transcripts."""
    from polymath_amd import circuits as PC, rng as R
    from polymath_amd.polymath import Polymath
    for curve in ("bls12_381", "bn254"):
        c = CURVES[curve]
        for tname in ("merlin", "keccak256", "blake3"):
            rng = R.StdRng.seed_from_u64(R.StdRng.test_rng().next_u64())
            pm = Polymath(curve, tname, device=0)
            a, b = R.fr_rand(rng, c.r), R.fr_rand(rng, c.r)
            circuit = PC.MiMCDemo(a, b, [R.fr_rand(rng, c.r) for _ in range(16)])
            pk = pm.setup(circuit, rng)
            vk = pm.make_vk(pk, *pm.last_trapdoors)
            proof = pm.prove(pk, circuit, rng)
            image = pm._synthesize(circuit)[1][1:]
            assert pm.verify(vk, image, proof)
            assert not pm.verify(vk, [(image[0] + 1) % c.r], proof)
            bad = bytearray(proof.to_bytes())
            bad[2 * (48 if curve == "bls12_381" else 32)] ^= 1          # a_at_x1 +- 1: still a canonical scalar, no longer the evaluation
            assert not pm.verify(vk, image, bytes(bad))
            pk.free()
            pm.ctx.close()
