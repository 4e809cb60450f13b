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
        for tname, ref in fx["proofs"].items():
            if not first and tname != "merlin":       # every fixture with Merlin, the first one with all three (a check is ~2 s of CPU)
                continue
            proof = bytes.fromhex(ref["bytes"])
            assert api.verify(curve, tname, vk, pub, proof), (fx["name"], tname)
        if not first:
            continue
        # tampering: a_at_x1 + 1 (bytes 2 x |G1| ..), a wrong public input, another transcript
        g1 = 48 if curve == "bls12_381" else 32
        bad = bytearray(proof)
        bad[2 * g1] ^= 1
        assert not api.verify(curve, tname, vk, pub, bytes(bad))
        wrong = CO.fr_to_mont_limbs(curve, [(I(v) + 1) % c.r for v in fx["instance"][1:]])
        assert not api.verify(curve, tname, vk, wrong, proof)
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
