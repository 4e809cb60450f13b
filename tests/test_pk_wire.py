"""ProvingKey / VerifyingKey wire format (data_structures.rs:25-73, common.rs:112-127).

CPU: the big-integer restatement against the committed fixture and public known-answer encodings, and the
Python twin's codec (pure host code) against the same bytes.  GPU: a key generated on the device serialises
to the fixture bytes, and a key LOADED from the fixture bytes reproduces the fixture proofs.
PARITY UNPINNED w.r.t. ark-serialize itself (no Rust here): the generator encodings below are the published
zcash test values, everything else follows SURVEY.md App. C.
"""
import pytest

from helpers import I, PT, load_golden, r1cs_from_json
from oracle.pyref import pairing as PA, protocol as PR, serialize as SE, transcripts as T
from oracle.pyref.fields import BLS12_381 as C381, BLS12_381_G2

# compressed generators as published with the BLS12-381 / zcash serialisation spec
G1_GEN = "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
G2_GEN = ("93e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
          "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8")


def _g2(j):
    return ((I(j[0][0]), I(j[0][1])), (I(j[1][0]), I(j[1][1])))


def test_point_encodings_known_answers():
    assert SE.ser_g1(C381, C381.g1).hex() == G1_GEN
    assert SE.ser_g2(C381, BLS12_381_G2).hex() == G2_GEN
    assert SE.deser_g1(C381, bytes.fromhex(G1_GEN)) == C381.g1
    assert SE.deser_g2(C381, bytes.fromhex(G2_GEN)) == BLS12_381_G2
    assert SE.ser_g1(C381, None).hex() == "c0" + "00" * 47 and SE.deser_g1(C381, bytes.fromhex("c0" + "00" * 47)) is None
    assert SE.ser_g2(C381, None).hex() == "c0" + "00" * 95 and SE.deser_g2(C381, bytes.fromhex("c0" + "00" * 95)) is None
    # the sign flag: -G encodes with the other root
    neg = (C381.g1[0], C381.p - C381.g1[1])
    b = SE.ser_g1(C381, neg)
    assert b[0] & 0x20 and not bytes.fromhex(G1_GEN)[0] & 0x20 and SE.deser_g1(C381, b) == neg
    nq = PA.g2_neg(BLS12_381_G2)
    assert SE.deser_g2(C381, SE.ser_g2(C381, nq)) == nq and SE.ser_g2(C381, nq) != bytes.fromhex(G2_GEN)


def test_pyref_pk_bytes_match_fixture_and_round_trip():
    fx = load_golden("pk_wire.json")
    assert fx["kats"]["g1_generator"] == G1_GEN and fx["kats"]["g2_generator"] == G2_GEN
    proofs = {e["name"]: e for e in load_golden("proofs.json")}
    for key in fx["keys"]:
        e = proofs[key["name"]]
        q = r1cs_from_json(e["r1cs"])
        pk = PR.generate_proving_key(C381, q, I(e["x_trapdoor"]), I(e["z_trapdoor"]))
        vk = PA.make_vk(pk)
        assert vk["x_g2"] == _g2(key["vk"]["x_g2"]) and vk["z_g2"] == _g2(key["vk"]["z_g2"])
        data = SE.ser_pk(C381, pk, vk)
        assert data.hex() == key["pk_bytes"] and SE.ser_vk(C381, vk).hex() == key["vk"]["bytes"]
        assert data.startswith(SE.ser_vk(C381, vk))                    # vk is the key's first field
        vk2, sap, vecs = SE.deser_pk(C381, data)
        assert vk2 == vk and sap == (q.m0, q.mw, q.nr, q.a, q.b, q.c)  # matrices as synthesised, duplicates kept
        for nm in SE.PK_VECTORS:
            assert vecs[nm] == [PT(p) for p in e["bases"][nm]], nm
        with pytest.raises(AssertionError):
            SE.deser_pk(C381, data[:-1])
        with pytest.raises(AssertionError):
            SE.deser_pk(C381, data + b"\x00")


def test_twin_codec_matches_pyref():
    """polymath_amd.polymath's host-side codec (no GPU needed for the point / vk / matrix parts)."""
    from polymath_amd import polymath as PM
    f = PM.Field("bls12_381")
    assert PM.ser_g2(f, BLS12_381_G2).hex() == G2_GEN and PM.deser_g2(f, bytes.fromhex(G2_GEN)) == BLS12_381_G2
    assert PM.deser_g1(f, bytes.fromhex(G1_GEN)) == C381.g1
    for key in load_golden("pk_wire.json")["keys"]:
        data = bytes.fromhex(key["pk_bytes"])
        vk = PM.VerifyingKey.read(f, PM._Reader(data))
        assert vk.to_bytes().hex() == key["vk"]["bytes"]
        assert (vk.x_g2, vk.z_g2) == (_g2(key["vk"]["x_g2"]), _g2(key["vk"]["z_g2"]))
    with pytest.raises(ValueError):
        PM.deser_g1(f, bytes([0x80]) + bytes(46) + b"\x01")           # x = 1: x^3 + 4 is a non-residue
    with pytest.raises(ValueError):
        PM.deser_g1(f, bytes(48))                                     # uncompressed flag


@pytest.mark.gpu
def test_gpu_key_serialises_to_fixture_and_loaded_key_proves(gpu_ctx):
    from polymath_amd import polymath as PM
    proofs = {e["name"]: e for e in load_golden("proofs.json")}
    pm = PM.Polymath("bls12_381", "merlin", ctx=gpu_ctx)
    f = pm.field
    for key in load_golden("pk_wire.json")["keys"]:
        e = proofs[key["name"]]
        q = r1cs_from_json(e["r1cs"])
        r1cs = PM.R1CS(q.m0, q.mw, q.a, q.b, q.c)
        inst, wit, r_a = [I(v) for v in e["instance"]], [I(v) for v in e["witness"]], [I(v) for v in e["r_a"]]
        pk = pm.setup((r1cs, inst, wit), I(e["x_trapdoor"]), I(e["z_trapdoor"]))
        vk = PM.VerifyingKey(f, C381.g1, BLS12_381_G2, _g2(key["vk"]["x_g2"]), _g2(key["vk"]["z_g2"]), pk.n, pk.m0, pk.sigma, pk.omega)
        assert pm.pk_to_bytes(pk, r1cs, vk).hex() == key["pk_bytes"], key["name"]
        pk.free()
        # the other direction: bytes -> pm_pk_load -> the fixture's proofs, all three transcripts
        for tname, ref in e["proofs"].items():
            pm_t = PM.Polymath("bls12_381", tname, ctx=gpu_ctx)
            pk2, vk2, r2 = pm_t.pk_from_bytes(bytes.fromhex(key["pk_bytes"]))
            assert (vk2.n, vk2.m0, vk2.sigma, vk2.omega) == (e["n"], q.m0, e["sigma"], I(e["omega"]))
            assert (r2.a, r2.b, r2.c) == (q.a, q.b, q.c)
            proof = pm_t.prove_limbs(pk2, inst, f.fr_limbs(inst), f.fr_limbs(wit), r_a)
            assert proof.to_bytes().hex() == ref["bytes"], (key["name"], tname)
            pk2.free()


# ------------------------------------------------------------------ C++ host mirror (polymath_amd/host/wire.hpp)
def _build_native(tmp_path, src, link=False):
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / src.replace(".cpp", ""))
    cmd = ["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(root, "tests", "native", src)]
    if link:
        lib = os.path.join(root, "polymath_amd")
        cmd += ["-L" + lib, "-lpolymath_hip", "-Wl,-rpath," + lib]
    subprocess.check_call(cmd)
    return exe


def _hex_files(tmp_path):
    files = {}
    for key in load_golden("pk_wire.json")["keys"]:
        p = tmp_path / ("pk_%s.hex" % key["name"])
        p.write_text(key["pk_bytes"])
        files[key["name"]] = str(p)
    return files


def test_cpp_codec_round_trips_fixture(tmp_path):
    """WireKey::parse (decompression: Fq and Fq2 square roots) -> to_bytes is the identity on the fixture keys;
    generator known answers, sign flags, off-curve / truncated / trailing input rejected.  CPU only."""
    import subprocess
    files = _hex_files(tmp_path)
    out = subprocess.run([_build_native(tmp_path, "wire_selftest.cpp")] + list(files.values()), capture_output=True, text=True)
    assert out.returncode == 0 and "wire selftest: 0 failures" in out.stdout, out.stdout + out.stderr
    assert out.stdout.count("wire ok") == len(files)


@pytest.mark.gpu
def test_cpp_loaded_key_proves_and_exports(tmp_path):
    """C++: bytes -> WireKey -> pm_pk_load -> Polymath::prove_with_assignment == the fixture proof (Merlin), and
    pm_pk_export_bases -> WireKey -> bytes is the identity."""
    import subprocess
    files = _hex_files(tmp_path)
    exe = _build_native(tmp_path, "host_wire.cpp", link=True)
    proofs = {e["name"]: e for e in load_golden("proofs.json")}
    for name, path in files.items():
        e = proofs[name]
        asg = tmp_path / ("asg_%s.txt" % name)
        asg.write_text("inst %s\nwit %s\nra %s\n" % (" ".join(e["instance"]), " ".join(e["witness"]), " ".join(e["r_a"])))
        out = subprocess.run([exe, path, str(asg)], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        lines = out.stdout.strip().splitlines()
        assert lines[0] == "proof " + e["proofs"]["merlin"]["bytes"], name
        assert lines[1] == "export identical=1", name
        assert lines[2] == "inconsistent keys rejected=3", name
