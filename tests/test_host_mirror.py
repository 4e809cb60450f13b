"""The C++ host mirror (polymath_amd/host/*.hpp: Polymath<Curve, T>::setup / prove, ConstraintSystem,
Merlin / Keccak256 / Blake3 transcripts, ark wire format) above the C ABI.

CPU: hash known-answer vectors.  GPU: the reference's own two tests restated in C++
(tests/native/host_polymath.cpp == tests/dummy.rs + tests/mimc.rs) -- the printed 176-byte proofs must
equal the CPU oracle's for the same SplitMix64 draws."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, "tests", "native")


def _build(tmp_path, src, link=False):
    exe = str(tmp_path / src.replace(".cpp", ""))
    cmd = ["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(NATIVE, src)]
    if link:
        lib = os.path.join(ROOT, "polymath_amd")
        cmd += ["-L" + lib, "-lpolymath_hip", "-Wl,-rpath," + lib]
    subprocess.check_call(cmd)
    return exe


def test_host_hash_known_answers(tmp_path):
    out = subprocess.run([_build(tmp_path, "host_selftest.cpp")], capture_output=True, text=True)
    assert out.returncode == 0 and "0 failures" in out.stdout, out.stdout
    # the random sources: C++ (host/rng.hpp) and Python (polymath_amd/rng.py) twins agree word for word
    import struct
    from polymath_amd import rng as R
    from polymath_amd.polymath import FIELDS
    key = list(struct.unpack("<8I", bytes(range(32))))                      # RFC 7539 section 2.3.2 (20 rounds)
    assert struct.pack("<16I", *R.chacha_block(key, [1, 0x09000000, 0x4A000000, 0], 20)).hex().startswith("10f1e7e4d13b5915500fdd1fa32071c4")
    # the 12-round function StdRng runs (rand 0.8: ChaCha12), pinned by a published vector: draft-strombergson-chacha-test-vectors-01,
    # TC1 (all-zero key and IV), 12 rounds, block 0
    assert struct.pack("<16I", *R.chacha_block([0] * 8, [0] * 4, 12)).hex() == (
        "9bf49a6a0755f953811fce125f2683d50429c3bb49e074147e0089a52eae155f0564f879d27ae3c02ce82834acfa8c793a629f2ca0de6919610be82f411326be")
    line = [l for l in out.stdout.splitlines() if l.startswith("rng ")][0].split()
    t0 = R.StdRng.test_rng().next_u64()
    assert line[1] == "test_rng_first=%016x" % t0
    r = R.StdRng.seed_from_u64(t0)
    assert line[3:23] == ["%016x" % r.next_u64() for _ in range(20)]
    assert line[23] == "fr_bls=%064x" % R.fr_rand_mont(r, FIELDS["bls12_381"]["r"])
    assert line[24] == "fr_bn=%064x" % R.fr_rand_mont(r, FIELDS["bn254"]["r"])


@pytest.mark.gpu
def test_cpp_dummy_and_mimc_match_oracle(tmp_path, oracle):
    from oracle import driver as DR
    from oracle.pyref import circuits as CI, serialize as SE, transcripts as T
    from oracle.pyref.fields import BLS12_381 as c
    rounds, samples = 40, 2
    out = subprocess.run([_build(tmp_path, "host_polymath.cpp", link=True), str(rounds), str(samples)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    all_lines = out.stdout.strip().splitlines()
    verdicts = [l for l in all_lines if l.startswith("verify")]
    # Polymath::verify (verifier.rs:19-62) in C++ with its own pairings (BLS12-381, then BN254): accepts, rejects tampering / wrong inputs
    assert verdicts == ["verify merlin accept=1 tampered=0 wrong_input=0", "verify keccak256 accept=1 tampered=0 wrong_input=0",
                        "verify blake3 accept=1 tampered=0 wrong_input=0", "verify mimc accept=1",
                        "verify_bn254 merlin accept=1 tampered=0 wrong_input=0", "verify_bn254 keccak256 accept=1 tampered=0 wrong_input=0",
                        "verify_bn254 blake3 accept=1 tampered=0 wrong_input=0"], verdicts
    bn_lines = [l for l in all_lines if l.startswith("dummy_bn254")]
    rng_lines = [l for l in all_lines if l.startswith("dummyrng")]
    lines = [l for l in all_lines if not l.startswith("verify") and not l.startswith("dummy_bn254") and not l.startswith("dummyrng")]
    # tests/dummy.rs:37-80 draw for draw on the reference's random sources (StdRng::seed_from_u64(test_rng().next_u64()), setup(c, rng),
    # a, b = rand, prove(pk, c, rng)): the Python twin of the RNG feeds the CPU oracle the same draws -> the same 176 bytes, accepted
    from polymath_amd import rng as R
    assert len(rng_lines) == 3
    for line, tname in zip(rng_lines, ["merlin", "keccak256", "blake3"]):
        rng = R.StdRng.seed_from_u64(R.StdRng.test_rng().next_u64())
        x, z = R.sample_element_outside_domain(rng, c.r, 8), R.sample_element_outside_domain(rng, c.r, 8)
        a, b = R.fr_rand(rng, c.r), R.fr_rand(rng, c.r)
        r_a = [R.fr_rand(rng, c.r), R.fr_rand(rng, c.r)]
        q, inst, wit = CI.dummy_circuit(c, a, b)
        opk = oracle.OraclePk("bls12_381", q, x, z, 1)
        assert opk.n == 8
        omega = oracle.fr_from_mont_limbs("bls12_381", opk.omega_limbs)[0]
        ref = DR.prove(opk, opk.n, opk.sigma, omega, inst, wit, r_a, T.make_transcripts(c)[tname])
        kind, name, hx, acc = line.split()
        assert (kind, name, acc) == ("dummyrng", tname, "accept=1") and hx == SE.ser_proof(c, ref).hex()
    # tests/dummy.rs on BN254: same proof bytes as the CPU oracle
    from oracle.pyref.fields import BN254 as cb
    TRB = T.make_transcripts(cb)
    assert len(bn_lines) == 3
    for line, (tname, seed) in zip(bn_lines, [("merlin", 201), ("keccak256", 202), ("blake3", 203)]):
        g = CI.SplitMix64(seed)
        a, b, x, z, r_a = g.fr(cb.r), g.fr(cb.r), g.fr(cb.r), g.fr(cb.r), [g.fr(cb.r), g.fr(cb.r)]
        q, inst, wit = CI.dummy_circuit(cb, a, b)
        opk = oracle.OraclePk("bn254", q, x, z, 1)
        omega = oracle.fr_from_mont_limbs("bn254", opk.omega_limbs)[0]
        ref = DR.prove(opk, opk.n, opk.sigma, omega, inst, wit, r_a, TRB[tname])
        kind, name, n, hx = line.split()
        assert (kind, name, n) == ("dummy_bn254", tname, "n=%d" % opk.n) and hx == SE.ser_proof(cb, ref).hex()
    TR = T.make_transcripts(c)
    # tests/dummy.rs
    for line, (tname, seed) in zip(lines[:3], [("merlin", 101), ("keccak256", 102), ("blake3", 103)]):
        g = CI.SplitMix64(seed)
        a, b, x, z, r_a = g.fr(c.r), g.fr(c.r), g.fr(c.r), g.fr(c.r), [g.fr(c.r), g.fr(c.r)]
        q, inst, wit = CI.dummy_circuit(c, a, b)
        opk = oracle.OraclePk("bls12_381", q, x, z, 1)
        omega = oracle.fr_from_mont_limbs("bls12_381", opk.omega_limbs)[0]
        ref = DR.prove(opk, opk.n, opk.sigma, omega, inst, wit, r_a, TR[tname])
        kind, name, n, hx = line.split()
        assert (kind, name, n) == ("dummy", tname, "n=%d" % opk.n) and hx == SE.ser_proof(c, ref).hex()
    # tests/mimc.rs
    g = CI.SplitMix64(322)
    consts = [g.fr(c.r) for _ in range(rounds)]
    x, z = g.fr(c.r), g.fr(c.r)
    opk = None
    for s in range(samples):
        xl, xr, r_a = g.fr(c.r), g.fr(c.r), [g.fr(c.r), g.fr(c.r)]
        q, inst, wit = CI.mimc_circuit(c, xl, xr, consts)
        if opk is None:
            opk = oracle.OraclePk("bls12_381", q, x, z, 4)
            omega = oracle.fr_from_mont_limbs("bls12_381", opk.omega_limbs)[0]
        ref = DR.prove(opk, opk.n, opk.sigma, omega, inst, wit, r_a, TR["merlin"])
        kind, idx, n, hx = lines[3 + s].split()
        assert (kind, idx) == ("mimc", str(s)) and hx == SE.ser_proof(c, ref).hex()
    assert lines[3 + samples] == "bad-witness rejected phase=1 status=4"    # == assert!(rem_poly.is_zero()), prover.rs:108
