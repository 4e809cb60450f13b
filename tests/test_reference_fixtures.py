"""Fixtures written by the REAL reference (tools/ark_crosscheck: sigma0-polymath on arkworks, run on a machine that has
cargo) -- the pin for "parity unpinned" (SURVEY.md §8c).  tests/golden/ref_*.json hold, per circuit: the R1CS the reference
synthesised, the assignment, the trapdoors and r_a it drew, its proof bytes under the three transcripts, its vk bytes and
(small circuits) its whole serialised ProvingKey.

  CPU  : the oracle (oracle/pyref) must reproduce every byte  -> pins the oracle, and with it every golden vector;
  GPU  : the HIP path through the C ABI must reproduce every byte -> pins the product directly.

No such file can be produced in the build image (no Rust toolchain): without them these tests SKIP, visibly, and parity
stays "unpinned"."""
import glob
import os

import numpy as np
import pytest

from helpers import GOLDEN, I, load_golden, r1cs_from_json
from oracle.pyref.fields import CURVES

REF_FILES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "ref_*.json")) if "cpu_baseline" not in p)
NEEDS = "no tests/golden/ref_*.json: run tools/ark_crosscheck/run.sh on a machine with cargo (parity stays UNPINNED until then)"


def _fixtures():
    return [fx for name in REF_FILES for fx in load_golden(name)]


def _check_oracle(fixtures):
    from oracle.pyref import pairing as PA, protocol as PR, serialize as SE, transcripts as T
    for fx in fixtures:
        c = CURVES[fx["curve"]]
        q = r1cs_from_json(fx["r1cs"])
        inst, wit = [I(v) for v in fx["instance"]], [I(v) for v in fx["witness"]]
        x, z, r_a = I(fx["x_trapdoor"]), I(fx["z_trapdoor"]), [I(v) for v in fx["r_a"]]
        pk = PR.generate_proving_key(c, q, x, z)
        assert (pk.n, pk.sigma, pk.omega) == (fx["n"], fx["sigma"], I(fx["omega"])), fx["name"]
        for name, pts in fx["bases"].items():
            got = getattr(pk, name)
            assert [None if p is None else (I(p[0]), I(p[1])) for p in pts] == list(got), (fx["name"], name)
        vk = PA.ENGINES[fx["curve"]].make_vk_from_trapdoors(pk.n, q.m0, pk.sigma, pk.omega, x, z)
        assert SE.ser_vk(c, vk).hex() == fx["vk_bytes"], fx["name"]
        if "pk_bytes" in fx:
            assert SE.ser_pk(c, pk, vk).hex() == fx["pk_bytes"], fx["name"]
        for tname, ref in fx["proofs"].items():
            proof = PR.create_proof_with_assignment(c, pk, inst, wit, r_a, T.make_transcripts(c)[tname])
            assert SE.ser_proof(c, proof).hex() == ref["bytes"], (fx["name"], tname)


@pytest.mark.skipif(not REF_FILES, reason=NEEDS)
def test_oracle_reproduces_the_reference_bytes():
    """oracle/pyref: setup from the reference's trapdoors == the reference's bases and vk bytes; prove with its r_a == its proof
    bytes, for all three transcripts; serialised ProvingKey == pk_bytes when the fixture carries it."""
    _check_oracle(_fixtures())


def test_the_checker_itself_on_a_fixture_in_the_reference_schema():
    """The harness, not parity: the committed `dummy` fixture (made by oracle/pyref) put into the schema tools/ark_crosscheck
    writes -- `vk_bytes`, `pk_bytes` next to the proof bytes -- goes through the same checker, so that the first real
    ref_*.json meets code that has run.  It proves nothing about the reference."""
    fx = dict(next(f for f in load_golden("proofs.json") if f["name"] == "dummy"))
    key = next(k for k in load_golden("pk_wire.json")["keys"] if k["name"] == "dummy")
    fx["vk_bytes"], fx["pk_bytes"], fx["source"] = key["vk"]["bytes"], key["pk_bytes"], "oracle/pyref (self-made: harness check only)"
    _check_oracle([fx])


@pytest.mark.gpu
@pytest.mark.skipif(not REF_FILES, reason=NEEDS)
def test_gpu_reproduces_the_reference_bytes():
    """The product: pm_pk_generate from the reference's trapdoors, pm_host_prove with its r_a -> the reference's proof bytes;
    pm_host_make_vk -> its vk bytes; the key exported from HBM serialises to the reference's pk bytes; and the reference's
    own serialised key, loaded through the wire format (pm_pk_load), proves the same bytes."""
    from polymath_amd import polymath as PM
    for fx in _fixtures():
        curve = fx["curve"]
        q = r1cs_from_json(fx["r1cs"])
        r1cs = PM.R1CS(q.m0, q.mw, q.a, q.b, q.c)
        inst, wit = [I(v) for v in fx["instance"]], [I(v) for v in fx["witness"]]
        x, z, r_a = I(fx["x_trapdoor"]), I(fx["z_trapdoor"]), [I(v) for v in fx["r_a"]]
        for tname, ref in fx["proofs"].items():
            pm = PM.Polymath(curve, tname, device=0)
            f = pm.field
            pk = pm.setup((r1cs, inst, wit), x, z)
            assert (pk.n, pk.sigma, pk.omega) == (fx["n"], fx["sigma"], I(fx["omega"])), fx["name"]
            assert pm.prove_native(pk, f.fr_limbs(inst), f.fr_limbs(wit), r_a).hex() == ref["bytes"], (fx["name"], tname)
            assert pm.make_vk(pk, x, z).hex() == fx["vk_bytes"], fx["name"]
            if "pk_bytes" in fx and tname == "merlin":
                data = bytes.fromhex(fx["pk_bytes"])
                vk = PM.VerifyingKey.read(f, PM._Reader(data))
                assert pm.pk_to_bytes(pk, r1cs, vk).hex() == fx["pk_bytes"], fx["name"]
                pk2, _vk2, _r2 = pm.pk_from_bytes(data)
                assert pm.prove_native(pk2, f.fr_limbs(inst), f.fr_limbs(wit), r_a).hex() == ref["bytes"], (fx["name"], "reference key bytes")
                pk2.free()
            pk.free()


def test_reference_fixture_status_is_reported():
    """Always runs: says in the test log whether the pin exists.  Not a pass/fail of parity."""
    if not REF_FILES:
        pytest.skip(NEEDS)
    assert all(fx["source"].startswith("sigma0-polymath") for fx in _fixtures())
