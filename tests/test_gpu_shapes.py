"""GPU parity on R1CS SHAPES: the reference accepts any ConstraintSynthesizer (lib.rs:52-91) and
its SAP matrices have arms for every m0 (common.rs:77-97, 138-207; prover.rs:156-166 takes the witness-only part by column
>= m0) -- the harness circuits and the synthetic gates all have m0 = 2 and one entry per row.  Here: m0 = 1 ... 20, both
branches of the witness-only part of u (the direct 2 m0-term sum and the fifth transform), rows with several entries on
column 0 and on instance columns, duplicate columns, zero coefficients, empty rows, unused witnesses, nr = 1 and domains filled
exactly / to two rows short / two rows over.  Everything through the C ABI, bit-compared with oracle/cpp (and, in
test_gpu_parity.test_golden_setup_prove_bytes, with the dense big-integer fixtures m0_1 / m0_3 / m0_12)."""
import numpy as np
import pytest

from helpers import I, load_golden, r1cs_from_json
from oracle import driver as DR
from oracle.pyref import circuits as CI, serialize as SE, transcripts as T
from oracle.pyref.fields import CURVES
from test_gpu_parity import _prove_both
from test_sharded_vector import _oracle_reference, _run_ranks, _sharded_proofs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from polymath_amd import api as _api
    return _api


def _limb_circuit(curve, q, inst, wit):
    from polymath_amd.polymath import Field, LimbCircuit, _csr
    f = Field(curve)
    return LimbCircuit(f, q.m0, q.mw, q.nr, (_csr(f, q.a), _csr(f, q.b), _csr(f, q.c)), f.fr_limbs(inst), f.fr_limbs(wit))


def fuzz_shape(seed):
    """seed -> (curve, m0, nr, tables): m0 = 1 .. 20; 2 (m0 + nr) = 2^k, 2^k - 2 or 2^k + 2 (k = 6 .. 9, n <= 2^10), every
    eighth shape has ONE constraint; BN254 every fourth; the three MSM pipelines in turn."""
    m0 = 1 + seed % 20
    k = 6 + (seed * 7) % 4
    d = (0, -1, 1)[seed % 3]
    nr = 1 if seed % 8 == 5 else (1 << (k - 1)) + d - m0
    return ("bn254" if seed % 4 == 3 else "bls12_381"), m0, nr, ("auto", "off", "wide")[(seed // 3) % 3]


@pytest.mark.parametrize("seed", range(40))
def test_differential_fuzz_of_r1cs_shapes(gpu_ctx, oracle, api, seed):
    """40 seeded random systems against oracle/cpp: bases, proof, challenges, all 8 intermediate vectors; pm_host_prove
    (its own pi(x1), common.rs:49-71) gives the same bytes; the library's verifier accepts them; an unsatisfied row gives
    PM_ERR_REMAINDER_NONZERO on both sides (prover.rs:108)."""
    from polymath_amd.polymath import Polymath, PolymathProverError
    curve, m0, nr, tables = fuzz_shape(seed)
    c = CURVES[curve]
    gpu_ctx.set_option("tables", tables)                                          # restored by conftest
    q, inst, wit = CI.random_r1cs(c, 0xF022 + seed, m0, nr)
    assert q.m0 == m0 and q.nr == nr
    opk, gpk, proof = _prove_both(api, gpu_ctx, oracle, curve, q, inst, wit, 9000 + seed)
    assert gpk.n <= 1 << 10
    g = CI.SplitMix64(9000 + seed)
    x, z, r_a = g.fr(c.r), g.fr(c.r), [g.fr(c.r), g.fr(c.r)]
    pm = Polymath(curve, "merlin", ctx=gpu_ctx)
    xl, wl = pm.field.fr_limbs(inst), pm.field.fr_limbs(wit)
    data = pm.prove_native(gpk, xl, wl, r_a)
    assert data == SE.ser_proof(c, proof)
    assert pm.verify(pm.make_vk(gpk, x, z), inst[1:], data)
    # one witness off by one
    zz = list(inst) + list(wit)
    used = sorted({j for rows in (q.a, q.b, q.c) for row in rows for v, j in row if j >= m0 and v})
    col = used[seed % len(used)] if used else None
    if col is not None:
        zz[col] = (zz[col] + 1) % c.r
        if not all(CI.first_entry_dot(c.r, a, zz) * CI.first_entry_dot(c.r, b, zz) % c.r == CI.first_entry_dot(c.r, cc, zz)
                   for a, b, cc in zip(q.a, q.b, q.c)):
            omega = oracle.fr_from_mont_limbs(curve, opk.omega_limbs)[0]
            for backend in (opk, gpk):
                with pytest.raises(DR.ProverError) as e:
                    DR.prove(backend, opk.n, opk.sigma, omega, inst, zz[m0:], r_a, T.make_transcripts(c)["merlin"])
                assert (e.value.phase, e.value.rc) == (1, 4)
            with pytest.raises(PolymathProverError) as e:
                pm.prove_native(gpk, xl, pm.field.fr_limbs(zz[m0:]), r_a)
            assert e.value.status == 4
    gpk.free()


def test_fuzz_shapes_cover_what_they_claim():
    """(bookkeeping, no device work) the 40 shapes hit: m0 = 1, both sides of 2 m0 = 16, nr = 1, the three domain fills, both
    curves, the three MSM pipelines."""
    shapes = [fuzz_shape(s) for s in range(40)]
    assert {m0 for _, m0, _, _ in shapes} == set(range(1, 21))
    fills = set()
    for _, m0, nr, _ in shapes:
        rows, n = 2 * (m0 + nr), 1
        while n < rows:
            n <<= 1
        fills.add(n - rows if nr > 1 else "nr=1")
        assert n <= 1 << 10
    assert {0, 2, "nr=1"} <= fills and any(isinstance(f, int) and f > 2 for f in fills)
    assert {t for _, _, _, t in shapes} == {"auto", "off", "wide"} and {cv for cv, _, _, _ in shapes} == {"bls12_381", "bn254"}


@pytest.mark.parametrize("curve,m0", [("bls12_381", 12), ("bn254", 12), ("bls12_381", 7), ("bls12_381", 1)])
def test_many_public_inputs_at_2p12_gates_one_gpu_and_four_ranks(gpu_ctx, oracle, api, curve, m0):
    """4096 gates (n = 2^13) with m0 = 12 (2 m0 > 16: the witness-only part of u takes a FIFTH transform -- prove.hip, and its
    distributed twin in prove_sharded.hip), m0 = 7 (the direct 14-term sum) and m0 = 1 (no public input: nothing to subtract):
    one GPU against oracle/cpp (bases, bytes, all taps), the library's verifier, then the SAME input as one proof over 4
    rank-threads with hundreds of tiny sub-segments (PM_OPT_MAX_SEG_LOG = 6), compared with the ORACLE's bytes and with its
    u / w / wit_u taps through the layout."""
    from polymath_amd.polymath import Polymath
    c = CURVES[curve]
    q, inst, wit = CI.random_r1cs(c, 0x2C12 + m0, m0, 4096)
    seed = 31000 + m0
    opk, gpk, proof = _prove_both(api, gpu_ctx, oracle, curve, q, inst, wit, seed)
    want = SE.ser_proof(c, proof)
    g = CI.SplitMix64(seed)
    x, z, r_a = g.fr(c.r), g.fr(c.r), [g.fr(c.r), g.fr(c.r)]
    pm = Polymath(curve, "merlin", ctx=gpu_ctx)
    assert pm.prove_native(gpk, pm.field.fr_limbs(inst), pm.field.fr_limbs(wit), r_a) == want
    assert pm.verify(pm.make_vk(gpk, x, z), inst[1:], want)
    if m0 > 1:
        wrong = list(inst[1:])
        wrong[-1] = (wrong[-1] + 1) % c.r
        assert not pm.verify(pm.make_vk(gpk, x, z), wrong, want)
    n = gpk.n
    gpk.free()
    N = 4
    lc = _limb_circuit(curve, q, inst, wit)
    pms, pks, comms, proofs = _sharded_proofs(curve, lc, x, z, r_a, N, options={"max_seg_log": 6, "ntt_overlap": m0 % 2})
    assert all(p == want for p in proofs)
    for w in (2, 3, 5):                                    # u, w, wit_u of the ranks, scattered through the layout == the oracle's
        got = np.zeros((n, 4), dtype=np.uint64)
        for r in range(N):
            got[api.layout_indices(n, N, r)] = pks[r].tap(w, n)
        ref = opk.tap(w, 11 * n)
        k = min(n, len(ref))
        assert np.array_equal(got[:k], ref[:k]) and not got[k:].any(), w
    for pk in pks:
        pk.free()


@pytest.mark.parametrize("fname", ["proofs.json", "proofs_bn254.json"])
def test_golden_m0_shapes_on_rank_threads(api, fname):
    """The dense big-integer fixtures with m0 = 1, 3 and 12 (tools/gen_golden.py, oracle/pyref ALONE) as ONE proof over rank
    threads (N = 4, or 2 where n = 8), sub-segments of 4 indices, the three transcripts: the fixture's bytes; u, w, wit_u of the
    ranks scattered through the layout equal the fixture's trace."""
    for fx in load_golden(fname):
        if fx["r1cs"]["m0"] == 2:
            continue
        curve = fx["curve"]
        q = r1cs_from_json(fx["r1cs"])
        inst, wit, r_a = [I(v) for v in fx["instance"]], [I(v) for v in fx["witness"]], [I(v) for v in fx["r_a"]]
        lc = _limb_circuit(curve, q, inst, wit)
        n = fx["n"]
        N = 4 if n % 16 == 0 else 2
        for tname, ref in fx["proofs"].items():
            pms, pks, comms, proofs = _sharded_proofs(curve, lc, I(fx["x_trapdoor"]), I(fx["z_trapdoor"]), r_a, N, transcript=tname,
                                                      options={"max_seg_log": 2, "ntt_overlap": int(tname == "blake3")})
            assert all(p.hex() == ref["bytes"] for p in proofs), (fx["name"], tname)
            if tname == "keccak256":
                for w, key in ((2, "u"), (3, "w"), (5, "wit_u")):
                    got = np.zeros((n, 4), dtype=np.uint64)
                    for r in range(N):
                        got[api.layout_indices(n, N, r)] = pks[r].tap(w, n)
                    have = pms[0].field
                    vals = [have.fr_int(row) for row in got]
                    want = [I(v) for v in fx["trace"][key]]
                    assert vals[:len(want)] == want and not any(vals[len(want):]), (fx["name"], key)
            for pk in pks:
                pk.free()


@pytest.mark.parametrize("seed", range(40, 52))
def test_differential_fuzz_of_r1cs_shapes_on_rank_threads(oracle, api, seed):
    """12 more random shapes as ONE proof over N = 2, 4 or 8 rank-threads (cyclic witness map with multi-entry rows on instance
    columns, the distributed head rows for every m0, both branches of the witness-only part of u), sub-segments of 8 or 64
    indices, the three transcripts in turn: the CPU oracle's bytes; an unsatisfied row gives status 4 on EVERY rank."""
    from polymath_amd.polymath import PolymathProverError
    curve, m0, nr, tables = fuzz_shape(seed)
    c = CURVES[curve]
    N = (2, 4, 8)[seed % 3]
    tname = ("merlin", "keccak256", "blake3")[(seed // 3) % 3]
    q, inst, wit = CI.random_r1cs(c, 0xF022 + seed, m0, nr)
    lc = _limb_circuit(curve, q, inst, wit)
    g = CI.SplitMix64(9000 + seed)
    x, z, r_a = g.fr(c.r), g.fr(c.r), [g.fr(c.r), g.fr(c.r)]
    want, opk = _oracle_reference(oracle, curve, lc, x, z, r_a, transcript=tname)
    assert opk.n % (N * N) == 0
    pms, pks, comms, proofs = _sharded_proofs(curve, lc, x, z, r_a, N, transcript=tname,
                                              options={"max_seg_log": 3 if seed % 2 else 6, "ntt_overlap": seed % 2, "tables": tables})
    assert all(p == want for p in proofs)
    zz = list(inst) + list(wit)
    used = sorted({j for rows in (q.a, q.b, q.c) for row in rows for v, j in row if j >= m0 and v})
    if used:
        col = used[seed % len(used)]
        zz[col] = (zz[col] + 1) % c.r
        if not all(CI.first_entry_dot(c.r, a, zz) * CI.first_entry_dot(c.r, b, zz) % c.r == CI.first_entry_dot(c.r, cc, zz)
                   for a, b, cc in zip(q.a, q.b, q.c)):
            bad = pms[0].field.fr_limbs(zz[m0:])

            def prove_bad(r):
                try:
                    pms[r].prove_native(pks[r], lc.inst_limbs, bad, r_a)
                except PolymathProverError as e:
                    return e.status
                return 0
            assert _run_ranks(N, prove_bad, comms) == [4] * N
            assert all(p == want for p in _run_ranks(N, lambda r: pms[r].prove_native(pks[r], lc.inst_limbs, lc.wit_limbs, r_a), comms))
    for pk in pks:
        pk.free()


@pytest.mark.parametrize("seed", range(60, 76))
def test_differential_fuzz_with_extreme_assignments(gpu_ctx, oracle, api, seed):
    """16 shapes whose FREE values (public inputs, seed and unused witnesses) are mostly 0, 1, r - 1, r - 2, 2: zero scalars in
    z_tail and in the MSMs (arkworks' msm skips them), 1 - x_i = 0 in the head rows (common.rs:77-97), gate products that vanish,
    -1 digits in every window.  Whatever the CPU oracle returns -- a proof or a status -- the HIP path returns the same."""
    from polymath_amd.polymath import Polymath, PolymathProverError
    curve, m0, nr, tables = fuzz_shape(seed)
    c = CURVES[curve]
    gpu_ctx.set_option("tables", tables)
    q, inst, wit = CI.random_r1cs(c, 0xE000 + seed, m0, nr, extreme=True)
    g = CI.SplitMix64(12000 + seed)
    x, z, r_a = g.fr(c.r), g.fr(c.r), [g.fr(c.r), g.fr(c.r)]
    if seed % 4 == 0:
        r_a = [0, 0]                      # the reference would draw this with probability 2^-510; A(X) = u(X) then
    opk = oracle.OraclePk(curve, q, x, z, 8)
    omega = oracle.fr_from_mont_limbs(curve, opk.omega_limbs)[0]
    TR = T.make_transcripts(c)["keccak256"]
    try:
        want, rc = SE.ser_proof(c, DR.prove(opk, opk.n, opk.sigma, omega, inst, wit, r_a, TR)), 0
    except DR.ProverError as e:
        want, rc = None, e.rc
    A, B, C = __import__("helpers").pm_csrs(curve, q)
    gpk = api.ProvingKey.generate(gpu_ctx, curve, q.m0, q.mw, q.nr, A, B, C, oracle.fr_to_mont_limbs(curve, [x])[0], oracle.fr_to_mont_limbs(curve, [z])[0])
    for i in range(6):
        assert np.array_equal(gpk.export_bases(i), opk.export_bases(i)), i
    pm = Polymath(curve, "keccak256", ctx=gpu_ctx)
    try:
        got, grc = pm.prove_native(gpk, pm.field.fr_limbs(inst), pm.field.fr_limbs(wit), r_a), 0
    except PolymathProverError as e:
        got, grc = None, e.status
    assert (grc, got) == (rc, want)
    if rc == 0:
        for which in range(8):
            a, b = gpk.tap(which, 1 << 16), opk.tap(which, 1 << 16)
            k = min(len(a), len(b))
            assert k > 0 and np.array_equal(a[:k], b[:k]) and not a[k:].any() and not b[k:].any(), which
        assert pm.verify(pm.make_vk(gpk, x, z), inst[1:], got)
    gpk.free()
