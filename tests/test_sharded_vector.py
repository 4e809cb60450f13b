"""PM_SHARD_VECTOR: ONE proof with the vector phases sharded over N ranks.

CPU (no GPU): the layout's index arithmetic (polymath_amd/host/layout.hpp through pm_layout_indices) -- the blocked /
cyclic maps are permutations, and the four-step transform built on them with the CPU oracle's size-n/N NTTs and a
numpy "all-to-all" equals the direct size-n transform.
GPU: N ranks as N threads of one process on one GPU (pm_comm_local_create), every rank holding ONLY its share of the
vectors and of the key; the proofs must be byte-identical to the single-GPU proof, for N = 2, 4, 8, both curves."""
import threading

import numpy as np
import pytest

from oracle.pyref import circuits as CI
from oracle.pyref.fields import CURVES


def _four_step_intt(oracle, curve, evals, n, N):
    """evals: list of n ints.  Inverse transform through the sharded algorithm, all ranks simulated here."""
    from polymath_amd import api
    c = CURVES[curve]
    m, B = n // N, n // N // N
    omega = pow(c.two_adic_root, 1 << (c.two_adicity - (n.bit_length() - 1)), c.r)
    rows = [api.layout_indices(n, N, q, coefficients=False) for q in range(N)]
    coef = [api.layout_indices(n, N, q, coefficients=True) for q in range(N)]
    assert sorted(np.concatenate(rows).tolist()) == list(range(n)) and sorted(np.concatenate(coef).tolist()) == list(range(n))
    log_m = m.bit_length() - 1
    local = []
    for q in range(N):     # N local size-m transforms on the cyclic sub-sequences
        x = oracle.fr_to_mont_limbs(curve, [evals[int(i)] for i in rows[q]])
        local.append(oracle.fr_from_mont_limbs(curve, oracle.ntt(curve, x, log_m, True)))
    out = [0] * n
    winv, ninv = pow(omega, -1, c.r), pow(N, -1, c.r)
    for q in range(N):     # after the all-to-all rank q holds block q of every rank's transform
        for b in range(B):
            k2 = q * B + b
            col = [local[r][k2] * pow(winv, r * k2, c.r) % c.r for r in range(N)]
            for k1 in range(N):
                v = sum(col[r] * pow(winv, m * r * k1, c.r) for r in range(N)) % c.r * ninv % c.r
                p = k1 * B + b
                out[int(coef[q][p])] = v
                assert int(coef[q][p]) == k1 * m + k2
    return out


@pytest.mark.parametrize("N", [2, 4])
def test_layout_four_step_equals_direct_transform(oracle, N):
    curve, n = "bls12_381", 64
    c = CURVES[curve]
    g = CI.SplitMix64(64 + N)
    evals = [g.fr(c.r) for _ in range(n)]
    direct = oracle.fr_from_mont_limbs(curve, oracle.ntt(curve, oracle.fr_to_mont_limbs(curve, evals), 6, True))
    assert _four_step_intt(oracle, curve, evals, n, N) == direct


def test_layout_selftest_native(tmp_path):
    """tests/native/layout_selftest.cpp: the index maps are bijections, every rank's quotient segments tile the numerator
    index space exactly once, the MSM piece lists cover every pair once (plus [c]'s N^2 - 1 shared block-end bases), and the
    per-rank counts the sharded prover assumes hold -- n = 4 ... 2^16, N = 1 ... 16, two sub-segment sizes."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "layout_selftest")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(root, "tests", "native", "layout_selftest.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "0 failures" in out.stdout, out.stdout


def test_layout_rejects_bad_shapes():
    from polymath_amd import api
    with pytest.raises(api.PolymathError):
        api.layout_indices(64, 3, 0)            # not a power of two
    with pytest.raises(api.PolymathError):
        api.layout_indices(32, 8, 0)            # N^2 > n


# ------------------------------------------------------------------------------------------------------ GPU
def _run_ranks(N, fn, comms=None, timeout=900):
    """fn(rank) on N threads; re-raises the first exception.  Fail-fast: a rank that raises aborts the group's communicators
    at once (its peers leave their collective with PM_ERR_COMM instead of waiting for it), and ranks still running after
    `timeout` seconds are aborted the same way -- a broken rank fails the test in seconds, it does not hang it."""
    errs, outs = [None] * N, [None] * N

    def body(r):
        try:
            outs[r] = fn(r)
        except BaseException as e:     # noqa: BLE001 -- reported below
            errs[r] = e
            if comms is not None:
                comms[r].abort("rank %d raised %s" % (r, type(e).__name__))
    th = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(N)]
    for t in th:
        t.start()
    import time
    t_end = time.time() + timeout
    for t in th:
        t.join(max(0.0, t_end - time.time()))
    if any(t.is_alive() for t in th):
        if comms is not None:
            for c in comms:
                c.abort("test timeout")
            for t in th:
                t.join(30)
        raise TimeoutError("ranks still running after %d s: %s" % (timeout, [r for r, t in enumerate(th) if t.is_alive()]))
    first = [e for e in errs if e is not None and getattr(e, "status", None) != 9] or [e for e in errs if e is not None]
    if first:
        raise first[0]              # the root cause, not a peer's PM_ERR_COMM
    return outs


def _oracle_reference(oracle, curve, lc, x, z, r_a, transcript="merlin"):
    """The CPU restatement (oracle/cpp) on the same circuit, trapdoors and r_a, with ITS OWN setup: proof bytes and the
    intermediate vectors (taps) the sharded GPU proof is compared with -- the (e)-row tests rest on the oracle, not on
    the single-GPU path."""
    from oracle import driver as DR
    from oracle.pyref import serialize as SE, transcripts as T
    import os
    c = CURVES[curve]

    class Shape:
        pass
    q = Shape()
    q.m0, q.mw, q.nr = lc.m0, lc.mw, lc.nr
    q.csr_arrays = [(a.rowptr, a.col, a.val) for a in lc.csrs]
    opk = oracle.OraclePk(curve, q, x, z, os.cpu_count() or 4)
    omega = oracle.fr_from_mont_limbs(curve, opk.omega_limbs)[0]
    po = DR.prove(opk, opk.n, opk.sigma, omega, lc.instance, None, r_a, T.make_transcripts(c)[transcript], w_limbs=lc.wit_limbs)
    return SE.ser_proof(c, po), opk


def _sharded_proofs(curve, lc, x, z, r_a, N, transcript="merlin", device_assignment=False, options=None):
    """options: {pm_option name: value} set on every rank's context (pm_ctx_set_option) before its key is generated"""
    from polymath_amd import api
    from polymath_amd.polymath import Polymath
    comms = api.Comm.local_group(N)
    pms = [Polymath(curve, transcript, device=0) for _ in range(N)]
    for r in range(N):
        pms[r].ctx.set_comm(comms[r])
        for k, v in (options or {}).items():
            pms[r].ctx.set_option(k, v)
    pks = [pms[r].setup(lc, x, z, shard_rank=r, shard_count=N, layout="vector") for r in range(N)]
    proofs = _run_ranks(N, lambda r: pms[r].prove_native(pks[r], lc.inst_limbs, lc.wit_limbs, r_a), comms)
    return pms, pks, comms, proofs


@pytest.mark.gpu
@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
@pytest.mark.parametrize("N", [2, 4, 8])
def test_vector_sharded_proof_equals_single_gpu(oracle, curve, N):
    """5000 gates (n = 16384): every rank proves on 1/N of the rows, coefficients, scans and MSM pairs; the N proofs are
    identical to the CPU ORACLE's proof of the same input (oracle/cpp, its own setup) and to the unsharded GPU proof.  The
    shards' local vectors, scattered through the layout, equal the oracle's taps."""
    from polymath_amd import api, circuits as PC
    from polymath_amd.polymath import Polymath
    c = CURVES[curve]
    lc = PC.synthetic_r1cs_native(curve, 5000)
    g = PC.SplitMix64(0x5A4D + N)
    x, z, r_a = g.fr(c.r), g.fr(c.r), [g.fr(c.r), g.fr(c.r)]
    ref_pm = Polymath(curve, "merlin", device=0)
    ref_pk = ref_pm.setup(lc, x, z)
    ref = ref_pm.prove_native(ref_pk, lc.inst_limbs, lc.wit_limbs, r_a)
    n = ref_pk.n
    oracle_proof, opk = _oracle_reference(oracle, curve, lc, x, z, r_a)
    assert ref == oracle_proof
    whole = {w: opk.tap(w, 11 * n) for w in (2, 3, 4, 5, 6, 7)}      # the ORACLE's vectors
    for w in whole:
        g_tap = ref_pk.tap(w, 11 * n)
        k = min(len(g_tap), len(whole[w]))
        assert np.array_equal(g_tap[:k], whole[w][:k]) and not g_tap[k:].any() and not whole[w][k:].any(), w
        whole[w] = g_tap                                              # == the oracle's over its whole length, zeros beyond
    pms, pks, comms, proofs = _sharded_proofs(curve, lc, x, z, r_a, N)
    assert all(p == oracle_proof for p in proofs)
    # local vectors -> global through the layout
    Lz = len(whole[6])
    for w in (2, 3, 5):                                    # u, w, wit_u: blocked coefficient layout
        got = np.zeros((n, 4), dtype=np.uint64)
        for r in range(N):
            got[api.layout_indices(n, N, r)] = pks[r].tap(w, n)
        assert np.array_equal(got, whole[w]), w
    got = np.zeros((n, 4), dtype=np.uint64)               # h: the same layout, index n - 1 does not exist
    for r in range(N):
        idx = api.layout_indices(n, N, r)
        loc = pks[r].tap(4, n)
        got[idx[:len(loc)]] = loc
    assert np.array_equal(got[:n - 1], whole[4])
    assert np.array_equal(np.concatenate([pks[r].tap(6, Lz) for r in range(N)]), whole[6])     # z_tail: contiguous slices
    # quotient: scatter every rank's scalars through its [d] pieces (bases y_gamma_z[k - 1] <-> H_k)
    qn = len(whole[7])
    got = np.zeros((qn + 1, 4), dtype=np.uint64)
    seen = np.zeros(qn + 1, dtype=np.int32)
    off_ygz = None
    for r in range(N):
        loc, at = pks[r].tap(7, 11 * n), 0
        pieces = pks[r].msm_pieces(2)
        if off_ygz is None:
            off_ygz = min(p[0] for rr in range(N) for p in pks[rr].msm_pieces(2))
        for lo, cnt in pieces:
            got[lo - off_ygz:lo - off_ygz + cnt] = loc[at:at + cnt]
            seen[lo - off_ygz:lo - off_ygz + cnt] += 1
            at += cnt
        assert at == len(loc)
    assert (seen[:qn] == 1).all() and np.array_equal(got[:qn], whole[7])
    for pk in pks:
        pk.free()
    ref_pk.free()


@pytest.mark.gpu
@pytest.mark.parametrize("N", [2, 8])
def test_vector_sharded_w_transform_beside_u_chain(oracle, monkeypatch, N):
    """PM_OPT_NTT_OVERLAP = 1 (the default since round 4): w's distributed transform on the helper stream beside u's chain, its all-to-all issued
    between u's exchanges on a second stream of the same communicator.  Three proofs in a row on the same contexts (the
    helper's buffers and events are reused) equal the CPU oracle's bytes; w's coefficients equal the oracle's tap; and the
    switch can be turned off again in the same process."""
    from polymath_amd import api, circuits as PC
    curve = "bls12_381"
    c = CURVES[curve]
    lc = PC.synthetic_r1cs_native(curve, 5000)
    g = PC.SplitMix64(0x0E11 + N)
    x, z, r_a = g.fr(c.r), g.fr(c.r), [g.fr(c.r), g.fr(c.r)]
    oracle_proof, opk = _oracle_reference(oracle, curve, lc, x, z, r_a)
    pms, pks, comms, proofs = _sharded_proofs(curve, lc, x, z, r_a, N, options={"ntt_overlap": 1})
    assert all(p == oracle_proof for p in proofs)
    n = pks[0].n
    for w in (2, 3):                                        # u, w: blocked coefficient layout
        got = np.zeros((n, 4), dtype=np.uint64)
        for r in range(N):
            got[api.layout_indices(n, N, r)] = pks[r].tap(w, n)
        ref = opk.tap(w, 11 * n)
        k = min(n, len(ref))
        assert np.array_equal(got[:k], ref[:k]) and not got[k:].any(), w
    r_b = [g.fr(c.r), g.fr(c.r)]
    oracle_b, _ = _oracle_reference(oracle, curve, lc, x, z, r_b)
    for ra, want in ((r_b, oracle_b), (r_a, oracle_proof)):
        out = _run_ranks(N, lambda r: pms[r].prove_native(pks[r], lc.inst_limbs, lc.wit_limbs, ra), comms)
        assert all(p == want for p in out)
    for pm in pms:
        pm.ctx.set_option("ntt_overlap", 0)
    out = _run_ranks(N, lambda r: pms[r].prove_native(pks[r], lc.inst_limbs, lc.wit_limbs, r_b), comms)
    assert all(p == oracle_b for p in out)
    for pk in pks:
        pk.free()


@pytest.mark.gpu
@pytest.mark.parametrize("tables", ["wide", "0"])
def test_vector_sharded_without_window_tables(oracle, monkeypatch, tables):
    """The shards' MSMs on the two table-less pipelines (PM_OPT_TABLES = wide: one bucket set per window, what a shard whose
    tables do not fit HBM runs; = off: the per-window pipeline): 4 ranks, both MSMs of phase 1 in flight together
    (msm_begin / msm_end), the same bytes as the CPU oracle."""
    from polymath_amd import circuits as PC
    curve, N = "bls12_381", 4
    c = CURVES[curve]
    lc = PC.synthetic_r1cs_native(curve, 5000)
    g = PC.SplitMix64(0x71DE)
    x, z, r_a = g.fr(c.r), g.fr(c.r), [g.fr(c.r), g.fr(c.r)]
    oracle_proof, _ = _oracle_reference(oracle, curve, lc, x, z, r_a)
    pms, pks, comms, proofs = _sharded_proofs(curve, lc, x, z, r_a, N, options={"tables": {"wide": "wide", "0": "off"}[tables]})
    assert all(not pk.msm_plan(k)[3] for pk in pks for k in range(3))          # no MSM of any shard holds tables
    assert all(p == oracle_proof for p in proofs)
    out = _run_ranks(N, lambda r: pms[r].prove_native(pks[r], lc.inst_limbs, lc.wit_limbs, r_a), comms)
    assert all(p == oracle_proof for p in out)
    for pk in pks:
        pk.free()


@pytest.mark.gpu
def test_vector_sharded_many_segments_public_inputs_and_errors(oracle, monkeypatch):
    """(Compared with the CPU ORACLE's bytes since round 5 -- the single-GPU proof of the same library is checked against them too.)
    Tiny sub-segments (PM_OPT_MAX_SEG_LOG = 6: hundreds of segments per rank, several per block), a circuit with 12 public
    inputs (2 m0 > 16: the witness-only part of u takes its own distributed transform), all three transcripts; an
    unsatisfied witness makes EVERY rank return PM_ERR_REMAINDER_NONZERO (no rank is left waiting in a collective)."""
    from polymath_amd import circuits as PC
    from polymath_amd.polymath import ConstraintSystem, LimbCircuit, Polymath, Field, _csr
    from polymath_amd import api
    curve, N = "bls12_381", 4
    c = CURVES[curve]
    f = Field(curve)
    cs = ConstraintSystem(c.r)
    g = PC.SplitMix64(1212)
    vals = [g.fr(c.r) for _ in range(300)]
    wv = [cs.new_witness_variable(v) for v in vals]
    for i in range(0, 290):
        prod = vals[i] * vals[i + 1] % c.r
        out = cs.new_input_variable(prod) if i < 11 else cs.new_witness_variable(prod)
        cs.enforce_constraint([(1, wv[i])], [(1, wv[i + 1])], [(1, out)])
    r1cs = cs.to_r1cs()
    assert r1cs.m0 == 12
    lc = LimbCircuit(f, r1cs.m0, r1cs.mw, r1cs.nr, (_csr(f, r1cs.a), _csr(f, r1cs.b), _csr(f, r1cs.c)), f.fr_limbs(cs.instance), f.fr_limbs(cs.witness))
    x, z, r_a = g.fr(c.r), g.fr(c.r), [g.fr(c.r), g.fr(c.r)]
    for tname in ("merlin", "keccak256", "blake3"):
        ref_pm = Polymath(curve, tname, device=0)
        ref_pk = ref_pm.setup(lc, x, z)
        ref = ref_pm.prove_native(ref_pk, lc.inst_limbs, lc.wit_limbs, r_a)
        oracle_proof, _ = _oracle_reference(oracle, curve, lc, x, z, r_a, transcript=tname)
        assert ref == oracle_proof, tname                   # m0 = 12 on one GPU: the fifth transform (prove.hip) against oracle/cpp
        assert ref_pm.verify(ref_pm.make_vk(ref_pk, x, z), lc.instance[1:], ref)
        ref = oracle_proof
        # the two-stream transforms on (keccak256) and off on this shape too
        pms, pks, comms, proofs = _sharded_proofs(curve, lc, x, z, r_a, N, transcript=tname,
                                                  options={"max_seg_log": 6, "ntt_overlap": 1 if tname == "keccak256" else 0})
        assert all(p == ref for p in proofs), tname
        ref_pk.free()
    bad = lc.wit_limbs.copy()
    bad[7, 0] ^= np.uint64(1)
    from polymath_amd.polymath import PolymathProverError

    def prove_bad(r):
        try:
            pms[r].prove_native(pks[r], lc.inst_limbs, bad, r_a)
        except PolymathProverError as e:
            return e.status
        return 0
    assert _run_ranks(N, prove_bad, comms) == [4] * N
    # the contexts AND the communicators are still usable afterwards: the verdict was exchanged, nobody aborted
    assert not any(cm.failed for cm in comms)
    assert all(p == ref for p in _run_ranks(N, lambda r: pms[r].prove_native(pks[r], lc.inst_limbs, lc.wit_limbs, r_a), comms))


@pytest.mark.gpu
def test_vector_sharded_phase3_needs_the_x1_of_phase2():
    """The division scan's x1-dependent sums are exchanged in phase 2; phase 3 with another x1 is refused on every rank with the
    same status (PM_ERR_INVALID_ARG: a verdict they share), the communicators stay usable and the proof still goes through."""
    from polymath_amd import circuits as PC
    curve, N = "bls12_381", 2
    c = CURVES[curve]
    lc = PC.synthetic_r1cs_native(curve, 3000)
    g = PC.SplitMix64(0x3A1)
    x, z, r_a = g.fr(c.r), g.fr(c.r), [g.fr(c.r), g.fr(c.r)]
    pms, pks, comms, proofs = _sharded_proofs(curve, lc, x, z, r_a, N)
    f = pms[0].field

    def body(r):
        pk = pks[r]
        rc, *_ = pk.phase1(lc.inst_limbs, lc.wit_limbs, f.fr_limbs(r_a))
        assert rc == 0
        x1 = f.fr_limbs([12345])[0]
        rc, _u = pk.phase2(x1)
        assert rc == 0
        other = f.fr_limbs([12346])[0]
        rc, _d, _i = pk.phase3(other, other, other, other)
        return rc
    assert _run_ranks(N, body, comms) == [1] * N                       # PM_ERR_INVALID_ARG on both ranks
    assert not any(cm.failed for cm in comms)
    again = _run_ranks(N, lambda r: pms[r].prove_native(pks[r], lc.inst_limbs, lc.wit_limbs, r_a), comms)
    assert all(p == proofs[0] for p in again)
    for pk in pks:
        pk.free()


@pytest.mark.gpu
def test_vector_sharded_mid_size_and_pairs_layout_agree(oracle):
    """2^16-100 gates on 8 ranks (B = 2048 coefficients per block, several sub-segments per stretch): the vector-sharded
    proof, the pairs-sharded proof (vector phases replicated, pm_comm combine) and the single-GPU proof are identical -- and
    equal to the CPU oracle's proof of the same input (oracle/cpp on the bases exported from the single-GPU key: its own
    setup is tested at the smaller sizes)."""
    from polymath_amd import api, circuits as PC
    from polymath_amd.polymath import Polymath
    curve, N = "bls12_381", 8
    c = CURVES[curve]
    lc = PC.synthetic_r1cs_native(curve, (1 << 16) - 100)
    g = PC.SplitMix64(0x1616)
    x, z, r_a = g.fr(c.r), g.fr(c.r), [g.fr(c.r), g.fr(c.r)]
    ref_pm = Polymath(curve, "merlin", device=0)
    ref_pk = ref_pm.setup(lc, x, z)
    ref = ref_pm.prove_native(ref_pk, lc.inst_limbs, lc.wit_limbs, r_a)
    import os
    from oracle import driver as DR
    from oracle.pyref import serialize as SE, transcripts as T

    class Shape:
        pass
    q = Shape()
    q.m0, q.mw, q.nr = lc.m0, lc.mw, lc.nr
    q.csr_arrays = [(a.rowptr, a.col, a.val) for a in lc.csrs]
    opk = oracle.OraclePk(curve, q, None, None, os.cpu_count() or 4)
    for i in range(6):
        opk.import_bases(i, ref_pk.export_bases(i))
    omega = oracle.fr_from_mont_limbs(curve, opk.omega_limbs)[0]
    po = DR.prove(opk, opk.n, opk.sigma, omega, lc.instance, None, r_a, T.make_transcripts(c)["merlin"], w_limbs=lc.wit_limbs)
    assert ref == SE.ser_proof(c, po)
    ref_pk.free()
    pms, pks, comms, proofs = _sharded_proofs(curve, lc, x, z, r_a, N)
    assert all(p == ref for p in proofs)
    for pk in pks:
        pk.free()
    # PM_SHARD_PAIRS keys on the same contexts and communicators: the native point combine replaces the callback
    pks = [pms[r].setup(lc, x, z, shard_rank=r, shard_count=N, layout="pairs") for r in range(N)]
    assert all(p == ref for p in _run_ranks(N, lambda r: pms[r].prove_native(pks[r], lc.inst_limbs, lc.wit_limbs, r_a), comms))


@pytest.mark.gpu
def test_sharded_prover_fails_fast_instead_of_hanging():
    """The fail-fast contract of pm_comm (include/polymath_hip.h) on the in-process group, 4 ranks:
    (i) a rank that never shows up: its peers leave the collective with PM_ERR_COMM once the deadline passes (2 s here);
    (ii) a rank that fails LOCALLY inside a phase (an assignment pointer of the wrong size class is not detectable, so the
         failure is injected as a communicator abort from the host -- what PhaseEnd does on a HIP error): the peers return
         at once, long before any deadline;
    a failed communicator stays failed; fresh communicators on the same contexts and keys prove again."""
    import time
    from polymath_amd import api, circuits as PC
    from polymath_amd.polymath import Polymath, PolymathProverError
    curve, N = "bls12_381", 4
    c = CURVES[curve]
    lc = PC.synthetic_r1cs_native(curve, 3000)
    g = PC.SplitMix64(0xFA57)
    x, z, r_a = g.fr(c.r), g.fr(c.r), [g.fr(c.r), g.fr(c.r)]
    pms, pks, comms, proofs = _sharded_proofs(curve, lc, x, z, r_a, N)
    assert all(p == proofs[0] for p in proofs)

    def status_of(r, skip):
        if r == skip:
            return "absent"
        try:
            pms[r].prove_native(pks[r], lc.inst_limbs, lc.wit_limbs, r_a)
        except PolymathProverError as e:
            return e.status
        return 0
    # (i) rank 2 never calls prove
    for cm in comms:
        cm.set_timeout_ms(2000)
    t0 = time.time()
    got = _run_ranks(N, lambda r: status_of(r, 2), None, timeout=120)
    assert got == [9, 9, "absent", 9], got
    assert 1.5 < time.time() - t0 < 60
    assert all(comms[r].failed for r in (0, 1, 3)) and "did not reach the collective" in comms[0].last_error()
    assert status_of(0, None) == 9                                   # sticky: no waiting this time
    # (ii) fresh communicators, long deadline; rank 1 gives up while the others are inside phase 1
    comms2 = api.Comm.local_group(N)
    for r in range(N):
        pms[r].ctx.set_comm(comms2[r])
        comms2[r].set_timeout_ms(600000)

    def body(r):
        if r == 1:
            time.sleep(0.2)
            comms2[1].abort("injected failure")
            return "aborted"
        return status_of(r, None)
    t0 = time.time()
    got = _run_ranks(N, body, None, timeout=120)
    assert got == [9, "aborted", 9, 9], got
    assert time.time() - t0 < 30
    assert "injected failure" in comms2[0].last_error()
    # and the keys / contexts are fine: a third group proves
    comms3 = api.Comm.local_group(N)
    for r in range(N):
        pms[r].ctx.set_comm(comms3[r])
    again = _run_ranks(N, lambda r: pms[r].prove_native(pks[r], lc.inst_limbs, lc.wit_limbs, r_a), comms3)
    assert all(p == proofs[0] for p in again)
    for pk in pks:
        pk.free()


def test_local_group_deadline_and_abort_without_a_gpu():
    """pm_comm fail-fast on host memory only (runs on CPU): an absent rank costs its peers the deadline, not for ever; an
    aborting rank wakes them at once; the failure is sticky and says why."""
    import time
    from polymath_amd import api
    comms = api.Comm.local_group(3)
    assert comms[0].kind == "local" and not comms[0].failed
    for cm in comms:
        cm.set_timeout_ms(700)

    def body(r):
        if r == 2:
            return "absent"
        try:
            comms[r].all_gather(np.array([r], dtype=np.int64))
        except api.PolymathError as e:
            return e.status
        return 0
    t0 = time.time()
    assert _run_ranks(3, body, None, timeout=60) == [9, 9, "absent"]
    assert 0.5 < time.time() - t0 < 30 and comms[0].failed and "did not reach" in comms[1].last_error()
    comms = api.Comm.local_group(3)                 # default deadline (120 s): the abort is what ends the wait

    def body2(r):
        if r == 2:
            time.sleep(0.3)
            comms[2].abort("rank 2 gives up")
            return "aborted"
        try:
            comms[r].all_gather(np.array([r], dtype=np.int64))
        except api.PolymathError as e:
            return e.status
        return 0
    t0 = time.time()
    assert _run_ranks(3, body2, None, timeout=60) == [9, 9, "aborted"]
    assert time.time() - t0 < 20 and "rank 2 gives up" in comms[0].last_error()
    good = api.Comm.local_group(2)
    out = _run_ranks(2, lambda r: good[r].all_gather(np.array([r + 5], dtype=np.int64)).reshape(-1).tolist(), good)
    assert out == [[5, 6], [5, 6]]


def test_run_ranks_helper_reports_a_broken_rank_in_seconds():
    """The test harness itself: a rank that raises before its first collective must fail the group at once."""
    import time
    from polymath_amd import api
    comms = api.Comm.local_group(3)

    def body(r):
        if r == 0:
            raise ValueError("broken rank")
        return comms[r].all_gather(np.array([r], dtype=np.int64))
    t0 = time.time()
    with pytest.raises(ValueError):
        _run_ranks(3, body, comms, timeout=60)
    assert time.time() - t0 < 20



class _Hip:
    """Device buffers for the exchange tests through the HIP runtime the library itself runs on (ctypes).  Not torch: torch ships
    its own HIP runtime and librccl, and bringing its GPU side up AFTER the library has used the system runtime in the same
    process fails ("No HIP GPUs are available"); bench.py and the tools initialise torch first, a test process cannot."""

    def __init__(self):
        import ctypes as ct
        self.ct = ct
        self.lib = ct.CDLL("libamdhip64.so")
        self.lib.hipMalloc.argtypes = [ct.POINTER(ct.c_void_p), ct.c_size_t]
        self.lib.hipFree.argtypes = [ct.c_void_p]
        self.lib.hipMemcpy.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_size_t, ct.c_int]

    def upload(self, arr):
        p = self.ct.c_void_p()
        assert self.lib.hipMalloc(self.ct.byref(p), arr.nbytes) == 0
        assert self.lib.hipMemcpy(p, arr.ctypes.data_as(self.ct.c_void_p), arr.nbytes, 1) == 0
        return p

    def download(self, p, like):
        out = np.empty_like(like)
        assert self.lib.hipDeviceSynchronize() == 0
        assert self.lib.hipMemcpy(out.ctypes.data_as(self.ct.c_void_p), p, out.nbytes, 2) == 0
        return out

    def free(self, *ps):
        for p in ps:
            self.lib.hipFree(p)

    def stream(self):
        st = self.ct.c_void_p()
        self.lib.hipStreamCreateWithFlags.argtypes = [self.ct.POINTER(self.ct.c_void_p), self.ct.c_uint]
        assert self.lib.hipStreamCreateWithFlags(self.ct.byref(st), 1) == 0          # hipStreamNonBlocking
        return st

    def sync(self, st):
        self.lib.hipStreamSynchronize.argtypes = [self.ct.c_void_p]
        assert self.lib.hipStreamSynchronize(st) == 0


def _two_stream_exchanges(comms, order):
    """Every rank issues the four all-to-alls of one proof's transforms (prover.rs:93-96: u and w are independent; then the
    forward / inverse pair of the square) on TWO streams of its ONE communicator, in `order` = a permutation of
    ("u", "w", "sq_fwd", "sq_inv") that all ranks share; "w" goes to the second stream.  -> nothing; asserts the delivered blocks."""
    N = len(comms)
    hip = _Hip()
    words = 512                                                    # 4 KiB blocks
    names = ("u", "w", "sq_fwd", "sq_inv")

    def tag(kind, src, dst):
        return (names.index(kind) << 40) | (src << 20) | dst

    def body(r):
        s_main, s_side = hip.stream(), hip.stream()
        bufs = {}
        for kind in names:
            send = np.concatenate([np.full(words, tag(kind, r, p), dtype=np.int64) for p in range(N)])
            bufs[kind] = (hip.upload(send), hip.upload(np.full(N * words, -1, dtype=np.int64)), send)
        for kind in order:
            d_send, d_recv, send = bufs[kind]
            comms[r].all_to_all_device(d_send.value, d_recv.value, words * 8, (s_side if kind == "w" else s_main).value)
        hip.sync(s_main)
        hip.sync(s_side)
        for kind in names:
            d_send, d_recv, send = bufs[kind]
            got = hip.download(d_recv, send).reshape(N, words)
            assert all((got[p] == tag(kind, p, r)).all() for p in range(N)), (kind, r)
            hip.free(d_send, d_recv)
        return True
    for c in comms:
        c.set_timeout_ms(20000)                                     # a deadlock would end as PM_ERR_COMM after 20 s, not as a hang
    assert _run_ranks(N, body, comms, timeout=120) == [True] * N


@pytest.mark.gpu
def test_exchanges_from_two_streams_of_one_communicator_in_both_orders():
    """VERDICT r3 item 6 (PM_OPT_NTT_OVERLAP is the default since round 4): the four exchanges of a proof's transforms issued from
    two streams of ONE communicator -- u first then w (the prover's order), and w first then u -- on 8 rank-threads of the
    in-process group, and on the RCCL communicator with a world of one; every block arrives, nothing deadlocks (each case runs
    under the communicators' fail-fast deadline)."""
    from polymath_amd import api
    for order in (("u", "w", "sq_fwd", "sq_inv"), ("w", "u", "sq_fwd", "sq_inv"), ("u", "sq_fwd", "w", "sq_inv")):
        comms = api.Comm.local_group(8)
        _two_stream_exchanges(comms, order)
        assert not any(c.failed for c in comms)
        for c in comms:
            c.close()
    for order in (("u", "w", "sq_fwd", "sq_inv"), ("w", "u", "sq_fwd", "sq_inv")):
        comm = api.Comm.rccl(api.Comm.rccl_unique_id(), 0, 1, 0)
        _two_stream_exchanges([comm], order)
        assert not comm.failed
        comm.close()

@pytest.mark.gpu
def test_rccl_comm_world_of_one():
    """The RCCL implementation of pm_comm (librccl dlopen'ed by the library, ncclCommInitRank / ncclAllToAll / ncclAllGather)
    on the single GPU of this box: a world of ONE rank -- the code path of an N-GPU job, with every collective degenerate.
    The PM_SHARD_VECTOR prover on it (B = n: one block) gives the single-GPU proof."""
    from polymath_amd import api, circuits as PC
    from polymath_amd.polymath import Polymath
    curve = "bls12_381"
    c = CURVES[curve]
    comm = api.Comm.rccl(api.Comm.rccl_unique_id(), 0, 1, 0)
    assert (comm.rank, comm.world) == (0, 1)
    assert comm.all_gather(np.arange(5, dtype=np.int64)).tolist() == [[0, 1, 2, 3, 4]]
    hip = _Hip()
    src = np.arange(64, dtype=np.int64)
    d_send, d_recv = hip.upload(src), hip.upload(np.full(64, -1, dtype=np.int64))
    comm.all_to_all_device(d_send.value, d_recv.value, src.nbytes)       # ncclAllToAll with one peer: block 0 -> rank 0
    assert np.array_equal(hip.download(d_recv, src), src)
    hip.free(d_send, d_recv)
    lc = PC.synthetic_r1cs_native(curve, 3000)
    ref_pm = Polymath(curve, "merlin", device=0)
    ref_pk = ref_pm.setup(lc, 11, 13)
    ref = ref_pm.prove_native(ref_pk, lc.inst_limbs, lc.wit_limbs, [3, 5])
    pm = Polymath(curve, "merlin", device=0)
    pm.ctx.set_comm(comm)
    pk = pm.setup(lc, 11, 13, shard_rank=0, shard_count=1, layout="vector")
    assert pm.prove_native(pk, lc.inst_limbs, lc.wit_limbs, [3, 5]) == ref
    # without a communicator the vector layout refuses to run (status, no crash)
    pm2 = Polymath(curve, "merlin", device=0)
    rc, _ = pk.view(pm2.ctx).host_prove("merlin", lc.inst_limbs, lc.inst_limbs, lc.wit_limbs, pm.field.fr_limbs([3, 5]))
    assert rc == 8
    pk.free()
    ref_pk.free()
    comm.close()


_WATCHDOG_SCRIPT = r"""
import ctypes as ct, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
from polymath_amd import api
hip = ct.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ct.POINTER(ct.c_void_p), ct.c_size_t]
hip.hipStreamCreate.argtypes = [ct.POINTER(ct.c_void_p)]
hip.hipMemcpyAsync.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_size_t, ct.c_int, ct.c_void_p]
hip.hipStreamSynchronize.argtypes = [ct.c_void_p]
ctx = api.Context(0)
st = ct.c_void_p()
assert hip.hipStreamCreate(ct.byref(st)) == 0
big, huge = 1 << 30, 16 << 30
a, b, s, r, hs, hr = (ct.c_void_p() for _ in range(6))
for p, n in ((a, big), (b, big), (s, 4096), (r, 4096), (hs, huge), (hr, huge)):
    assert hip.hipMalloc(ct.byref(p), n) == 0

def wait_failed(comm, seconds=20):
    deadline = time.time() + seconds
    while not comm.failed and time.time() < deadline:
        time.sleep(0.01)
    return comm.failed

# 1. healthy, with work queued AHEAD of the collective that takes several deadlines: the clock of a collective starts when the
#    collective reaches the head of its stream, so it completes (the round-3 watchdog counted from the enqueue and aborted here)
comm = api.Comm.rccl(api.Comm.rccl_unique_id(), 0, 1, 0)
assert comm.kind == "rccl" and not comm.failed
comm.set_timeout_ms(60)
for _ in range(150):                                                 # ~0.2 s of copies ahead, 3 deadlines' worth
    assert hip.hipMemcpyAsync(b, a, big, 3, st) == 0
comm.all_to_all_device(s.value, r.value, 4096, st.value)
assert hip.hipStreamSynchronize(st) == 0
time.sleep(0.1)
assert not comm.failed, comm.last_error()
print("HEALTHY behind queued work")
# 2. the collective ITSELF misses its deadline: 16 GiB through a world-of-one all-to-all take ~15 ms, the deadline is 2 ms
comm.set_timeout_ms(2)
t0 = time.time()
comm.all_to_all_device(hs.value, hr.value, huge, st.value)
hip.hipStreamSynchronize(st)
assert wait_failed(comm), "the watchdog did not fire"
print("FIRED after %.2f s: %s" % (time.time() - t0, comm.last_error()))
try:
    comm.all_gather(np.arange(3, dtype=np.int64))
    print("STILL ALIVE")
except api.PolymathError as e:
    print("STICKY", e.status)
# 3. a stream that never gets to the collective is bounded too: 8 deadlines from the enqueue
comm2 = api.Comm.rccl(api.Comm.rccl_unique_id(), 0, 1, 0)
comm2.set_timeout_ms(10)
for _ in range(600):                                                 # ~0.8 s of copies ahead of an 80 ms ceiling
    assert hip.hipMemcpyAsync(b, a, big, 3, st) == 0
comm2.all_to_all_device(s.value, r.value, 4096, st.value)
assert wait_failed(comm2), "the ceiling did not fire"
print("CEILING: %s" % comm2.last_error())
hip.hipStreamSynchronize(st)
"""


@pytest.mark.gpu
def test_rccl_watchdog_aborts_a_collective_that_misses_its_deadline(tmp_path):
    """The RCCL half of the fail-fast contract, on the one GPU of this box (world of one).  (1) A collective queued behind several
    deadlines' worth of other work on its stream completes: its clock starts when it reaches the head of the stream (ADVICE r3:
    it used to start at the enqueue).  (2) A collective that itself takes longer than the deadline -- 16 GiB against 2 ms -- makes
    the watchdog thread call ncclCommAbort and mark the communicator failed; later collectives return PM_ERR_COMM at once.
    (3) A stream that does not reach the collective within 8 deadlines fails it as well.
    (A real dead PEER cannot be staged on one GPU -- RCCL refuses two ranks on one device; this exercises the watchdog, the abort
    and the sticky failure on the real library.)  Runs in a child process: a misbehaving abort must not take pytest down."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "watchdog.py"
    script.write_text(_WATCHDOG_SCRIPT)
    run = subprocess.run([sys.executable, str(script), root], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-3000:])
    assert "HEALTHY behind queued work" in run.stdout
    assert "FIRED" in run.stdout and "did not complete within 2 ms" in run.stdout and "STICKY 9" in run.stdout, run.stdout
    assert "CEILING" in run.stdout and "was not reached by its stream within 80 ms" in run.stdout, run.stdout
    assert run.stderr.count("aborting the RCCL communicator") == 2


@pytest.mark.gpu
def test_config_2p22_eight_ranks_vector_sharded():
    """BASELINE configs[2] as named -- the 2^22-100-gate circuit (n = 2^23, 117 M MSM pairs) proved as ONE proof by 8 ranks
    (threads of this process on the one GPU of the box, pm_comm_local_create), each holding 1/8 of the key, of the rows, of
    the coefficients (B = 2^17 per block), of the quotient and of the MSM pairs: byte-identical to the single-GPU proof,
    which the pairing verifier accepts (test_gpu_configs.py::test_config_2p22_one_gpu)."""
    from polymath_amd import circuits as PC
    from polymath_amd.polymath import Polymath
    curve, N = "bls12_381", 8
    c = CURVES[curve]
    lc = PC.synthetic_r1cs_native(curve, (1 << 22) - 100)
    g = PC.SplitMix64(0x2222)
    x, z, r_a = g.fr(c.r), g.fr(c.r), [g.fr(c.r), g.fr(c.r)]
    ref_pm = Polymath(curve, "merlin", device=0)
    ref_pk = ref_pm.setup(lc, x, z)
    ref = ref_pm.prove_native(ref_pk, lc.inst_limbs, lc.wit_limbs, r_a)
    ref_pk.free()
    ref_pm.ctx.close()
    pms, pks, comms, proofs = _sharded_proofs(curve, lc, x, z, r_a, N)
    assert all(p == ref for p in proofs)
    assert sum(pk.msm_plan(2)[0] for pk in pks) == 10 * (1 << 23) + 22          # the quotient's pairs, each exactly once
    for pk in pks:
        pk.free()


@pytest.mark.gpu
def test_bench_multi_gpu_launch_path_with_a_world_of_one():
    """What a one-GPU box can check of the driver's multi-GPU launch: bench.py under torch.distributed.run with the `nccl`
    (RCCL) process group, BENCH_FORCE_VECTOR=1 -- the library's own RCCL communicator (dlopen, ncclCommInitRank from the id
    broadcast over torch.distributed) next to torch's, a PM_SHARD_VECTOR key, ncclAllToAll / ncclAllGather inside the phases.
    Same proof bytes as the plain single-GPU run, and the JSON line says the native communicator was used."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "1", "--warmup", "0", "--log-constraints", "12", "--no-cpu-baseline", "--msm-micro", "", "--no-live-traffic", "--other-configs", ""]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + common, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    env = dict(os.environ, BENCH_FORCE_VECTOR="1")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                          "--master-port", "29546", os.path.join(root, "bench.py"), "--gpus", "1"] + common,
                         capture_output=True, text=True, timeout=600, env=env)
    assert two.returncode == 0, two.stderr[-3000:]
    j1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    j2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert j1["proof_bytes"] == j2["proof_bytes"]
    assert "rccl (native" in j2["config"]["parallelism"], j2["config"]["parallelism"]
    # the collective fallback when RCCL cannot be used inside the library: pm_comm callbacks over the same nccl process group
    env["BENCH_NO_RCCL"] = "1"
    three = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                            "--master-port", "29547", os.path.join(root, "bench.py"), "--gpus", "1"] + common,
                           capture_output=True, text=True, timeout=600, env=env)
    assert three.returncode == 0, three.stderr[-3000:]
    j3 = json.loads([l for l in three.stdout.splitlines() if l.startswith("{")][-1])
    assert j3["proof_bytes"] == j1["proof_bytes"] and "torch.distributed callbacks (nccl)" in j3["config"]["parallelism"]


@pytest.mark.gpu
def test_config_2p24_eight_ranks_vector_sharded(monkeypatch):
    """BASELINE configs[3] as named: the 2^24-100-gate circuit (n = 2^25, 470 M MSM pairs), "NTT domain + MSM both 8-way
    partitioned" -- 8 ranks (threads on the one GPU of the box) each holding 1/8 of the key (5.6 GB of bases), 2^22 rows,
    2^22 coefficients in 8 blocks of 2^19, 42 M quotient pairs; local transforms of 2^22 points, four all-to-alls of 16 MB
    blocks per rank.  Byte-identical to the single-GPU proof (which test_gpu_configs.py::test_config_2p24_one_gpu_piece_split
    checks with the pairing verifier).  The ranks' keys are built without window tables: eight ranks' tables do not fit ONE
    GPU's 288 GB (on eight GPUs they do); the table pipeline is covered at the smaller sizes."""
    from polymath_amd import circuits as PC
    from polymath_amd.polymath import Polymath
    curve, N = "bls12_381", 8
    c = CURVES[curve]
    lc = PC.synthetic_r1cs_native(curve, (1 << 24) - 100)
    g = PC.SplitMix64(0x2424)
    x, z, r_a = g.fr(c.r), g.fr(c.r), [g.fr(c.r), g.fr(c.r)]
    ref_pm = Polymath(curve, "merlin", device=0)
    ref_pk = ref_pm.setup(lc, x, z)
    ref = ref_pm.prove_native(ref_pk, lc.inst_limbs, lc.wit_limbs, r_a)
    ref_pk.free()
    ref_pm.ctx.close()
    pms, pks, comms, proofs = _sharded_proofs(curve, lc, x, z, r_a, N, options={"tables": "off"})
    assert all(p == ref for p in proofs)
    for pk in pks:
        pk.free()
    for pm in pms:
        pm.ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_vector_sharded_tiny_domains_and_ragged_shapes(curve):
    """The smallest shapes the layout admits: the reference's dummy circuit (tests/dummy.rs: n = 8, N = 2, blocks of B = 2),
    5 gates on n = 16 with N = 4 (B = 1: every block is one coefficient, rank 3's h block is empty), the reference's bench
    shape with unused witnesses (bases at infinity) and 7 gates, and a domain with a ragged tail of zero rows (nr = 9 -> 22 rows
    of 32).  Byte-identical to the unsharded proof each time."""
    from polymath_amd import circuits as PC
    from polymath_amd.polymath import Field, LimbCircuit, Polymath, _csr
    c = CURVES[curve]
    f = Field(curve)
    g = PC.SplitMix64(0x717)

    def limb_circuit(r1cs, inst, wit):
        return LimbCircuit(f, r1cs.m0, r1cs.mw, r1cs.nr, (_csr(f, r1cs.a), _csr(f, r1cs.b), _csr(f, r1cs.c)), f.fr_limbs(inst), f.fr_limbs(wit))
    pm0 = Polymath(curve, "keccak256", device=0)
    cases = []
    cases.append((limb_circuit(*pm0._synthesize(PC.DummyCircuit(g.fr(c.r), g.fr(c.r)))), [2]))
    cases.append((limb_circuit(*PC.synthetic_r1cs(c.r, 5)), [2, 4]))
    cases.append((limb_circuit(*pm0._synthesize(PC.BenchCircuit(g.fr(c.r), g.fr(c.r), 9, 7))), [2, 4]))
    cases.append((limb_circuit(*PC.synthetic_r1cs(c.r, 9)), [2, 4]))
    for lc, ranks in cases:
        x, z, r_a = g.fr(c.r), g.fr(c.r), [g.fr(c.r), g.fr(c.r)]
        ref_pk = pm0.setup(lc, x, z)
        ref = pm0.prove_native(ref_pk, lc.inst_limbs, lc.wit_limbs, r_a)
        for N in ranks:
            assert ref_pk.n % (N * N) == 0
            pms, pks, comms, proofs = _sharded_proofs(curve, lc, x, z, r_a, N, transcript="keccak256")
            assert all(p == ref for p in proofs), (lc.nr, N)
            for pk in pks:
                pk.free()
        ref_pk.free()


@pytest.mark.gpu
def test_vector_sharded_keys_loaded_from_an_existing_key():
    """pm_pk_load_sharded(layout = PM_SHARD_VECTOR): every rank uploads only ITS pieces of an existing ProvingKey given in
    arkworks' 104-byte G1Affine layout (x, y, infinity flag) -- the path a Rust host takes -- and the
    4 ranks' proof equals the proof of the key generated on one GPU.  Exported bases of a shard match the whole key's."""
    from polymath_amd import api, circuits as PC
    from polymath_amd.polymath import Polymath
    curve, N = "bls12_381", 4
    c = CURVES[curve]
    lc = PC.synthetic_r1cs_native(curve, 1000)
    g = PC.SplitMix64(0x10AD)
    x, z, r_a = g.fr(c.r), g.fr(c.r), [g.fr(c.r), g.fr(c.r)]
    pm0 = Polymath(curve, "merlin", device=0)
    whole = pm0.setup(lc, x, z)
    ref = pm0.prove_native(whole, lc.inst_limbs, lc.wit_limbs, r_a)
    arrays = []
    for i in range(6):
        b = whole.export_bases(i)
        wide = np.zeros((b.shape[0], 13), dtype=np.uint64)
        wide[:, :12] = b
        wide[:, 12] = (~b.any(axis=1)).astype(np.uint64)
        arrays.append(wide)
    comms = api.Comm.local_group(N)
    pms = [Polymath(curve, "merlin", device=0) for _ in range(N)]
    pks = []
    for r in range(N):
        pms[r].ctx.set_comm(comms[r])
        pk = api.ProvingKey.load(pms[r].ctx, curve, whole.n, lc.m0, lc.mw, lc.nr, whole.sigma, *lc.csrs, arrays, shard_rank=r,
                                 shard_count=N, layout="vector")
        pk.omega = whole.omega
        pks.append(pk)
    assert all(p == ref for p in _run_ranks(N, lambda r: pms[r].prove_native(pks[r], lc.inst_limbs, lc.wit_limbs, r_a)))
    # a shard can export exactly the bases it holds: piece (cat_lo, count) of MSM [a] <-> x_powers[...]
    off_xp = 2 * lc.m0 + lc.mw + lc.nr + (whole.n - 1)
    lo, cnt = pks[1].msm_pieces(0)[0]
    assert np.array_equal(pks[1].export_bases(api.X_POWERS, lo - off_xp, cnt), arrays[api.X_POWERS][lo - off_xp:lo - off_xp + cnt, :12])
    with pytest.raises(api.PolymathError):
        pks[1].export_bases(api.X_POWERS, 0, 4)            # not resident on rank 1
    for pk in pks:
        pk.free()
    whole.free()


@pytest.mark.gpu
def test_vector_sharded_reference_bench_circuit_skew():
    """The reference's own bench circuit (benches/bench.rs:38-61: every padding witness carries the same value, so one bucket per
    window of the [c] MSM is hot) at 2^14 - 100 constraints on 4 vector-sharded ranks: the z_tail slices are pure repetitions, the
    hot buckets go through the parallel fold on every rank; same proof as on one GPU."""
    from polymath_amd import circuits as PC
    from polymath_amd.polymath import Field, LimbCircuit, Polymath, _csr
    curve, N = "bls12_381", 4
    c = CURVES[curve]
    f = Field(curve)
    g = PC.SplitMix64(0xBEAC4)
    nc = (1 << 14) - 100
    pm0 = Polymath(curve, "merlin", device=0)
    r1cs, inst, wit = pm0._synthesize(PC.BenchCircuit(g.fr(c.r), g.fr(c.r), nc, nc))
    assert len(set(wit[2:])) == 1
    lc = LimbCircuit(f, r1cs.m0, r1cs.mw, r1cs.nr, (_csr(f, r1cs.a), _csr(f, r1cs.b), _csr(f, r1cs.c)), f.fr_limbs(inst), f.fr_limbs(wit))
    x, z, r_a = g.fr(c.r), g.fr(c.r), [g.fr(c.r), g.fr(c.r)]
    ref_pk = pm0.setup(lc, x, z)
    ref = pm0.prove_native(ref_pk, lc.inst_limbs, lc.wit_limbs, r_a)
    pms, pks, comms, proofs = _sharded_proofs(curve, lc, x, z, r_a, N)
    assert all(p == ref for p in proofs)
    for pk in pks:
        pk.free()
    ref_pk.free()


@pytest.mark.gpu
def test_all_to_all_block_order_local_group():
    """The block order every pm_comm implementation must have, and the one distributed.make_comm probes an RCCL communicator
    for before a prover depends on it: block p of the send buffer goes to rank p, block r of the receive buffer comes from
    rank r.  Here: the in-process group (4 rank threads on one GPU), 64-byte blocks tagged (sender, receiver)."""
    import threading
    from polymath_amd import api
    N = 4
    comms = api.Comm.local_group(N)
    hip = _Hip()
    got, errs = [None] * N, []

    def rank_main(r):
        try:
            src = np.repeat(np.arange(r * N, r * N + N, dtype=np.int64), 8)
            d_send, d_recv = hip.upload(src), hip.upload(np.full(8 * N, -1, dtype=np.int64))
            comms[r].all_to_all_device(d_send.value, d_recv.value, 64)
            got[r] = hip.download(d_recv, src).reshape(N, 8)[:, 0].tolist()
            hip.free(d_send, d_recv)
        except Exception as e:   # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(N)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert not errs, errs
    assert got == [[q * N + r for q in range(N)] for r in range(N)]
    for c in comms:
        c.close()
