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
def _run_ranks(N, fn):
    """fn(rank) on N threads; re-raises the first exception."""
    errs, outs = [None] * N, [None] * N

    def body(r):
        try:
            outs[r] = fn(r)
        except BaseException as e:     # noqa: BLE001 -- reported below
            errs[r] = e
    th = [threading.Thread(target=body, args=(r,)) for r in range(N)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e in errs:
        if e is not None:
            raise e
    return outs


def _sharded_proofs(curve, lc, x, z, r_a, N, transcript="merlin", device_assignment=False):
    from polymath_amd import api
    from polymath_amd.polymath import Polymath
    comms = api.Comm.local_group(N)
    pms = [Polymath(curve, transcript, device=0) for _ in range(N)]
    for r in range(N):
        pms[r].ctx.set_comm(comms[r])
    pks = [pms[r].setup(lc, x, z, shard_rank=r, shard_count=N, layout="vector") for r in range(N)]
    proofs = _run_ranks(N, lambda r: pms[r].prove_native(pks[r], lc.inst_limbs, lc.wit_limbs, r_a))
    return pms, pks, comms, proofs


@pytest.mark.gpu
@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
@pytest.mark.parametrize("N", [2, 4, 8])
def test_vector_sharded_proof_equals_single_gpu(curve, N):
    """5000 gates (n = 16384): every rank proves on 1/N of the rows, coefficients, scans and MSM pairs; the N proofs are
    identical to the unsharded proof.  The shards' local vectors, scattered through the layout, equal the whole ones."""
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
    whole = {w: ref_pk.tap(w, 11 * n) for w in (2, 3, 4, 5, 6, 7)}
    pms, pks, comms, proofs = _sharded_proofs(curve, lc, x, z, r_a, N)
    assert all(p == ref for p in proofs)
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
def test_vector_sharded_many_segments_public_inputs_and_errors(monkeypatch):
    """Tiny sub-segments (PM_MAX_SEG_LOG=6: hundreds of segments per rank, several per block), a circuit with 12 public
    inputs (2 m0 > 16: the witness-only part of u takes its own distributed transform), all three transcripts; an
    unsatisfied witness makes EVERY rank return PM_ERR_REMAINDER_NONZERO (no rank is left waiting in a collective)."""
    from polymath_amd import circuits as PC
    from polymath_amd.polymath import ConstraintSystem, LimbCircuit, Polymath, Field, _csr
    from polymath_amd import api
    curve, N = "bls12_381", 4
    c = CURVES[curve]
    monkeypatch.setenv("PM_MAX_SEG_LOG", "6")
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
        pms, pks, comms, proofs = _sharded_proofs(curve, lc, x, z, r_a, N, transcript=tname)
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
    assert _run_ranks(N, prove_bad) == [4] * N
    # the contexts are still usable afterwards
    assert all(p == ref for p in _run_ranks(N, lambda r: pms[r].prove_native(pks[r], lc.inst_limbs, lc.wit_limbs, r_a)))


@pytest.mark.gpu
def test_vector_sharded_mid_size_and_pairs_layout_agree():
    """2^16-100 gates on 8 ranks (B = 2048 coefficients per block, several sub-segments per stretch): the vector-sharded
    proof, the pairs-sharded proof (vector phases replicated, pm_comm combine) and the single-GPU proof are identical."""
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
    ref_pk.free()
    pms, pks, comms, proofs = _sharded_proofs(curve, lc, x, z, r_a, N)
    assert all(p == ref for p in proofs)
    for pk in pks:
        pk.free()
    # PM_SHARD_PAIRS keys on the same contexts and communicators: the native point combine replaces the callback
    pks = [pms[r].setup(lc, x, z, shard_rank=r, shard_count=N, layout="pairs") for r in range(N)]
    assert all(p == ref for p in _run_ranks(N, lambda r: pms[r].prove_native(pks[r], lc.inst_limbs, lc.wit_limbs, r_a)))



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
    common = ["--steps", "1", "--warmup", "0", "--log-constraints", "12", "--no-cpu-baseline", "--msm-micro", "", "--no-live-traffic"]
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
    monkeypatch.setenv("PM_TABLES", "0")
    pms, pks, comms, proofs = _sharded_proofs(curve, lc, x, z, r_a, N)
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
