#!/usr/bin/env python3
"""Generates tests/golden/*.json from the big-integer restatement oracle/pyref ALONE (no C++ oracle,
no GPU code): the vectors every other implementation is pinned against.

PARITY UNPINNED w.r.t. the reference binary (no Rust toolchain, no reference golden vectors); the
proof fixtures are additionally checked here by the pairing verifier before being written.

Run: python tools/gen_golden.py        (about one minute)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyref import circuits as CI, pairing as PA, protocol as PR, serialize as SE, transcripts as T  # noqa: E402
from oracle.pyref.fields import CURVES, g1_msm_naive, g1_mul  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
H = lambda v: hex(v)
PT = lambda P: None if P is None else [hex(P[0]), hex(P[1])]


def field_and_curve_kats():
    out = {}
    for name, c in CURVES.items():
        g = CI.SplitMix64(0xF1E1D)
        fr = [[H(a), H(b), H(a * b % c.r), H((a + b) % c.r), H((a - b) % c.r), H(pow(a, -1, c.r))]
              for a, b in [(g.fr(c.r), g.fr(c.r)) for _ in range(8)] + [(c.r - 1, c.r - 1), (1, c.r - 1), (2, (c.r + 1) // 2)]]
        fq = []
        for _ in range(8):
            a = (g.fr(c.r) * g.fr(c.r)) % c.p
            b = (g.fr(c.r) * g.fr(c.r) + 12345) % c.p
            fq.append([H(a), H(b), H(a * b % c.p), H((a + b) % c.p), H((a - b) % c.p)])
        fq.append([H(c.p - 1), H(c.p - 1), H(1), H(c.p - 2), H(0)])
        muls = []
        for k in [1, 2, 3, c.r - 1, g.fr(c.r), g.fr(c.r)]:
            muls.append([H(k), PT(g1_mul(c, c.g1, k))])
        out[name] = dict(p=H(c.p), r=H(c.r), fr_R=H(c.fr_R), fq_R=H(c.fq_R), two_adic_root=H(c.two_adic_root),
                         fr=fr, fq=fq, g1=PT(c.g1), g1_muls=muls)
    return out


def ntt_msm_vectors():
    out = {}
    for name, c in CURVES.items():
        g = CI.SplitMix64(0xA77)
        ntts = []
        for log_n in [0, 1, 2, 3, 5]:
            n = 1 << log_n
            vals = [g.fr(c.r) for _ in range(n)]
            ntts.append(dict(log_n=log_n, input=[H(v) for v in vals],
                             fwd=[H(v) for v in PR.ntt_naive(c, vals, n)],
                             inv=[H(v) for v in PR.ntt_naive(c, vals, n, True)]))
        msms = []
        for ln in [1, 2, 7, 33]:
            pts = [g1_mul(c, c.g1, g.fr(c.r)) for _ in range(ln)]
            sc = [g.fr(c.r) for _ in range(ln)]
            if ln >= 7:
                pts[2] = None          # base at infinity
                sc[3] = 0              # zero scalar
                sc[4] = 1
                sc[5] = c.r - 1        # == -1
                pts[6] = pts[1]        # repeated base
                sc[6] = sc[1]          # ... with the same scalar (forces a doubling inside a bucket)
            msms.append(dict(bases=[PT(p) for p in pts], scalars=[H(s) for s in sc], result=PT(g1_msm_naive(c, pts, sc))))
        out[name] = dict(ntt=ntts, msm=msms)
    return out


def r1cs_json(q):
    row = lambda rw: [[H(v), j] for v, j in rw]
    return dict(m0=q.m0, mw=q.mw, a=[row(r) for r in q.a], b=[row(r) for r in q.b], c=[row(r) for r in q.c])


def proof_fixtures():
    c = CURVES["bls12_381"]
    TR = T.make_transcripts(c)
    g = CI.SplitMix64(0xD00D)
    cases = [("dummy", ) + CI.dummy_circuit(c, g.fr(c.r), g.fr(c.r)),
             ("mimc2", ) + CI.mimc_circuit(c, g.fr(c.r), g.fr(c.r), [g.fr(c.r) for _ in range(2)]),
             ("synthetic6", ) + CI.synthetic_r1cs(c, 6),
             ("bench_shape", ) + CI.bench_circuit(c, g.fr(c.r), g.fr(c.r), 7, 6)]
    # a row with a duplicated column: m_at (common.rs:100-105) only sees the first entry
    q, inst, wit = CI.dummy_circuit(c, 3, 5)
    q.a[0] = [(1, 2), (7, 2)]
    cases.append(("dup_column", q, inst, wit))
    # m0 != 2: no public input at all; two; eleven (2 m0 > 16: the witness-only part of u is no longer a
    # short direct sum).  Rows with several entries that touch column 0 and the instance columns, duplicate columns, zero
    # coefficients, unused witnesses (CI.random_r1cs); 2 (m0 + nr) = 2^k exactly, 2^k + 2, 2^k exactly.
    for name, seed, m0, nr in [("m0_1", 0x101, 1, 3), ("m0_3", 0x103, 3, 6), ("m0_12", 0x10C, 12, 4)]:
        cases.append((name, ) + CI.random_r1cs(c, seed, m0, nr))
    out = []
    for name, q, inst, wit in cases:
        x, z = g.fr(c.r), g.fr(c.r)
        pk = PR.generate_proving_key(c, q, x, z)
        vk = PA.make_vk(pk)
        r_a = [g.fr(c.r), g.fr(c.r)]
        entry = dict(name=name, curve="bls12_381", r1cs=r1cs_json(q), instance=[H(v) for v in inst], witness=[H(v) for v in wit],
                     x_trapdoor=H(x), z_trapdoor=H(z), r_a=[H(v) for v in r_a], n=pk.n, sigma=pk.sigma, omega=H(pk.omega),
                     bases={nm: [PT(p) for p in getattr(pk, nm)] for nm in
                            ["x_powers_g1", "x_powers_y_alpha_g1", "x_powers_y_gamma_g1", "x_powers_y_gamma_z_g1",
                             "x_powers_zh_by_y_alpha_g1", "uj_wj_lcs_by_y_alpha_g1"]}, proofs={})
        for tname in ["merlin", "keccak256", "blake3"]:
            tr = {}
            proof = PR.create_proof_with_assignment(c, pk, inst, wit, r_a, TR[tname], tr)
            assert PR.verify_proof(c, vk, proof, inst[1:], TR[tname], PA.pairing_check), (name, tname)
            entry["proofs"][tname] = dict(a_g1=PT(proof["a_g1"]), c_g1=PT(proof["c_g1"]), a_at_x1=H(proof["a_at_x1"]),
                                          d_g1=PT(proof["d_g1"]), x1=H(tr["x1"]), x2=H(tr["x2"]),
                                          bytes=SE.ser_proof(c, proof).hex())
            if tname == "keccak256":
                entry["trace"] = {k: [H(v) for v in tr[k]] for k in ["u_evals", "w_evals", "u", "w", "h", "wit_u", "z_tail", "quotient"]}
        print("fixture", name, "n =", pk.n, "verified x3")
        out.append(entry)
    return out


def proof_fixtures_bn254():
    """BN254 (BASELINE configs[4]: second curve, 254-bit limb path): same shape as proofs.json; every proof is checked
    by the verifier (verifier.rs:19-62) over the BN254 optimal-ate pairing of oracle/pyref/pairing.py before it is
    written, and a tampered one must be rejected."""
    E = PA.ENGINES["bn254"]
    c = CURVES["bn254"]
    TR = T.make_transcripts(c)
    g = CI.SplitMix64(0xB254)
    cases = [("dummy", ) + CI.dummy_circuit(c, g.fr(c.r), g.fr(c.r)),
             ("synthetic6", ) + CI.synthetic_r1cs(c, 6),
             ("bench_shape", ) + CI.bench_circuit(c, g.fr(c.r), g.fr(c.r), 7, 6),
             ("m0_12", ) + CI.random_r1cs(c, 0x20C, 12, 3),      # 2 (m0 + nr) = 2^k - 2
             ("m0_1", ) + CI.random_r1cs(c, 0x201, 1, 5)]
    out = []
    for name, q, inst, wit in cases:
        x, z = g.fr(c.r), g.fr(c.r)
        pk = PR.generate_proving_key(c, q, x, z)
        r_a = [g.fr(c.r), g.fr(c.r)]
        entry = dict(name=name, curve="bn254", r1cs=r1cs_json(q), instance=[H(v) for v in inst], witness=[H(v) for v in wit],
                     x_trapdoor=H(x), z_trapdoor=H(z), r_a=[H(v) for v in r_a], n=pk.n, sigma=pk.sigma, omega=H(pk.omega),
                     bases={nm: [PT(p) for p in getattr(pk, nm)] for nm in
                            ["x_powers_g1", "x_powers_y_alpha_g1", "x_powers_y_gamma_g1", "x_powers_y_gamma_z_g1",
                             "x_powers_zh_by_y_alpha_g1", "uj_wj_lcs_by_y_alpha_g1"]}, proofs={})
        for tname in ["merlin", "keccak256", "blake3"]:
            tr = {}
            proof = PR.create_proof_with_assignment(c, pk, inst, wit, r_a, TR[tname], tr)
            entry["proofs"][tname] = dict(a_g1=PT(proof["a_g1"]), c_g1=PT(proof["c_g1"]), a_at_x1=H(proof["a_at_x1"]),
                                          d_g1=PT(proof["d_g1"]), x1=H(tr["x1"]), x2=H(tr["x2"]),
                                          bytes=SE.ser_proof(c, proof).hex())
            vk = E.make_vk(pk)
            assert PR.verify_proof(c, vk, proof, inst[1:], TR[tname], E.pairing_check), (name, tname)
            assert not PR.verify_proof(c, vk, dict(proof, a_at_x1=(proof["a_at_x1"] + 1) % c.r), inst[1:], TR[tname], E.pairing_check)
            if tname == "keccak256":
                entry["trace"] = {k: [H(v) for v in tr[k]] for k in ["u_evals", "w_evals", "u", "w", "h", "wit_u", "z_tail", "quotient"]}
        print("bn254 fixture", name, "n =", pk.n, "verified x3")
        out.append(entry)
    return out


def pk_wire_fixtures():
    """ProvingKey / VerifyingKey wire bytes of the small proof fixtures: same circuits and
    trapdoors as proofs.json (which must exist), so a key loaded from these bytes reproduces those proofs."""
    c = CURVES["bls12_381"]
    cases = {e["name"]: e for e in json.load(open(os.path.join(OUT, "proofs.json")))}
    I = lambda h: int(h, 16)
    out = []
    for name in ["dummy", "dup_column", "mimc2", "m0_1", "m0_12"]:      # round 5: keys whose SAP header says m0 = 1 and 12
        e = cases[name]
        rows = lambda m: [[(I(v), j) for v, j in rw] for rw in m]
        q = PR.R1CS(e["r1cs"]["m0"], e["r1cs"]["mw"], rows(e["r1cs"]["a"]), rows(e["r1cs"]["b"]), rows(e["r1cs"]["c"]))
        pk = PR.generate_proving_key(c, q, I(e["x_trapdoor"]), I(e["z_trapdoor"]))
        vk = PA.make_vk(pk)
        pkb = SE.ser_pk(c, pk, vk)
        vk2, sap, vecs = SE.deser_pk(c, pkb)
        assert vk2 == vk and sap == (q.m0, q.mw, q.nr, q.a, q.b, q.c) and all(vecs[n] == getattr(pk, n) for n in SE.PK_VECTORS)
        G2 = lambda Q: [[H(Q[0][0]), H(Q[0][1])], [H(Q[1][0]), H(Q[1][1])]]
        out.append(dict(name=name, vk=dict(one_g2=G2(vk["one_g2"]), x_g2=G2(vk["x_g2"]), z_g2=G2(vk["z_g2"]), bytes=SE.ser_vk(c, vk).hex()),
                        pk_bytes=pkb.hex()))
        print("pk wire fixture", name, len(pkb), "bytes, round trip ok")
    kats = dict(g1_generator=SE.ser_g1(c, c.g1).hex(), g2_generator=SE.ser_g2(c, PA.make_vk(pk)["one_g2"]).hex(),
                g1_infinity=SE.ser_g1(c, None).hex(), g2_infinity=SE.ser_g2(c, None).hex())
    return dict(kats=kats, keys=out)


def main():
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1:]
    for fname, fn in [("field_curve_kats.json", field_and_curve_kats), ("ntt_msm.json", ntt_msm_vectors),
                      ("proofs.json", proof_fixtures), ("proofs_bn254.json", proof_fixtures_bn254), ("pk_wire.json", pk_wire_fixtures)]:
        if only and fname not in only:
            continue
        with open(os.path.join(OUT, fname), "w") as f:
            json.dump(fn(), f, separators=(",", ":"))
        print("wrote", fname)


if __name__ == "__main__":
    main()
