// C++ host mirror of the reference's prover-side API above the C ABI (include/polymath_hip.h):
//
//   reference (Rust)                                   here
//   Polymath<E, T>::setup(circuit, rng)   lib.rs:63    Polymath<Curve, T>::setup(circuit, rng)  or  setup(circuit, x, z)      trapdoors = the two rng draws
//   Polymath<E, T>::prove(pk, circuit, rng) lib.rs:72  Polymath<Curve, T>::prove(pk, circuit, rng)  or  prove(pk, circuit, r_a)   r_a = the two F::rand of prover.rs:110
//   rand::rngs::StdRng, ark_std::test_rng, F::rand     StdRng (rng.hpp), FrOps<Curve>::rand
//   trait ConstraintSynthesizer::generate_constraints  struct with generate_constraints(ConstraintSystem&)
//   trait Transcript {new, append_message, challenge}  MerlinFieldTranscript / Keccak256Transcript / Blake3Transcript
//   Proof { a_g1, c_g1, a_at_x1, d_g1 } + CanonicalSerialize (compressed)   Proof::to_bytes()
//
// Everything O(n) runs on the GPU behind pm_pk_generate / pm_prove_phase{1,2,3}; this header does the
// O(m0) scalar glue of common.rs:21-98 and the two Fiat-Shamir calls with the same field templates the
// kernels use (csrc/field.cuh compiles as plain C++).  Header-only; link with -lpolymath_hip.
#pragma once
#include <functional>
#include <stdexcept>
#include <type_traits>
#include <string>
#include <utility>
#include <vector>

#include "../../include/polymath_hip.h"
#include "../csrc/ec.cuh"
#include "hashes.hpp"
#include "pairing.hpp"
#include "rng.hpp"

namespace pmhost {

struct PolymathError : std::runtime_error {
    int phase, status;
    PolymathError(int phase_, int status_, const std::string &what) : std::runtime_error(what), phase(phase_), status(status_) {}
};

// ------------------------------------------------------------------------------- Fr helper
template <class C>
struct FrOps {
    typedef typename C::FrP P;
    typedef pm::Fp<P> Fr;
    static Fr from_u64(uint64_t v) { return pm::from_u64<P>(v); }
    static Fr zero() { return Fr::zero(); }
    static Fr one() { return Fr::one(); }
    static Fr add(const Fr &a, const Fr &b) { return pm::add<P>(a, b); }
    static Fr sub(const Fr &a, const Fr &b) { return pm::sub<P>(a, b); }
    static Fr mul(const Fr &a, const Fr &b) { return pm::mul<P>(a, b); }
    static Fr inv(const Fr &a) { return pm::inverse<P>(a); }
    static Fr pow(const Fr &a, uint64_t e) { return pm::pow_u64<P>(a, e); }
    static Fr neg(const Fr &a) { return pm::neg<P>(a); }
    template <class Rng>
    static Fr rand(Rng &rng) { return fr_rand<P, Fr>(rng); }   // F::rand(rng) (ark-ff UniformRand)
    // canonical little-endian 32 bytes (ark-serialize Fp)
    static void to_le_bytes(const Fr &a, uint8_t out[32]) {
        Fr c = pm::from_mont<P>(a);
        memcpy(out, c.l, 32);
    }
    // CanonicalDeserialize of Fp: 32 little-endian bytes, rejected when >= r
    static Fr from_le_bytes_canonical(const uint8_t b[32]) {
        Fr v;
        memcpy(v.l, b, 32);
        for (int i = 7; i >= 0; --i) {
            if (v.l[i] < P::MOD[i]) break;
            if (v.l[i] > P::MOD[i] || i == 0) throw std::runtime_error("Fr: non-canonical encoding");
        }
        return pm::to_mont<P>(v);
    }
    // F::from_be_bytes_mod_order on a 32-byte digest (keccak256.rs:36, blake3.rs:36)
    static Fr from_be_bytes_mod_order(const uint8_t d[32]) {
        Fr v;
        for (int i = 0; i < 8; ++i)
            v.l[i] = (uint32_t)d[31 - 4 * i] | ((uint32_t)d[30 - 4 * i] << 8) | ((uint32_t)d[29 - 4 * i] << 16) | ((uint32_t)d[28 - 4 * i] << 24);
        return pm::to_mont<P>(v);   // the Montgomery product reduces any 256-bit input mod r
    }
    // F::from_random_bytes on 64 bytes (merlin.rs:33): first 32 bytes little-endian, masked to the modulus
    // bit length, None if >= r  [ark-ff, from memory]
    static bool from_random_bytes(const uint8_t *buf, Fr *out) {
        Fr v;
        memcpy(v.l, buf, 32);
        int top_bits = P::BITS - 32 * 7;
        v.l[7] &= top_bits >= 32 ? 0xffffffffu : ((1u << top_bits) - 1);
        for (int i = 7; i >= 0; --i) {
            if (v.l[i] < P::MOD[i]) break;
            if (v.l[i] > P::MOD[i] || i == 0) return false;
        }
        *out = pm::to_mont<P>(v);
        return true;
    }
};

// ------------------------------------------------------------------------------ transcripts
template <class C>
struct MerlinFieldTranscript {   // transcript/merlin.rs:13-37 (the reference's default)
    typedef typename FrOps<C>::Fr Fr;
    MerlinTranscript m;
    explicit MerlinFieldTranscript(const std::string &name) : m(name) {}
    void append_message(const char *label, const Bytes &msg) { m.append_message(label, msg.data(), msg.size()); }
    Fr challenge(const char *label) {
        for (;;) {
            uint8_t buf[64];
            m.challenge_bytes(label, buf, 64);
            Fr r;
            if (FrOps<C>::from_random_bytes(buf, &r)) return r;
        }
    }
};
template <class C, Bytes (*H)(const Bytes &)>
struct HashTranscript {          // transcript/keccak256.rs:12-43, blake3.rs:12-43; `name` ignored (:19-24)
    typedef typename FrOps<C>::Fr Fr;
    Bytes t;
    explicit HashTranscript(const std::string &) {}
    void append_message(const char *label, const Bytes &msg) {
        t.insert(t.end(), label, label + strlen(label));
        t.insert(t.end(), msg.begin(), msg.end());
    }
    Fr challenge(const char *label) {
        Bytes in(t);
        in.insert(in.end(), label, label + strlen(label));
        t = H(in);
        return FrOps<C>::from_be_bytes_mod_order(t.data());
    }
};
template <class C> using Keccak256Transcript = HashTranscript<C, keccak256>;
template <class C> using Blake3Transcript = HashTranscript<C, blake3>;

// ------------------------------------------------------------------------ constraint system
struct Variable { int kind; size_t index; };   // kind 0 = One, 1 = instance, 2 = witness
static const Variable ONE{0, 0};

template <class C>
struct ConstraintSystem {   // ark-relations ConstraintSystem, the part the reference uses (generator.rs:31-54, prover.rs:33-59)
    typedef typename FrOps<C>::Fr Fr;
    typedef std::vector<std::pair<Fr, Variable>> LC;
    std::vector<Fr> instance{Fr::one()}, witness;
    std::vector<LC> a, b, c;
    Variable new_input_variable(const Fr &v) { instance.push_back(v); return Variable{1, instance.size() - 1}; }
    Variable new_witness_variable(const Fr &v) { witness.push_back(v); return Variable{2, witness.size() - 1}; }
    void enforce_constraint(const LC &la, const LC &lb, const LC &lc) { a.push_back(la); b.push_back(lb); c.push_back(lc); }
    size_t column(const Variable &v) const { return v.kind == 2 ? instance.size() + v.index : v.index; }
};

struct CsrHost {
    std::vector<uint64_t> rowptr{0};
    std::vector<uint32_t> col;
    std::vector<uint64_t> val;
    pm_csr view() const { return pm_csr{rowptr.size() - 1, rowptr.data(), col.empty() ? nullptr : col.data(), val.empty() ? nullptr : val.data()}; }
};

// ------------------------------------------------------------------------------- wire format
template <class C>
struct G1Point {
    pm::Affine<C> p;   // Montgomery, as the ABI returns it
    bool inf;
};

template <class C>
void ser_fr(const typename FrOps<C>::Fr &v, Bytes &out) {
    uint8_t b[32];
    FrOps<C>::to_le_bytes(v, b);
    out.insert(out.end(), b, b + 32);
}
inline void ser_u64(uint64_t v, Bytes &out) { for (int i = 0; i < 8; ++i) out.push_back((uint8_t)(v >> (8 * i))); }

// ark-serialize compressed G1 [ark, from memory]: BLS12-381 = 48 B big-endian x, bit7 compressed, bit6
// infinity, bit5 y lexicographically largest; BN254 = 32 B little-endian x, top byte bit7 y > -y, bit6 infinity.
template <class C>
void ser_g1(const G1Point<C> &g, Bytes &out) {
    typedef typename C::FqP Q;
    const int NB = Q::N * 4;
    if (C::ID == 0) {
        if (g.inf) { out.push_back(0xC0); out.insert(out.end(), NB - 1, 0); return; }
    } else if (g.inf) { out.insert(out.end(), NB - 1, 0); out.push_back(0x40); return; }
    pm::Fp<Q> x = pm::from_mont<Q>(g.p.x), y = pm::from_mont<Q>(g.p.y), ny = pm::from_mont<Q>(pm::neg<Q>(g.p.y));
    bool y_larger = false;   // y > -y as integers
    for (int i = Q::N - 1; i >= 0; --i)
        if (y.l[i] != ny.l[i]) { y_larger = y.l[i] > ny.l[i]; break; }
    uint8_t le[64];
    memcpy(le, x.l, NB);
    if (C::ID == 0) {
        size_t at = out.size();
        for (int i = NB - 1; i >= 0; --i) out.push_back(le[i]);
        out[at] |= 0x80 | (y_larger ? 0x20 : 0);
    } else {
        out.insert(out.end(), le, le + NB);
        if (y_larger) out.back() |= 0x80;
    }
}

template <class C>
struct Proof {   // data_structures.rs:10-19
    G1Point<C> a_g1, c_g1, d_g1;
    typename FrOps<C>::Fr a_at_x1;
    Bytes to_bytes() const {
        Bytes o;
        ser_g1<C>(a_g1, o);
        ser_g1<C>(c_g1, o);
        ser_fr<C>(a_at_x1, o);
        ser_g1<C>(d_g1, o);
        return o;
    }
};

inline std::string to_hex(const Bytes &b) {
    static const char *d = "0123456789abcdef";
    std::string s;
    for (uint8_t v : b) { s.push_back(d[v >> 4]); s.push_back(d[v & 15]); }
    return s;
}

// ------------------------------------------------------------------------------------ facade
struct Context {   // pm_ctx RAII (or a borrowed handle: pm_host_prove runs the mirror on the caller's context)
    pm_ctx *h = nullptr;
    bool owned = true;
    explicit Context(int device = 0) {
        int st = pm_ctx_create(device, &h);
        if (st) throw PolymathError(0, st, "pm_ctx_create failed (no GPU? status " + std::to_string(st) + ")");
    }
    struct Borrow {};
    Context(pm_ctx *borrowed, Borrow) : h(borrowed), owned(false) {}
    ~Context() { if (owned) pm_ctx_destroy(h); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
};

template <class C>
struct ProvingKey {   // device-resident ProvingKey (data_structures.rs:56-73) + the vk scalars the prover needs
    pm_pk *h = nullptr;
    uint64_t n = 0, m0 = 0, sigma = 0;
    typename FrOps<C>::Fr omega;
    ProvingKey() = default;
    ProvingKey(ProvingKey &&o) noexcept { *this = std::move(o); }
    ProvingKey &operator=(ProvingKey &&o) noexcept { std::swap(h, o.h); n = o.n; m0 = o.m0; sigma = o.sigma; omega = o.omega; return *this; }
    ~ProvingKey() { if (h) pm_pk_free(h); }
};

// VerifyingKey (data_structures.rs:38-52) with PairingVK (:25-35), per pairing engine: BLS12-381 (the reference's,
// Cargo.toml:35) and BN254 (BASELINE.json configs[4]).
template <class C>
struct VerifyingKeyT {
    typedef typename PairingOf<C>::type Pairing;
    G1Point<C> one_g1;
    typename Pairing::G2 one_g2, x_g2, z_g2;
    uint64_t n = 0, m0 = 0, sigma = 0;
    pm::Fp<typename C::FrP> omega;
};
typedef VerifyingKeyT<pm::BlsCurve> VerifyingKey;   // the wire format (wire.hpp) is the BLS12-381 one

template <class C, class T>
class Polymath {
public:
    typedef FrOps<C> F;
    typedef typename F::Fr Fr;
    static constexpr uint64_t MINUS_ALPHA = 3, MINUS_GAMMA = 5;   // common.rs:11,14

    explicit Polymath(Context &ctx) : ctx_(ctx) {}

    // circuit_specific_setup (lib.rs:63-70) -> generate_proving_key (generator.rs:24-167)
    template <class Circuit>
    ProvingKey<C> setup(const Circuit &circuit, const Fr &x_trapdoor, const Fr &z_trapdoor, int shard_rank = 0, int shard_count = 1) {
        ConstraintSystem<C> cs;
        circuit.generate_constraints(cs);                          // generator.rs:37
        CsrHost A = to_csr(cs, cs.a), B = to_csr(cs, cs.b), Cm = to_csr(cs, cs.c);   // cs.to_matrices(), :46
        pm_csr va = A.view(), vb = B.view(), vc = Cm.view();
        ProvingKey<C> pk;
        int st = pm_pk_generate(ctx_.h, C::ID, cs.instance.size(), cs.witness.size(), cs.a.size(), &va, &vb, &vc,
                                (const uint64_t *)x_trapdoor.l, (const uint64_t *)z_trapdoor.l, shard_rank, shard_count, &pk.h);
        if (st) throw PolymathError(0, st, std::string("pm_pk_generate: ") + pm_last_error(ctx_.h));
        uint64_t lens[PM_NUM_BASE_VECS];
        pm_pk_info(pk.h, &pk.n, &pk.m0, &pk.sigma, (uint64_t *)pk.omega.l, lens);
        return pk;
    }

    // the reference's own signature: the two trapdoors are drawn from `rng` exactly as generate_proving_key does
    // (generator.rs:72,77: sample_element_outside_domain twice, x then z)
    template <class Circuit, class Rng, class = decltype(std::declval<Rng &>().next_u64())>
    ProvingKey<C> setup(const Circuit &circuit, Rng &rng) {
        ConstraintSystem<C> cs;
        circuit.generate_constraints(cs);
        uint64_t n = 1;
        while (n < 2 * (cs.instance.size() + cs.a.size())) n <<= 1;          // Radix2EvaluationDomain::new, generator.rs:60
        auto outside = [&] {
            for (;;) {
                Fr t = F::rand(rng);
                if (!F::pow(t, n).eq(Fr::one())) return t;                    // evaluate_vanishing_polynomial(t) != 0
            }
        };
        const Fr x = outside(), z = outside();
        last_trapdoors_[0] = x;
        last_trapdoors_[1] = z;
        return setup(circuit, x, z);
    }
    // the trapdoors the rng form of setup drew (the reference returns the vk instead; make_vk needs them here)
    const Fr *last_trapdoors() const { return last_trapdoors_; }

    // prove (lib.rs:72-78) with the reference's signature: r_a = two F::rand draws (prover.rs:110), constant term first
    template <class Circuit, class Rng, class = decltype(std::declval<Rng &>().next_u64())>
    Proof<C> prove(const ProvingKey<C> &pk, const Circuit &circuit, Rng &rng) {
        ConstraintSystem<C> cs;
        circuit.generate_constraints(cs);                          // prover.rs:44 (synthesis draws nothing)
        const Fr r_a[2] = {F::rand(rng), F::rand(rng)};
        return prove_with_assignment(pk, cs.instance, cs.witness, r_a);
    }

    // prove (lib.rs:72-78) -> create_proof (prover.rs:27-64) -> create_proof_with_assignment (:66-237)
    template <class Circuit>
    Proof<C> prove(const ProvingKey<C> &pk, const Circuit &circuit, const Fr r_a[2]) {
        ConstraintSystem<C> cs;
        circuit.generate_constraints(cs);                          // prover.rs:44
        return prove_with_assignment(pk, cs.instance, cs.witness, r_a);
    }

    Proof<C> prove_with_assignment(const ProvingKey<C> &pk, const std::vector<Fr> &instance, const std::vector<Fr> &witness, const Fr r_a[2]) {
        Fr dummy = Fr::zero();
        return prove_raw(pk, instance, (const uint64_t *)instance.data(), witness.empty() ? (const uint64_t *)&dummy : (const uint64_t *)witness.data(),
                         false, r_a);
    }

    // The same with the assignment given as raw Montgomery limbs, on the host or already resident in HBM
    // (pm_prove_phase1_device); `instance` is the host copy of the public inputs the transcript hashes.
    // combine(points, count): replace this rank's PARTIAL points (sharded key) by their sums over all ranks -- the
    // all-gather + pm_g1_sum of SURVEY.md §8e; null for an unsharded key.  Non-zero return aborts the proof.
    typedef std::function<int(G1Point<C> *points, int count)> Combine;

    Proof<C> prove_raw(const ProvingKey<C> &pk, const std::vector<Fr> &instance, const uint64_t *x, const uint64_t *w, bool on_device,
                       const Fr r_a[2], const Combine &combine = nullptr) {
        Proof<C> proof;
        int ai = 0, ci = 0, di = 0;
        int st = on_device ? pm_prove_phase1_device(ctx_.h, pk.h, x, w, (const uint64_t *)r_a, (uint64_t *)&proof.a_g1.p, &ai, (uint64_t *)&proof.c_g1.p, &ci)
                           : pm_prove_phase1(ctx_.h, pk.h, x, w, (const uint64_t *)r_a, (uint64_t *)&proof.a_g1.p, &ai, (uint64_t *)&proof.c_g1.p, &ci);
        if (st) throw PolymathError(1, st, "prove phase 1 failed: status " + std::to_string(st));   // == the asserts of prover.rs:107,108
        proof.a_g1.inf = ai != 0;
        proof.c_g1.inf = ci != 0;
        if (combine) {
            G1Point<C> ac[2] = {proof.a_g1, proof.c_g1};
            if (int rc = combine(ac, 2)) throw PolymathError(1, rc, "combining the phase-1 partial points failed");
            proof.a_g1 = ac[0];
            proof.c_g1 = ac[1];
        }
        T t("polymath");                                                                    // prover.rs:125, B_POLYMATH
        Fr x1 = compute_x1(t, instance, proof.a_g1, proof.c_g1);                            // :126
        Fr y1 = F::pow(x1, pk.sigma), y1_inv = F::inv(y1);                                  // :128
        Fr y1_alpha = F::pow(y1_inv, MINUS_ALPHA);                                          // :130
        Fr u_at_x1;
        st = pm_prove_phase2(ctx_.h, (const uint64_t *)x1.l, (uint64_t *)u_at_x1.l);
        if (st) throw PolymathError(2, st, "prove phase 2 failed");
        Fr ra_at = F::add(r_a[0], F::mul(r_a[1], x1));
        proof.a_at_x1 = F::add(u_at_x1, F::mul(ra_at, y1_alpha));                           // :132
        Fr y1_gamma = F::pow(y1_inv, MINUS_GAMMA);                                          // :134
        Fr pi_at_x1 = compute_pi_at_x1(pk, instance, x1, y1_gamma);                         // :135
        Fr c_at_x1 = F::mul(F::sub(F::mul(F::add(proof.a_at_x1, y1_gamma), proof.a_at_x1), pi_at_x1), F::inv(y1_alpha));   // :138, common.rs:73-75
        Fr x2 = compute_x2(t, x1, proof.a_at_x1, c_at_x1);                                  // :189
        st = pm_prove_phase3(ctx_.h, (const uint64_t *)x1.l, (const uint64_t *)x2.l, (const uint64_t *)proof.a_at_x1.l,
                             (const uint64_t *)c_at_x1.l, (uint64_t *)&proof.d_g1.p, &di);
        if (st) throw PolymathError(3, st, "prove phase 3 failed: status " + std::to_string(st));   // prover.rs:221,222
        proof.d_g1.inf = di != 0;
        if (combine)
            if (int rc = combine(&proof.d_g1, 1)) throw PolymathError(3, rc, "combining the phase-3 partial point failed");
        return proof;                                                                       // :231-236
    }

    // generator.rs:139-157: the verifying key of a proving key made from trapdoors (x, z)
    typedef typename PairingOf<C>::type Pairing;
    static VerifyingKeyT<C> make_vk(const ProvingKey<C> &pk, const Fr &x_trapdoor, const Fr &z_trapdoor) {
        VerifyingKeyT<C> vk;
        for (int i = 0; i < C::FqP::N; ++i) { vk.one_g1.p.x.l[i] = C::GX_MONT[i]; vk.one_g1.p.y.l[i] = C::GY_MONT[i]; }
        vk.one_g1.inf = false;
        vk.one_g2 = Pairing::g2_generator();
        Fr xc = pm::from_mont<typename C::FrP>(x_trapdoor), zc = pm::from_mont<typename C::FrP>(z_trapdoor);
        vk.x_g2 = Pairing::g2_mul(vk.one_g2, xc.l, 8);
        vk.z_g2 = Pairing::g2_mul(vk.one_g2, zc.l, 8);
        vk.n = pk.n; vk.m0 = pk.m0; vk.sigma = pk.sigma; vk.omega = pk.omega;
        return vk;
    }

    // verify (lib.rs:80-90) -> verify_proof (verifier.rs:19-62).  `public_inputs` WITHOUT the leading one (:26).
    static bool verify(const VerifyingKeyT<C> &vk, const std::vector<Fr> &public_inputs, const Proof<C> &proof) {
        typedef pm::XYZZ<C> J;
        T t("polymath");                                                                   // :24
        std::vector<Fr> pub{Fr::one()};
        pub.insert(pub.end(), public_inputs.begin(), public_inputs.end());                 // :26
        Fr x1 = compute_x1(t, pub, proof.a_g1, proof.c_g1);                                // :29
        Fr y1 = F::pow(x1, vk.sigma), y1_inv = F::inv(y1);                                 // :32
        Fr y1_gamma = F::pow(y1_inv, MINUS_GAMMA);                                         // :34
        ProvingKey<C> view;                                                                // n / omega carrier for compute_pi_at_x1
        view.n = vk.n; view.omega = vk.omega;
        Fr pi_at_x1 = compute_pi_at_x1(view, pub, x1, y1_gamma);                           // :35
        Fr y1_alpha = F::pow(y1_inv, MINUS_ALPHA);                                         // :37
        Fr c_at_x1 = F::mul(F::sub(F::mul(F::add(proof.a_at_x1, y1_gamma), proof.a_at_x1), pi_at_x1), F::inv(y1_alpha));   // :40
        Fr x2 = compute_x2(t, x1, proof.a_at_x1, c_at_x1);                                 // :42
        // commitments_minus_evals_in_g1 = a + x2 c - (a_at_x1 + x2 c_at_x1) [1]_1         :44-47
        auto smul = [](const G1Point<C> &g, const Fr &k_mont) {
            J acc = J::identity();
            if (g.inf) return acc;
            Fr k = pm::from_mont<typename C::FrP>(k_mont);
            for (int i = 7; i >= 0; --i)
                for (int b = 31; b >= 0; --b) {
                    acc = pm::xyzz_dbl<C>(acc);
                    if ((k.l[i] >> b) & 1) pm::xyzz_madd<C>(acc, g.p, false);
                }
            return acc;
        };
        J lhs = J::identity();
        if (!proof.a_g1.inf) pm::xyzz_madd<C>(lhs, proof.a_g1.p, false);
        lhs = pm::xyzz_add<C>(lhs, smul(proof.c_g1, x2));
        lhs = pm::xyzz_add<C>(lhs, smul(vk.one_g1, F::neg(F::add(proof.a_at_x1, F::mul(x2, c_at_x1)))));
        Fr x1c = pm::from_mont<typename C::FrP>(x1);
        typename Pairing::G2 x_minus_x1 = Pairing::g2_add(vk.x_g2, Pairing::g2_neg(Pairing::g2_mul(vk.one_g2, x1c.l, 8)));   // :48
        pm::Affine<C> lhs_aff = pm::xyzz_to_affine<C>(lhs), neg_d = proof.d_g1.p;
        neg_d.y = pm::neg<typename C::FqP>(neg_d.y);                                        // proof.d_g1 * (-1)  :53
        std::vector<typename Pairing::Pair> pairs{{lhs_aff, lhs.is_identity(), vk.z_g2}, {neg_d, proof.d_g1.inf, x_minus_x1}};
        return Pairing::product_is_one(pairs);                                             // :50-61
    }

    // common.rs:21-30
    static Fr compute_x1(T &t, const std::vector<Fr> &public_inputs, const G1Point<C> &a, const G1Point<C> &c) {
        Bytes m;
        ser_u64(public_inputs.size(), m);
        for (const Fr &v : public_inputs) ser_fr<C>(v, m);
        t.append_message("public_inputs", m);
        Bytes g;
        ser_u64(2, g);
        ser_g1<C>(a, g);
        ser_g1<C>(c, g);
        t.append_message("commitments", g);
        return t.challenge("x1");
    }
    // common.rs:32-37
    static Fr compute_x2(T &t, const Fr &x1, const Fr &a_at_x1, const Fr &c_at_x1) {
        Bytes m;
        ser_fr<C>(x1, m);
        t.append_message("x1", m);
        Bytes v;
        ser_u64(2, v);
        ser_fr<C>(a_at_x1, v);
        ser_fr<C>(c_at_x1, v);
        t.append_message("values", v);
        return t.challenge("x2");
    }
    // common.rs:49-71 with z_tilde_i :77-97
    static Fr compute_pi_at_x1(const ProvingKey<C> &pk, const std::vector<Fr> &pub, const Fr &x1, const Fr &y1_gamma) {
        const size_t m0 = pub.size();
        Fr one = Fr::one(), sum = Fr::zero();
        Fr num = F::mul(F::sub(F::pow(x1, pk.n), one), F::inv(F::from_u64(pk.n)));
        Fr w_i = one;
        for (size_t i = 0; i < 2 * m0; ++i) {
            Fr zt = i == 0 ? F::add(one, one) : i < m0 ? F::add(one, pub[i]) : i == m0 ? Fr::zero() : F::sub(one, pub[i - m0]);
            Fr li = F::mul(num, F::inv(F::sub(x1, w_i)));
            sum = F::add(sum, F::mul(zt, li));
            num = F::mul(num, pk.omega);
            w_i = F::mul(w_i, pk.omega);
        }
        return F::mul(sum, y1_gamma);
    }

private:
    Context &ctx_;
    Fr last_trapdoors_[2];
    static CsrHost to_csr(const ConstraintSystem<C> &cs, const std::vector<typename ConstraintSystem<C>::LC> &rows) {
        CsrHost m;
        for (const auto &row : rows) {
            for (const auto &term : row) {
                m.col.push_back((uint32_t)cs.column(term.second));
                const uint64_t *limbs = (const uint64_t *)term.first.l;
                m.val.insert(m.val.end(), limbs, limbs + 4);
            }
            m.rowptr.push_back(m.col.size());
        }
        return m;
    }
};

}  // namespace pmhost
