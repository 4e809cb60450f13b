// The reference's own tests, restated on the C++ host mirror (GPU required):
//   tests/dummy.rs:37-80  -- 1-constraint circuit, setup -> prove for Merlin / Keccak256 / Blake3
//   tests/mimc.rs:145-227 -- MiMC (rounds from argv), setup once, then proofs for several preimages
// RNG draws come from SplitMix64 so tests/test_gpu_host_mirror.py can regenerate the same inputs and
// compare the printed proof bytes with the CPU oracle (the reference itself only checks verify()).
#include <cstdio>
#include <cstdlib>
#include "../../polymath_amd/host/polymath.hpp"

using namespace pmhost;
typedef pm::BlsCurve Curve;
typedef FrOps<Curve> F;
typedef F::Fr Fr;

struct SplitMix64 {
    uint64_t s;
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    template <class CC>
    typename FrOps<CC>::Fr fr_of() {   // 4 draws, limb 0 first, masked to the field's bit length, rejection; canonical value -> Montgomery
        for (;;) {
            uint64_t l[4];
            for (int i = 0; i < 4; ++i) l[i] = next();
            l[3] &= (1ull << (CC::FrP::BITS - 192)) - 1;
            uint8_t buf[64] = {0};
            memcpy(buf, l, 32);
            typename FrOps<CC>::Fr out;
            if (FrOps<CC>::from_random_bytes(buf, &out)) return out;
        }
    }
    Fr fr() { return fr_of<Curve>(); }
};

template <class CC>
struct DummyCircuitT {   // tests/dummy.rs:20-35
    typename FrOps<CC>::Fr a, b;
    void generate_constraints(ConstraintSystem<CC> &cs) const {
        typedef typename FrOps<CC>::Fr FrC;
        Variable va = cs.new_witness_variable(a), vb = cs.new_witness_variable(b);
        Variable vc = cs.new_input_variable(FrOps<CC>::mul(a, b));
        cs.enforce_constraint({{FrC::one(), va}}, {{FrC::one(), vb}}, {{FrC::one(), vc}});
    }
};

struct MiMCDemo {       // tests/mimc.rs:66-143
    Fr xl, xr;
    const std::vector<Fr> *constants;
    void generate_constraints(ConstraintSystem<Curve> &cs) const {
        Fr xl_v = xl, xr_v = xr, one = Fr::one();
        Variable vxl = cs.new_witness_variable(xl_v), vxr = cs.new_witness_variable(xr_v);
        const size_t rounds = constants->size();
        for (size_t i = 0; i < rounds; ++i) {
            const Fr &ci = (*constants)[i];
            Fr t = F::add(xl_v, ci), tmp_v = F::mul(t, t);
            Variable tmp = cs.new_witness_variable(tmp_v);
            ConstraintSystem<Curve>::LC lc{{one, vxl}, {ci, ONE}};
            cs.enforce_constraint(lc, lc, {{one, tmp}});                      // mimc.rs:98-102
            Fr new_v = F::add(F::mul(t, tmp_v), xr_v);
            Variable nv = (i == rounds - 1) ? cs.new_input_variable(new_v) : cs.new_witness_variable(new_v);   // :114-121
            cs.enforce_constraint({{one, tmp}}, lc, {{one, nv}, {F::neg(one), vxr}});   // :123-127
            vxr = vxl; xr_v = xl_v;
            vxl = nv;  xl_v = new_v;
        }
    }
};

// CC = pairing engine: BLS12-381 (the reference's) or BN254 (BASELINE.json configs[4]); `tag` prefixes the printed lines
template <class CC, class T>
static void run_dummy(Context &ctx, const char *tag, const char *tname, uint64_t seed) {
    typedef FrOps<CC> FC;
    typedef typename FC::Fr FrC;
    SplitMix64 g{seed};
    FrC a = g.fr_of<CC>(), b = g.fr_of<CC>(), x = g.fr_of<CC>(), z = g.fr_of<CC>();
    FrC r_a[2] = {g.fr_of<CC>(), g.fr_of<CC>()};
    Polymath<CC, T> pm(ctx);
    DummyCircuitT<CC> c{a, b};
    ProvingKey<CC> pk = pm.setup(c, x, z);
    Proof<CC> proof = pm.prove(pk, c, r_a);
    printf("dummy%s %s n=%llu %s\n", tag, tname, (unsigned long long)pk.n, to_hex(proof.to_bytes()).c_str());
    // tests/dummy.rs:69-72: assert!(Polymath::verify(&vk, &[product], &proof))
    VerifyingKeyT<CC> vk = Polymath<CC, T>::make_vk(pk, x, z);
    std::vector<FrC> pub{FC::mul(a, b)};
    bool ok = Polymath<CC, T>::verify(vk, pub, proof);
    Proof<CC> bad = proof;
    bad.a_at_x1 = FC::add(bad.a_at_x1, FrC::one());
    bool ok_bad = Polymath<CC, T>::verify(vk, pub, bad);
    pub[0] = FC::add(pub[0], FrC::one());
    bool ok_wrong_input = Polymath<CC, T>::verify(vk, pub, proof);
    printf("verify%s %s accept=%d tampered=%d wrong_input=%d\n", tag, tname, ok, ok_bad, ok_wrong_input);
}

// tests/dummy.rs:37-80 draw for draw, on the reference's own random sources: rng = StdRng::seed_from_u64(test_rng().next_u64())
// (:44), setup(c, &mut rng) (:52), a, b = rand (:56-57), prove(&pk, circuit, &mut rng) (:67), verify (:69-72)
template <class T>
static void run_dummy_reference_rng(Context &ctx, const char *tname) {
    StdRng seed_source = StdRng::test_rng();
    StdRng rng = StdRng::seed_from_u64(seed_source.next_u64());
    Polymath<Curve, T> pm(ctx);
    DummyCircuitT<Curve> shape{Fr::zero(), Fr::zero()};                  // `DummyCircuit { a: None, b: None }`: shape only
    ProvingKey<Curve> pk = pm.setup(shape, rng);
    const Fr x = pm.last_trapdoors()[0], z = pm.last_trapdoors()[1];
    Fr a = F::rand(rng), b = F::rand(rng);
    DummyCircuitT<Curve> c{a, b};
    Proof<Curve> proof = pm.prove(pk, c, rng);
    VerifyingKey vk = Polymath<Curve, T>::make_vk(pk, x, z);
    std::vector<Fr> pub{F::mul(a, b)};
    printf("dummyrng %s %s accept=%d\n", tname, to_hex(proof.to_bytes()).c_str(), (int)Polymath<Curve, T>::verify(vk, pub, proof));
}

int main(int argc, char **argv) {
    int rounds = argc > 1 ? atoi(argv[1]) : 322, samples = argc > 2 ? atoi(argv[2]) : 3;
    try {
        Context ctx(0);
        run_dummy<Curve, MerlinFieldTranscript<Curve>>(ctx, "", "merlin", 101);
        run_dummy<Curve, Keccak256Transcript<Curve>>(ctx, "", "keccak256", 102);
        run_dummy<Curve, Blake3Transcript<Curve>>(ctx, "", "blake3", 103);
        // tests/mimc.rs: constants, setup once, then SAMPLES x (random preimage, prove)
        SplitMix64 g{322};
        std::vector<Fr> constants(rounds);
        for (auto &c : constants) c = g.fr();
        Fr x = g.fr(), z = g.fr();
        Polymath<Curve, MerlinFieldTranscript<Curve>> pm(ctx);
        MiMCDemo shape{Fr::zero(), Fr::zero(), &constants};
        ProvingKey<Curve> pk = pm.setup(shape, x, z);
        for (int s = 0; s < samples; ++s) {
            MiMCDemo c{g.fr(), g.fr(), &constants};
            Fr r_a[2] = {g.fr(), g.fr()};
            Proof<Curve> proof = pm.prove(pk, c, r_a);
            printf("mimc %d n=%llu %s\n", s, (unsigned long long)pk.n, to_hex(proof.to_bytes()).c_str());
            if (s == 0) {   // tests/mimc.rs:214: assert!(Polymath::verify(&pvk, &[image], &proof))
                ConstraintSystem<Curve> cs0;
                c.generate_constraints(cs0);
                VerifyingKey vk = Polymath<Curve, MerlinFieldTranscript<Curve>>::make_vk(pk, x, z);
                std::vector<Fr> image{cs0.instance[1]};
                printf("verify mimc accept=%d\n", (int)Polymath<Curve, MerlinFieldTranscript<Curve>>::verify(vk, image, proof));
            }
        }
        // unsatisfied witness -> the reference's assert!(rem_poly.is_zero()) (prover.rs:108)
        ConstraintSystem<Curve> cs;
        MiMCDemo c{g.fr(), g.fr(), &constants};
        c.generate_constraints(cs);
        cs.witness[3] = F::add(cs.witness[3], Fr::one());
        Fr r_a[2] = {Fr::one(), Fr::one()};
        try {
            pm.prove_with_assignment(pk, cs.instance, cs.witness, r_a);
            printf("bad-witness NOT rejected\n");
        } catch (const PolymathError &e) {
            printf("bad-witness rejected phase=%d status=%d\n", e.phase, e.status);
        }
        run_dummy_reference_rng<MerlinFieldTranscript<Curve>>(ctx, "merlin");
        run_dummy_reference_rng<Keccak256Transcript<Curve>>(ctx, "keccak256");
        run_dummy_reference_rng<Blake3Transcript<Curve>>(ctx, "blake3");
        // the same dummy test on the second pairing engine (BN254: 254-bit limb path, its own optimal-ate verifier)
        run_dummy<pm::BnCurve, MerlinFieldTranscript<pm::BnCurve>>(ctx, "_bn254", "merlin", 201);
        run_dummy<pm::BnCurve, Keccak256Transcript<pm::BnCurve>>(ctx, "_bn254", "keccak256", 202);
        run_dummy<pm::BnCurve, Blake3Transcript<pm::BnCurve>>(ctx, "_bn254", "blake3", 203);
    } catch (const std::exception &e) {
        printf("ERROR %s\n", e.what());
        return 1;
    }
    return 0;
}
