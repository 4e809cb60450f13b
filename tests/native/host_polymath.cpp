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
    Fr fr() {   // 4 draws, limb 0 first, masked to 255 bits, rejection; canonical value -> Montgomery
        for (;;) {
            uint64_t l[4];
            for (int i = 0; i < 4; ++i) l[i] = next();
            l[3] &= (1ull << 63) - 1;
            uint8_t buf[64] = {0};
            memcpy(buf, l, 32);
            Fr out;
            if (F::from_random_bytes(buf, &out)) return out;
        }
    }
};

struct DummyCircuit {   // tests/dummy.rs:20-35
    Fr a, b;
    void generate_constraints(ConstraintSystem<Curve> &cs) const {
        Variable va = cs.new_witness_variable(a), vb = cs.new_witness_variable(b);
        Variable vc = cs.new_input_variable(F::mul(a, b));
        cs.enforce_constraint({{Fr::one(), va}}, {{Fr::one(), vb}}, {{Fr::one(), vc}});
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

template <class T>
static void run_dummy(Context &ctx, const char *tname, uint64_t seed) {
    SplitMix64 g{seed};
    Fr a = g.fr(), b = g.fr(), x = g.fr(), z = g.fr();
    Fr r_a[2] = {g.fr(), g.fr()};
    Polymath<Curve, T> pm(ctx);
    DummyCircuit c{a, b};
    ProvingKey<Curve> pk = pm.setup(c, x, z);
    Proof<Curve> proof = pm.prove(pk, c, r_a);
    printf("dummy %s n=%llu %s\n", tname, (unsigned long long)pk.n, to_hex(proof.to_bytes()).c_str());
    // tests/dummy.rs:69-72: assert!(Polymath::verify(&vk, &[product], &proof))
    VerifyingKey vk = Polymath<Curve, T>::make_vk(pk, x, z);
    std::vector<Fr> pub{F::mul(a, b)};
    bool ok = Polymath<Curve, T>::verify(vk, pub, proof);
    Proof<Curve> bad = proof;
    bad.a_at_x1 = F::add(bad.a_at_x1, Fr::one());
    bool ok_bad = Polymath<Curve, T>::verify(vk, pub, bad);
    pub[0] = F::add(pub[0], Fr::one());
    bool ok_wrong_input = Polymath<Curve, T>::verify(vk, pub, proof);
    printf("verify %s accept=%d tampered=%d wrong_input=%d\n", tname, ok, ok_bad, ok_wrong_input);
}

int main(int argc, char **argv) {
    int rounds = argc > 1 ? atoi(argv[1]) : 322, samples = argc > 2 ? atoi(argv[2]) : 3;
    try {
        Context ctx(0);
        run_dummy<MerlinFieldTranscript<Curve>>(ctx, "merlin", 101);
        run_dummy<Keccak256Transcript<Curve>>(ctx, "keccak256", 102);
        run_dummy<Blake3Transcript<Curve>>(ctx, "blake3", 103);
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
    } catch (const std::exception &e) {
        printf("ERROR %s\n", e.what());
        return 1;
    }
    return 0;
}
