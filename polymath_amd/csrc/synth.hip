// Harness workloads built natively (SURVEY.md §8 a-H): the synthetic "random A*B=C gates" R1CS the headline
// metric is quoted on (BASELINE.json configs[1..4]) -- the same draws, in the same order, as
// polymath_amd/circuits.py: synthetic_r1cs (the Python loop needs minutes at 2^24 gates) -- and the
// reference's own bench circuit (/root/reference/benches/bench.rs:38-61).  Host code, no GPU needed.
#include <cstring>
#include <vector>

#include "internal.h"

namespace {

struct SplitMix64 {
    uint64_t s;
    uint64_t next() {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
};

// 4 draws (little-endian limbs) masked to the field's bit length, rejected while >= r; -> Montgomery form
template <class P>
static pm::Fp<P> draw_fr(SplitMix64 &g) {
    for (;;) {
        pm::Fp<P> v;
        for (int i = 0; i < 4; ++i) {
            uint64_t d = g.next();
            v.l[2 * i] = (uint32_t)d;
            v.l[2 * i + 1] = (uint32_t)(d >> 32);
        }
        const unsigned top_bits = (unsigned)P::BITS - 224;   // bits kept in the top 32-bit limb
        if (top_bits < 32) v.l[7] &= (1u << top_bits) - 1;
        bool lt = false;
        for (int i = 7; i >= 0; --i) {
            if (v.l[i] != P::MOD[i]) { lt = v.l[i] < P::MOD[i]; break; }
        }
        if (lt) return pm::to_mont<P>(v);
    }
}

template <class P>
static int synth_impl(uint64_t nr, uint64_t seed, uint64_t *a_val, uint32_t *a_col, uint64_t *b_val, uint32_t *b_col,
                      uint64_t *c_val, uint32_t *c_col, uint64_t *instance, uint64_t *witness) {
    typedef pm::Fp<P> Fr;
    if (nr < 1 || nr + 4 > 0xFFFFFFFFull) return PM_ERR_INVALID_ARG;
    SplitMix64 g{seed};
    const uint32_t m0 = 2;
    // defined variables in definition order: column ids and values (column 0 = One, then the witnesses)
    std::vector<uint32_t> cols;
    std::vector<Fr> vals;
    cols.reserve(nr + 3);
    vals.reserve(nr + 3);
    Fr *wit = (Fr *)witness;
    uint64_t nwit = 0;
    wit[nwit++] = draw_fr<P>(g);
    wit[nwit++] = draw_fr<P>(g);
    cols.push_back(0); vals.push_back(Fr::one());
    cols.push_back(m0); vals.push_back(wit[0]);
    cols.push_back(m0 + 1); vals.push_back(wit[1]);
    const Fr one = Fr::one();
    Fr pub = Fr::zero();
    for (uint64_t i = 0; i < nr; ++i) {
        const Fr alpha = draw_fr<P>(g), beta = draw_fr<P>(g);
        const uint64_t pi = g.next() % cols.size(), qi = g.next() % cols.size();
        const Fr t = pm::mul<P>(pm::mul<P>(alpha, vals[pi]), pm::mul<P>(beta, vals[qi]));
        uint32_t col;
        if (i == nr - 1) {
            col = 1;
            pub = t;
        } else {
            col = m0 + (uint32_t)nwit;
            wit[nwit++] = t;
            cols.push_back(col);
            vals.push_back(t);
        }
        memcpy(a_val + 4 * i, alpha.l, 32); a_col[i] = cols[pi];
        memcpy(b_val + 4 * i, beta.l, 32);  b_col[i] = cols[qi];
        memcpy(c_val + 4 * i, one.l, 32);   c_col[i] = col;
    }
    memcpy(instance, one.l, 32);
    memcpy(instance + 4, pub.l, 32);
    return PM_OK;
}

}  // namespace

extern "C" int pm_synth_r1cs(int curve, uint64_t nr, uint64_t seed, uint64_t *a_val, uint32_t *a_col, uint64_t *b_val, uint32_t *b_col,
                             uint64_t *c_val, uint32_t *c_col, uint64_t *instance, uint64_t *witness) {
    if (!a_val || !a_col || !b_val || !b_col || !c_val || !c_col || !instance || !witness) return PM_ERR_INVALID_ARG;
    if (curve == PM_BLS12_381) return synth_impl<pm::BlsFrP>(nr, seed, a_val, a_col, b_val, b_col, c_val, c_col, instance, witness);
    if (curve == PM_BN254) return synth_impl<pm::BnFrP>(nr, seed, a_val, a_col, b_val, b_col, c_val, c_col, instance, witness);
    return PM_ERR_INVALID_ARG;
}
