// The reference's random sources, restated for the host mirror ("StdRng + Fr::rand"), so that
// Polymath<E, T>::setup(circuit, rng) / prove(pk, circuit, rng) (/root/reference/src/lib.rs:63-78) can be mirrored with the
// SAME signature and the reference's tests (tests/dummy.rs:44-67, tests/mimc.rs:153-210, benches/bench.rs:65) restated draw for
// draw.  None of these crates is vendored under /root/reference; the published algorithms are restated here:
//   rand 0.8 `StdRng` = rand_chacha `ChaCha12Rng`: the ChaCha block function (RFC 7539 section 2.3 with 12 rounds instead of
//     20) on [constants | 256-bit seed | 64-bit block counter | 64-bit stream id = 0], output words consumed in order;
//     next_u64 = two consecutive words, low first.  The 20-round form is pinned by RFC 7539's test vector (section 2.3.2).
//   rand_core 0.6 `SeedableRng::seed_from_u64`: a PCG32 stream expands the u64 into the 32 seed bytes.
//   ark-std 0.4 `test_rng()`: StdRng::from_seed of a fixed 32-byte array                      [ark, from memory]
//   ark-ff `Fp::rand`: N x next_u64 into the limbs (limb 0 first), top limb masked to the modulus' bit length, rejected
//     while >= modulus, and the limbs ARE the Montgomery representation                        [ark, from memory]
//   ark-poly `sample_element_outside_domain`: F::rand until the vanishing polynomial is non-zero.
// RNG stays outside the C ABI (r_a and the trapdoors are inputs of pm_prove_phase1 / pm_pk_generate).
#pragma once
#include <cstdint>
#include <cstring>

namespace pmhost {

inline uint32_t chacha_rotl(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }

// one ChaCha block: state = "expand 32-byte k" | key[8] | w12 w13 w14 w15 ; out = state + rounds(state)
inline void chacha_block(const uint32_t key[8], const uint32_t tail[4], int rounds, uint32_t out[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
    for (int i = 0; i < 8; ++i) s[4 + i] = key[i];
    for (int i = 0; i < 4; ++i) s[12 + i] = tail[i];
    uint32_t x[16];
    memcpy(x, s, sizeof(x));
    auto qr = [&](int a, int b, int c, int d) {
        x[a] += x[b]; x[d] = chacha_rotl(x[d] ^ x[a], 16);
        x[c] += x[d]; x[b] = chacha_rotl(x[b] ^ x[c], 12);
        x[a] += x[b]; x[d] = chacha_rotl(x[d] ^ x[a], 8);
        x[c] += x[d]; x[b] = chacha_rotl(x[b] ^ x[c], 7);
    };
    for (int r = 0; r < rounds; r += 2) {
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
    }
    for (int i = 0; i < 16; ++i) out[i] = x[i] + s[i];
}

struct StdRng {   // rand::rngs::StdRng (rand 0.8) == ChaCha12Rng
    uint32_t key[8];
    uint64_t counter = 0;
    uint32_t buf[16];
    int index = 16;   // words consumed of buf

    static StdRng from_seed(const uint8_t seed[32]) {
        StdRng r;
        for (int i = 0; i < 8; ++i) r.key[i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) | ((uint32_t)seed[4 * i + 2] << 16) | ((uint32_t)seed[4 * i + 3] << 24);
        return r;
    }
    static StdRng seed_from_u64(uint64_t state) {   // rand_core::SeedableRng::seed_from_u64 (PCG32 expansion)
        const uint64_t MUL = 6364136223846793005ull, INC = 11634580027462260723ull;
        uint8_t seed[32];
        for (int c = 0; c < 8; ++c) {
            state = state * MUL + INC;
            const uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27), rot = (uint32_t)(state >> 59);
            const uint32_t x = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
            seed[4 * c] = (uint8_t)x; seed[4 * c + 1] = (uint8_t)(x >> 8); seed[4 * c + 2] = (uint8_t)(x >> 16); seed[4 * c + 3] = (uint8_t)(x >> 24);
        }
        return from_seed(seed);
    }
    static StdRng test_rng() {   // ark_std::test_rng()
        const uint8_t seed[32] = {1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        return from_seed(seed);
    }
    uint32_t next_u32() {
        if (index >= 16) {
            const uint32_t tail[4] = {(uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
            chacha_block(key, tail, 12, buf);
            ++counter;
            index = 0;
        }
        return buf[index++];
    }
    uint64_t next_u64() {
        const uint64_t lo = next_u32();
        return lo | ((uint64_t)next_u32() << 32);
    }
};

// F::rand(rng) (ark-ff UniformRand for Fp) -> Montgomery limbs; FrLike = pm::Fp<P> with 32-bit limbs l[8], P::MOD, P::BITS
template <class P, class FrLike, class Rng>
inline FrLike fr_rand(Rng &rng) {
    for (;;) {
        FrLike v;
        for (int i = 0; i < 4; ++i) {
            const uint64_t w = rng.next_u64();
            v.l[2 * i] = (uint32_t)w;
            v.l[2 * i + 1] = (uint32_t)(w >> 32);
        }
        const int top_bits = P::BITS - 224;
        if (top_bits < 32) v.l[7] &= (1u << top_bits) - 1;
        bool lt = false;
        for (int i = 7; i >= 0; --i)
            if (v.l[i] != P::MOD[i]) { lt = v.l[i] < P::MOD[i]; break; }
        if (lt) return v;     // the limbs are taken as the Montgomery representation
    }
}

}  // namespace pmhost
