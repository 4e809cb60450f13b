// CPU-only known-answer tests of the C++ host mirror's hashes and field glue (no GPU, no library link).
#include <cstdio>
#include "../../polymath_amd/host/hashes.hpp"
#include "../../polymath_amd/csrc/field.cuh"
#include "../../polymath_amd/host/pairing.hpp"
#include "../../polymath_amd/host/rng.hpp"
using namespace pmhost;
int main() {
    int fails = 0;
    auto hex = [](const Bytes &b) { static const char *d = "0123456789abcdef"; std::string s; for (uint8_t v : b) { s.push_back(d[v >> 4]); s.push_back(d[v & 15]); } return s; };
    if (hex(keccak256(Bytes())) != "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470") { fails++; printf("keccak256 KAT\n"); }
    if (hex(blake3(Bytes())) != "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262") { fails++; printf("blake3 KAT\n"); }
    Bytes pat(2049);
    for (size_t i = 0; i < pat.size(); ++i) pat[i] = (uint8_t)(i % 251);
    if (hex(blake3(pat)).substr(0, 32) != "5f4d72f40d7a5f82b15ca2b2e44b1de3") { fails++; printf("blake3 2049 KAT\n"); }
    pat.resize(1025);
    if (hex(blake3(pat)).substr(0, 32) != "d00278ae47eb27b34faecf67b4fe263f") { fails++; printf("blake3 1025 KAT\n"); }
    MerlinTranscript m("test protocol");   // merlin 3.0.0 transcript.rs equivalence_simple
    m.append_message("some label", (const uint8_t *)"some data", 9);
    Bytes ch(32);
    m.challenge_bytes("challenge", ch.data(), 32);
    if (hex(ch) != "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615") { fails++; printf("merlin KAT %s\n", hex(ch).c_str()); }
    {   // pairing: generators on their curves, bilinearity e(aP, bQ) e(-(ab)P, Q) == 1, and a negative
        typedef Bls12Pairing B;
        B::G2 g2 = B::g2_generator();
        if (!B::g2_on_curve(g2)) { fails++; printf("G2 generator off curve\n"); }
        pm::Affine<pm::BlsCurve> g1;
        for (int i = 0; i < 12; ++i) { g1.x.l[i] = pm::BlsCurve::GX_MONT[i]; g1.y.l[i] = pm::BlsCurve::GY_MONT[i]; }
        auto g1mul = [&](uint32_t k) {
            pm::XYZZ<pm::BlsCurve> acc = pm::XYZZ<pm::BlsCurve>::identity();
            for (int b = 31; b >= 0; --b) { acc = pm::xyzz_dbl<pm::BlsCurve>(acc); if ((k >> b) & 1) pm::xyzz_madd<pm::BlsCurve>(acc, g1, false); }
            return pm::xyzz_to_affine<pm::BlsCurve>(acc);
        };
        uint32_t a = 0x1234567u, b = 0x89abcdu, ab_lo, ab_hi;
        uint64_t ab = (uint64_t)a * b;
        uint32_t kb[2] = {b, 0}, kab[2] = {(uint32_t)ab, (uint32_t)(ab >> 32)};
        (void)ab_lo; (void)ab_hi;
        pm::Affine<pm::BlsCurve> aP = g1mul(a);
        B::G2 bQ = B::g2_mul(g2, kb, 1);
        // -(ab) P via G2 side instead: e(aP, bQ) * e(-P, (ab) Q) == 1
        pm::Affine<pm::BlsCurve> negP = g1;
        negP.y = pm::neg<pm::BlsFqP>(negP.y);
        B::G2 abQ = B::g2_mul(g2, kab, 2);
        if (!B::product_is_one({{aP, false, bQ}, {negP, false, abQ}})) { fails++; printf("pairing bilinearity\n"); }
        kab[0] += 1;
        B::G2 wrong = B::g2_mul(g2, kab, 2);
        if (B::product_is_one({{aP, false, bQ}, {negP, false, wrong}})) { fails++; printf("pairing false accept\n"); }
    }
    {   // the same on BN254 (optimal ate: 6x + 2 loop and the two Frobenius steps)
        typedef Bn254Pairing B;
        typedef pm::BnCurve CC;
        B::G2 g2 = B::g2_generator();
        if (!B::g2_on_curve(g2)) { fails++; printf("BN254 G2 generator off the twist\n"); }
        uint32_t rm1[8];
        for (int i = 0; i < 8; ++i) rm1[i] = pm::BnFrP::MOD[i];
        rm1[0] -= 1;                                                     // r - 1 (r is odd)
        B::G2 m = B::g2_mul(g2, rm1, 8), ng = B::g2_neg(g2);
        if (m.inf || !m.x.eq(ng.x) || !m.y.eq(ng.y)) { fails++; printf("BN254 G2 generator order\n"); }
        pm::Affine<CC> g1;
        for (int i = 0; i < 8; ++i) { g1.x.l[i] = CC::GX_MONT[i]; g1.y.l[i] = CC::GY_MONT[i]; }
        auto g1mul = [&](uint32_t k) {
            pm::XYZZ<CC> acc = pm::XYZZ<CC>::identity();
            for (int b = 31; b >= 0; --b) { acc = pm::xyzz_dbl<CC>(acc); if ((k >> b) & 1) pm::xyzz_madd<CC>(acc, g1, false); }
            return pm::xyzz_to_affine<CC>(acc);
        };
        uint32_t a = 0x1234567u, b = 0x89abcdu;
        uint64_t ab = (uint64_t)a * b;
        uint32_t kb[2] = {b, 0}, kab[2] = {(uint32_t)ab, (uint32_t)(ab >> 32)};
        pm::Affine<CC> aP = g1mul(a), negP = g1;
        negP.y = pm::neg<pm::BnFqP>(negP.y);
        B::G2 bQ = B::g2_mul(g2, kb, 1), abQ = B::g2_mul(g2, kab, 2);
        if (!B::product_is_one({{aP, false, bQ}, {negP, false, abQ}})) { fails++; printf("BN254 pairing bilinearity\n"); }
        kab[0] += 1;
        B::G2 wrong = B::g2_mul(g2, kab, 2);
        if (B::product_is_one({{aP, false, bQ}, {negP, false, wrong}})) { fails++; printf("BN254 pairing false accept\n"); }
        if (B::product_is_one({{g1, false, g2}})) { fails++; printf("BN254 pairing degenerate\n"); }
    }
    {   // random sources (rng.hpp): the ChaCha block function against RFC 7539 section 2.3.2 (20 rounds), then the streams the
        // Python twin (polymath_amd/rng.py) must reproduce word for word (tests/test_host_mirror.py)
        uint32_t key[8], out[16];
        for (int i = 0; i < 8; ++i) key[i] = (uint32_t)(4 * i) | ((uint32_t)(4 * i + 1) << 8) | ((uint32_t)(4 * i + 2) << 16) | ((uint32_t)(4 * i + 3) << 24);
        const uint32_t tail[4] = {1u, 0x09000000u, 0x4a000000u, 0u};
        chacha_block(key, tail, 20, out);
        Bytes ob(64);
        for (int i = 0; i < 16; ++i) for (int b = 0; b < 4; ++b) ob[4 * i + b] = (uint8_t)(out[i] >> (8 * b));
        if (hex(ob) != "10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4ed2826446079faa0914c2d705d98b02a2b5129cd1de164eb9cbd083e8a2503c4e") { fails++; printf("chacha20 RFC 7539 KAT\n"); }
        // ... and the TWELVE-round function StdRng actually runs, against a published known answer: draft-strombergson-chacha-test-
        // vectors-01, TC1 (all-zero 256-bit key and IV), 12 rounds, keystream block 0 -- and the 8-round block of the same table
        {
            uint32_t zk[8] = {0}, zt[4] = {0};
            chacha_block(zk, zt, 12, out);
            for (int i = 0; i < 16; ++i) for (int b = 0; b < 4; ++b) ob[4 * i + b] = (uint8_t)(out[i] >> (8 * b));
            if (hex(ob) != "9bf49a6a0755f953811fce125f2683d50429c3bb49e074147e0089a52eae155f0564f879d27ae3c02ce82834acfa8c793a629f2ca0de6919610be82f411326be") { fails++; printf("chacha12 KAT\n"); }
            chacha_block(zk, zt, 8, out);
            for (int i = 0; i < 16; ++i) for (int b = 0; b < 4; ++b) ob[4 * i + b] = (uint8_t)(out[i] >> (8 * b));
            if (hex(ob) != "3e00ef2f895f40d67f5bb8e81f09a5a12c840ec3ce9a7f3b181be188ef711a1e984ce172b9216f419f445367456d5619314a42a3da86b001387bfdb80e0cfe42") { fails++; printf("chacha8 KAT\n"); }
        }
        StdRng t = StdRng::test_rng();
        const unsigned long long t0 = t.next_u64();
        StdRng r = StdRng::seed_from_u64(t0);
        printf("rng test_rng_first=%016llx seeded:", t0);
        for (int i = 0; i < 20; ++i) printf(" %016llx", (unsigned long long)r.next_u64());
        pm::Fp<pm::BlsFrP> f = fr_rand<pm::BlsFrP, pm::Fp<pm::BlsFrP>>(r);
        pm::Fp<pm::BnFrP> g = fr_rand<pm::BnFrP, pm::Fp<pm::BnFrP>>(r);
        printf(" fr_bls=");
        for (int i = 7; i >= 0; --i) printf("%08x", f.l[i]);
        printf(" fr_bn=");
        for (int i = 7; i >= 0; --i) printf("%08x", g.l[i]);
        printf("\n");
    }
    printf("host selftest: %d failures\n", fails);
    return fails ? 1 : 0;
}
