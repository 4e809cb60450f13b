// Reduced-radix base-field arithmetic for the MSM inner loop on gfx950.
//
// Measured on MI355X (profiles/r01_microbench_valu.txt): v_mad_u64_u32, v_add_co_u32 / v_addc_co_u32
// and v_lshl_add_u64 all issue at the same (half) rate, so on dense 32-bit limbs every carry costs as
// much as a multiply and the compiler's Montgomery multiplier spends 1328 instructions per Fq product
// (288 mads + 299 64-bit adds + 640 register moves).  Here an element is N limbs of W = 28 bits held in
// u32 registers; the 64-bit accumulator of v_mad_u64_u32 then absorbs 2N products without overflow, so a
// product is N*N mads for a*b plus N*N for the interleaved Montgomery reduction and NO carry chain:
// 463 instructions for BLS12-381 (N = 14; 392 mads, 14 + 27 digit / limb masks, 14 v_mul_lo, 27 carry shifts).
// Additions and subtractions are limb-wise v_add_u32 with lazy carries (14 instructions instead of ~200).
//
// Bounds discipline (R = 2^(W N) is >= 2^8 larger than p, so magnitudes up to ~45p are harmless):
//   T ("tight")  limbs < 2^28,            value < 2p      -- every mul/sqr output
//   W ("weak")   limbs < 2^28,            value < 16p     -- output of weak_norm (carry propagation only)
//   L ("loose")  limbs < 2^(28+e), e <= 3 (fits u32)      -- outputs of add / sub
//   mul(a, b) requires e_a + e_b <= 4  (14 * 2^(56+4) + 14 * 2^56 + carry < 2^64)
//   and value(a) * value(b) < 2^(W N) * p  (then the result is < 2p); the callers in this file
//   (the XYZZ mixed add) document their bounds inline.
// Values are in Montgomery form with radix 2^(W N) ("internal" form); bases are converted once when
// they are uploaded / generated (api.hip), results are converted back before they leave the kernel.
#pragma once
#include "ec.cuh"

namespace pm {

template <class RR>
struct F28 {
    static constexpr int N = RR::N;
    uint32_t l[N];
};

template <class RR>
PM_HD F28<RR> f28_zero() {
    F28<RR> r;
#pragma unroll
    for (int i = 0; i < RR::N; ++i) r.l[i] = 0;
    return r;
}
template <class RR>
PM_HD F28<RR> f28_one() {
    F28<RR> r;
#pragma unroll
    for (int i = 0; i < RR::N; ++i) r.l[i] = RR::ONE[i];
    return r;
}
template <class RR>
PM_HD bool f28_all_zero(const F28<RR> &a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < RR::N; ++i) o |= a.l[i];
    return o == 0;
}
// a is T (tight limbs, value < 2p): a == 0 (mod p)  <=>  a in {0, p}
template <class RR>
PM_HD bool f28_is_zero_mod_p(const F28<RR> &a) {
    uint32_t o = 0, q = 0;
#pragma unroll
    for (int i = 0; i < RR::N; ++i) {
        o |= a.l[i];
        q |= a.l[i] ^ RR::MOD[i];
    }
    return o == 0 || q == 0;
}

// ---- Montgomery products by product scanning (column by column): ONE 64-bit accumulator walks the 2N columns of a b + m p; the
// carry out of a column (acc >> W) is the addend of the next column's first v_mad_u64_u32, so a product needs no 64-bit addition
// at all.  (Rounds 1-4 ran the operand-scanning form -- N accumulators, one row of a b_i + m_i p per step -- which pays one
// v_lshl_add_u64 per row and one per output limb: 26 of its 489 instructions, 234 of the mixed addition's 4 711; same digits m_k,
// same column sums, same bounds, bit-identical outputs.  Same-box A/B: profiles/r04_product_scanning_ab.txt.)
// (PM_PIN64, field.cuh, keeps every chain in the order written here.)
template <class RR>
PM_HD F28<RR> f28_mul(const F28<RR> &a, const F28<RR> &b) {
    constexpr int N = RR::N;
    uint32_t m[N];
    F28<RR> r;
    uint64_t acc = 0;
    // p = 1 (mod 2^W) (BLS12-381's scalar field on 29-bit limbs): the compiler folds m[k] * 1 into a 64-bit ADD of the zero-extended
    // digit -- a v_mov for the high half plus a v_lshl_add_u64, N times per product (72 of the transforms' 1 144-instruction two-stage
    // body).  Multiplying by a 1 it cannot see keeps the step what it is for every other modulus: one v_mad_u64_u32 on the chain.
    uint32_t mod0 = RR::MOD[0];
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PM_NO_OPAQUE_MOD0)
    if (RR::MOD[0] == 1u) asm volatile("" : "+v"(mod0));
#endif
#pragma unroll
    for (int k = 0; k < 2 * N - 1; ++k) {
#pragma unroll
        for (int i = (k < N ? 0 : k - N + 1); i <= (k < N ? k : N - 1); ++i) { acc += (uint64_t)a.l[i] * b.l[k - i]; PM_PIN64(acc); }
#pragma unroll
        for (int i = (k < N ? 0 : k - N + 1); i <= (k < N ? k - 1 : N - 1); ++i) { acc += (uint64_t)m[i] * RR::MOD[k - i]; PM_PIN64(acc); }
        if (k < N) {
            m[k] = ((uint32_t)acc * RR::INV) & RR::MASK;
            { acc += (uint64_t)m[k] * mod0; PM_PIN64(acc); }
        } else {
            r.l[k - N] = (uint32_t)acc & RR::MASK;
        }
        acc >>= RR::W;
    }
    r.l[N - 1] = (uint32_t)acc;   // value < 2p < 2^(W N): the last column is the top limb itself
    return r;
}

// a*b + c*d with ONE Montgomery reduction (3 N^2 mads instead of 4 N^2).  Column bound:
// N 2^(56+ea+eb) + N 2^(56+ec+ed) + N 2^56 < 2^64; value bound (ab + cd) < 2^(W N) p.  Output T.
template <class RR>
PM_HD F28<RR> f28_mul2_add(const F28<RR> &a, const F28<RR> &b, const F28<RR> &c, const F28<RR> &d) {
    constexpr int N = RR::N;
    uint32_t m[N];
    F28<RR> r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * N - 1; ++k) {
#pragma unroll
        for (int i = (k < N ? 0 : k - N + 1); i <= (k < N ? k : N - 1); ++i) { acc += (uint64_t)a.l[i] * b.l[k - i]; PM_PIN64(acc); }
#pragma unroll
        for (int i = (k < N ? 0 : k - N + 1); i <= (k < N ? k : N - 1); ++i) { acc += (uint64_t)c.l[i] * d.l[k - i]; PM_PIN64(acc); }
#pragma unroll
        for (int i = (k < N ? 0 : k - N + 1); i <= (k < N ? k - 1 : N - 1); ++i) { acc += (uint64_t)m[i] * RR::MOD[k - i]; PM_PIN64(acc); }
        if (k < N) {
            m[k] = ((uint32_t)acc * RR::INV) & RR::MASK;
            { acc += (uint64_t)m[k] * RR::MOD[0]; PM_PIN64(acc); }
        } else {
            r.l[k - N] = (uint32_t)acc & RR::MASK;
        }
        acc >>= RR::W;
    }
    r.l[N - 1] = (uint32_t)acc;
    return r;
}

// Montgomery square: the N(N-1)/2 cross products are taken once against the doubled operand
// (limbs < 2^(29+e)), so a square costs N(N+1)/2 + N*N mads instead of 2 N*N.  Requires e_a <= 1.6
// (2 e_a + 1 <= 4.2 under the same 64-bit column bound as f28_mul).  Output T.
template <class RR>
PM_HD F28<RR> f28_sqr(const F28<RR> &a) {
    constexpr int N = RR::N;
    uint32_t a2[N];
#pragma unroll
    for (int j = 0; j < N; ++j) a2[j] = a.l[j] << 1;
    uint32_t m[N];
    F28<RR> r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * N - 1; ++k) {
        // column k of a^2: 2 a_i a_(k-i) for i < k - i, plus a_(k/2)^2 on the even columns
#pragma unroll
        for (int i = (k < N ? 0 : k - N + 1); 2 * i < k; ++i) { acc += (uint64_t)a2[i] * a.l[k - i]; PM_PIN64(acc); }
        if ((k & 1) == 0) { acc += (uint64_t)a.l[k / 2] * a.l[k / 2]; PM_PIN64(acc); }
#pragma unroll
        for (int i = (k < N ? 0 : k - N + 1); i <= (k < N ? k - 1 : N - 1); ++i) { acc += (uint64_t)m[i] * RR::MOD[k - i]; PM_PIN64(acc); }
        if (k < N) {
            m[k] = ((uint32_t)acc * RR::INV) & RR::MASK;
            { acc += (uint64_t)m[k] * RR::MOD[0]; PM_PIN64(acc); }
        } else {
            r.l[k - N] = (uint32_t)acc & RR::MASK;
        }
        acc >>= RR::W;
    }
    r.l[N - 1] = (uint32_t)acc;
    return r;
}

template <class RR>
PM_HD F28<RR> f28_add(const F28<RR> &a, const F28<RR> &b) {
    F28<RR> r;
#pragma unroll
    for (int i = 0; i < RR::N; ++i) r.l[i] = a.l[i] + b.l[i];
    return r;
}
// a + K - b with K = 4p (K4: b tight-limbed, value <= 2p), 8p (K8: b limbs < 2^29, value <= 4p)
// or 16p (K16: b tight-limbed, value < 14p); every limb of K dominates the matching limb of b.
template <class RR>
PM_HD F28<RR> f28_sub_k4(const F28<RR> &a, const F28<RR> &b) {
    F28<RR> r;
#pragma unroll
    for (int i = 0; i < RR::N; ++i) r.l[i] = a.l[i] + (RR::K4[i] - b.l[i]);
    return r;
}
template <class RR>
PM_HD F28<RR> f28_sub_k8(const F28<RR> &a, const F28<RR> &b) {
    F28<RR> r;
#pragma unroll
    for (int i = 0; i < RR::N; ++i) r.l[i] = a.l[i] + (RR::K8[i] - b.l[i]);
    return r;
}
template <class RR>
PM_HD F28<RR> f28_sub_k16(const F28<RR> &a, const F28<RR> &b) {
    F28<RR> r;
#pragma unroll
    for (int i = 0; i < RR::N; ++i) r.l[i] = a.l[i] + (RR::K16[i] - b.l[i]);
    return r;
}
// carry propagation only: limbs back below 2^28, value unchanged (top limb keeps the excess)
template <class RR>
PM_HD F28<RR> f28_weak_norm(const F28<RR> &a) {
    F28<RR> r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < RR::N - 1; ++i) {
        uint32_t v = a.l[i] + c;
        r.l[i] = v & RR::MASK;
        c = v >> RR::W;
    }
    r.l[RR::N - 1] = a.l[RR::N - 1] + c;
    return r;
}

// dense 32-bit limbs (value < p, already in internal Montgomery form) -> 28-bit limbs
template <class RR>
PM_HD F28<RR> f28_unpack(const uint32_t *d) {
    F28<RR> r;
#pragma unroll
    for (int i = 0; i < RR::N; ++i) {
        const int bit = RR::W * i, w = bit >> 5, s = bit & 31;
        uint64_t two = w < RR::N32 ? d[w] : 0u;
        if (w + 1 < RR::N32) two |= (uint64_t)d[w + 1] << 32;
        r.l[i] = (uint32_t)(two >> s) & RR::MASK;
    }
    return r;
}

// T value -> canonical (< p) -> dense 32-bit limbs
template <class RR>
PM_HD void f28_pack_reduced(const F28<RR> &a, uint32_t *d) {
    constexpr int N = RR::N;
    uint32_t t[N];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {  // t = a - p
        uint32_t v = a.l[i] - RR::MOD[i] - borrow;
        borrow = v >> 31;            // limbs < 2^28, so a negative difference sets bit 31
        t[i] = v & RR::MASK;
    }
    uint32_t c[N];
#pragma unroll
    for (int i = 0; i < N; ++i) c[i] = borrow ? a.l[i] : t[i];
#pragma unroll
    for (int w = 0; w < RR::N32; ++w) {
        const int bit = 32 * w, i = bit / RR::W, s = bit % RR::W;
        uint64_t v = (uint64_t)c[i] >> s;
        if (i + 1 < N) v |= (uint64_t)c[i + 1] << (RR::W - s);
        if (i + 2 < N && 2 * RR::W - s < 32) v |= (uint64_t)c[i + 2] << (2 * RR::W - s);
        d[w] = (uint32_t)v;
    }
}

// internal T value -> standard Montgomery form (radix 2^(32 N32)), canonical, dense
template <class RR>
PM_HD Fp<typename RR::Dense> f28_to_std(const F28<RR> &a) {
    F28<RR> k;
#pragma unroll
    for (int i = 0; i < RR::N; ++i) k.l[i] = RR::TO_STD[i];
    Fp<typename RR::Dense> out;
    f28_pack_reduced<RR>(f28_mul<RR>(a, k), out.l);
    return out;
}


// ---- scalar-field values that have grown lazily (NTT tiles, the division scan's Horner chains): back to the canonical range.
// tight limbs (after f28_weak_norm; the top limb keeps the excess), value < 2^(JMAX+1) p  ->  canonical (< p): conditional subtraction
// of 2^JMAX p, ..., 2p, p (limbs of p << j by constant shifts).  JMAX <= 5: the top limb of p << j stays below 2^28 for both scalar
// fields (255 / 254 bits on 9 limbs of 29).  After a product the value is < 2p and JMAX = 0 does; after <= 7 butterfly stages < 30p
// and JMAX = 4.
template <class RR, int JMAX = 5>
PM_HD F28<RR> f28_canonical_lazy(F28<RR> x) {
#pragma unroll
    for (int j = JMAX; j >= 0; --j) {
        uint32_t t[RR::N];
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < RR::N; ++i) {
            uint32_t m = (RR::MOD[i] << j) & RR::MASK;
            if (i > 0 && j > 0) m |= RR::MOD[i - 1] >> (RR::W - j);
            const uint32_t v = x.l[i] - m - borrow;
            borrow = v >> 31;
            t[i] = v & RR::MASK;
        }
        if (!borrow) {
#pragma unroll
            for (int i = 0; i < RR::N; ++i) x.l[i] = t[i];
        }
    }
    return x;
}
// The same map with ONE quotient step instead of JMAX + 1 conditional subtractions (round 6; the transforms' passes canonicalise every
// element once per pass, and the chain above was a quarter of a pass's instructions).  With X = the top limb, P = the top limb of p:
//   q' = floor(X M / 2^45), M = floor(2^45 / (P + 1))   satisfies   q - 1 <= q' <= q = floor(x / p):
//   x / p >= X / (P + 1) >= X M / 2^45 (lower limbs are tight: x < (X + 1) 2^(W (N - 1))), and
//   x / p - X M / 2^45 < (X + P + 1) / (P (P + 1)) + X / 2^45 < 2^-15 for x < 64p (P > 2^21, X < 2^29),
// so x - q' p lies in [0, 2p) and one conditional subtraction finishes.  q' p is formed limb by limb (q' < 64: one v_mad_u64_u32 and a
// carry per limb) and subtracted with the borrow chain of the step above.  Same inputs, same (unique) canonical outputs.
template <class RR, int JMAX = 5>
PM_HD F28<RR> f28_canonical_quot(F28<RR> x) {
    static_assert(JMAX <= 5, "value < 64p");
    constexpr int N = RR::N;
    constexpr uint64_t M = ((uint64_t)1 << 45) / ((uint64_t)RR::MOD[N - 1] + 1);
    static_assert(RR::MOD[N - 1] > (1u << 21) && M < ((uint64_t)1 << 32), "quotient estimate: top limb of p too short");
    const uint32_t q = (uint32_t)(((uint64_t)x.l[N - 1] * (uint32_t)M) >> 45);
    uint64_t carry = 0;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint64_t t = (uint64_t)q * RR::MOD[i] + carry;
        carry = t >> RR::W;
        const uint32_t lo = i + 1 < N ? ((uint32_t)t & RR::MASK) : (uint32_t)t;
        const uint32_t v = x.l[i] - lo - borrow;
        if (i + 1 < N) {
            borrow = v >> 31;
            x.l[i] = v & RR::MASK;
        } else {
            x.l[i] = v;            // >= 0: q' <= q
        }
    }
    return f28_canonical_lazy<RR, 0>(x);
}
// canonical W-bit limbs -> dense 32-bit limbs
template <class RR>
PM_HD void f28_pack_canonical(const F28<RR> &c, uint32_t *d) {
#pragma unroll
    for (int w = 0; w < RR::N32; ++w) {
        const int bit = 32 * w, i = bit / RR::W, s = bit % RR::W;
        uint64_t v = (uint64_t)c.l[i] >> s;
        if (i + 1 < RR::N) v |= (uint64_t)c.l[i + 1] << (RR::W - s);
        if (i + 2 < RR::N && 2 * RR::W - s < 32) v |= (uint64_t)c.l[i + 2] << (2 * RR::W - s);
        d[w] = (uint32_t)v;
    }
}

// ---------------------------------------------------------------------------- XYZZ on F28
template <class C>
struct XYZZ28 {
    typedef typename C::FqRR RR;
    F28<RR> X, Y, ZZ, ZZZ;  // X: W (< 14p), Y: W (< 6p), ZZ, ZZZ: T; identity <=> ZZ all-zero
};

// dense-side conversions between the standard and the internal Montgomery radix (one dense mul)
template <class C>
PM_HD Fp<typename C::FqP> fq_std_to_int(const Fp<typename C::FqP> &x) {
    Fp<typename C::FqP> c;
    for (int i = 0; i < C::FqP::N; ++i) c.l[i] = C::FqRR::STD2INT[i];
    return mul<typename C::FqP>(x, c);
}
template <class C>
PM_HD Fp<typename C::FqP> fq_int_to_std(const Fp<typename C::FqP> &x) {
    Fp<typename C::FqP> c;
    for (int i = 0; i < C::FqP::N; ++i) c.l[i] = C::FqRR::INT2STD[i];
    return mul<typename C::FqP>(x, c);
}

// acc += (x2, +-y2): madd-2008-s (8M + 2S) with lazy reductions.  `q` holds INTERNAL-form dense
// coordinates (pre-converted bases).  Returns false when the pair hits the exceptional case
// P == 0 (same x: doubling or cancellation), which the caller resolves on the dense path.
template <class C>
PM_HD bool xyzz28_madd_limbs(XYZZ28<C> &acc, const F28<typename C::FqRR> &x2, F28<typename C::FqRR> y2, bool negate);

template <class C>
PM_HD bool xyzz28_madd(XYZZ28<C> &acc, const Affine<C> &q, bool negate) {
    typedef typename C::FqRR RR;
    return xyzz28_madd_limbs<C>(acc, f28_unpack<RR>(q.x.l), f28_unpack<RR>(q.y.l), negate);
}

// the same with the point already on 28-bit limbs (x2, y2 < p, tight): what the window tables store
template <class C>
PM_HD bool xyzz28_madd_limbs(XYZZ28<C> &acc, const F28<typename C::FqRR> &x2, F28<typename C::FqRR> y2, bool negate) {
    typedef typename C::FqRR RR;
    typedef F28<RR> F;
    // 4p - y as a MULTIPLICAND needs no carry propagation: limbs < 2^29 (e = 1) against a tight ZZZ stay inside the column bound
    // (e_a + e_b <= 4), value < 4p.  (Round 4: the 42-instruction weak normalisation moved to the rare first-point branch.)
    if (negate) y2 = f28_sub_k4<RR>(f28_zero<RR>(), y2);
    if (f28_all_zero<RR>(acc.ZZ)) {
        acc.X = x2;
        acc.Y = f28_weak_norm<RR>(y2);                     // the accumulator's Y is W: tight limbs (K16 must dominate them)
        acc.ZZ = f28_one<RR>();
        acc.ZZZ = f28_one<RR>();
        return true;
    }
    const F U2 = f28_mul<RR>(x2, acc.ZZ);                  // T
    const F S2 = f28_mul<RR>(y2, acc.ZZZ);                 // y2 (limbs < 2^29, < 4p) x T -> T
    const F P = f28_sub_k16<RR>(U2, acc.X);                // L e<=2, < 18p
    const F R = f28_sub_k16<RR>(S2, acc.Y);                // L e<=2, < 18p
    const F PP = f28_sqr<RR>(P);                           // e 2+2, 18p*18p < 2^8.4 p^2 -> T
    // P == 0 (mod p) <=> PP in {0, p}: the low limb decides in all but 2^-27 of the cases, the 42-instruction compare is behind it
    if ((PP.l[0] == 0u || PP.l[0] == RR::MOD[0]) && f28_is_zero_mod_p<RR>(PP)) return false;   // exceptional
    const F PPP = f28_mul<RR>(P, PP);                      // T
    const F Q = f28_mul<RR>(acc.X, PP);                    // W(<14p) x T -> T
    const F RR2 = f28_sqr<RR>(R);                          // T
    // X3 = R^2 - PPP - 2Q
    F X3 = f28_sub_k4<RR>(RR2, PPP);                       // < 6p,  limbs < 2^28 + 2^29
    X3 = f28_sub_k8<RR>(X3, f28_add<RR>(Q, Q));            // < 14p, limbs < 2^31
    X3 = f28_weak_norm<RR>(X3);                            // W, < 14p
    // Y3 = R (Q - X3) - Y1 PPP
    const F QX = f28_sub_k16<RR>(Q, X3);                   // L e<=2, < 18p
    // Y3 = R QX - Y1 PPP = R QX + Y1 (4p - PPP): both products under one reduction
    // (columns 14 (2^59.2 + 2^57.6 + 2^56) < 2^63.5; (18p)^2 + 6p 6p < 2^(392) p 0.15 -> result < 2p)
    const F nPPP = f28_sub_k4<RR>(f28_zero<RR>(), PPP);    // 4p - PPP, limbs < 2^29
    acc.Y = f28_mul2_add<RR>(R, QX, acc.Y, nPPP);          // T (so also W < 6p)
    acc.X = X3;
    acc.ZZ = f28_mul<RR>(acc.ZZ, PP);                      // T
    acc.ZZZ = f28_mul<RR>(acc.ZZZ, PPP);                   // T
    return true;
}

// XYZZ28 (internal) -> dense XYZZ in STANDARD Montgomery form (what the reduce kernels consume)
template <class C>
PM_HD_COLD XYZZ<C> xyzz28_to_std(XYZZ28<C> a) {
    typedef typename C::FqRR RR;
    XYZZ<C> r;
    if (f28_all_zero<RR>(a.ZZ)) return XYZZ<C>::identity();
    r.X = f28_to_std<RR>(a.X);
    r.Y = f28_to_std<RR>(a.Y);
    r.ZZ = f28_to_std<RR>(a.ZZ);
    r.ZZZ = f28_to_std<RR>(a.ZZZ);
    return r;
}
// dense STANDARD XYZZ -> internal XYZZ28 (after resolving an exceptional case on the dense path)
template <class C>
PM_HD_COLD XYZZ28<C> xyzz28_from_std(const XYZZ<C> &a) {
    typedef typename C::FqRR RR;
    XYZZ28<C> r;
    if (a.is_identity()) {
        r.X = r.Y = r.ZZ = r.ZZZ = f28_zero<RR>();
        return r;
    }
    r.X = f28_unpack<RR>(fq_std_to_int<C>(a.X).l);
    r.Y = f28_unpack<RR>(fq_std_to_int<C>(a.Y).l);
    r.ZZ = f28_unpack<RR>(fq_std_to_int<C>(a.ZZ).l);
    r.ZZZ = f28_unpack<RR>(fq_std_to_int<C>(a.ZZZ).l);
    return r;
}

// a^(p-2) for a T input (Fermat); cold: table construction only
template <class RR>
PM_HD_COLD F28<RR> f28_inverse(F28<RR> a) {
    constexpr int N = RR::N;
    uint32_t e[N];
    uint32_t borrow = 2;
    for (int i = 0; i < N; ++i) {
        uint32_t v = RR::MOD[i] - borrow;
        borrow = v >> 31;
        e[i] = v & RR::MASK;
    }
    F28<RR> acc = f28_one<RR>();
    bool started = false;
    for (int i = N - 1; i >= 0; --i)
        for (int b = RR::W - 1; b >= 0; --b) {
            if (started) acc = f28_sqr<RR>(acc);
            if ((e[i] >> b) & 1) {
                acc = started ? f28_mul<RR>(acc, a) : a;
                started = true;
            }
        }
    return acc;
}

// ---- internal-form points in memory ---------------------------------------------------------------
// Task partials and every intermediate of the bucket reduction stay in the INTERNAL Montgomery radix:
// an XYZZ<C> record whose four coordinates are canonical (< p) dense words of the internal form.
template <class C>
PM_HD XYZZ28<C> xyzz28_load(const XYZZ<C> &m) {
    typedef typename C::FqRR RR;
    XYZZ28<C> r;
    r.X = f28_unpack<RR>(m.X.l);
    r.Y = f28_unpack<RR>(m.Y.l);
    r.ZZ = f28_unpack<RR>(m.ZZ.l);
    r.ZZZ = f28_unpack<RR>(m.ZZZ.l);
    return r;
}
template <class C>
PM_HD XYZZ<C> xyzz28_store(const XYZZ28<C> &a) {
    typedef typename C::FqRR RR;
    XYZZ<C> r;
    if (f28_all_zero<RR>(a.ZZ)) return XYZZ<C>::identity();
    const F28<RR> one = f28_one<RR>();
    f28_pack_reduced<RR>(f28_mul<RR>(a.X, one), r.X.l);   // X is W (< 14p): one Montgomery product by R brings it below 2p
    f28_pack_reduced<RR>(f28_mul<RR>(a.Y, one), r.Y.l);
    f28_pack_reduced<RR>(a.ZZ, r.ZZ.l);                   // T already
    f28_pack_reduced<RR>(a.ZZZ, r.ZZZ.l);
    return r;
}
// internal-form record <-> standard-form record (dense conversions; cold / host)
template <class C>
PM_HD_COLD XYZZ<C> xyzz_internal_to_std(XYZZ<C> a) {
    if (a.is_identity()) return a;
    a.X = fq_int_to_std<C>(a.X); a.Y = fq_int_to_std<C>(a.Y); a.ZZ = fq_int_to_std<C>(a.ZZ); a.ZZZ = fq_int_to_std<C>(a.ZZZ);
    return a;
}
template <class C>
PM_HD_COLD XYZZ<C> xyzz_std_to_internal(XYZZ<C> a) {
    if (a.is_identity()) return a;
    a.X = fq_std_to_int<C>(a.X); a.Y = fq_std_to_int<C>(a.Y); a.ZZ = fq_std_to_int<C>(a.ZZ); a.ZZZ = fq_std_to_int<C>(a.ZZZ);
    return a;
}

// a + b on reduced-radix registers: add-2008-s (12M + 2S), same invariants as xyzz28_madd
// (X: W < 14p, Y: W < 6p, ZZ, ZZZ: T).  Returns false on the exceptional case (same x).
template <class C>
PM_HD bool xyzz28_add(XYZZ28<C> &a, const XYZZ28<C> &b) {
    typedef typename C::FqRR RR;
    typedef F28<RR> F;
    if (f28_all_zero<RR>(b.ZZ)) return true;
    if (f28_all_zero<RR>(a.ZZ)) { a = b; return true; }
    const F U1 = f28_mul<RR>(a.X, b.ZZ), U2 = f28_mul<RR>(b.X, a.ZZ);      // W x T -> T
    const F S1 = f28_mul<RR>(a.Y, b.ZZZ), S2 = f28_mul<RR>(b.Y, a.ZZZ);    // T
    const F P = f28_sub_k4<RR>(U2, U1), R = f28_sub_k4<RR>(S2, S1);        // L e<=1.6, < 6p
    const F PP = f28_sqr<RR>(P);
    if (f28_is_zero_mod_p<RR>(PP)) return false;
    const F PPP = f28_mul<RR>(P, PP), Q = f28_mul<RR>(U1, PP), RR2 = f28_sqr<RR>(R);
    F X3 = f28_sub_k4<RR>(RR2, PPP);
    X3 = f28_weak_norm<RR>(f28_sub_k8<RR>(X3, f28_add<RR>(Q, Q)));         // W, < 14p
    const F QX = f28_sub_k16<RR>(Q, X3);                                   // L e<=1.6, < 18p
    const F Y3 = f28_mul2_add<RR>(R, QX, S1, f28_sub_k4<RR>(f28_zero<RR>(), PPP));             // R QX + S1 (4p - PPP): T
    a.X = X3;
    a.Y = Y3;
    a.ZZ = f28_mul<RR>(f28_mul<RR>(a.ZZ, b.ZZ), PP);
    a.ZZZ = f28_mul<RR>(f28_mul<RR>(a.ZZZ, b.ZZZ), PPP);
    return true;
}
// 2a on reduced-radix registers: dbl-2008-s-1 (6M + 3S... here 5M + 4S incl. the fused last product).
// No point of order 2 exists on either curve (odd group order), so Y != 0 for every finite curve point.
template <class C>
PM_HD void xyzz28_dbl(XYZZ28<C> &a) {
    typedef typename C::FqRR RR;
    typedef F28<RR> F;
    if (f28_all_zero<RR>(a.ZZ)) return;
    const F U = f28_add<RR>(a.Y, a.Y);                         // limbs < 2^29, < 12p
    const F V = f28_sqr<RR>(U), Wv = f28_mul<RR>(U, V), S = f28_mul<RR>(a.X, V);   // T
    const F X2 = f28_sqr<RR>(a.X);                             // T
    const F M = f28_add<RR>(f28_add<RR>(X2, X2), X2);          // limbs < 3 2^28, < 6p
    F X3 = f28_sub_k8<RR>(f28_sqr<RR>(M), f28_add<RR>(S, S));  // < 10p
    X3 = f28_weak_norm<RR>(X3);                                // W
    const F SX = f28_sub_k16<RR>(S, X3);                       // L e<=1.6, < 18p
    const F Y3 = f28_mul2_add<RR>(M, SX, a.Y, f28_sub_k4<RR>(f28_zero<RR>(), Wv));   // M (S - X3) + Y1 (4p - W): T
    a.ZZ = f28_mul<RR>(V, a.ZZ);
    a.ZZZ = f28_mul<RR>(Wv, a.ZZZ);
    a.X = X3;
    a.Y = Y3;
}

template <class C>
PM_HD_COLD XYZZ28<C> xyzz28_add_exceptional(XYZZ28<C> a, XYZZ28<C> b);

// mem += a, where `mem` lives in LDS (or any memory) and is streamed coordinate by coordinate, `a` stays
// intact in registers.  Same formulas and invariants as xyzz28_add; only ~one operand plus the temporaries are
// live at a time, which is what lets the bucket-reduction kernels run two waves per SIMD.
template <class C>
PM_HD bool xyzz28_add_into(XYZZ28<C> *mem, const XYZZ28<C> &a) {
    typedef typename C::FqRR RR;
    typedef F28<RR> F;
    if (f28_all_zero<RR>(a.ZZ)) return true;
    if (f28_all_zero<RR>(mem->ZZ)) { *mem = a; return true; }
    const F U1 = f28_mul<RR>(a.X, mem->ZZ), U2 = f28_mul<RR>(mem->X, a.ZZ);
    const F S1 = f28_mul<RR>(a.Y, mem->ZZZ), S2 = f28_mul<RR>(mem->Y, a.ZZZ);
    const F P = f28_sub_k4<RR>(U2, U1), R = f28_sub_k4<RR>(S2, S1);
    const F PP = f28_sqr<RR>(P);
    if (f28_is_zero_mod_p<RR>(PP)) return false;
    const F PPP = f28_mul<RR>(P, PP), Q = f28_mul<RR>(U1, PP), RR2 = f28_sqr<RR>(R);
    F X3 = f28_sub_k4<RR>(RR2, PPP);
    X3 = f28_weak_norm<RR>(f28_sub_k8<RR>(X3, f28_add<RR>(Q, Q)));
    const F QX = f28_sub_k16<RR>(Q, X3);
    const F Y3 = f28_mul2_add<RR>(R, QX, S1, f28_sub_k4<RR>(f28_zero<RR>(), PPP));
    mem->X = X3;
    mem->Y = Y3;
    mem->ZZ = f28_mul<RR>(f28_mul<RR>(a.ZZ, mem->ZZ), PP);
    mem->ZZZ = f28_mul<RR>(f28_mul<RR>(a.ZZZ, mem->ZZZ), PPP);
    return true;
}
template <class C>
PM_HD void xyzz28_add_into_full(XYZZ28<C> *mem, const XYZZ28<C> &a) {
    if (!xyzz28_add_into<C>(mem, a)) *mem = xyzz28_add_exceptional<C>(*mem, a);
}

// exceptional case of xyzz28_add, complete dense formulas (cold)
template <class C>
PM_HD_COLD XYZZ28<C> xyzz28_add_exceptional(XYZZ28<C> a, XYZZ28<C> b) {
    XYZZ<C> da = xyzz_internal_to_std<C>(xyzz28_store<C>(a)), db = xyzz_internal_to_std<C>(xyzz28_store<C>(b));
    return xyzz28_load<C>(xyzz_std_to_internal<C>(xyzz_add<C>(da, db)));
}
template <class C>
PM_HD void xyzz28_add_full(XYZZ28<C> &a, const XYZZ28<C> &b) {
    if (!xyzz28_add<C>(a, b)) a = xyzz28_add_exceptional<C>(a, b);
}

// ---------------------------------------------------------------------------- table points
// A window-table entry (setup.hip: tables_build): affine x, y < p in the INTERNAL radix, already on 28-bit
// limbs and padded to one 128-byte line.  The bucket accumulation gathers these at random: a 96-byte dense
// point straddles 1.5 lines on average and costs 97 unpack instructions per mixed add; this record is one
// aligned line and no unpacking (measured on the 21 M-pair MSM: -5 % of k_accumulate for the alignment alone).
// The point at infinity is all-zero.
template <class C>
struct alignas(128) TablePoint {
    uint32_t x[C::FqRR::N], y[C::FqRR::N];
};

// T (< 2p, tight limbs) -> canonical (< p, tight limbs)
template <class RR>
PM_HD F28<RR> f28_canonical(const F28<RR> &a) {
    constexpr int N = RR::N;
    F28<RR> t, r;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {   // t = a - p
        uint32_t v = a.l[i] - RR::MOD[i] - borrow;
        borrow = v >> 31;
        t.l[i] = v & RR::MASK;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = borrow ? a.l[i] : t.l[i];
    return r;
}

template <class C>
PM_HD TablePoint<C> table_point_from_affine(const Affine<C> &p_internal) {
    typedef typename C::FqRR RR;
    TablePoint<C> t;
    const F28<RR> x = f28_unpack<RR>(p_internal.x.l), y = f28_unpack<RR>(p_internal.y.l);
#pragma unroll
    for (int i = 0; i < RR::N; ++i) { t.x[i] = x.l[i]; t.y[i] = y.l[i]; }
    return t;
}

template <class C>
PM_HD Affine<C> table_point_to_affine(const TablePoint<C> &t) {
    typedef typename C::FqRR RR;
    F28<RR> x, y;
#pragma unroll
    for (int i = 0; i < RR::N; ++i) { x.l[i] = t.x[i]; y.l[i] = t.y[i]; }
    Affine<C> a;
    f28_pack_reduced<RR>(x, a.x.l);
    f28_pack_reduced<RR>(y, a.y.l);
    return a;
}

// The exceptional case of xyzz28_madd (acc == +-point), resolved with the complete dense formulas.
template <class C>
PM_HD_COLD XYZZ28<C> xyzz28_madd_exceptional(XYZZ28<C> acc, Affine<C> q_internal, bool negate) {
    // by value on purpose: a by-reference accumulator would have to live in scratch memory in the hot loop
    XYZZ<C> d = xyzz28_to_std<C>(acc);
    Affine<C> ps{fq_int_to_std<C>(q_internal.x), fq_int_to_std<C>(q_internal.y)};
    xyzz_madd<C>(d, ps, negate);
    return xyzz28_from_std<C>(d);
}

}  // namespace pm
