// Prime-field arithmetic for CDNA4 (gfx950): Montgomery form on 32-bit limbs.
//
// gfx950 has no 64x64 multiplier; the widest integer multiply-add is v_mad_u64_u32
// (32x32 + 64 -> 64, carry-out).  A field element is therefore N x u32 limbs in VGPRs
// (N = 8 for the 255/254-bit scalar fields, 12 for BLS12-381's 381-bit base field), which is
// byte-identical to arkworks' N/2 x u64 little-endian Montgomery limbs in memory
// (ark-ff MontBackend) -- no conversion at the FFI boundary.
//
// The same templates compile for the host (plain C++), where the library needs a handful of
// field operations for glue (pm_g1_sum, setup scalars).
#pragma once
#include <stdint.h>

#include "constants.cuh"

#if defined(__HIPCC__)
#define PM_HD __host__ __device__ __forceinline__
// cold paths (inversions, full additions, doublings): real function calls keep the hot kernels'
// code size and the build time down
#define PM_HD_COLD __host__ __device__ __noinline__
#else
#define PM_HD inline
#define PM_HD_COLD inline
#endif

namespace pm {

template <class P>
struct Fp {
    static constexpr int N = P::N;
    uint32_t l[N];

    PM_HD static Fp zero() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.l[i] = 0;
        return r;
    }
    PM_HD static Fp one() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.l[i] = P::ONE[i];
        return r;
    }
    PM_HD static Fp r2() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.l[i] = P::R2[i];
        return r;
    }
    PM_HD bool is_zero() const {
        uint32_t a = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) a |= l[i];
        return a == 0;
    }
    PM_HD bool eq(const Fp &o) const {
        uint32_t a = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) a |= l[i] ^ o.l[i];
        return a == 0;
    }
};

// r = a - MOD if a >= MOD else a   (a < 2*MOD, optional incoming carry bit `hi`)
template <class P>
PM_HD void reduce_once(uint32_t *r, const uint32_t *a, uint32_t hi) {
    constexpr int N = P::N;
    uint32_t t[N];
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        uint64_t d = (uint64_t)a[i] - P::MOD[i] - borrow;
        t[i] = (uint32_t)d;
        borrow = (d >> 32) & 1;
    }
    // a >= MOD  <=>  no final borrow, or the incoming carry covers it
    bool ge = (borrow == 0) || (hi != 0);
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = ge ? t[i] : a[i];
}

template <class P>
PM_HD Fp<P> add(const Fp<P> &a, const Fp<P> &b) {
    constexpr int N = P::N;
    uint32_t s[N];
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        c += (uint64_t)a.l[i] + b.l[i];
        s[i] = (uint32_t)c;
        c >>= 32;
    }
    Fp<P> r;
    reduce_once<P>(r.l, s, (uint32_t)c);
    return r;
}

template <class P>
PM_HD Fp<P> sub(const Fp<P> &a, const Fp<P> &b) {
    constexpr int N = P::N;
    uint32_t d[N];
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        uint64_t t = (uint64_t)a.l[i] - b.l[i] - borrow;
        d[i] = (uint32_t)t;
        borrow = (t >> 32) & 1;
    }
    uint32_t mask = (uint32_t)0 - (uint32_t)borrow;
    Fp<P> r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        c += (uint64_t)d[i] + (P::MOD[i] & mask);
        r.l[i] = (uint32_t)c;
        c >>= 32;
    }
    return r;
}

template <class P>
PM_HD Fp<P> neg(const Fp<P> &a) {
    return a.is_zero() ? a : sub<P>(Fp<P>::zero(), a);
}

template <class P>
PM_HD Fp<P> dbl(const Fp<P> &a) {
    return add<P>(a, a);
}

// Montgomery multiplication, CIOS over 32-bit limbs: every inner step is one
// v_mad_u64_u32 (a_j * b_i + t_j) plus the carry add.  Result fully reduced (< MOD).
// Reference implementation: hipcc lowers it to ~600 instructions for 8 limbs, 340 of them register moves
// around the carries; the product the kernels use is mul() below.  Kept for cross-checks (tests/native).
template <class P>
PM_HD Fp<P> mul_cios(const Fp<P> &a, const Fp<P> &b) {
    constexpr int N = P::N;
    uint32_t t[N + 2];
#pragma unroll
    for (int i = 0; i < N + 2; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        uint64_t c = 0;
        const uint32_t bi = b.l[i];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            c += (uint64_t)a.l[j] * bi + t[j];
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        c += t[N];
        t[N] = (uint32_t)c;
        t[N + 1] = (uint32_t)(c >> 32);
        const uint32_t m = t[0] * P::INV;
        c = (uint64_t)m * P::MOD[0] + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < N; ++j) {
            c += (uint64_t)m * P::MOD[j] + t[j];
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t[N];
        t[N - 1] = (uint32_t)c;
        t[N] = t[N + 1] + (uint32_t)(c >> 32);
    }
    Fp<P> r;
    reduce_once<P>(r.l, t, t[N]);
    return r;
}

// The same Montgomery product for HOST code on 64-bit limbs (unsigned __int128): the 32-bit CIOS above costs a CPU ~250 ns for the
// 12-limb base field, this one a sixth of that -- and a proof's three host-side point normalisations (one Fermat inversion each)
// sit between its phases with the GPU idle.  Same radix 2^(32 N), same canonical result, word for word.
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__SIZEOF_INT128__)
template <class P>
inline Fp<P> mul_cios64(const Fp<P> &a, const Fp<P> &b) {
    static_assert(P::N % 2 == 0, "whole 64-bit limbs");
    constexpr int M = P::N / 2;
    typedef unsigned __int128 u128;
    uint64_t A[M], B[M], Q[M], t[M + 2];
    for (int i = 0; i < M; ++i) {
        A[i] = a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
        B[i] = b.l[2 * i] | ((uint64_t)b.l[2 * i + 1] << 32);
        Q[i] = P::MOD[2 * i] | ((uint64_t)P::MOD[2 * i + 1] << 32);
    }
    // -p^-1 mod 2^64 from the 32-bit constant: one Newton step on p^-1 (x <- x (2 - p x)) doubles the valid bits
    const uint64_t pinv32 = (uint64_t)(uint32_t)(0u - P::INV);
    const uint64_t inv64 = 0 - pinv32 * (2 - Q[0] * pinv32);
    for (int i = 0; i < M + 2; ++i) t[i] = 0;
    for (int i = 0; i < M; ++i) {
        u128 c = 0;
        for (int j = 0; j < M; ++j) {
            c += (u128)A[j] * B[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[M];
        t[M] = (uint64_t)c;
        t[M + 1] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * inv64;
        c = (u128)m * Q[0] + t[0];
        c >>= 64;
        for (int j = 1; j < M; ++j) {
            c += (u128)m * Q[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[M];
        t[M - 1] = (uint64_t)c;
        t[M] = t[M + 1] + (uint64_t)(c >> 64);
    }
    uint32_t w[P::N];
    for (int i = 0; i < M; ++i) { w[2 * i] = (uint32_t)t[i]; w[2 * i + 1] = (uint32_t)(t[i] >> 32); }
    Fp<P> r;
    reduce_once<P>(r.l, w, (uint32_t)t[M]);
    return r;
}
#define PM_HOST_MUL64 1
#endif

// acc += x * y, pinned: LLVM's reassociation orders a sum by dependency depth and would add the carry (the deepest operand) LAST,
// as a separate 64-bit addition.  llvm.annotation is opaque to the IR optimiser and vanishes at instruction selection (an empty
// inline asm does the same job but makes the hazard recogniser put an s_nop behind every one of them): the chain keeps the order
// written here and each step is one v_mad_u64_u32 whose addend is the previous step.
#if defined(__clang__)
#define PM_PIN64(v) ((v) = __builtin_annotation((v), "pm_chain"))
#else
#define PM_PIN64(v) ((void)0)
#endif

// The same product a b 2^(-32 N) mod p, same canonical result, computed on W-bit limbs (Radix28<P>::RR: W = 28, or 29 for the
// 254/255-bit scalar fields = 9 limbs): a 64-bit accumulator of v_mad_u64_u32 absorbs a whole column of W x W-bit products
// (2 L 2^(2W) < 2^64), so there is no carry chain inside the loop (fq28.cuh uses the idea with its own radix; here the radix
// stays the dense one, 2^(32 N)).  L - 1 full Montgomery steps of W bits and one partial step of TAIL = 32 N - W (L - 1) bits
// (24 for the scalar fields, 20 for BLS12-381 Fq), then a TAIL-bit right shift.
template <class P>
PM_HD Fp<P> mul_r28(const Fp<P> &a, const Fp<P> &b) {
    typedef typename Radix28<P>::RR RR;
    constexpr int N = P::N, L = RR::N, W = RR::W, TAIL = 32 * N - W * (L - 1);
    static_assert(TAIL > 0 && TAIL <= W && 2 * W + 5 <= 63 && 2 * L <= 32, "limb layout: 2 L products of 2 W bits per 64-bit column");
    constexpr uint32_t MASK = RR::MASK;
    uint32_t A[L], B[L];
#pragma unroll
    for (int i = 0; i < L; ++i) {   // 32-bit words -> 28-bit limbs (the top limb holds TAIL bits)
        const int bit = W * i, w = bit >> 5, s = bit & 31;
        uint64_t ta = w < N ? a.l[w] : 0u, tb = w < N ? b.l[w] : 0u;
        if (w + 1 < N) { ta |= (uint64_t)a.l[w + 1] << 32; tb |= (uint64_t)b.l[w + 1] << 32; }
        A[i] = (uint32_t)(ta >> s) & MASK;
        B[i] = (uint32_t)(tb >> s) & MASK;
    }
    // product scanning: one accumulator walks the columns, the carry of a column is the addend of the next column's first
    // v_mad_u64_u32 (fq28.cuh has the reasoning and PM_PIN64).  Columns 0 .. L-2 clear W bits each, column L-1 only TAIL bits;
    // columns L-1 .. 2L-2 are the result before the TAIL-bit shift (the top one keeps its overflow).
    uint32_t m[L];
    uint64_t t[L];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * L - 1; ++k) {
#pragma unroll
        for (int i = (k < L ? 0 : k - L + 1); i <= (k < L ? k : L - 1); ++i) { acc += (uint64_t)A[i] * B[k - i]; PM_PIN64(acc); }
#pragma unroll
        for (int i = (k < L ? 0 : k - L + 1); i <= (k < L ? k - 1 : L - 1); ++i) { acc += (uint64_t)m[i] * RR::MOD[k - i]; PM_PIN64(acc); }
        if (k < L - 1) {
            m[k] = ((uint32_t)acc * RR::INV) & MASK;
            acc += (uint64_t)m[k] * RR::MOD[0]; PM_PIN64(acc);
        } else if (k == L - 1) {   // partial step: clear the low TAIL bits only
            uint32_t mt = ((uint32_t)acc * RR::INV) & ((1u << TAIL) - 1u);
#if defined(__HIP_DEVICE_COMPILE__)
            // TAIL = 24 (9 limbs of 29 bits) makes mt a known-24-bit value: hipcc (ROCm 7.2) then forms a 24-bit multiply with the
            // top limb of p, drops the mask as redundant for it, and fuses the product into v_mad_u64_u32 -- which does not truncate
            // its operands (seen in the ISA: the unmasked -acc times MOD[8]; 4 081 of 4 096 random products wrong on the device, none
            // on the host).  Keeping the masked value opaque makes the compiler multiply the register it was given.
            asm volatile("" : "+v"(mt));
#endif
            m[k] = mt;
            acc += (uint64_t)mt * RR::MOD[0]; PM_PIN64(acc);
        }
        if (k >= L - 1) t[k - (L - 1)] = k < 2 * L - 2 ? (acc & MASK) : acc;
        acc >>= W;
    }
    // shift right by TAIL: value < 2p
    uint32_t r28[L];
#pragma unroll
    for (int j = 0; j < L; ++j) {
        uint64_t v = t[j] >> TAIL;
        if (j + 1 < L) v |= (t[j + 1] << (W - TAIL)) & MASK;
        r28[j] = (uint32_t)v;
    }
    // conditional subtraction of p on the limbs, then 28-bit limbs -> 32-bit words
    uint32_t d28[L];
    uint32_t borrow = 0;
#pragma unroll
    for (int j = 0; j < L; ++j) {
        const uint32_t v = r28[j] - RR::MOD[j] - borrow;
        borrow = v >> 31;               // limbs < 2^W <= 2^29: a negative difference sets bit 31
        d28[j] = v & MASK;
    }
#pragma unroll
    for (int j = 0; j < L; ++j) r28[j] = borrow ? r28[j] : d28[j];
    Fp<P> r;
#pragma unroll
    for (int w = 0; w < N; ++w) {
        const int bit = 32 * w, i = bit / W, s = bit % W;
        uint64_t v = (uint64_t)r28[i] >> s;
        if (i + 1 < L) v |= (uint64_t)r28[i + 1] << (W - s);
        if (i + 2 < L && 2 * W - s < 32) v |= (uint64_t)r28[i + 2] << (2 * W - s);
        r.l[w] = (uint32_t)v;
    }
    return r;
}

// Device code takes the 28-bit-limb product (fewer instructions on gfx950); host code (key generation glue,
// transcripts, verifier) the 32-bit CIOS, which a CPU's carry flags run faster.  Identical results.
template <class P>
PM_HD Fp<P> mul(const Fp<P> &a, const Fp<P> &b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return mul_r28<P>(a, b);
#elif defined(PM_HOST_MUL64)
    return mul_cios64<P>(a, b);
#else
    return mul_cios<P>(a, b);
#endif
}

template <class P>
PM_HD Fp<P> sqr(const Fp<P> &a) {
    return mul<P>(a, a);
}

// Montgomery form -> canonical integer limbs (multiply by 1).
template <class P>
PM_HD Fp<P> from_mont(const Fp<P> &a) {
    Fp<P> o = Fp<P>::zero();
    o.l[0] = 1;
    return mul<P>(a, o);
}
template <class P>
PM_HD Fp<P> to_mont(const Fp<P> &a) {
    return mul<P>(a, Fp<P>::r2());
}

template <class P>
PM_HD_COLD Fp<P> pow_u64(const Fp<P> &a, uint64_t e) {
    Fp<P> acc = Fp<P>::one(), base = a;
    while (e) {
        if (e & 1) acc = mul<P>(acc, base);
        base = sqr<P>(base);
        e >>= 1;
    }
    return acc;
}

// a^(MOD-2): Fermat inverse (host glue and one-off device normalisations only).
template <class P>
PM_HD_COLD Fp<P> inverse(const Fp<P> &a) {
    constexpr int N = P::N;
    uint32_t e[N];
    uint64_t borrow = 2;
    for (int i = 0; i < N; ++i) {
        uint64_t d = (uint64_t)P::MOD[i] - borrow;
        e[i] = (uint32_t)d;
        borrow = (d >> 32) & 1;
    }
    Fp<P> acc = Fp<P>::one();
    bool started = false;
    for (int i = N - 1; i >= 0; --i)
        for (int b = 31; b >= 0; --b) {
            if (started) acc = sqr<P>(acc);
            if ((e[i] >> b) & 1) {
                acc = started ? mul<P>(acc, a) : a;
                started = true;
            }
        }
    return acc;
}

template <class P>
PM_HD Fp<P> from_u64(uint64_t v) {
    Fp<P> r = Fp<P>::zero();
    r.l[0] = (uint32_t)v;
    r.l[1] = (uint32_t)(v >> 32);
    return to_mont<P>(r);
}

}  // namespace pm
