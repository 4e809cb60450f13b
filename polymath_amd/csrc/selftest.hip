// pm_selftest_field: the device's field products against the host's textbook ones, word for word (tight operands, and the
// lazy ones -- limbs up to 2^(W+1.6), values up to 18p -- that the mixed addition and the butterflies really multiply).
//
// The kernels multiply in reduced radix (field.cuh: mul_r28, fq28.cuh: f28_mul / f28_sqr); the host computes the same
// canonical product with the 32-bit CIOS (mul_cios).  Every proof already depends on the two agreeing, but a proof tells
// little about WHERE they stopped agreeing: this entry point isolates it.  It exists because they once did not -- hipcc
// dropped the mask of a 24-bit Montgomery digit when it fused the product into v_mad_u64_u32 (field.cuh: mul_r28, the
// opaque `m`), 4 081 of 4 096 products wrong on the device and none on the host.
#include <vector>

#include "internal.h"
#include "fq28.cuh"

namespace pm {

template <class P, class RR>
__global__ void k_selftest_field(const Fp<P> *a, const Fp<P> *b, Fp<P> *prod, Fp<P> *square, Fp<P> *prod28, Fp<P> *lazy, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    prod[i] = mul<P>(a[i], b[i]);
    square[i] = sqr<P>(a[i]);
    // the MSM / NTT radix: standard Montgomery operands -> internal form (x STD2INT as a plain integer), product, back
    Fp<P> s2i, i2s;
    for (int k = 0; k < P::N; ++k) { s2i.l[k] = RR::STD2INT[k]; i2s.l[k] = RR::INT2STD[k]; }
    const Fp<P> ai = mul<P>(a[i], s2i), bi = mul<P>(b[i], s2i);
    Fp<P> pi;
    f28_pack_reduced<RR>(f28_mul<RR>(f28_unpack<RR>(ai.l), f28_unpack<RR>(bi.l)), pi.l);
    prod28[i] = mul<P>(pi, i2s);
    // the same radix on LAZY operands, as the mixed addition and the butterflies feed them: limbs up to 2^(W+1.6), values up to 18p
    const F28<RR> fa = f28_unpack<RR>(ai.l), fb = f28_unpack<RR>(bi.l);
    const F28<RR> d = f28_sub_k16<RR>(fa, fb);                                                  // a - b + 16p
    f28_pack_reduced<RR>(f28_mul<RR>(d, fb), pi.l);                                             // (a - b) b
    lazy[3 * i] = mul<P>(pi, i2s);
    if (RR::W == 28) {   // the 29-bit tiles only ever multiply a lazy value by a tight twiddle
        f28_pack_reduced<RR>(f28_sqr<RR>(d), pi.l);                                             // (a - b)^2
        lazy[3 * i + 1] = mul<P>(pi, i2s);
        f28_pack_reduced<RR>(f28_mul2_add<RR>(d, f28_sub_k16<RR>(fb, fa), fb, f28_sub_k4<RR>(f28_zero<RR>(), fa)), pi.l);   // (a - b)(b - a) + b (-a)
        lazy[3 * i + 2] = mul<P>(pi, i2s);
    }
}

template <class P, class RR>
static int selftest_field(pm_ctx *ctx, size_t n, uint64_t seed, uint64_t *bad) {
    typedef Fp<P> F;
    std::vector<F> a(n), b(n), r(6 * n);
    uint64_t s = seed | 1;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    const int topbits = P::BITS - 32 * (P::N - 1);
    for (size_t it = 0; it < n; ++it) {
        for (int i = 0; i < P::N; ++i) { a[it].l[i] = rnd(); b[it].l[i] = rnd(); }
        a[it].l[P::N - 1] &= (1u << (topbits - 1)) - 1;     // < 2^(BITS-1) < p
        b[it].l[P::N - 1] &= (1u << (topbits - 1)) - 1;
    }
    F pm1;
    for (int i = 0; i < P::N; ++i) pm1.l[i] = P::MOD[i];
    pm1.l[0] -= 1;                                          // p is odd
    if (n > 4) {
        a[0] = pm1; b[0] = pm1;
        a[1] = F::zero();
        a[2] = F::one();
        a[3] = pm1; b[3] = F::one();
    }
    DevBuf da, db, dr;
    PM_HIP(ctx, da.reserve(n * sizeof(F)));
    PM_HIP(ctx, db.reserve(n * sizeof(F)));
    PM_HIP(ctx, dr.reserve(6 * n * sizeof(F)));
    int st = PM_OK;
    do {
        if (hipMemcpyAsync(da.p, a.data(), n * sizeof(F), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(db.p, b.data(), n * sizeof(F), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { st = PM_ERR_HIP; break; }
        hipLaunchKernelGGL((k_selftest_field<P, RR>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, da.as<F>(), db.as<F>(),
                           dr.as<F>(), dr.as<F>() + n, dr.as<F>() + 2 * n, dr.as<F>() + 3 * n, n);
        if (hipGetLastError() != hipSuccess ||
            hipMemcpyAsync(r.data(), dr.p, 6 * n * sizeof(F), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) { st = PM_ERR_HIP; break; }
    } while (false);
    da.release();
    db.release();
    dr.release();
    if (st) { ctx->err = "pm_selftest_field: HIP call failed"; return st; }
    uint64_t m = 0;
    for (size_t it = 0; it < n; ++it) {
        const F p = mul_cios<P>(a[it], b[it]);
        m += !r[it].eq(p);
        m += !r[n + it].eq(mul_cios<P>(a[it], a[it]));
        m += !r[2 * n + it].eq(p);
        const F d = sub<P>(a[it], b[it]), dd = mul_cios<P>(d, d);
        m += !r[3 * n + 3 * it].eq(mul_cios<P>(d, b[it]));
        if (RR::W == 28) {
            m += !r[3 * n + 3 * it + 1].eq(dd);
            m += !r[3 * n + 3 * it + 2].eq(neg<P>(add<P>(dd, p)));
        }
    }
    *bad = m;
    return PM_OK;
}

}  // namespace pm

extern "C" int pm_selftest_field(pm_ctx *ctx, size_t products_per_field, uint64_t seed, uint64_t mismatches[4]) {
    using namespace pm;
    if (!ctx || !mismatches || products_per_field == 0 || products_per_field > ((size_t)1 << 24)) return PM_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return PM_ERR_HIP;
    uint64_t more = 0;
    PM_TRY((selftest_field<BlsFrP, BlsFrRR>(ctx, products_per_field, seed, &mismatches[0])));
    PM_TRY((selftest_field<BlsFrP, BlsFrRR29>(ctx, products_per_field, seed + 4, &more)));      // the NTT tiles' radix
    mismatches[0] += more;
    PM_TRY((selftest_field<BnFrP, BnFrRR>(ctx, products_per_field, seed + 1, &mismatches[1])));
    PM_TRY((selftest_field<BnFrP, BnFrRR29>(ctx, products_per_field, seed + 5, &more)));
    mismatches[1] += more;
    PM_TRY((selftest_field<BlsFqP, BlsFqRR>(ctx, products_per_field, seed + 2, &mismatches[2])));
    PM_TRY((selftest_field<BnFqP, BnFqRR>(ctx, products_per_field, seed + 3, &mismatches[3])));
    return PM_OK;
}
