// Prices a BLS12-381 Fq mixed addition on 13 BALANCED signed limbs of 30 bits (v_mad_i64_i32, Montgomery radix 2^390) against
// fq28.cuh's 14 unsigned limbs of 28 bits: 3 224 + 52 instead of 3 542 multiplier products per addition, every sum that enters a
// product carry-normalised or lifted into a reduction (a signed 64-bit column holds 26 products of 2^58 and nothing more).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Ipolymath_amd/csrc tools/f30_bench.hip -o tools/f30_bench
// Checks f30_mul / f30_mul_lift against the big-integer model's known answers (tools/f30_model.py -> f30_tables.h), then times
// ITER additions per lane at two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "fq28.cuh"
#include "f30_tables.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

namespace f30 {
struct F { int32_t l[N]; };
__device__ __forceinline__ int32_t sext30(uint32_t x) { return (int32_t)(x << 2) >> 2; }
__device__ __forceinline__ int32_t modl(int i) { constexpr int32_t M[N] = {F30_MOD_LITERALS}; return M[i]; }

// (a b + m p) / 2^390 [- lift]: product scanning on one signed accumulator.  Inputs |limb| <= 2^29 (26 products of 2^58 per column),
// output limbs s + carry-bit in [-2^29, 2^29]; LIFT: limb j of `lift` (|.| < 2^31) is subtracted in column N + j.
template <bool LIFT, bool SQR>
__device__ __forceinline__ F mul_impl(const F &a, const F &b, const F &lift) {
    int32_t m[N], a2[N];
    if (SQR) {
#pragma unroll
        for (int i = 0; i < N; ++i) a2[i] = a.l[i] * 2;
    }
    int32_t minus_one = -1;
    if (LIFT) asm("" : "+s"(minus_one));
    F r;
    int64_t acc = 0;
    int32_t bit = 0;
#pragma unroll
    for (int k = 0; k < 2 * N - 1; ++k) {
        if (SQR) {
#pragma unroll
            for (int i = (k < N ? 0 : k - N + 1); 2 * i < k; ++i) { acc += (int64_t)a2[i] * a.l[k - i]; PM_PIN64(acc); }
            if ((k & 1) == 0) { acc += (int64_t)a.l[k / 2] * a.l[k / 2]; PM_PIN64(acc); }
        } else {
#pragma unroll
            for (int i = (k < N ? 0 : k - N + 1); i <= (k < N ? k : N - 1); ++i) { acc += (int64_t)a.l[i] * b.l[k - i]; PM_PIN64(acc); }
        }
#pragma unroll
        for (int i = (k < N ? 0 : k - N + 1); i <= (k < N ? k - 1 : N - 1); ++i) { acc += (int64_t)m[i] * modl(k - i); PM_PIN64(acc); }
        if (k < N) {
            m[k] = sext30((uint32_t)acc * INV);
            acc += (int64_t)m[k] * modl(0); PM_PIN64(acc);
        } else {
            if (LIFT) { acc += (int64_t)lift.l[k - N] * minus_one; PM_PIN64(acc); }
            const uint32_t lo = (uint32_t)acc;
            r.l[k - N] = sext30(lo) + bit;
            bit = (int32_t)((lo >> 29) & 1u);
        }
        acc >>= W;
    }
    r.l[N - 1] = (int32_t)acc + bit - (LIFT ? lift.l[N - 1] : 0);
    return r;
}
__device__ __forceinline__ F mul(const F &a, const F &b) { return mul_impl<false, false>(a, b, a); }
__device__ __forceinline__ F sqr(const F &a) { return mul_impl<false, true>(a, a, a); }
__device__ __forceinline__ F mul_lift(const F &a, const F &b, const F &lift) { return mul_impl<true, false>(a, b, lift); }
__device__ __forceinline__ F sqr_lift(const F &a, const F &lift) { return mul_impl<true, true>(a, a, lift); }
// a - b, carry-normalised: limbs back into [-2^29, 2^29), the top limb takes the rest
__device__ __forceinline__ F sub_norm(const F &a, const F &b) {
    F r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < N - 1; ++i) {
        const int32_t t = a.l[i] - b.l[i] + c;
        c = (t + (1 << 29)) >> 30;
        r.l[i] = sext30((uint32_t)t);
    }
    r.l[N - 1] = a.l[N - 1] - b.l[N - 1] + c;
    return r;
}
struct XYZZ { F X, Y, ZZ, ZZZ; };
__device__ __forceinline__ bool all_zero(const F &a) {
    int32_t o = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) o |= a.l[i];
    return o == 0;
}
// acc += (x2, +-y2): madd-2008-s.  P, R, X3 and Y3 come out of reductions that carry their linear terms (no separate carry pass).
__device__ __forceinline__ bool madd(XYZZ &acc, const F &x2, F y2, bool negate, const F &one) {
    if (negate) {
#pragma unroll
        for (int i = 0; i < N; ++i) y2.l[i] = -y2.l[i];
    }
    if (all_zero(acc.ZZ)) { acc.X = x2; acc.Y = y2; acc.ZZ = one; acc.ZZZ = one; return true; }
    const F P = mul_lift(x2, acc.ZZ, acc.X);      // U2 - X1
    const F R = mul_lift(y2, acc.ZZZ, acc.Y);     // S2 - Y1
    const F PP = sqr(P);
    if (PP.l[0] == 0 && all_zero(PP)) return false;
    const F PPP = mul(P, PP), Q = mul(acc.X, PP);
    F w;                                           // PPP + 2Q, |limb| <= 3 2^29
#pragma unroll
    for (int i = 0; i < N; ++i) w.l[i] = PPP.l[i] + 2 * Q.l[i];
    const F X3 = sqr_lift(R, w);                   // R^2 - PPP - 2Q
    const F QX = sub_norm(Q, X3);
    const F YP = mul(acc.Y, PPP);
    acc.Y = mul_lift(R, QX, YP);                   // R (Q - X3) - Y1 PPP
    acc.X = X3;
    acc.ZZ = mul(acc.ZZ, PP);
    acc.ZZZ = mul(acc.ZZZ, PPP);
    return true;
}
}  // namespace f30

__global__ void k_check(const int32_t *vec, int32_t *out) {
    const int t = threadIdx.x;
    if (t >= f30::NVEC) return;
    f30::F a, b, lift;
    for (int i = 0; i < f30::N; ++i) { a.l[i] = vec[(t * 5 + 0) * f30::N + i]; b.l[i] = vec[(t * 5 + 1) * f30::N + i]; lift.l[i] = vec[(t * 5 + 2) * f30::N + i]; }
    const f30::F r = f30::mul(a, b), rl = f30::mul_lift(a, b, lift);
    for (int i = 0; i < f30::N; ++i) { out[(t * 2) * f30::N + i] = r.l[i]; out[(t * 2 + 1) * f30::N + i] = rl.l[i]; }
}

__global__ __launch_bounds__(128) void k_madd30(const int32_t *in, int32_t *out, int iters) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    f30::F x2, y2, one;
    for (int i = 0; i < f30::N; ++i) { x2.l[i] = in[t % 64 * 39 + i]; y2.l[i] = in[t % 64 * 39 + 13 + i]; one.l[i] = in[t % 64 * 39 + 26 + i]; }
    f30::XYZZ acc;
    for (int i = 0; i < f30::N; ++i) acc.X.l[i] = acc.Y.l[i] = acc.ZZ.l[i] = acc.ZZZ.l[i] = 0;
    int bad = 0;
    for (int it = 0; it < iters; ++it) {
        x2.l[0] ^= (it & 0xff);
        y2.l[1] ^= (it & 0x3f);
        if (!f30::madd(acc, x2, y2, (it & 1) != 0, one)) ++bad;
    }
    int32_t s = bad;
    for (int i = 0; i < f30::N; ++i) s ^= acc.X.l[i] ^ acc.Y.l[i] ^ acc.ZZ.l[i] ^ acc.ZZZ.l[i];
    out[t] = s;
}

__global__ __launch_bounds__(128) void k_madd28(const uint32_t *in, uint32_t *out, int iters) {
    using namespace pm;
    typedef BlsCurve::FqRR RR;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    F28<RR> x2, y2;
    for (int i = 0; i < 14; ++i) { x2.l[i] = in[t % 64 * 28 + i]; y2.l[i] = in[t % 64 * 28 + 14 + i]; }
    XYZZ28<BlsCurve> acc;
    acc.X = acc.Y = acc.ZZ = acc.ZZZ = f28_zero<RR>();
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        x2.l[0] ^= (it & 0xff);
        y2.l[1] ^= (it & 0x3f);
        if (!xyzz28_madd_limbs<BlsCurve>(acc, x2, y2, (it & 1) != 0)) ++bad;
    }
    uint32_t s = bad;
    for (int i = 0; i < 14; ++i) s ^= acc.X.l[i] ^ acc.Y.l[i] ^ acc.ZZ.l[i] ^ acc.ZZZ.l[i];
    out[t] = s;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    // known answers
    {
        int32_t *dv, *dout;
        CK(hipMalloc(&dv, sizeof(f30::VEC)));
        CK(hipMalloc(&dout, f30::NVEC * 2 * f30::N * 4));
        CK(hipMemcpy(dv, f30::VEC, sizeof(f30::VEC), hipMemcpyHostToDevice));
        k_check<<<1, 64>>>(dv, dout);
        std::vector<int32_t> o(f30::NVEC * 2 * f30::N);
        CK(hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int t = 0; t < f30::NVEC; ++t)
            for (int i = 0; i < f30::N; ++i) {
                bad += o[(t * 2) * f30::N + i] != f30::VEC[t][3][i];
                bad += o[(t * 2 + 1) * f30::N + i] != f30::VEC[t][4][i];
            }
        printf("known answers: %d products and %d lifted products, %d limb mismatches\n", f30::NVEC, f30::NVEC, bad);
    }
    const size_t lanes = (size_t)cus * 4 * 2 * 64;   // two waves per SIMD
    std::vector<int32_t> in30(64 * 39);
    std::vector<uint32_t> in28(64 * 28);
    srand(7);
    for (int l = 0; l < 64; ++l) {
        for (int i = 0; i < 26; ++i) in30[l * 39 + i] = (i % 13 == 12) ? (rand() % (1 << 20)) : ((rand() % (1 << 30)) - (1 << 29));
        for (int i = 0; i < 13; ++i) in30[l * 39 + 26 + i] = f30::ONE[i];
        for (int i = 0; i < 28; ++i) in28[l * 28 + i] = (i % 14 == 13) ? (rand() % (1 << 16)) : (rand() % (1 << 28));
    }
    int32_t *d30, *o30;
    uint32_t *d28, *o28;
    CK(hipMalloc(&d30, in30.size() * 4)); CK(hipMalloc(&o30, lanes * 4));
    CK(hipMalloc(&d28, in28.size() * 4)); CK(hipMalloc(&o28, lanes * 4));
    CK(hipMemcpy(d30, in30.data(), in30.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d28, in28.data(), in28.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 4; ++rep) {
        for (int which = 0; which < 2; ++which) {
            CK(hipEventRecord(e0));
            if (which == 0) k_madd28<<<lanes / 128, 128>>>(d28, o28, iters);
            else k_madd30<<<lanes / 128, 128>>>(d30, o30, iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("rep %d %s: %.3f ms for %d additions per lane = %.2f ns per addition and SIMD-wave pair, %.1f M additions/s\n", rep,
                   which ? "13 x 30 signed" : "14 x 28 unsigned", ms, iters, ms * 1e6 / iters, (double)lanes * iters / ms / 1e3);
        }
    }
    return 0;
}
