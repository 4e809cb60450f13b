#include <cstdio>
#include <hip/hip_runtime.h>
#include "../polymath_amd/csrc/field.cuh"
using namespace pm;
struct Dbg { uint32_t A[10], B[10]; uint64_t acc_full[10], acc_part[10], t[10]; uint32_t r28[10], borrow; };
template <class P>
__host__ __device__ void mul_dbg(const Fp<P> &a, const Fp<P> &b, Dbg &d) {
    typedef typename Radix28<P>::RR RR;
    constexpr int N = P::N, L = RR::N, W = RR::W, TAIL = 32 * N - W * (L - 1);
    constexpr uint32_t MASK = RR::MASK;
    uint32_t A[L], B[L];
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const int bit = W * i, w = bit >> 5, s = bit & 31;
        uint64_t ta = w < N ? a.l[w] : 0u, tb = w < N ? b.l[w] : 0u;
        if (w + 1 < N) { ta |= (uint64_t)a.l[w + 1] << 32; tb |= (uint64_t)b.l[w + 1] << 32; }
        A[i] = (uint32_t)(ta >> s) & MASK;
        B[i] = (uint32_t)(tb >> s) & MASK;
        d.A[i] = A[i]; d.B[i] = B[i];
    }
    uint64_t acc[L];
#pragma unroll
    for (int j = 0; j < L; ++j) acc[j] = 0;
#pragma unroll
    for (int i = 0; i < L - 1; ++i) {
        const uint32_t bi = B[i];
#pragma unroll
        for (int j = 0; j < L; ++j) acc[j] += (uint64_t)A[j] * bi;
        const uint32_t m = ((uint32_t)acc[0] * RR::INV) & MASK;
#pragma unroll
        for (int j = 0; j < L; ++j) acc[j] += (uint64_t)m * RR::MOD[j];
        const uint64_t carry = acc[0] >> W;
#pragma unroll
        for (int j = 0; j < L - 1; ++j) acc[j] = acc[j + 1];
        acc[L - 1] = 0;
        acc[0] += carry;
    }
    for (int j = 0; j < L; ++j) d.acc_full[j] = acc[j];
    {
        const uint32_t bi = B[L - 1];
#pragma unroll
        for (int j = 0; j < L; ++j) acc[j] += (uint64_t)A[j] * bi;
        const uint32_t m = ((uint32_t)acc[0] * RR::INV) & ((1u << TAIL) - 1u);
#pragma unroll
        for (int j = 0; j < L; ++j) acc[j] += (uint64_t)m * RR::MOD[j];
    }
    for (int j = 0; j < L; ++j) d.acc_part[j] = acc[j];
    uint64_t t[L];
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < L; ++j) {
        c += acc[j];
        t[j] = j + 1 < L ? (c & MASK) : c;
        c >>= W;
        d.t[j] = t[j];
    }
    uint32_t r28[L];
#pragma unroll
    for (int j = 0; j < L; ++j) {
        uint64_t v = t[j] >> TAIL;
        if (j + 1 < L) v |= (t[j + 1] << (W - TAIL)) & MASK;
        r28[j] = (uint32_t)v;
        d.r28[j] = r28[j];
    }
}
template <class P>
__global__ void k(const Fp<P> *a, const Fp<P> *b, Dbg *d) { mul_dbg<P>(a[0], b[0], d[0]); }
template <class P>
__global__ void k2(const Fp<P> *a, const Fp<P> *b, Fp<P> *r) { r[0] = mul_r28<P>(a[0], b[0]); r[1] = mul_cios<P>(a[0], b[0]); }
int main() {
    typedef BlsFrP P;
    Fp<P> *a, *b; Dbg *d;
    hipMallocManaged(&a, sizeof(Fp<P>)); hipMallocManaged(&b, sizeof(Fp<P>)); hipMallocManaged(&d, sizeof(Dbg));
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    for (int i = 0; i < P::N; ++i) { a->l[i] = rnd(); b->l[i] = rnd(); }
    a->l[7] &= 0x3fffffff; b->l[7] &= 0x3fffffff;
    hipLaunchKernelGGL(k<P>, dim3(1), dim3(1), 0, 0, a, b, d);
    hipDeviceSynchronize();
    Dbg h; mul_dbg<P>(*a, *b, h);
    for (int j = 0; j < 9; ++j)
        printf("%d A %08x/%08x B %08x/%08x full %016llx/%016llx part %016llx/%016llx t %016llx/%016llx r %08x/%08x\n", j, d->A[j], h.A[j], d->B[j], h.B[j],
               (unsigned long long)d->acc_full[j], (unsigned long long)h.acc_full[j], (unsigned long long)d->acc_part[j], (unsigned long long)h.acc_part[j],
               (unsigned long long)d->t[j], (unsigned long long)h.t[j], d->r28[j], h.r28[j]);
    Fp<P> *r; hipMallocManaged(&r, 2 * sizeof(Fp<P>));
    hipLaunchKernelGGL(k2<P>, dim3(1), dim3(1), 0, 0, a, b, r);
    hipDeviceSynchronize();
    Fp<P> h1 = mul_r28<P>(*a, *b), h2 = mul_cios<P>(*a, *b);
    for (int i = 0; i < 8; ++i) printf("%d dev r28 %08x dev cios %08x host r28 %08x host cios %08x\n", i, r[0].l[i], r[1].l[i], h1.l[i], h2.l[i]);
    return 0;
}
