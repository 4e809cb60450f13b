#include <cstdio>
#include <hip/hip_runtime.h>
#include "../polymath_amd/csrc/field.cuh"
using namespace pm;
template <class P>
__global__ void k(const Fp<P> *a, const Fp<P> *b, Fp<P> *r1, Fp<P> *r2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { r1[i] = mul_r28<P>(a[i], b[i]); r2[i] = mul_cios<P>(a[i], b[i]); }
}
template <class P>
int run(const char *name) {
    const int n = 4096;
    Fp<P> *a, *b, *r1, *r2;
    hipMallocManaged(&a, n * sizeof(Fp<P>)); hipMallocManaged(&b, n * sizeof(Fp<P>)); hipMallocManaged(&r1, n * sizeof(Fp<P>)); hipMallocManaged(&r2, n * sizeof(Fp<P>));
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    const int topbits = P::BITS - 32 * (P::N - 1);
    for (int it = 0; it < n; ++it) {
        for (int i = 0; i < P::N; ++i) { a[it].l[i] = rnd(); b[it].l[i] = rnd(); }
        a[it].l[P::N - 1] &= (1u << (topbits - 1)) - 1; b[it].l[P::N - 1] &= (1u << (topbits - 1)) - 1;
    }
    hipLaunchKernelGGL(k<P>, dim3(n / 256), dim3(256), 0, 0, a, b, r1, r2, n);
    hipDeviceSynchronize();
    int bad = 0, badh = 0;
    for (int it = 0; it < n; ++it) { if (!r1[it].eq(r2[it])) bad++; if (!r2[it].eq(mul_cios<P>(a[it], b[it]))) badh++; }
    for (int it = 0; it < 6; ++it) { Fp<P> h = mul_cios<P>(a[it], b[it]); printf("%s[%d] dev r28:", name, it); for (int i = 0; i < P::N; ++i) printf(" %08x", r1[it].l[i]); printf("\n         cios   :"); for (int i = 0; i < P::N; ++i) printf(" %08x", h.l[i]); printf("\n"); }
    printf("%s device mul_r28 vs device cios: %d mismatches; device cios vs host cios: %d\n", name, bad, badh);
    return bad;
}
int main() { return run<BlsFrP>("BlsFr"); }
