#include <cstdio>
#include <hip/hip_runtime.h>
#include "../polymath_amd/csrc/field.cuh"   // build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -o tools/dense_mul_device_check tools/dense_mul_device_check.hip
using namespace pm;
template <class P>
__global__ void k_only(const Fp<P> *a, const Fp<P> *b, Fp<P> *r1, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) r1[i] = mul_r28<P>(a[i], b[i]);
}
template <class P>
int run(const char *name, int threads) {
    const int n = 4096;
    Fp<P> *a, *b, *r1;
    hipMallocManaged(&a, n * sizeof(Fp<P>)); hipMallocManaged(&b, n * sizeof(Fp<P>)); hipMallocManaged(&r1, n * sizeof(Fp<P>));
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    const int topbits = P::BITS - 32 * (P::N - 1);
    for (int it = 0; it < n; ++it) {
        for (int i = 0; i < P::N; ++i) { a[it].l[i] = rnd(); b[it].l[i] = rnd(); }
        a[it].l[P::N - 1] &= (1u << (topbits - 1)) - 1; b[it].l[P::N - 1] &= (1u << (topbits - 1)) - 1;
    }
    hipLaunchKernelGGL(k_only<P>, dim3(n / threads), dim3(threads), 0, 0, a, b, r1, n);
    hipDeviceSynchronize();
    int bad = 0, badr = 0;
    for (int it = 0; it < n; ++it) { if (!r1[it].eq(mul_cios<P>(a[it], b[it]))) bad++; if (!mul_r28<P>(a[it], b[it]).eq(mul_cios<P>(a[it], b[it]))) badr++; }
    printf("%s threads %d: device mul_r28 vs host cios: %d mismatches; host r28 vs host cios %d\n", name, threads, bad, badr);
    return bad;
}
int main() { run<BlsFrP>("BlsFr", 1); run<BlsFrP>("BlsFr", 64); run<BlsFrP>("BlsFr", 256); run<BnFrP>("BnFr", 256); return 0; }
