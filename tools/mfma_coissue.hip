// Do v_mfma_i32_32x32x32_i8 and v_mad_u64_u32 overlap on one SIMD of gfx950?  
// Loop bodies built from inline asm so that the compiler cannot reorder them:
//   mfma   16 x MFMA (4 independent accumulators)
//   mad    16 x G v_mad_u64_u32 (8 independent accumulators)
//   both   16 x { MFMA ; G x v_mad_u64_u32 }  -- G = 4, 6, 8
//   swap   64 x v_permlane32_swap_b32
// at 1 and 2 waves per SIMD.  If the pipes overlap, both ~ max(mfma, mad); if not, both ~ mfma + mad.
//   build: hipcc --offload-arch=gfx950 -O3 tools/mfma_coissue.hip -o tools/mfma_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define MAD(r) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r) : "v"(a), "v"(b) : "vcc");
#define MFMA(d) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(d) : "v"(A), "v"(B));

template <int MODE, int G>
__global__ __launch_bounds__(256, 2) void k(uint32_t *out, int iters) {
    uint32_t a = threadIdx.x * 2654435761u + 1, b = a ^ 0x9e3779b9u;
    uint64_t r0 = a, r1 = b, r2 = a + 1, r3 = b + 1, r4 = a + 2, r5 = b + 2, r6 = a + 3, r7 = b + 3;
    v4i A = {(int)a, (int)b, (int)(a * 3), (int)(b * 5)}, B = {(int)(a * 7), (int)(b * 11), (int)(a * 13), (int)(b * 17)};
    v16i d0 = {}, d1 = {}, d2 = {}, d3 = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (MODE & 1) MFMA(d0)
            if (MODE & 2) { MAD(r0) MAD(r1) MAD(r2) MAD(r3) if (G > 4) { MAD(r4) MAD(r5) } if (G > 6) { MAD(r6) MAD(r7) } }
            if (MODE & 1) MFMA(d1)
            if (MODE & 2) { MAD(r0) MAD(r1) MAD(r2) MAD(r3) if (G > 4) { MAD(r4) MAD(r5) } if (G > 6) { MAD(r6) MAD(r7) } }
            if (MODE & 1) MFMA(d2)
            if (MODE & 2) { MAD(r0) MAD(r1) MAD(r2) MAD(r3) if (G > 4) { MAD(r4) MAD(r5) } if (G > 6) { MAD(r6) MAD(r7) } }
            if (MODE & 1) MFMA(d3)
            if (MODE & 2) { MAD(r0) MAD(r1) MAD(r2) MAD(r3) if (G > 4) { MAD(r4) MAD(r5) } if (G > 6) { MAD(r6) MAD(r7) } }
        }
        if (MODE & 4) {
            uint32_t x = (uint32_t)r0, y = (uint32_t)r1;
#pragma unroll
            for (int s = 0; s < 64; ++s) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
            r0 += x; r1 += y;
        }
    }
    uint32_t acc = (uint32_t)(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7);
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += d0[i] + d1[i] + d2[i] + d3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE, int G>
static void run(const char *name, int cus, int w, uint32_t *out, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE, G><<<cus * w, 256>>>(out, 16);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        k<MODE, G><<<cus * w, 256>>>(out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double cyc = best * 1e-3 * 2.4e9 / ((double)iters * w);   // nominal cycles per loop body per wave slot
    printf("{\"body\": \"%s\", \"mads_per_mfma\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"nominal_cycles_per_body_per_simd\": %.1f}\n", name, G, w, best, cyc);
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
    const int cus = prop.multiProcessorCount, iters = 20000;
    uint32_t *out;
    (void)hipMalloc(&out, (size_t)cus * 2 * 256 * 4);
    for (int w : {1, 2}) {
        run<1, 4>("16 mfma", cus, w, out, iters);
        run<2, 4>("16 x 4 mad", cus, w, out, iters);
        run<3, 4>("16 x (mfma + 4 mad)", cus, w, out, iters);
        run<2, 6>("16 x 6 mad", cus, w, out, iters);
        run<3, 6>("16 x (mfma + 6 mad)", cus, w, out, iters);
        run<2, 8>("16 x 8 mad", cus, w, out, iters);
        run<3, 8>("16 x (mfma + 8 mad)", cus, w, out, iters);
        run<4, 4>("64 permlane32_swap", cus, w, out, iters);
    }
    return 0;
}
