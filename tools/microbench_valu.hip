// Measures per-instruction VALU issue rates on gfx950 for the integer building blocks of the
// Montgomery multiplier (DESIGN.md §Field arithmetic): result = wave-instructions per cycle per CU
// and lane-ops/s chip-wide.  Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_valu.hip -o tools/microbench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITERS 4096
#define CHAINS 8

#define BENCH_KERNEL(NAME, BODY)                                                             \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed) {              \
        uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;                  \
        uint64_t r0 = a, r1 = b, r2 = a + 1, r3 = b + 1, r4 = a + 2, r5 = b + 2, r6 = a + 3, r7 = b + 3; \
        uint32_t x0 = a, x1 = b, x2 = a + 5, x3 = b + 5, x4 = a + 7, x5 = b + 7, x6 = a + 9, x7 = b + 9; \
        double d0 = a, d1 = b, d2 = a + 1, d3 = b + 1, d4 = a + 2, d5 = b + 2, d6 = a + 3, d7 = b + 3;   \
        double dm = 1.0000001, da = 0.5;                                                     \
        for (int i = 0; i < ITERS; ++i) { BODY }                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7) + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7); \
    }

#define MAD64(r) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r) : "v"(a), "v"(b) : "vcc");
#define MULLO(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(a));
#define MULHI(x) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(a));
#define ADDU(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(a));
#define ADDCO(x) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(x) : "v"(a) : "vcc");
#define ADDC(x) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x) : "v"(a) : "vcc");
#define ADD64(r) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(r) : "v"(r1));
#define MAD24(x) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define MULHI24(x) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x) : "v"(a));
#define ADD3(x) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define FMA64(d) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d) : "v"(dm), "v"(da));
#define FMA32(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define CNDMASK(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(a));
#define MADI32(x) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));

#define ALL8(M, p) M(p##0) M(p##1) M(p##2) M(p##3) M(p##4) M(p##5) M(p##6) M(p##7)

BENCH_KERNEL(k_mad64, ALL8(MAD64, r))
BENCH_KERNEL(k_mullo, ALL8(MULLO, x))
BENCH_KERNEL(k_mulhi, ALL8(MULHI, x))
BENCH_KERNEL(k_addu, ALL8(ADDU, x))
BENCH_KERNEL(k_addco, ALL8(ADDCO, x))
BENCH_KERNEL(k_addc, ALL8(ADDC, x))
BENCH_KERNEL(k_add64, ALL8(ADD64, r))
BENCH_KERNEL(k_mad24, ALL8(MAD24, x))
BENCH_KERNEL(k_mulhi24, ALL8(MULHI24, x))
BENCH_KERNEL(k_add3, ALL8(ADD3, x))
BENCH_KERNEL(k_fma64, ALL8(FMA64, d))
BENCH_KERNEL(k_fma32, ALL8(FMA32, x))
BENCH_KERNEL(k_cndmask, ALL8(CNDMASK, x))
// mixed: one mad64 + one addc per product (the multiplier's inner step)
BENCH_KERNEL(k_mad64_addc, MAD64(r0) ADDC(x0) MAD64(r1) ADDC(x1) MAD64(r2) ADDC(x2) MAD64(r3) ADDC(x3))

typedef void (*kern_t)(uint32_t *, uint32_t);
struct Case { const char *name; kern_t k; int per_iter; };

int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
    int cus = prop.multiProcessorCount;
    double clk_ghz = prop.clockRate / 1e6;
    printf("device %s CUs %d clock %.2f GHz\n", prop.name, cus, clk_ghz);
    Case cases[] = {{"v_mad_u64_u32", k_mad64, 8}, {"v_mul_lo_u32", k_mullo, 8}, {"v_mul_hi_u32", k_mulhi, 8},
                    {"v_add_u32", k_addu, 8}, {"v_add_co_u32", k_addco, 8}, {"v_addc_co_u32", k_addc, 8},
                    {"v_lshl_add_u64", k_add64, 8}, {"v_mad_u32_u24", k_mad24, 8}, {"v_mul_hi_u32_u24", k_mulhi24, 8},
                    {"v_add3_u32", k_add3, 8}, {"v_fma_f64", k_fma64, 8}, {"v_fma_f32", k_fma32, 8},
                    {"v_cndmask_b32", k_cndmask, 8}, {"mad64+addc pair", k_mad64_addc, 8}};
    uint32_t *out;
    for (int waves_per_simd : {1, 2, 4}) {
        int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD
        hipMalloc(&out, (size_t)blocks * 256 * 4);
        printf("--- %d wave(s) per SIMD\n", waves_per_simd);
        for (auto &c : cases) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            c.k<<<blocks, 256>>>(out, 1);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int rep = 0; rep < 5; ++rep) c.k<<<blocks, 256>>>(out, rep);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double wave_instr = 5.0 * blocks * 4.0 * ITERS * c.per_iter;   // wave-instructions
            double per_s = wave_instr / (ms * 1e-3);
            double cyc_per_instr_per_simd = (cus * 4.0 * clk_ghz * 1e9) / per_s;
            printf("%-18s %8.3f ms  %7.2f G wave-instr/s  %6.2f cycles/instr/SIMD (at nominal clk)  %7.2f T lane-ops/s\n",
                   c.name, ms / 5, per_s / 1e9, cyc_per_instr_per_simd, per_s * 64 / 1e12);
        }
        hipFree(out);
    }
    return 0;
}
