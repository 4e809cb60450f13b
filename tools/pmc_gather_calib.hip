// Calibrates rocprofv3's FETCH_SIZE on gfx950 for k_accumulate's access pattern: every lane gathers a
// 96-byte affine point (6 x 16 B loads) from a random index of a table far larger than the 256 MiB
// Infinity Cache, each point read exactly once (a permutation).  Known unique bytes = 96 * N; at 64-byte
// request granularity the fabric moves 128 B per point.  (MI355X_MICROARCH.md §HBM: FETCH_SIZE halves wide
// coalesced streams and is uncalibrated for other shapes -- this is the calibration it asks for.)
// Build: hipcc --offload-arch=gfx950 -O3 tools/pmc_gather_calib.hip -o tools/pmc_gather_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct P96 { uint4 v[6]; };
__global__ void k_gather96(const P96 *tab, const uint32_t *idx, size_t n, uint32_t *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    P96 p = tab[idx[i]];
    uint32_t a = 0;
    for (int k = 0; k < 6; ++k) a ^= p.v[k].x ^ p.v[k].y ^ p.v[k].z ^ p.v[k].w;
    if (a == 0x12345678u) out[0] = a;   // keep the loads alive
}
__global__ void k_perm(uint32_t *idx, size_t n, uint32_t mul, uint32_t add) {   // n a power of two, mul odd: bijection
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = (uint32_t)((i * mul + add) & (n - 1));
}
__global__ void k_stream(const uint4 *src, size_t n16, uint32_t *out) {        // coalesced 16 B/lane reference stream
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n16) return;
    uint4 v = src[i];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) out[0] = v.x;
}
int main() {
    const size_t n = (size_t)1 << 24;   // 16 Mi points = 1.5 GiB table
    P96 *tab; uint32_t *idx, *out;
    hipMalloc(&tab, n * sizeof(P96)); hipMalloc(&idx, n * 4); hipMalloc(&out, 64);
    hipMemset(tab, 1, n * sizeof(P96));
    k_perm<<<(unsigned)(n / 256), 256>>>(idx, n, 2654435761u, 12345u);
    hipDeviceSynchronize();
    k_gather96<<<(unsigned)(n / 256), 256>>>(tab, idx, n, out);
    k_stream<<<(unsigned)(n * 6 / 256), 256>>>((const uint4 *)tab, n * 6, out);
    hipDeviceSynchronize();
    printf("points %zu unique_bytes %zu sector_model_bytes %zu stream_bytes %zu\n", n, n * 96, n * 128, n * 96);
    return 0;
}
