// How fast can 33.6 MB of pinned host memory reach HBM?  (the scalar H2D inside every timed proof)
//   a. one hipMemcpyAsync            b. two halves on two streams            c. a copy KERNEL reading the mapped host pointer
//   hipcc --offload-arch=gfx950 -O3 tools/h2d_bench.hip -o tools/h2d_bench && tools/h2d_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_copy(const uint4 *src, uint4 *dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t bytes = 33554496;   // (2^20 + 2) Fr
    void *h = nullptr, *d = nullptr;
    CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
    memset(h, 0x5a, bytes);
    CK(hipMalloc(&d, bytes));
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1));
    CK(hipStreamCreate(&s2));
    void *hd = nullptr;
    CK(hipHostGetDevicePointer(&hd, h, 0));
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s1));
        CK(hipStreamSynchronize(s1));
        double a = now() - t0;
        t0 = now();
        CK(hipMemcpyAsync(d, h, bytes / 2, hipMemcpyHostToDevice, s1));
        CK(hipMemcpyAsync((char *)d + bytes / 2, (char *)h + bytes / 2, bytes - bytes / 2, hipMemcpyHostToDevice, s2));
        CK(hipStreamSynchronize(s1));
        CK(hipStreamSynchronize(s2));
        double b = now() - t0;
        double c[3];
        const int grids[3] = {256, 1024, 4096};
        for (int g = 0; g < 3; ++g) {
            t0 = now();
            hipLaunchKernelGGL(k_copy, dim3(grids[g]), dim3(256), 0, s1, (const uint4 *)hd, (uint4 *)d, bytes / 16);
            CK(hipStreamSynchronize(s1));
            c[g] = now() - t0;
        }
        printf("rep %d: memcpy %.3f ms (%.1f GB/s) | two streams %.3f ms (%.1f GB/s) | kernel 256/1024/4096 wg: %.3f / %.3f / %.3f ms (%.1f / %.1f / %.1f GB/s)\n", rep,
               a, bytes / a / 1e6, b, bytes / b / 1e6, c[0], c[1], c[2], bytes / c[0] / 1e6, bytes / c[1] / 1e6, bytes / c[2] / 1e6);
    }
    return 0;
}
