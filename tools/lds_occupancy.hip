// How many workgroups with a given dynamic LDS size fit one gfx950 CU (160 KiB of LDS): sizing aid for ntt.hip's tiles.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(unsigned *o) { extern __shared__ unsigned s[]; s[threadIdx.x] = threadIdx.x; __syncthreads(); o[threadIdx.x] = s[255 - threadIdx.x]; }
int main() {
    for (int kb : {64, 72, 76, 79, 80, 81, 96, 128, 160}) {
        int nb = -1;
        hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 256, (size_t)kb * 1024);
        printf("%d KiB: %d workgroups per CU (%s)\n", kb, nb, hipGetErrorString(e));
    }
    return 0;
}
