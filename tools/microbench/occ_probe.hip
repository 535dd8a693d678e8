// occupancy probe: resident 256-thread blocks per CU as a function of LDS per block (static 24 KB + dynamic)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256, 6) void k(float* out)
{
    __shared__ float s[6144];
    extern __shared__ float d[];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    out[threadIdx.x] = s[255 - threadIdx.x] + d[0];
}
int main()
{
    hipDeviceProp_t p{};
    hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerMultiprocessor %zu maxSharedMemoryPerMultiProcessor %zu sharedMemPerBlock %zu CUs %d\n", (size_t)p.sharedMemPerMultiprocessor,
           (size_t)p.maxSharedMemoryPerMultiProcessor, (size_t)p.sharedMemPerBlock, p.multiProcessorCount);
    for (int dyn : {0, 1024, 2048, 2560, 4096, 8192, 16384, 32768})
    {
        int n = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, dyn);
        printf("static 24576 + dynamic %5d -> %d blocks/CU\n", dyn, n);
    }
    return 0;
}
