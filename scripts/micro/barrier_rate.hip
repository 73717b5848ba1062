// Cost of one workgroup barrier per loop trip (with a small LDS read in the trip) for 4, 8 and 16 waves per workgroup, on one CU
// and with every CU busy.  hipcc --offload-arch=gfx950 -O3 scripts/micro/barrier_rate.hip -o /tmp/barrier_rate && /tmp/barrier_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void k(float *out, unsigned long long *cyc, int iters)
{
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    float a = 0.0f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 1) a += lds[(threadIdx.x * 4 + it) & 4095];
        if (MODE >= 2) lds[(threadIdx.x + it * 64) & 4095] = a;
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x + blockIdx.x * blockDim.x] = a;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
    float *out; unsigned long long *cyc, h[2];
    (void)hipMalloc(&out, 64 << 20); (void)hipMalloc(&cyc, 16);
    const int iters = 4096;
    for (int blocks = 1; blocks <= 256; blocks *= 256)
        for (int lds_kb = 16; lds_kb <= 128; lds_kb *= 8)
            for (int waves = 4; waves <= 16; waves *= 2)
                for (int m = 0; m < 3; ++m) {
                    const size_t lb = (size_t)lds_kb * 1024;
                    switch (m) {
                        case 0: (void)hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb); hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64 * waves), lb, 0, out, cyc, iters); break;
                        case 1: (void)hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb); hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64 * waves), lb, 0, out, cyc, iters); break;
                        case 2: (void)hipFuncSetAttribute((const void *)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb); hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64 * waves), lb, 0, out, cyc, iters); break;
                    }
                    (void)hipDeviceSynchronize();
                    (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
                    printf("blocks %3d  LDS %3d KB  waves %2d  %-22s %.1f cycles per trip\n", blocks, lds_kb, waves,
                           m == 0 ? "barrier only" : m == 1 ? "LDS read + barrier" : "read + write + barrier", (double)h[0] / iters);
                }
    return 0;
}
