// Issue rate / dependent latency of the f32 MFMAs on gfx950 (cycles per instruction seen by one wave, and per SIMD with 1, 2, 4
// waves per SIMD).  hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: 16x16x4 one dependent chain, 1: 16x16x4 four independent chains, 2: 32x32x2 one chain, 3: 32x32x2 two chains, 4: 4x4x1? no
__global__ void k(float *out, unsigned long long *cyc, float a0)
{
    f4 c[4]; f16v d[2];
    for (int i = 0; i < 4; ++i) c[i] = f4{a0, a0, a0, a0};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) d[i][j] = a0;
    const float a = a0 * 0.5f + threadIdx.x * 1e-6f, b = a0 * 0.25f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 2048; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) c[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[0], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) c[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i & 3], 0, 0, 0);
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) d[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d[0], 0, 0, 0);
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) d[i & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d[i & 1], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0; for (int i = 0; i < 4; ++i) r += c[i][0] + c[i][3]; for (int i = 0; i < 2; ++i) r += d[i][0] + d[i][15];
    out[threadIdx.x + blockIdx.x * blockDim.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
    float *out; unsigned long long *cyc, h[2];
    (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 16);
    const char *names[4] = {"16x16x4f32, one dependent chain", "16x16x4f32, four chains", "32x32x2f32, one dependent chain", "32x32x2f32, two chains"};
    for (int waves = 4; waves <= 16; waves *= 2) {
        for (int m = 0; m < 4; ++m) {
            float ms = 0; hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                switch (m) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0f); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0f); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0f); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0f); break;
                }
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
            }
            (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
            const double n = 2048.0 * 16;
            printf("waves/SIMD %d  %-34s per wave: %.1f ticks/MFMA   per SIMD: %.1f ticks/MFMA (kernel %.3f ms -> %.2f ns/MFMA/SIMD)\n", waves / 4, names[m],
                   (double)h[0] / n, (double)h[0] / n / (waves / 4), ms, ms * 1e6 / n / (waves / 4));
        }
    }
    return 0;
}
