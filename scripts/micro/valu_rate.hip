// Issue rate of the f32 vector fma forms on gfx950, one wave on one SIMD: cycles per instruction for 16 independent chains.
// hipcc --offload-arch=gfx950 -O3 scripts/micro/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float *out, unsigned long long *cyc, float a0, float b0)
{
    f2 acc[8]; float s[16];
    for (int i = 0; i < 8; ++i) acc[i] = f2{a0 + i, a0 - i};
    for (int i = 0; i < 16; ++i) s[i] = a0 + i;
    f2 w = f2{b0, b0 + 1.0f}, x = f2{a0 * 0.5f, a0 * 0.25f};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long r0 = wall_clock64();
    for (int it = 0; it < 4096; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i]) : "v"(w.x), "v"(x.x));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(w), "v"(x));
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "v"(w), "v"(x));
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[i]) : "v"(w), "v"(x));
        } else if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(w), "v"(x));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = wall_clock64();
    float r = 0; for (int i = 0; i < 8; ++i) r += acc[i].x + acc[i].y; for (int i = 0; i < 16; ++i) r += s[i];
    out[threadIdx.x + blockIdx.x * blockDim.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}

int main()
{
    float *out; unsigned long long *cyc, h[2];
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 16);
    const char *names[5] = {"v_fma_f32 (16/iter)", "v_pk_fma_f32 (8/iter)", "v_pk_fma_f32 op_sel_hi broadcast lo", "v_pk_fma_f32 op_sel broadcast hi", "v_pk_mul_f32"};
    for (int waves = 1; waves <= 8; waves *= 2) {
        for (int m = 0; m < 5; ++m) {
            float ms = 0; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                switch (m) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0f, 1e-9f); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0f, 1e-9f); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0f, 1e-9f); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0f, 1e-9f); break;
                    case 4: hipLaunchKernelGGL(k<4>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0f, 1e-9f); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
            const int n = m == 0 ? 16 : 8;
            printf("waves/block %d  %-38s memtime ticks/instr %.3f  wall_clock ticks/instr %.4f  kernel %.3f ms -> ns/instr %.3f\n", waves, names[m],
                   (double)h[0] / (4096.0 * n), (double)h[1] / (4096.0 * n), ms, ms * 1e6 / (4096.0 * n));
        }
    }
    return 0;
}
