// How long is one hop of a wave-uniform pointer chase through LDS (ds_read_b32 -> v_readfirstlane -> scalar compare + branch -> next
// address), the inner loop of the self-play walker's descent?  One wave chases, in a workgroup of four: the others (a) have left,
// (b) wait at the workgroup barrier, (c) spin on an LDS flag.   hipcc --offload-arch=gfx950 -O3 lds_chase_probe.hip -o lds_chase_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) uint32_t lds_u32;
template <bool VEC>
__global__ void __launch_bounds__(256) chase(int mode, int hops, int extra_salu, unsigned long long *out)
{
    extern __shared__ uint32_t lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    volatile lds_u32 *flag = (volatile lds_u32 *)(lds + 3000);        // (inside the smallest allocation)
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = (uint32_t)((i * 1237 + 17) & 2047);
    if (threadIdx.x == 0) *flag = 0;
    __syncthreads();
    if (wave == 0) {
        uint32_t p = 0, acc = 0;
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int h = 0; h < hops; ++h) {
            p = (uint32_t)__builtin_amdgcn_readfirstlane((int)((lds_u32 *)lds)[p]);
            if (VEC) asm volatile("" : "+v"(acc));                                       // (the same chain on the vector ALU: the compiler no longer knows it is uniform)
            for (int e = 0; e < extra_salu; ++e) acc = (acc >> 3) ^ (acc * 5u + p);      // dependent work beside the chase: shift, multiply, add, xor
            if (p == 0xffffffffu) break;
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        if (lane == 0) { out[0] = t1 - t0; out[1] = p + acc; }
        if (mode == 2) { *flag = 1; }
    } else {
        if (mode == 0) return;
        if (mode == 2) { for (int spin = 0; spin < (1 << 22) && *flag == 0; ++spin) __builtin_amdgcn_s_sleep(2); }     // (bounded)
    }
    if (mode == 1) __syncthreads();
}
int main()
{
    unsigned long long *d, h[2];
    hipMalloc(&d, 16);
    hipFuncSetAttribute((const void *)chase<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    const char *names[3] = {"others left", "others at s_barrier", "others spin on an LDS flag (s_sleep 2)"};
    for (int lds_kb : {16, 130})
        for (int mode = 0; mode < 3; ++mode)
            for (int extra : {0, 8, 24}) {
                for (int rep = 0; rep < 2; ++rep) {
                    hipLaunchKernelGGL(chase<false>, dim3(1), dim3(256), lds_kb * 1024, 0, mode, 2000, extra, d);
                    hipDeviceSynchronize();
                }
                hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
                fflush(stdout); printf("LDS %3d KB, %-40s extra scalar ops per hop %2d: %.1f cycles per hop\n", lds_kb, names[mode], extra, (double)h[0] / 2000.0);
            }
    for (int extra : {0, 8, 24}) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(chase<true>, dim3(1), dim3(256), 16 * 1024, 0, 1, 2000, extra, d); hipDeviceSynchronize(); }
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("the chain on the VECTOR ALU (others at s_barrier), extra ops per hop %2d: %.1f cycles per hop\n", extra, (double)h[0] / 2000.0);
    }
    return 0;
}
