// Cost of a hand-off between workgroups on DIFFERENT compute units through L2 -- the step a "team" forward would need (VERDICT
// r02 item 5: four CUs share 16 episodes, each keeps a slice of W1 resident in LDS and they exchange the hidden units twice per
// time step).  One workgroup per CU (100 KB of LDS each), teams of four; per exchange every member stores its 4 KB slice, releases
// a flag (agent scope), waits for the flags of the other three (acquire), and reads their slices (12 KB) into LDS.  Two team
// formations: the four members on ONE XCD (workgroup ids congruent mod 8 land on the same XCD) and on four different XCDs.
// Every wait is bounded (a member that never shows up ends the run with an error flag, nobody spins for ever).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/xcu_handoff.hip -o /tmp/xcu_handoff && /tmp/xcu_handoff
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int TEAM = 4, SLICE_FLOATS = 1024;          // 16 columns x 64 hidden units

__global__ void __launch_bounds__(256) k(float *xbuf, unsigned *flags, unsigned long long *out, unsigned *err, int iters, int same_xcd, int n_groups, int payload)
{
    extern __shared__ float lds[];
    // team of this workgroup and its rank in it
    int team, rank;
    if (same_xcd) { const int x = blockIdx.x & 7, q = blockIdx.x >> 3; team = (q >> 2) * 8 + x; rank = q & 3; }
    else { team = blockIdx.x >> 2; rank = blockIdx.x & 3; }
    if (team >= n_groups) return;
    float *tx = xbuf + (size_t)team * 2 * TEAM * SLICE_FLOATS;          // two exchange buffers (ping-pong between exchanges)
    unsigned *tf = flags + (size_t)team * TEAM * 32;                     // one flag per member, 128 bytes apart
    float acc = (float)threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (int it = 1; it <= iters; ++it) {
        float *buf = tx + (size_t)(it & 1) * TEAM * SLICE_FLOATS;
        if (payload) for (int i = threadIdx.x; i < SLICE_FLOATS; i += 256) buf[rank * SLICE_FLOATS + i] = acc + (float)i;
        __syncthreads();                                                  // (the whole slice is written ...)
        if (threadIdx.x == 0) __hip_atomic_store(tf + rank * 32, (unsigned)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // ... before the flag goes up
        if (threadIdx.x < TEAM) {
            int spins = 0;
            while (__hip_atomic_load(tf + threadIdx.x * 32, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) {
                if (++spins > (1 << 22)) { atomicExch(err, 1u); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (*(volatile unsigned *)err) break;
        if (payload) {
            for (int m = 0; m < TEAM; ++m) {
                if (m == rank) continue;
                for (int i = threadIdx.x; i < SLICE_FLOATS; i += 256) lds[m * SLICE_FLOATS + i] = __builtin_nontemporal_load(buf + m * SLICE_FLOATS + i);
            }
            __syncthreads();
            acc += lds[((rank + 1) & 3) * SLICE_FLOATS + threadIdx.x];
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; xbuf[(size_t)n_groups * 2 * TEAM * SLICE_FLOATS + blockIdx.x] = acc; }
}

int main()
{
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, groups = cus / TEAM;
    int wc_khz = 0; (void)hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0);
    float *xbuf; unsigned *flags, *err; unsigned long long *out;
    (void)hipMalloc(&xbuf, ((size_t)groups * 2 * TEAM * SLICE_FLOATS + cus) * 4); (void)hipMalloc(&flags, (size_t)groups * TEAM * 128);
    (void)hipMalloc(&err, 4); (void)hipMalloc(&out, (size_t)cus * 8);
    const size_t lds = 100 * 1024;
    (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int iters = 4000;
    for (int same = 1; same >= 0; --same)
        for (int payload = 0; payload <= 1; ++payload)
            for (int g = 1; g <= groups; g *= groups) {
                (void)hipMemset(flags, 0, (size_t)groups * TEAM * 128); (void)hipMemset(err, 0, 4);
                hipLaunchKernelGGL(k, dim3(cus), dim3(256), lds, 0, xbuf, flags, out, err, iters, same, g, payload);
                if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
                unsigned long long h[1024]; unsigned e = 0;
                (void)hipMemcpy(h, out, (size_t)cus * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
                const int b0 = same ? 0 : 0;
                printf("%-14s teams %3d  %-22s %s %.2f us per exchange (wall clock %d kHz)\n", same ? "one XCD" : "four XCDs", g,
                       payload ? "4 KB out + 12 KB in" : "flags only", e ? "TIMED OUT" : "", (double)h[b0] / iters / (wc_khz * 1e-3), wc_khz);
            }
    return 0;
}
