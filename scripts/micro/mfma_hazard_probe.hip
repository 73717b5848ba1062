// mfma_hazard_probe.hip -- does the hardware interlock an MFMA result against the next VALU read, and what does an
// independent MFMA in between stand for?  (Ground truth for scripts/scan_mfma_hazards.py: the inline-asm MFMAs of the engines get
// no padding from hipcc, and the scanner must neither miss a real hazard nor demand nops the matrix pipe already provides.)
//
// Every lane runs:   D = sentinel;  D = MFMA(a, b, D);  [M independent MFMAs of the same kind];  [s_nop K];  out = D[first], D[last]
// entirely inside ONE asm statement (physical registers, so nothing is padded or reordered by the compiler).  A lane is WRONG when
// `out` still holds the sentinel-only value instead of sentinel + a.b.  Printed: wrong lanes out of 64 x blocks for K = none, 0..15
// and M = 0, 1, 2, for the three MFMA kinds the engines issue from inline asm.
//
//   hipcc -O2 --offload-arch=gfx950 scripts/micro/mfma_hazard_probe.hip -o /tmp/mfma_hazard_probe && /tmp/mfma_hazard_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

#define INIT16 "v_mov_b32 v100, %4\n v_mov_b32 v101, %4\n v_mov_b32 v102, %4\n v_mov_b32 v103, %4\n v_mov_b32 v104, %4\n v_mov_b32 v105, %4\n v_mov_b32 v106, %4\n v_mov_b32 v107, %4\n" \
               "v_mov_b32 v108, %4\n v_mov_b32 v109, %4\n v_mov_b32 v110, %4\n v_mov_b32 v111, %4\n v_mov_b32 v112, %4\n v_mov_b32 v113, %4\n v_mov_b32 v114, %4\n v_mov_b32 v115, %4\n" \
               "v_mov_b32 v116, %4\n v_mov_b32 v117, %4\n v_mov_b32 v118, %4\n v_mov_b32 v119, %4\n v_mov_b32 v120, %4\n v_mov_b32 v121, %4\n v_mov_b32 v122, %4\n v_mov_b32 v123, %4\n" \
               "v_mov_b32 v124, %4\n v_mov_b32 v125, %4\n v_mov_b32 v126, %4\n v_mov_b32 v127, %4\n v_mov_b32 v128, %4\n v_mov_b32 v129, %4\n v_mov_b32 v130, %4\n v_mov_b32 v131, %4\n" \
               "s_nop 15\n s_nop 15\n"
#define CLOB "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115", \
             "v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131"
#define TAIL(last) "v_mov_b32 %0, v100\n v_mov_b32 %1, " last "\n s_nop 15\n s_nop 15\n s_nop 15\n"

// KIND 0: v_mfma_f32_32x32x2_f32 (16 passes)   1: v_mfma_f32_16x16x4_f32 (8 passes)   2: v_mfma_f32_32x32x16_f16 (8 passes)
template <int KIND, int M, int K>
__global__ void probe(const float *a, const float *b, float *out, float sentinel)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    float o0, o1;
    if constexpr (KIND == 0) {
        const float av = a[t], bv = b[t];
#define P0(MID, NOP) asm volatile(INIT16 "v_mfma_f32_32x32x2_f32 v[100:115], %2, %3, v[100:115]\n" MID NOP TAIL("v115") : "=&v"(o0), "=&v"(o1) : "v"(av), "v"(bv), "v"(sentinel), "n"(K < 0 ? 0 : K) : CLOB)
#define MID0 "v_mfma_f32_32x32x2_f32 v[116:131], %2, %3, v[116:131]\n"
        if constexpr (M == 0 && K < 0) P0("", ""); else if constexpr (M == 0) P0("", "s_nop %5\n");
        else if constexpr (M == 1 && K < 0) P0(MID0, ""); else if constexpr (M == 1) P0(MID0, "s_nop %5\n");
        else if constexpr (K < 0) P0(MID0 MID0, ""); else P0(MID0 MID0, "s_nop %5\n");
    } else if constexpr (KIND == 1) {
        const float av = a[t], bv = b[t];
#define P1(MID, NOP) asm volatile(INIT16 "v_mfma_f32_16x16x4_f32 v[100:103], %2, %3, v[100:103]\n" MID NOP TAIL("v103") : "=&v"(o0), "=&v"(o1) : "v"(av), "v"(bv), "v"(sentinel), "n"(K < 0 ? 0 : K) : CLOB)
#define MID1 "v_mfma_f32_16x16x4_f32 v[116:119], %2, %3, v[116:119]\n"
        if constexpr (M == 0 && K < 0) P1("", ""); else if constexpr (M == 0) P1("", "s_nop %5\n");
        else if constexpr (M == 1 && K < 0) P1(MID1, ""); else if constexpr (M == 1) P1(MID1, "s_nop %5\n");
        else if constexpr (K < 0) P1(MID1 MID1, ""); else P1(MID1 MID1, "s_nop %5\n");
    } else {
        h16x8 av, bv;
        for (int i = 0; i < 8; ++i) { av[i] = (_Float16)a[t]; bv[i] = (_Float16)b[t]; }
#define P2(MID, NOP) asm volatile(INIT16 "v_mfma_f32_32x32x16_f16 v[100:115], %2, %3, v[100:115]\n" MID NOP TAIL("v115") : "=&v"(o0), "=&v"(o1) : "v"(av), "v"(bv), "v"(sentinel), "n"(K < 0 ? 0 : K) : CLOB)
#define MID2 "v_mfma_f32_32x32x16_f16 v[116:131], %2, %3, v[116:131]\n"
        if constexpr (M == 0 && K < 0) P2("", ""); else if constexpr (M == 0) P2("", "s_nop %5\n");
        else if constexpr (M == 1 && K < 0) P2(MID2, ""); else if constexpr (M == 1) P2(MID2, "s_nop %5\n");
        else if constexpr (K < 0) P2(MID2 MID2, ""); else P2(MID2 MID2, "s_nop %5\n");
    }
    out[2 * t] = o0; out[2 * t + 1] = o1;
}

static const int BLOCKS = 1024;
static float *d_a, *d_b, *d_out;
static std::vector<float> h_out(2 * 64 * BLOCKS);

template <int KIND, int M, int K> static void run(int *wrong_first, int *wrong_last)
{
    const float sentinel = 1024.0f;                      // a.b is a sum of products of 1..2: never 0, so "still the sentinel" = stale
    hipMemset(d_out, 0, h_out.size() * 4);
    hipLaunchKernelGGL((probe<KIND, M, K>), dim3(BLOCKS), dim3(64), 0, 0, d_a, d_b, d_out, sentinel);
    hipMemcpy(h_out.data(), d_out, h_out.size() * 4, hipMemcpyDeviceToHost);
    int w0 = 0, w1 = 0;
    for (int t = 0; t < 64 * BLOCKS; ++t) { w0 += h_out[2 * t] == sentinel; w1 += h_out[2 * t + 1] == sentinel; }
    *wrong_first = w0; *wrong_last = w1;
}

template <int KIND, int M> static void sweep(const char *name)
{
    int w0[17], w1[17];
    run<KIND, M, -1>(&w0[0], &w1[0]);
    run<KIND, M, 0>(&w0[1], &w1[1]);   run<KIND, M, 1>(&w0[2], &w1[2]);   run<KIND, M, 2>(&w0[3], &w1[3]);   run<KIND, M, 3>(&w0[4], &w1[4]);
    run<KIND, M, 4>(&w0[5], &w1[5]);   run<KIND, M, 5>(&w0[6], &w1[6]);   run<KIND, M, 6>(&w0[7], &w1[7]);   run<KIND, M, 7>(&w0[8], &w1[8]);
    run<KIND, M, 8>(&w0[9], &w1[9]);   run<KIND, M, 9>(&w0[10], &w1[10]); run<KIND, M, 10>(&w0[11], &w1[11]); run<KIND, M, 11>(&w0[12], &w1[12]);
    run<KIND, M, 12>(&w0[13], &w1[13]); run<KIND, M, 13>(&w0[14], &w1[14]); run<KIND, M, 14>(&w0[15], &w1[15]); run<KIND, M, 15>(&w0[16], &w1[16]);
    printf("%-26s %d independent MFMA(s) between | wait states 0..16 -> stale lanes of %d (first register / last register):\n   ", name, M, 64 * BLOCKS);
    for (int k = 0; k < 17; ++k) printf(" %d:%d/%d", k, w0[k], w1[k]);
    printf("\n");
}

int main()
{
    std::vector<float> a(64 * BLOCKS), b(64 * BLOCKS);
    for (size_t i = 0; i < a.size(); ++i) { a[i] = 1.0f + (float)(i % 7) / 8.0f; b[i] = 1.0f + (float)(i % 5) / 4.0f; }
    hipMalloc(&d_a, a.size() * 4); hipMalloc(&d_b, b.size() * 4); hipMalloc(&d_out, h_out.size() * 4);
    hipMemcpy(d_a, a.data(), a.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_b, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    sweep<0, 0>("v_mfma_f32_32x32x2_f32"); sweep<0, 1>("v_mfma_f32_32x32x2_f32"); sweep<0, 2>("v_mfma_f32_32x32x2_f32");
    sweep<1, 0>("v_mfma_f32_16x16x4_f32"); sweep<1, 1>("v_mfma_f32_16x16x4_f32"); sweep<1, 2>("v_mfma_f32_16x16x4_f32");
    sweep<2, 0>("v_mfma_f32_32x32x16_f16"); sweep<2, 1>("v_mfma_f32_32x32x16_f16"); sweep<2, 2>("v_mfma_f32_32x32x16_f16");
    return 0;
}
