// Probe: how does v_mfma_f32_32x32x16_f16 round?  Compares one MFMA on random operands with candidate models.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(const _Float16 *A, const _Float16 *B, const float *C, float *D)
{   // A[32][16], B[16][32], C/D[32][32] row-major
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    h8 a, b; f16v c;
    for (int j = 0; j < 8; ++j) { a[j] = A[r * 16 + 8 * h + j]; b[j] = B[(8 * h + j) * 32 + r]; }
    for (int g = 0; g < 16; ++g) c[g] = C[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r];
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int g = 0; g < 16; ++g) D[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r] = c[g];
}
static float f32(double x) { return (float)x; }
int main()
{
    srand(1);
    std::vector<_Float16> A(32 * 16), B(16 * 32); std::vector<float> C(1024), D(1024);
    int trials = 64, bad[8] = {0};
    _Float16 *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, 4096); hipMalloc(&dD, 4096);
    for (int t = 0; t < trials; ++t) {
        const float scaleC = (t & 1) ? 1.0f : 64.0f;
        for (auto &x : A) x = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.0f);
        for (auto &x : B) x = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.0f);
        for (auto &x : C) x = (rand() / (float)RAND_MAX - 0.5f) * scaleC;
        hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double p[16];
            for (int kk = 0; kk < 16; ++kk) p[kk] = (double)(float)A[i * 16 + kk] * (double)(float)B[kk * 32 + j];
            const float c = C[i * 32 + j], d = D[i * 32 + j];
            float m[8];
            auto two = [&](const int (&g0)[8], const int (&g1)[8]) { double s0 = 0, s1 = 0; for (int q = 0; q < 8; ++q) { s0 += p[g0[q]]; s1 += p[g1[q]]; } return f32((double)f32((double)c + s0) + s1); };
            const int lo8[8] = {0,1,2,3,4,5,6,7}, hi8[8] = {8,9,10,11,12,13,14,15};
            const int ev[8] = {0,2,4,6,8,10,12,14}, od[8] = {1,3,5,7,9,11,13,15};
            const int qa[8] = {0,1,2,3,8,9,10,11}, qb[8] = {4,5,6,7,12,13,14,15};
            const int pa[8] = {0,1,4,5,8,9,12,13}, pb[8] = {2,3,6,7,10,11,14,15};
            m[0] = two(lo8, hi8); m[1] = two(hi8, lo8); m[2] = two(ev, od); m[3] = two(od, ev); m[4] = two(qa, qb); m[5] = two(qb, qa);
            m[6] = two(pa, pb); m[7] = two(pb, pa);
            for (int z = 0; z < 8; ++z) if (memcmp(&m[z], &d, 4) != 0) bad[z]++;
        }
    }
    const char *names[8] = {"lo8,hi8", "hi8,lo8", "even,odd", "odd,even", "{0-3,8-11},{4-7,12-15}", "reverse", "{0,1,4,5,..},{2,3,6,7,..}", "reverse"};
    for (int z = 0; z < 8; ++z) printf("model %-24s mismatches %d of %d\n", names[z], bad[z], trials * 1024);
    return 0;
}
