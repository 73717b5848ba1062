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
            { double s = c; for (int kk = 0; kk < 16; ++kk) s += p[kk]; m[0] = f32(s); }                                   // fused: one rounding
            { float s = c; for (int kk = 0; kk < 16; ++kk) s = f32((double)s + p[kk]); m[1] = s; }                        // sequential
            { double s0 = 0, s1 = 0; for (int kk = 0; kk < 8; ++kk) { s0 += p[kk]; s1 += p[8 + kk]; } m[2] = f32((double)f32((double)c + s0) + s1); }   // two halves, c first
            { float s = c; for (int g = 0; g < 4; ++g) { double q = 0; for (int kk = 0; kk < 4; ++kk) q += p[4 * g + kk]; s = f32((double)s + q); } m[3] = s; }   // groups of 4
            { double q = 0; for (int kk = 0; kk < 16; ++kk) q += p[kk]; m[4] = f32((double)c + (double)f32(q)); }       // dot rounded, then + c
            { double s0 = 0, s1 = 0; for (int kk = 0; kk < 8; ++kk) { s0 += p[kk]; s1 += p[8 + kk]; } m[5] = f32((double)c + (double)f32(s0) + (double)f32(s1)); }
            { float s = c; for (int g = 0; g < 8; ++g) { double q = p[2 * g] + p[2 * g + 1]; s = f32((double)s + q); } m[6] = s; }   // groups of 2
            { double q[4]; for (int g = 0; g < 4; ++g) { q[g] = 0; for (int kk = 0; kk < 4; ++kk) q[g] += p[4 * g + kk]; } m[7] = f32((double)c + (double)f32(f32(q[0] + q[1]) + (double)f32(q[2] + q[3]))); }
            for (int z = 0; z < 8; ++z) if (memcmp(&m[z], &d, 4) != 0) bad[z]++;
        }
    }
    const char *names[8] = {"fused (one rounding)", "sequential", "two halves", "groups of 4", "round(dot)+c", "c+round(h0)+round(h1)", "groups of 2", "tree"};
    for (int z = 0; z < 8; ++z) printf("model %-24s mismatches %d of %d\n", names[z], bad[z], trials * 1024);
    return 0;
}
