// Probe: is v_mfma_f32_16x16x4_f32 a k-ordered fma chain, like v_mfma_f32_32x32x2_f32?  (It would let the exact engine run
// 16 episodes per wave group: twice the workgroups for tiny batches.)  One MFMA on random operands vs candidate models.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *A, const float *B, const float *C, float *D)
{   // A[16][4], B[4][16], C/D[16][16] row-major.  Lane l: A[i = l%16][k = l/16], B[k = l/16][j = l%16], D[4*(l/16)+r][l%16]
    const int l = threadIdx.x, i = l & 15, q = l >> 4;
    f4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[(4 * q + r) * 16 + i];
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * 4 + q], B[q * 16 + i], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * q + r) * 16 + i] = c[r];
}
int main()
{
    srand(3);
    std::vector<float> A(64), B(64), C(256), D(256);
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
    const int trials = 256;
    int bad[5] = {0, 0, 0, 0, 0};
    for (int t = 0; t < trials; ++t) {
        const float sc = (t & 1) ? 1.0f : 37.0f;
        for (auto &x : A) x = (rand() / (float)RAND_MAX - 0.5f) * 2.0f;
        for (auto &x : B) x = (rand() / (float)RAND_MAX - 0.5f) * 2.0f;
        for (auto &x : C) x = (rand() / (float)RAND_MAX - 0.5f) * sc;
        hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            const float c = C[i * 16 + j], d = D[i * 16 + j];
            float a[4], b[4];
            for (int kk = 0; kk < 4; ++kk) { a[kk] = A[i * 4 + kk]; b[kk] = B[kk * 16 + j]; }
            float m[5];
            { float s = c; for (int kk = 0; kk < 4; ++kk) s = fmaf(a[kk], b[kk], s); m[0] = s; }                 // k-ordered fma chain
            { float s = c; for (int kk = 3; kk >= 0; --kk) s = fmaf(a[kk], b[kk], s); m[1] = s; }                // reversed chain
            { double s = c; for (int kk = 0; kk < 4; ++kk) s += (double)a[kk] * b[kk]; m[2] = (float)s; }        // one rounding
            { float s = c; for (int kk = 0; kk < 4; ++kk) s = s + a[kk] * b[kk]; m[3] = s; }                     // un-fused, ordered
            { double q = 0; for (int kk = 0; kk < 4; ++kk) q += (double)a[kk] * b[kk]; m[4] = c + (float)q; }    // round(dot) + c
            for (int z = 0; z < 5; ++z) if (memcmp(&m[z], &d, 4) != 0) bad[z]++;
        }
    }
    const char *names[5] = {"k-ordered fma chain", "reversed fma chain", "fused (one rounding)", "un-fused ordered", "round(dot)+c"};
    for (int z = 0; z < 5; ++z) printf("model %-22s mismatches %d of %d\n", names[z], bad[z], trials * 256);
    return 0;
}
