#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
// MFMA D -> VALU read
extern "C" __global__ void k_f32_32x32x2_valu(float *o, float a, float b) {
    f32x16 c = {}; c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    o[threadIdx.x] = c[0] + 1.0f;
}
extern "C" __global__ void k_f32_16x16x4_valu(float *o, float a, float b) {
    f32x4 c = {}; c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    o[threadIdx.x] = c[0] + 1.0f;
}
extern "C" __global__ void k_f16_32x32x16_valu(float *o, h16x8 a, h16x8 b) {
    f32x16 c = {}; c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    o[threadIdx.x] = c[0] + 1.0f;
}
// MFMA D -> store (VMEM read of D)
extern "C" __global__ void k_f32_16x16x4_store(f32x4 *o, float a, float b) {
    f32x4 c = {}; c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    o[threadIdx.x] = c;
}
// VALU write -> MFMA A
extern "C" __global__ void k_valu_f32_16x16x4(f32x4 *o, float a, float b, f32x4 c) {
    float a2 = a * (float)threadIdx.x;
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b, c, 0, 0, 0);
    o[threadIdx.x] = c;
}
extern "C" __global__ void k_valu_f16_32x32x16(f32x16 *o, h16x8 a, h16x8 b, f32x16 c) {
    h16x8 a2 = a * (_Float16)threadIdx.x;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b, c, 0, 0, 0);
    o[threadIdx.x] = c;
}
// VALU write -> MFMA C
extern "C" __global__ void k_valuC_f32_16x16x4(f32x4 *o, float a, float b, f32x4 c) {
    c = c * (float)threadIdx.x;
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    o[threadIdx.x] = c;
}
// MFMA D -> MFMA A/B
extern "C" __global__ void k_f32_16x16x4_to_AB(f32x4 *o, float a, float b, f32x4 c) {
    f32x4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(d[0], b, c, 0, 0, 0);
    o[threadIdx.x] = c;
}
extern "C" __global__ void k_f32_32x32x2_to_AB(f32x16 *o, float a, float b, f32x16 c) {
    f32x16 d = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(d[0], b, c, 0, 0, 0);
    o[threadIdx.x] = c;
}
// MFMA D -> MFMA C overlapped, different size: 32x32 D (16 regs), then 16x16 with C = first 4 regs
extern "C" __global__ void k_f32_32_to_C16(f32x4 *o, float a, float b, f32x16 c) {
    f32x16 d = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    f32x4 c4 = {d[0], d[1], d[2], d[3]};
    c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c4, 0, 0, 0);
    o[threadIdx.x] = c4;
}
extern "C" __global__ void k_f16_32_to_C16(f32x4 *o, h16x8 a, h16x8 b, f32x16 c) {
    f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    f32x4 c4 = {d[0], d[1], d[2], d[3]};
    c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c4, 0, 0, 0);
    o[threadIdx.x] = c4;
}
// same-C chain
extern "C" __global__ void k_f32_16x16x4_chain(f32x4 *o, float a, float b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c, 0, 0, 0);
    o[threadIdx.x] = c;
}
// MFMA D -> VALU write (WAW)
extern "C" __global__ void k_f32_16x16x4_waw(f32x4 *o, float a, float b, f32x4 c, float *p) {
    f32x4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    p[threadIdx.x] = d[1];
    o[threadIdx.x] = c * 2.0f;
}
