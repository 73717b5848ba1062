// tw_engine_generic.hpp -- EngineV: the policy-forward engine for ANY Sequential stack the reference's Policy can hold
// (rust/src/nn/modules.rs:28-34 runs any list of Linear layers; src/twisterl/nn/utils.py:17-42 exports any): N common
// layers, non-empty policy_layers / value_layers, widths up to 512, any obs_size -- the shapes the MFMA engines of
// tw_engine.hpp (embedding -> one common Linear of 32..256 units -> linear heads: both Puzzle configs) do not cover.
//
// Same interface as Engine3T (16 episode columns per 256-thread workgroup, lane = (h = lane >> 4, j = lane & 15), all four
// waves carry every column's state), so the rollout / self-play / solve kernels instantiate it unchanged.  Arithmetic = the
// numeric spec of DESIGN.md: EmbeddingBag = bias + rows in cell order (plain adds), every Linear a k-ordered fma chain from
// 0 with the bias added last -- on the vector ALU, which on gfx950 has the same f32 rate as the f32 MFMA (both use the
// SIMD's FMA lanes): thread (column c, group g of 16) computes output quads g, g+16, .. of a layer for its column with one
// float4 weight load (wave-broadcast, L1/L2 resident) and four v_fma_f32 per k.  Activations ping-pong through LDS as
// [unit][column].  Bit-equal to the oracle's TWO_ARITH_CHAIN forward.
#pragma once
#include "tw_common.hpp"

namespace tw {

constexpr int GEN_MAX_WIDTH = 512;        // widest layer (and embedding) the engine's LDS buffers hold
constexpr int GEN_COLS = 16;

template <int NC>
struct EngineV {
    static constexpr int NW = 4, THREADS = 256, EPB = GEN_COLS, NS = 4;
    static constexpr bool SPLIT = true;

    PolicyDev pol;
    int tid, lane, wave, j, h, g;
    float *lds0, *lds_out, *lds_user;
    const uint8_t *perm_obs, *perm_act;
#ifdef TW_ABLATE
    unsigned long long stq[6] = {0, 0, 0, 0, 0, 0};      // (cycle stamps of the diagnostic build: none inside this engine)
#endif

    // three activation buffers (the common output stays put while the two heads run) | head outputs [16][8] | kernel use
    __host__ __device__ static size_t lds_floats(int) { return (size_t)3 * GEN_MAX_WIDTH * GEN_COLS + GEN_COLS * 8 + 256; }
    __device__ __forceinline__ bool primary() const { return wave == 0; }
    __device__ __forceinline__ int  ep_lane() const { return j; }
    __device__ __forceinline__ bool owns_lane() const { return wave == j / (EPB / NS); }

    __device__ __forceinline__ void begin1(const PolicyDev &p, float *lds)
    {
        pol = p;
        tid = threadIdx.x; lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        j = lane & 15; h = lane >> 4; g = tid >> 4;
        lds0 = lds;
        lds_out = lds + (size_t)3 * GEN_MAX_WIDTH * GEN_COLS;
        lds_user = lds_out + GEN_COLS * 8;
        perm_obs = pol.obs_perms; perm_act = pol.act_perms;
    }
    __device__ __forceinline__ void begin2() {}
    __device__ __forceinline__ void end() {}

    __device__ __forceinline__ float *bufp(int i) const { return lds0 + (size_t)i * (GEN_MAX_WIDTH * GEN_COLS); }

    // obs ids of the board's cells (after the twist); -1 for cells the board does not have
    __device__ __forceinline__ void rows_of(uint64_t board, int n_cells, int perm, int (&rowoff)[NC]) const
    {
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            int row = -1;
            if (i < n_cells) {
                const int id = i * n_cells + (int)nib(board, i);
                row = perm >= 0 ? (int)perm_obs[perm * pol.obs_size + id] : id;
            }
            rowoff[i] = row;
        }
    }

    __device__ __forceinline__ void act_perm(int perm, float (&lg)[4]) const
    {
        if (perm < 0) return;
        const float l0 = lg[0], l1 = lg[1], l2 = lg[2], l3 = lg[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int src = perm_act[perm * 4 + i];
            lg[i] = src == 0 ? l0 : (src == 1 ? l1 : (src == 2 ? l2 : l3));
        }
    }

    // one Linear (layers.rs:31-37) for the 16 columns: x [in][16] -> y [out][16]
    __device__ __forceinline__ void layer(const LayerDev &L, const float *x, float *y) const
    {
        const int in = L.in, out = L.out;
        if ((out & 3) == 0) {
            for (int q = g; q < out / 4; q += 16) {
                float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
                const float *wp = L.w + 4 * q;
#pragma unroll 4
                for (int k = 0; k < in; ++k) {
                    const float4 w = *reinterpret_cast<const float4 *>(wp + (size_t)k * out);
                    const float xv = x[k * GEN_COLS + j];
                    a0 = __builtin_fmaf(w.x, xv, a0); a1 = __builtin_fmaf(w.y, xv, a1);
                    a2 = __builtin_fmaf(w.z, xv, a2); a3 = __builtin_fmaf(w.w, xv, a3);
                }
                const float4 b = *reinterpret_cast<const float4 *>(L.b + 4 * q);
                a0 = a0 + b.x; a1 = a1 + b.y; a2 = a2 + b.z; a3 = a3 + b.w;
                if (L.relu) {         // layers.rs:89-91: `if x > 0.0 { x } else { 0.0 }`
                    a0 = a0 > 0.0f ? a0 : 0.0f; a1 = a1 > 0.0f ? a1 : 0.0f; a2 = a2 > 0.0f ? a2 : 0.0f; a3 = a3 > 0.0f ? a3 : 0.0f;
                }
                y[(4 * q + 0) * GEN_COLS + j] = a0; y[(4 * q + 1) * GEN_COLS + j] = a1;
                y[(4 * q + 2) * GEN_COLS + j] = a2; y[(4 * q + 3) * GEN_COLS + j] = a3;
            }
        } else {
            for (int o = g; o < out; o += 16) {
                float a = 0.0f;
                for (int k = 0; k < in; ++k) a = __builtin_fmaf(L.w[(size_t)k * out + o], x[k * GEN_COLS + j], a);
                a = a + L.b[o];
                y[o * GEN_COLS + j] = L.relu ? (a > 0.0f ? a : 0.0f) : a;
            }
        }
        __syncthreads();
    }

    // runs a stack from buffer `src`; returns the buffer holding its output (never `keep`)
    __device__ __forceinline__ int stack(const LayerDev *ls, int n, int src, int keep)
    {
        int cur = src;
        for (int l = 0; l < n; ++l) {
            int dst = 0;
            while (dst == cur || dst == keep) ++dst;
            layer(ls[l], bufp(cur), bufp(dst));
            cur = dst;
        }
        return cur;
    }

    __device__ __forceinline__ void forward(const int (&rowoff)[NC], float (&lg)[4], float &value)
    {
        // EmbeddingBag (layers.rs:56-62,82-84): bias + the rows of the cells, in cell order
        const int E = pol.emb;
        const float *tab = pol.emb_rows;
        const float *bias = tab + (size_t)pol.obs_size * E;
        for (int q = g; q < E / 4; q += 16) {
            float4 a = *reinterpret_cast<const float4 *>(bias + 4 * q);
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                if (rowoff[i] >= 0) {
                    const float4 r = *reinterpret_cast<const float4 *>(tab + (size_t)rowoff[i] * E + 4 * q);
                    a.x = a.x + r.x; a.y = a.y + r.y; a.z = a.z + r.z; a.w = a.w + r.w;
                }
            }
            if (pol.emb_relu) { a.x = a.x > 0.0f ? a.x : 0.0f; a.y = a.y > 0.0f ? a.y : 0.0f; a.z = a.z > 0.0f ? a.z : 0.0f; a.w = a.w > 0.0f ? a.w : 0.0f; }
            float *y = bufp(0);
            y[(4 * q + 0) * GEN_COLS + j] = a.x; y[(4 * q + 1) * GEN_COLS + j] = a.y;
            y[(4 * q + 2) * GEN_COLS + j] = a.z; y[(4 * q + 3) * GEN_COLS + j] = a.w;
        }
        __syncthreads();
        const LayerDev *ls = pol.layers;
        const int co = stack(ls, pol.n_common, 0, -1);                                         // policy.rs:86
        const int vo = stack(ls + pol.n_common + pol.n_action, pol.n_value, co, co);           // policy.rs:89
        if (g == 0) {                                                                          // .sum() of the value head's outputs
            float s = 0.0f;
            for (int i = 0; i < pol.value_out; ++i) s = s + bufp(vo)[i * GEN_COLS + j];
            lds_out[j * 8 + 4] = s;
        }
        __syncthreads();
        const int ao = stack(ls + pol.n_common, pol.n_action, co, co);                         // policy.rs:92
        if (g == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) lds_out[j * 8 + i] = bufp(ao)[i * GEN_COLS + j];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) lg[i] = lds_out[j * 8 + i];
        value = lds_out[j * 8 + 4];
        __syncthreads();           // (the buffers and lds_out are rewritten by the next forward)
    }
};

}  // namespace tw
