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
constexpr int GEN_PF = 16;                // weight quads a lane keeps in flight inside a layer

typedef float fx4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) fx4 gfx4;

template <int NC>
struct EngineV {
    static constexpr int NW = 4, THREADS = 256, EPB = GEN_COLS, NS = 4;
    static constexpr bool SPLIT = true;

    PolicyDev pol;
    int tid, lane, wave, j, h, g;
    float *lds0, *lds_out, *lds_user;
    const uint8_t *perm_obs, *perm_act;
#ifdef TW_ABLATE
    unsigned long long stq[6] = {0, 0, 0, 0, 0, 0};      // diagnostic build: cycles in embedding | common | value head | action head
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

    // one Linear (layers.rs:31-37) for the 16 columns: x [in][16] -> y [out][16].  Inside a layer lane = (column group cg =
    // lane >> 4, quad ql = lane & 15): a lane owns output quad 16*wave + ql (+64 per pass) for the FOUR columns 4cg..4cg+3 --
    // 16 independent k-ordered fma chains, two per v_pk_fma_f32, fed per k by ONE float4 weight load (the wave reads 256
    // contiguous bytes of the weight row) and ONE ds_read_b128 of the four columns' activations: the vector ALU, not the
    // memory pipe, is the limit.
    __device__ __forceinline__ void layer(const LayerDev &L, const float *x, float *y) const
    {
        typedef float f2 __attribute__((ext_vector_type(2)));
        const int in = L.in, out = L.out;       // (out: padded to a multiple of four by tw_policy_create, zero weights / bias)
        {
            const int nq = out >> 2, cg = lane >> 4, ql = lane & 15;
            for (int q = 16 * wave + ql; q < nq; q += 64) {           // (no barrier inside this loop)
                f2 acc[4][2];
#pragma unroll
                for (int c = 0; c < 4; ++c) { acc[c][0] = f2{0.0f, 0.0f}; acc[c][1] = f2{0.0f, 0.0f}; }
                // the weight pointer comes out of the layer table in memory: say it is global, or the loads are flat_load, which
                // count on lgkmcnt as well and every LDS wait then drains the whole weight prefetch
                const gfx4 *wp = (const gfx4 *)(L.w + 4 * q);
                const size_t wstride = (size_t)(out >> 2);      // (one k = out floats = out / 4 quads)
                const float *xp = x + 4 * cg;
                auto fma4 = [&](const fx4 w, const float4 xv) {
                    const f2 w01 = f2{w.x, w.y}, w23 = f2{w.z, w.w};
                    acc[0][0] = __builtin_elementwise_fma(w01, f2{xv.x, xv.x}, acc[0][0]); acc[0][1] = __builtin_elementwise_fma(w23, f2{xv.x, xv.x}, acc[0][1]);
                    acc[1][0] = __builtin_elementwise_fma(w01, f2{xv.y, xv.y}, acc[1][0]); acc[1][1] = __builtin_elementwise_fma(w23, f2{xv.y, xv.y}, acc[1][1]);
                    acc[2][0] = __builtin_elementwise_fma(w01, f2{xv.z, xv.z}, acc[2][0]); acc[2][1] = __builtin_elementwise_fma(w23, f2{xv.z, xv.z}, acc[2][1]);
                    acc[3][0] = __builtin_elementwise_fma(w01, f2{xv.w, xv.w}, acc[3][0]); acc[3][1] = __builtin_elementwise_fma(w23, f2{xv.w, xv.w}, acc[3][1]);
                };
                // GEN_PF weight quads in flight per lane (one wave per SIMD: nothing else hides the L2 latency); a slot is
                // refilled for k + GEN_PF right after its fma group, clamped to the last row so every address is valid
                fx4 w[GEN_PF];
#pragma unroll
                for (int i = 0; i < GEN_PF; ++i) w[i] = wp[(size_t)(i < in ? i : in - 1) * wstride];
                int k0 = 0;
                for (; k0 + GEN_PF <= in; k0 += GEN_PF) {
#pragma unroll
                    for (int i = 0; i < GEN_PF; ++i) {
                        fma4(w[i], *reinterpret_cast<const float4 *>(xp + (k0 + i) * GEN_COLS));
                        const int kn = k0 + i + GEN_PF;
                        w[i] = wp[(size_t)(kn < in ? kn : in - 1) * wstride];
                    }
                }
#pragma unroll
                for (int i = 0; i < GEN_PF; ++i)
                    if (k0 + i < in) fma4(w[i], *reinterpret_cast<const float4 *>(xp + (k0 + i) * GEN_COLS));
                const fx4 bb = *(const gfx4 *)(L.b + 4 * q);
                float r[4][4];          // [output e][column c]
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    r[0][c] = acc[c][0][0] + bb.x; r[1][c] = acc[c][0][1] + bb.y; r[2][c] = acc[c][1][0] + bb.z; r[3][c] = acc[c][1][1] + bb.w;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (L.relu) {         // layers.rs:89-91: `if x > 0.0 { x } else { 0.0 }`
#pragma unroll
                        for (int c = 0; c < 4; ++c) r[e][c] = r[e][c] > 0.0f ? r[e][c] : 0.0f;
                    }
                    *reinterpret_cast<float4 *>(y + (4 * q + e) * GEN_COLS + 4 * cg) = make_float4(r[e][0], r[e][1], r[e][2], r[e][3]);
                }
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
#ifdef TW_ABLATE
        const unsigned long long c0 = __builtin_readcyclecounter();
#endif
        const int E = pol.emb;
        const float *tab = pol.emb_rows;
        const float *bias = tab + (size_t)pol.obs_size * E;
        for (int q = g; q < E / 4; q += 16) {
            // every row's load is issued before the first add (a missing cell loads row 0 and is not added): one latency per
            // quad, not one per cell
            fx4 r[NC];
#pragma unroll
            for (int i = 0; i < NC; ++i)
                r[i] = *(const gfx4 *)(tab + (size_t)(rowoff[i] >= 0 ? rowoff[i] : 0) * E + 4 * q);
            fx4 a = *(const gfx4 *)(bias + 4 * q);
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                if (rowoff[i] >= 0) { a.x = a.x + r[i].x; a.y = a.y + r[i].y; a.z = a.z + r[i].z; a.w = a.w + r[i].w; }
            }
            if (pol.emb_relu) { a.x = a.x > 0.0f ? a.x : 0.0f; a.y = a.y > 0.0f ? a.y : 0.0f; a.z = a.z > 0.0f ? a.z : 0.0f; a.w = a.w > 0.0f ? a.w : 0.0f; }
            float *y = bufp(0);
            y[(4 * q + 0) * GEN_COLS + j] = a.x; y[(4 * q + 1) * GEN_COLS + j] = a.y;
            y[(4 * q + 2) * GEN_COLS + j] = a.z; y[(4 * q + 3) * GEN_COLS + j] = a.w;
        }
        __syncthreads();
#ifdef TW_ABLATE
        const unsigned long long c1 = __builtin_readcyclecounter();
#endif
        const LayerDev *ls = pol.layers;
        const int co = stack(ls, pol.n_common, 0, -1);                                         // policy.rs:86
#ifdef TW_ABLATE
        const unsigned long long c2 = __builtin_readcyclecounter();
#endif
        const int vo = stack(ls + pol.n_common + pol.n_action, pol.n_value, co, co);           // policy.rs:89
        if (g == 0) {                                                                          // .sum() of the value head's outputs
            float s = 0.0f;
            for (int i = 0; i < pol.value_out; ++i) s = s + bufp(vo)[i * GEN_COLS + j];
            lds_out[j * 8 + 4] = s;
        }
        __syncthreads();
#ifdef TW_ABLATE
        const unsigned long long c3 = __builtin_readcyclecounter();
#endif
        const int ao = stack(ls + pol.n_common, pol.n_action, co, co);                         // policy.rs:92
        if (g == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) lds_out[j * 8 + i] = bufp(ao)[i * GEN_COLS + j];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) lg[i] = lds_out[j * 8 + i];
        value = lds_out[j * 8 + 4];
#ifdef TW_ABLATE
        const unsigned long long c4 = __builtin_readcyclecounter();
        stq[0] += c1 - c0; stq[1] += c2 - c1; stq[2] += c3 - c2; stq[3] += c4 - c3;
#endif
        __syncthreads();           // (the buffers and lds_out are rewritten by the next forward)
    }
};

}  // namespace tw
