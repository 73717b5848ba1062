// tw_engine_generic.hpp -- EngineV: the policy-forward engine for ANY Sequential stack the reference's Policy can hold
// (rust/src/nn/modules.rs:28-34 runs any list of Linear layers; src/twisterl/nn/utils.py:17-42 exports any): N common
// layers, non-empty policy_layers / value_layers, widths up to 512, any obs_size -- the shapes the MFMA engines of
// tw_engine.hpp (embedding -> one common Linear of 32..256 units -> linear heads: both Puzzle configs) do not cover.
//
// Same interface as Engine3T (16 episode columns per 256-thread workgroup, lane = (h = lane >> 4, j = lane & 15), all four
// waves carry every column's state), so the rollout / self-play / solve kernels instantiate it unchanged.  Arithmetic = the
// numeric spec of DESIGN.md: EmbeddingBag = bias + rows in cell order (plain adds), every Linear a k-ordered fma chain from
// 0 with the bias added last -- v_mfma_f32_16x16x4_f32 per group of four inputs and tile of 16 outputs, weights read from a
// per-layer image in global memory (L2-resident), activations ping-pong through LDS as [unit][column].  Bit-equal to the
// oracle's TWO_ARITH_CHAIN forward.
#pragma once
#include "tw_common.hpp"

namespace tw {

constexpr int GEN_MAX_WIDTH = 512;        // widest layer (and embedding) the engine's LDS buffers hold
constexpr int GEN_COLS = 16;
constexpr int GEN_STRIDE = 17;             // floats per unit row of an activation buffer: 16 columns + 1 (the embedding's writes -- 16 lanes, 4 units apart -- then hit 16 different banks)

typedef float fx4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) fx4 gfx4;

// ASM_MFMA: the MFMAs as inline asm with the accumulators pinned to VGPRs (with the intrinsic the allocator shuttles them between VGPRs
// and AGPRs around every k-group).  NO kernel uses that form any more (round 3).  hipcc pads the wait states an MFMA needs after a
// vector-ALU write of one of its operands for its own MFMAs, not for an asm string: the accumulator's zero-initialisation (v_mov) or the
// reload of a value the allocator parked in an AGPR can land right in front of the asm MFMA.  Seen twice as deterministic wrong sums:
// the 9-cell rollout at two workgroups per CU, and self-play of a 6 x 4 board with a 32 -> 48 -> 32 stack (one visit count off in 14 of
// 126 records; every other shape tested was right).  scripts/scan_mfma_hazards.py looks for the pattern in the compiled kernels (an
// operand written by one of the two vector instructions in front of an MFMA, no s_nop between): none with the intrinsic form; the
// asm MFMAs of the other engines (tw_engine.hpp) take their operands from LDS / global loads and ReLUs that carry their own s_nop.
template <int NC, bool ASM_MFMA = false>
struct EngineV {
    static constexpr int NW = 4, THREADS = 256, EPB = GEN_COLS, NS = 4;
    static constexpr bool SPLIT = true;

    PolicyDev pol;
    int tid, lane, wave, j, h, g;
    float *lds0, *lds_out, *lds_user;
    uint64_t boffs;                        // float offsets of the three activation buffers, 21 bits each (one value: selecting among
                                           // three adjacent members made the compiler index them in memory -- the whole engine went to scratch)
    int *lds_rows;                         // [16 columns][NC] obs ids of the forward (the embedding gather works on other columns than its lane's)
    const uint8_t *perm_obs, *perm_act;
#ifdef TW_ABLATE
    unsigned long long stq[6] = {0, 0, 0, 0, 0, 0};      // diagnostic build: cycles in embedding | common | value head | action head
#endif

    // three activation buffers (the common output stays put while the two heads run), each as many rows as the widest layer it
    // ever holds (PolicyDev::gen_rows, laid out by tw_policy_create with the buffer choices of stack() below): the usual
    // 256 -> 512 -> 256 policy needs 57 KB instead of the 108 KB of three full-width buffers, and two workgroups share a CU
    // | head outputs [16][8] | kernel use
    __host__ __device__ static size_t lds_floats(const PolicyDev &p)
    {
        return (size_t)(p.gen_rows0 + p.gen_rows1 + p.gen_rows2) * GEN_STRIDE + GEN_COLS * 8 + 256 + GEN_COLS * NC + 544;   // (+ 2 KiB of read slack, see layer_tb)
    }
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
        boffs = ((uint64_t)(pol.gen_rows0 * GEN_STRIDE) << 21) | ((uint64_t)((pol.gen_rows0 + pol.gen_rows1) * GEN_STRIDE) << 42);
        lds_out = lds + (size_t)(pol.gen_rows0 + pol.gen_rows1 + pol.gen_rows2) * GEN_STRIDE;
        lds_user = lds_out + GEN_COLS * 8;
        lds_rows = reinterpret_cast<int *>(lds_user + 256);
        perm_obs = pol.obs_perms; perm_act = pol.act_perms;
    }
    __device__ __forceinline__ void begin2() {}
    __device__ __forceinline__ void end() {}

    __device__ __forceinline__ float *bufp(int i) const { return lds0 + (size_t)((boffs >> (21 * i)) & 0x1fffffu); }

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

    // one Linear (layers.rs:31-37) for the 16 columns: x [in][16] -> y [out][16], on the matrix cores.  v_mfma_f32_16x16x4_f32
    // is a k-ordered fma chain per output element (scripts/mfma_probe/probe_f32_16x16x4.hip), so one MFMA per group of four
    // inputs and 16-output tile continues the layer's chains exactly as the reference's dot product runs; the padding of the
    // image (inputs up to a multiple of four, outputs up to whole tiles) multiplies -0.0 into the chain, an exact identity.
    // Lane = (kq = lane >> 4, jj = lane & 15): A = W[4g + kq][tile row jj] (ONE load of tb floats per lane and k-group: the
    // image interleaves the block's tiles), B = x[4g + kq][column jj] (one LDS read), D rows 4kq .. 4kq+3 of each tile.
    // A wave works on blocks wave, wave + 4, ..; per k-group: one weight load, one LDS read, tb MFMAs -- the matrix pipe is the
    // limit (the vector-ALU form of this loop spent 10 instructions per 128 fmas and ran at a quarter of this rate).
    template <int TB>
    __device__ __forceinline__ void layer_tb(const LayerDev &L, const float *x, float *y, int w0, int nwv) const      // waves w0 .. w0+nwv-1 work
    {
        typedef float f4v __attribute__((ext_vector_type(4)));
        const int kq = lane >> 4, jj = lane & 15;
        const int NB = L.nb, KG = L.kg;
        constexpr int PF = 8;                                                       // k-groups of weights in flight
        for (int b = wave - w0; b >= 0 && wave < w0 + nwv && b < NB; b += nwv) {     // (wave-uniform; no barrier inside)
            f4v acc[TB];
#pragma unroll
            for (int t = 0; t < TB; ++t) acc[t] = f4v{0.0f, 0.0f, 0.0f, 0.0f};
            const size_t gstride = (size_t)4 * NB * 16 * TB;                         // floats per k-group
            // running pointers, PF groups ahead of the MFMAs: the image carries PF k-groups of slack behind the last one and the
            // LDS buffers 2 KiB, so nothing is clamped; the slack is loaded and never used
            const float *wq = L.wm + (((size_t)kq * NB + b) * 16 + jj) * TB;
            const float *xq = x + kq * GEN_STRIDE + jj;
            auto ldw = [&](float (&w)[TB]) {
                if constexpr (TB == 4) { const fx4 v = *(const gfx4 *)wq; w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
                else {
                    typedef const __attribute__((address_space(1))) float gf1;
#pragma unroll
                    for (int t = 0; t < TB; ++t) w[t] = ((gf1 *)wq)[t];
                }
                wq += gstride;
            };
            float w[PF][TB], xv[PF];                                                 // weights AND activations PF k-groups ahead: no LDS
#pragma unroll                                                                       // or L2 round trip between two MFMAs
            for (int i = 0; i < PF; ++i) { ldw(w[i]); xv[i] = xq[i * (4 * GEN_STRIDE)]; }
            xq += PF * (4 * GEN_STRIDE);
            int g0 = 0;
            for (; g0 + PF <= KG; g0 += PF) {
#pragma unroll
                for (int i = 0; i < PF; ++i) {
#pragma unroll
                    for (int t = 0; t < TB; ++t) {
                        if constexpr (ASM_MFMA) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(w[i][t]), "v"(xv[i]));
                        else acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i][t], xv[i], acc[t], 0, 0, 0);
                        if (t == TB - 1) xv[i] = xq[i * (4 * GEN_STRIDE)];            // (after the last MFMA that reads it was issued)
                    }
                    ldw(w[i]);
                }
                xq += PF * (4 * GEN_STRIDE);
            }
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                if (g0 + i < KG) {
#pragma unroll
                    for (int t = 0; t < TB; ++t) {
                        if constexpr (ASM_MFMA) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(w[i][t]), "v"(xv[i]));
                        else acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i][t], xv[i], acc[t], 0, 0, 0);
                    }
                }
            }
            // (the MFMAs are inline asm with the accumulators pinned to VGPRs -- with the intrinsic the allocator shuttles them
            //  between VGPRs and AGPRs around every group; the wait states between the last MFMA and the reads below are ours)
            if constexpr (ASM_MFMA) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
            // D row 4kq + r of tile t is output (b * TB + t) * 16 + 4kq + r
#pragma unroll
            for (int t = 0; t < TB; ++t) {
                const int o0 = (b * TB + t) * 16 + 4 * kq;
                const fx4 bb = *(const gfx4 *)(L.b + o0);
                float r[4] = {acc[t][0] + bb.x, acc[t][1] + bb.y, acc[t][2] + bb.z, acc[t][3] + bb.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (L.relu) r[e] = r[e] > 0.0f ? r[e] : 0.0f;                   // layers.rs:89-91: `if x > 0.0 { x } else { 0.0 }`
                    y[(o0 + e) * GEN_STRIDE + jj] = r[e];
                }
            }
        }
    }

    // (no barrier: the caller publishes y)
    __device__ __forceinline__ void layer_on(const LayerDev &L, const float *x, float *y, int w0, int nwv) const
    {
        switch (L.tb) {                                                              // (wave-uniform)
            case 1: layer_tb<1>(L, x, y, w0, nwv); break;
            case 2: layer_tb<2>(L, x, y, w0, nwv); break;
            case 3: layer_tb<3>(L, x, y, w0, nwv); break;
            default: layer_tb<4>(L, x, y, w0, nwv); break;
        }
    }
    __device__ __forceinline__ void layer(const LayerDev &L, const float *x, float *y) const
    {
        layer_on(L, x, y, 0, NW);
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
        // Thread t gathers for column t >> 4 (not its lane's: the ids go round through LDS) the output quads t & 15, + 16, ..: the 16
        // lanes of a column read 256 contiguous bytes of a row per load instruction -- half the cache lines of a mapping with one
        // column per lane.  QT quads per trip: all QT x NC row loads are requested before the first add (a missing cell loads
        // row 0 and is not added), so a trip costs one L2 round trip.
        if (g == 0) {
#pragma unroll
            for (int i = 0; i < NC; ++i) lds_rows[j * NC + i] = rowoff[i];
        }
        __syncthreads();
        const int col = tid >> 4, ql = tid & 15;
        const int nq = E / 4;
        if constexpr (NC <= 25) {
        int ro[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) ro[i] = lds_rows[col * NC + i];
        constexpr int QT = NC > 16 ? 1 : (NC > 9 ? 2 : 4);     // QT x NC row quads are in flight per lane: at most 128 registers (two workgroups per CU)
        for (int q0 = ql; q0 < nq; q0 += 16 * QT) {
            int qs[QT];
#pragma unroll
            for (int u = 0; u < QT; ++u) qs[u] = q0 + 16 * u < nq ? q0 + 16 * u : q0;  // (past the end: a repeat of the trip's first quad)
            fx4 r[QT][NC];
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const float *row = tab + (size_t)(ro[i] >= 0 ? ro[i] : 0) * E;
#pragma unroll
                for (int u = 0; u < QT; ++u) r[u][i] = *(const gfx4 *)(row + 4 * qs[u]);
            }
            fx4 a[QT];
#pragma unroll
            for (int u = 0; u < QT; ++u) a[u] = *(const gfx4 *)(bias + 4 * qs[u]);
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                if (ro[i] >= 0) {
#pragma unroll
                    for (int u = 0; u < QT; ++u) { a[u].x = a[u].x + r[u][i].x; a[u].y = a[u].y + r[u][i].y; a[u].z = a[u].z + r[u][i].z; a[u].w = a[u].w + r[u][i].w; }
                }
            }
            float *y = bufp(0);
#pragma unroll
            for (int u = 0; u < QT; ++u) {
                if (pol.emb_relu) { a[u].x = a[u].x > 0.0f ? a[u].x : 0.0f; a[u].y = a[u].y > 0.0f ? a[u].y : 0.0f; a[u].z = a[u].z > 0.0f ? a[u].z : 0.0f; a[u].w = a[u].w > 0.0f ? a[u].w : 0.0f; }
                y[(4 * qs[u] + 0) * GEN_STRIDE + col] = a[u].x; y[(4 * qs[u] + 1) * GEN_STRIDE + col] = a[u].y;
                y[(4 * qs[u] + 2) * GEN_STRIDE + col] = a[u].z; y[(4 * qs[u] + 3) * GEN_STRIDE + col] = a[u].w;
            }
        }
        } else {
        // boards of 26 .. 64 cells: the cells in blocks of 16 (the row loads of a block are in flight together, the adds stay in cell
        // order), two output quads per trip; the ids are read from LDS where they are used
        constexpr int QT = 2, CB = 16;
        for (int q0 = ql; q0 < nq; q0 += 16 * QT) {
            int qs[QT];
#pragma unroll
            for (int u = 0; u < QT; ++u) qs[u] = q0 + 16 * u < nq ? q0 + 16 * u : q0;
            fx4 a[QT];
#pragma unroll
            for (int u = 0; u < QT; ++u) a[u] = *(const gfx4 *)(bias + 4 * qs[u]);
#pragma unroll
            for (int c0 = 0; c0 < NC; c0 += CB) {
                fx4 r[QT][CB]; int ro[CB];
#pragma unroll
                for (int i = 0; i < CB; ++i) {
                    ro[i] = c0 + i < NC ? lds_rows[col * NC + c0 + i] : -1;
                    const float *row = tab + (size_t)(ro[i] >= 0 ? ro[i] : 0) * E;
#pragma unroll
                    for (int u = 0; u < QT; ++u) r[u][i] = *(const gfx4 *)(row + 4 * qs[u]);
                }
#pragma unroll
                for (int i = 0; i < CB; ++i) {
                    if (ro[i] >= 0) {
#pragma unroll
                        for (int u = 0; u < QT; ++u) { a[u].x = a[u].x + r[u][i].x; a[u].y = a[u].y + r[u][i].y; a[u].z = a[u].z + r[u][i].z; a[u].w = a[u].w + r[u][i].w; }
                    }
                }
            }
            float *y = bufp(0);
#pragma unroll
            for (int u = 0; u < QT; ++u) {
                if (pol.emb_relu) { a[u].x = a[u].x > 0.0f ? a[u].x : 0.0f; a[u].y = a[u].y > 0.0f ? a[u].y : 0.0f; a[u].z = a[u].z > 0.0f ? a[u].z : 0.0f; a[u].w = a[u].w > 0.0f ? a[u].w : 0.0f; }
                y[(4 * qs[u] + 0) * GEN_STRIDE + col] = a[u].x; y[(4 * qs[u] + 1) * GEN_STRIDE + col] = a[u].y;
                y[(4 * qs[u] + 2) * GEN_STRIDE + col] = a[u].z; y[(4 * qs[u] + 3) * GEN_STRIDE + col] = a[u].w;
            }
        }
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
        int vo, ao;
        if (pol.n_value == 1 && pol.n_action == 1) {
            // single-Linear heads (BasicPolicy's default): both at once, the value head on waves 2..3, the action head on waves
            // 0..1 (a head is one or two blocks of tiles: one after the other they keep one wave busy each)
            vo = co == 0 ? 1 : 0; ao = 3 - co - vo;
            layer_on(ls[pol.n_common + pol.n_action], bufp(co), bufp(vo), 2, 2);               // policy.rs:89
            layer_on(ls[pol.n_common], bufp(co), bufp(ao), 0, 2);                              // policy.rs:92
            __syncthreads();
        } else {
            vo = stack(ls + pol.n_common + pol.n_action, pol.n_value, co, co);                 // policy.rs:89
            ao = -1;
        }
        if (g == 0) {                                                                          // .sum() of the value head's outputs
            float s = 0.0f;
            for (int i = 0; i < pol.value_out; ++i) s = s + bufp(vo)[i * GEN_STRIDE + j];
            lds_out[j * 8 + 4] = s;
        }
        __syncthreads();
#ifdef TW_ABLATE
        const unsigned long long c3 = __builtin_readcyclecounter();
#endif
        if (ao < 0) ao = stack(ls + pol.n_common, pol.n_action, co, co);                       // policy.rs:92
        if (g == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) lds_out[j * 8 + i] = bufp(ao)[i * GEN_STRIDE + j];
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
