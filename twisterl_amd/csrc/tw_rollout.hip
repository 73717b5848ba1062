// tw_rollout.hip -- fused PPO rollout kernel for gfx950 (MI355X), f32 "exact" arithmetic.
//
// Replaces the hot loop of PPOCollector::single_collect (reference rust/src/collector/ppo.rs:
// 54-80): per record observe + masks + reward (envs/puzzle.rs:162-185), Policy::forward_with_perm
// (nn/policy.rs:56-100: twist, EmbeddingBag, common Linear+ReLU, value/action heads, act-perm,
// -1e10 mask), sample_from_logits (policy.rs:169-172) and Env::step (puzzle.rs:135-160).
//
// Mapping to CDNA4
//   * one workgroup = 8 waves (2 per SIMD) = 256 episodes, resident for the WHOLE episode: no
//     inter-workgroup communication (episodes are independent, ppo.rs:59).  Board state lives in
//     registers as 16 packed nibbles; both lanes (j, j+32) of an MFMA column hold the same episode.
//   * the network is evaluated TRANSPOSED: h1^T[hidden x 32 episodes] = W1^T . h0^T on
//     v_mfma_f32_32x32x2_f32, episode = MFMA column = lane&31.  The B operand of k-step s is ONE
//     f32 per lane: h0[episode][2s + (lane>>5)], which the lane computes itself as the EmbeddingBag
//     gather-sum (bias + sum over cells of table[id][k], in cell order) from a KC-column chunk of
//     the table held in LDS.  The chunk image is column-permuted ([even k | odd k]) so the four
//     k-steps a lane handles next sit in ONE 16-byte word: one ds_read_b128 per cell feeds four
//     independent add chains.  Row stride KC+4 floats: (16i+tile)*(KC+4) mod 64 takes 16 distinct
//     16-byte bank slots for the 16 tiles of a cell (conflict-free; equal tiles broadcast).
//   * the inner loop is software-pipelined by hand: while the 4*NT MFMAs of one group of four
//     k-steps issue, the gather of the NEXT group is interleaved between them (issue order pinned
//     with sched_barrier), so one wave alone can keep its SIMD's matrix pipe busy.
//   * an f32 MFMA chain IS a k-ordered fmaf chain, so the result is bit-equal to the oracle's
//     TWO_ARITH_CHAIN forward; rows of W1 are fed in an order (hid()) that makes the accumulator
//     registers come out in natural hidden order for the head product, which consumes the
//     accumulators directly as its B operand (no LDS round trip, no shuffles).
//   * both weight streams are double-buffered in LDS, one barrier per 32-column chunk: the W1
//     chunk arrives by LDS-DMA (global_load_lds_dwordx4, its image is lane-linear), the padded
//     table chunk through registers (loads issued before the chunk's MFMAs, ds_write after).
//     The streams run ahead across timesteps (chunk 0 of step t+1 is fetched during chunk 15 of t).
//   * heads: [4 logits + value] x hidden on the same MFMA shape (rows 5..31 are zero).
#include "tw_common.hpp"

#include <cstdlib>
#include <cstring>

namespace tw {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const float lds_cfloat;

constexpr int EPW = 32;            // episodes per wave (MFMA columns)
// Geometry variants (template parameters NW = waves per workgroup, KC = embedding columns per LDS
// chunk; padded LDS row stride of the table chunk is KC+1 floats):
//   NW=4, KC=16: two 256-thread workgroups per CU (75 KB LDS each); the two waves of a SIMD belong
//                to different workgroups, so their barriers and stalls are not correlated
//   NW=8, KC=32: one 512-thread workgroup per CU (141 KB LDS), half the L2->LDS stream per episode

// MFMA row i of row-tile r carries hidden unit hid(r,i) = 32r + 2g + h with
// g = (i&3) + 4*(i>>3), h = (i>>2)&1: the C/D layout (row = (g&3) + 8*(g>>2) + 4h for
// accumulator register g on lane half h) then holds hidden unit 32r + 2g + h in register g.
// (tw_api.hip builds the W1 image [k][q][i][4] = W1[k][hid(4q+c, i)] with the same formula.)

// One 1-KiB LDS-DMA piece: every lane copies 16 bytes global -> LDS (destination = wave-uniform
// base in M0 + lane*16).  Issued through inline asm on purpose: the builtin form is FLAT-encoded
// with an LDS memory operand, which makes hipcc treat it as a pending flat access and degrade
// EVERY later LDS wait to lgkmcnt(0) until the DMA has been waited for.  The copy is waited for
// with the explicit vmcnt(0) in front of the chunk barrier.
__device__ __forceinline__ void glds16(const float *gsrc, float *lds_dst)
{
    unsigned keep;
    const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)lds_dst;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(dst))
                 : "memory");
}

template <int NT> struct Tiles { static constexpr int NQ = (NT + 3) / 4; };

// LDS carve (floats): W[2][KC*NQ*128] | T[2][n_rows*LSTR] | b1[NT*32] | wh8[NT*32*8]
template <int NT, int KC>
__host__ __device__ inline size_t rollout_lds_floats(int obs_size)
{
    return (size_t)2 * KC * Tiles<NT>::NQ * 128 + (size_t)2 * (obs_size + 2) * (KC + 4) + (size_t)NT * 32 * 9;
}

// DBG != 0 builds are timing-only ablations (wrong results): 1 no gather, 2 no A-operand reads,
// 4 no weight streams, 8 no heads/sampling/record stores.  Never used by the product path.
template <int NT, int NC, int NW, int KC, int DBG = 0>
__global__ void __launch_bounds__(NW * 64, 2) rollout_f32_kernel(const RolloutArgs a)
{
    constexpr int THREADS = NW * 64;
    constexpr int EPB     = NW * EPW;
    constexpr int LSTR    = KC + 4;                               // padded row stride (floats), 16-B aligned rows
    constexpr int NG      = KC / 8;                               // groups of four k-steps per chunk
    constexpr int NQ      = Tiles<NT>::NQ;
    constexpr int WCHUNK  = KC * NQ * 128;                       // floats per W1 chunk
    constexpr int WPIECES = WCHUNK / 256;                        // 1-KiB LDS-DMA pieces per chunk
    constexpr int TITER   = (NC * NC * (KC / 4) + THREADS - 1) / THREADS;   // float4 loads per thread per table chunk
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const PuzzleConsts env = a.env;
    const PolicyDev    pol = a.pol;
    const int n_rows   = pol.obs_size + 2;
    const int bias_row = pol.obs_size, zero_row = pol.obs_size + 1;
    const int n_chunks = pol.emb / KC;

    float *lds_w  = lds;                                  // [2][WCHUNK]
    float *lds_t  = lds + 2 * WCHUNK;                     // [2][n_rows*LSTR]
    float *lds_b1 = lds_t + 2 * n_rows * LSTR;
    float *lds_wh = lds_b1 + NT * 32;
    const int tbuf = n_rows * LSTR;

    for (int i = tid; i < NT * 32; i += THREADS) lds_b1[i] = pol.b1[i];
    for (int i = tid; i < NT * 32 * 8; i += THREADS) lds_wh[i] = pol.wh8[i];
    if (tid < 2 * LSTR) lds_t[(tid / LSTR) * tbuf + zero_row * LSTR + (tid % LSTR)] = 0.0f;   // zero rows, never restaged

    const __amdgpu_buffer_rsrc_t rs_emb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(pol.emb_rows), 0, n_rows * pol.emb * (int)sizeof(float), 0x00020000);

    // ---- weight-stream helpers ----------------------------------------------------------------
    f32x4 tst[TITER];   // table chunk in flight (registers)
    f32x4 tsb;          // bias-row piece (threads 0..7)
    auto stream_issue = [&](int chunk, int buf) {
        if constexpr (DBG & 4) return;
        // W1 chunk: contiguous 4*WCHUNK bytes -> lane-linear LDS image by LDS-DMA
#pragma unroll
        for (int p = 0; p < (WPIECES + NW - 1) / NW; ++p) {
            const int piece = wave + NW * p;
            if (piece < WPIECES) {
                const float *src = pol.w1p + (size_t)chunk * WCHUNK + piece * 256 + lane * 4;
                glds16(src, lds_w + buf * WCHUNK + piece * 256);
            }
        }
        // table chunk: rows 0..obs_size-1 (+ bias row by threads 0..7) to registers
#pragma unroll
        for (int it = 0; it < TITER; ++it) {
            const int idx = tid + it * THREADS, row = idx / (KC / 4), q = idx % (KC / 4);
            if (row < pol.obs_size)
                tst[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    rs_emb, (row * pol.emb + q * 4) * (int)sizeof(float), chunk * KC * (int)sizeof(float), 0));
        }
        if (tid < KC / 4)
            tsb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rs_emb, (bias_row * pol.emb + tid * 4) * (int)sizeof(float), chunk * KC * (int)sizeof(float), 0));
    };
    auto stream_commit = [&](int buf) {
        if constexpr (DBG & 4) return;
        float *tb = lds_t + buf * tbuf;
#pragma unroll
        for (int it = 0; it < TITER; ++it) {
            const int idx = tid + it * THREADS, row = idx / (KC / 4), q = idx % (KC / 4);
            if (row < pol.obs_size) {   // k = 4q..4q+3 -> even k at 2q,2q+1; odd k at KC/2+2q,+1
                float *d = tb + row * LSTR + q * 2;
                *reinterpret_cast<float2 *>(d)          = make_float2(tst[it][0], tst[it][2]);
                *reinterpret_cast<float2 *>(d + KC / 2) = make_float2(tst[it][1], tst[it][3]);
            }
        }
        if (tid < KC / 4) {
            float *d = tb + bias_row * LSTR + tid * 2;
            *reinterpret_cast<float2 *>(d)          = make_float2(tsb[0], tsb[2]);
            *reinterpret_cast<float2 *>(d + KC / 2) = make_float2(tsb[1], tsb[3]);
        }
    };

    // ---- episode state -------------------------------------------------------------------------
    const uint64_t e_local  = (uint64_t)blockIdx.x * EPB + (uint64_t)(wave * EPW + j);
    const bool     valid    = e_local < a.num_episodes;
    const uint64_t e_global = a.episode_offset + e_local;

    stream_issue(0, 0);                                   // overlaps the scramble below

    PuzzleLane st;
    st.board = env.ident; st.zx = 0; st.zy = 0; st.depth = 0;
    if (valid) puzzle_reset(st, env, a.seed, e_global);

    bool     alive = valid;
    int      t = 0;
    uint32_t len = 0;
    const uint64_t rec_base = e_local * (uint64_t)a.out.t_pad;

    stream_commit(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int cur = 0;                                          // LDS buffer holding the chunk about to be consumed
    // (the barrier inside __syncthreads_or publishes buffer 0 and the b1/wh8/zero-row stores)

    while (__syncthreads_or(alive ? 1 : 0)) {
        // ---- observe (puzzle.rs:183-185) + twist of the obs ids (policy.rs:67-83) -------------
        int perm = -1;
        if (pol.n_perms > 0) {
            const u32x4 w = rng_draw(a.seed, e_global, (uint32_t)t, STREAM_PERM);
            perm = (int)u32_below(w.x, (uint32_t)pol.n_perms);
        }
        int rowoff[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            int row = zero_row;
            if (i < env.n_cells) {
                const int id = i * env.n_cells + (int)nib(st.board, i);
                row = perm >= 0 ? (int)pol.obs_perms[perm * pol.obs_size + id] : id;
            }
            rowoff[i] = row * LSTR + h * (KC / 2);
        }

        f32x16 acc[NT];
#pragma unroll
        for (int r = 0; r < NT; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[r][g] = 0.0f;

        // ---- EmbeddingBag (layers.rs:56-62,82-84) fused into common Linear (layers.rs:31-37) --
        for (int c = 0; c < n_chunks; ++c) {
            // run the streams one chunk ahead (wrapping to chunk 0 of the next timestep)
            stream_issue(c + 1 == n_chunks ? 0 : c + 1, cur ^ 1);

            const float *tb = lds_t + cur * tbuf;
            const float *wl = lds_w + cur * WCHUNK + (h * NQ * 32 + j) * 4;   // A operand base of this lane
            const float *bias_p = tb + bias_row * LSTR + h * (KC / 2);

            // MFMAs per group, and the issue slots of the next group's gather reads / adds
            constexpr int M   = 4 * NT;
            constexpr int LAT = M >= 16 ? 4 : 1;                  // MFMA slots between a read and its adds
            auto rd_slot  = [](int q) constexpr { return M >= 16 ? (q * (M - 6)) / (NC + 1) : 0; };
            auto add_slot = [&](int q) constexpr { int v = rd_slot(q) + LAT; return v > M - 1 ? M - 1 : v; };
            auto gather_ptr = [&](int q, int g) -> const f32x4 * {   // q = 0: bias row, q = i+1: cell i
                return reinterpret_cast<const f32x4 *>((q == 0 ? bias_p : tb + rowoff[q - 1]) + 4 * g);
            };
            auto a_ptr = [&](int kp, int q) -> const f32x4 * {        // kp = k-step within the chunk
                return reinterpret_cast<const f32x4 *>(wl + (kp * 2 * NQ + q) * 128);
            };

            // prologue: B operands of group 0 (exposed once per chunk) and the first A operands
            f32x4 bq = *gather_ptr(0, 0);
            if constexpr (!(DBG & 1))
#pragma unroll
                for (int q = 1; q <= NC; ++q) bq = bq + *gather_ptr(q, 0);
            if (pol.emb_relu)
#pragma unroll
                for (int u = 0; u < 4; ++u) bq[u] = bq[u] > 0.0f ? bq[u] : 0.0f;
            f32x4 aw[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) aw[q] = *a_ptr(0, q);

#pragma unroll
            for (int g = 0; g < NG; ++g) {
                f32x4 nb, rd[NC + 1];
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const int u = m / NT, r = m % NT;               // k-step 4g+u, row-tile r
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[r / 4][r % 4], bq[u], acc[r], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    // A operands of the next k-step, as soon as their registers are free
                    if constexpr (!(DBG & 2))
                        if ((r % 4 == 3 || r == NT - 1) && !(g == NG - 1 && u == 3))
                            aw[r / 4] = *a_ptr(4 * g + u + 1, r / 4);
                    // gather of the next group: reads, then (LAT slots later) four independent adds
                    if (g + 1 < NG) {
#pragma unroll
                        for (int q = 0; q <= ((DBG & 1) ? 0 : NC); ++q)
                            if (rd_slot(q) == m) rd[q] = *gather_ptr(q, g + 1);
#pragma unroll
                        for (int q = 0; q <= ((DBG & 1) ? 0 : NC); ++q)
                            if (add_slot(q) == m) nb = (q == 0) ? rd[0] : nb + rd[q];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (g + 1 < NG) {
                    if (pol.emb_relu)
#pragma unroll
                        for (int u = 0; u < 4; ++u) nb[u] = nb[u] > 0.0f ? nb[u] : 0.0f;
                    bq = nb;
                }
            }
            stream_commit(cur ^ 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces have landed
            __syncthreads();     // chunk c fully consumed by every wave; chunk c+1 (DMA + ds_write) landed
            cur ^= 1;
        }

        float lg[4]; float value, rew; int action;
        if constexpr (DBG & 8) {
            value = 0.0f;
#pragma unroll
            for (int r = 0; r < NT; ++r) value += acc[r][0];   // keeps the GEMM alive (stored below)
            lg[0] = lg[1] = lg[2] = lg[3] = 0.0f; rew = 0.0f; action = t & 3;
        } else {
        // ---- bias + ReLU of the common layer, then both heads (policy.rs:86-92) ---------------
        f32x16 hacc;
#pragma unroll
        for (int g = 0; g < 16; ++g) hacc[g] = 0.0f;
        // one per-lane LDS base each (kept opaque so every access is base + immediate offset)
        // (explicit LDS address space: an opaque GENERIC pointer would turn these into flat loads, and a
        //  pending flat op makes hipcc wait lgkmcnt(0) everywhere in the loop nest)
        lds_cfloat *b1_lane = (lds_cfloat *)(lds_b1 + h);
        lds_cfloat *wh_lane = (lds_cfloat *)(lds_wh + h * 8 + (j & 7));
        asm volatile("" : "+v"(b1_lane), "+v"(wh_lane));
#pragma unroll
        for (int r = 0; r < NT; ++r) {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int n0 = 32 * r + 2 * g;   // hidden unit n = n0 + h
                float hv = acc[r][g] + b1_lane[n0];
                if (pol.common_relu) hv = hv > 0.0f ? hv : 0.0f;
                const float awl = wh_lane[n0 * 8];
                const float aw  = j < 8 ? awl : 0.0f;
                hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(aw, hv, hacc, 0, 0, 0);
            }
        }
        // rows 0..3 (logits) sit in registers 0..3 of lane (j,0); row 4 (value) in register 0 of lane (j,1)
#pragma unroll
        for (int i = 0; i < 4; ++i) lg[i] = __shfl(hacc[i], j, 64) + pol.bh8[i];
        value = __shfl(hacc[0], j + 32, 64) + pol.bh8[4];

        if (perm >= 0) {   // logits'[i] = logits[act_perm[i]]  (policy.rs:95-97)
            const float l0 = lg[0], l1 = lg[1], l2 = lg[2], l3 = lg[3];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int src = pol.act_perms[perm * 4 + i];
                lg[i] = src == 0 ? l0 : (src == 1 ? l1 : (src == 2 ? l2 : l3));
            }
        }
        const uint32_t mb = puzzle_maskbits(st, env);
#pragma unroll
        for (int i = 0; i < 4; ++i) lg[i] = ((mb >> i) & 1u) ? lg[i] : -1e10f;   // policy.rs:62

        rew = puzzle_reward(st, env);
        const u32x4 gw  = rng_draw(a.seed, e_global, (uint32_t)t, STREAM_GUMBEL);
        action = gumbel_argmax4(lg, gw);

        }
        // ---- push the record (ppo.rs:71-76), then is_final / step (ppo.rs:78-79) --------------
        if (alive) {
            if (h == 0) {
                const uint64_t rec = rec_base + (uint64_t)t;
                uint32_t pk[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int i = 0; i < NC; ++i)
                    if (i < env.n_cells) pk[i >> 2] |= (uint32_t)(i * env.n_cells + (int)nib(st.board, i)) << (8 * (i & 3));
                reinterpret_cast<uint4 *>(a.out.obs)[rec]      = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                reinterpret_cast<float4 *>(a.out.logits)[rec]  = make_float4(lg[0], lg[1], lg[2], lg[3]);
                a.out.values[rec]  = value;
                a.out.rewards[rec] = rew;
                a.out.actions[rec] = (uint8_t)action;
                a.out.perms[rec]   = (int8_t)perm;
            }
            if (puzzle_final(st, env)) { alive = false; len = (uint32_t)t + 1u; }
            else { puzzle_step(st, env, action); ++t; }
        }
    }
    if (valid && h == 0) a.out.ep_len[e_local] = len;
    // drain the stream that ran ahead of the last timestep before the LDS is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int NT, int NC, int NW, int KC, int DBG = 0>
static int launch_geom(const RolloutArgs &a, hipStream_t s, uint32_t *blocks, uint32_t *threads)
{
    constexpr int EPB = NW * EPW, THREADS = NW * 64;
    const uint64_t nb = (a.num_episodes + EPB - 1) / EPB;
    if (nb == 0 || nb > 0x7fffffffull) { set_error("rollout: bad episode count %llu", (unsigned long long)a.num_episodes); return TW_ERR_INVALID; }
    const size_t lds_bytes = rollout_lds_floats<NT, KC>(a.pol.obs_size) * sizeof(float);
    if (lds_bytes > 159 * 1024) { set_error("rollout: %zu bytes of LDS needed, 159 KiB available", lds_bytes); return TW_ERR_UNSUPPORTED; }
    static size_t attr_bytes = 0;   // per instantiation: raise the dynamic-LDS limit above the 64 KiB default
    if (lds_bytes > attr_bytes) {
        TW_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&rollout_f32_kernel<NT, NC, NW, KC, DBG>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        attr_bytes = lds_bytes;
    }
    hipLaunchKernelGGL((rollout_f32_kernel<NT, NC, NW, KC, DBG>), dim3((unsigned)nb), dim3(THREADS), lds_bytes, s, a);
    TW_HIP(hipGetLastError());
    if (blocks) *blocks = (uint32_t)nb;
    if (threads) *threads = THREADS;
    return TW_OK;
}

// TW_ROLLOUT_GEOM selects the geometry (default 8x32; 4x16 and the diagnostic 4x32 for A/B runs)
static int geom_sel()
{
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("TW_ROLLOUT_GEOM");
        v = (e && strcmp(e, "4x16") == 0) ? 1 : ((e && strcmp(e, "4x32") == 0) ? 2 : 0);
    }
    return v;
}

template <int NT, int NC>
static int launch_one(const RolloutArgs &a, hipStream_t s, uint32_t *blocks, uint32_t *threads)
{
    const int g = geom_sel();
    if (g == 1) return launch_geom<NT, NC, 4, 16>(a, s, blocks, threads);
    if (g == 2) return launch_geom<NT, NC, 4, 32>(a, s, blocks, threads);   // diagnostic: one wave per SIMD
#ifdef TW_ABLATE
    if constexpr (NT == 8 && NC == 16) {
        const char *d = getenv("TW_ROLLOUT_DBG");
        const int dbg = d ? atoi(d) : 0;
        switch (dbg) {
            case 1: return launch_geom<NT, NC, 8, 32, 1>(a, s, blocks, threads);
            case 2: return launch_geom<NT, NC, 8, 32, 2>(a, s, blocks, threads);
            case 3: return launch_geom<NT, NC, 8, 32, 3>(a, s, blocks, threads);
            case 4: return launch_geom<NT, NC, 8, 32, 4>(a, s, blocks, threads);
            case 8: return launch_geom<NT, NC, 8, 32, 8>(a, s, blocks, threads);
            default: break;
        }
    }
#endif
    return launch_geom<NT, NC, 8, 32>(a, s, blocks, threads);
}

template <int NT>
static int launch_nt(const RolloutArgs &a, hipStream_t s, uint32_t *blocks, uint32_t *threads)
{
    const int nc = a.env.n_cells;
    if (nc <= 4) return launch_one<NT, 4>(a, s, blocks, threads);
    if (nc <= 9) return launch_one<NT, 9>(a, s, blocks, threads);
    return launch_one<NT, 16>(a, s, blocks, threads);
}

int launch_rollout_f32(const RolloutArgs &a, hipStream_t s, uint32_t *blocks, uint32_t *threads)
{
    // host-side shape checks: everything the kernel indexes with is validated here
    if (a.env.n_cells < 1 || a.env.n_cells > 16 || a.pol.obs_size != a.env.n_cells * a.env.n_cells ||
        a.pol.obs_size > 256 || a.pol.n_actions != 4 || a.pol.emb % 32 != 0 || a.pol.emb < 32 ||
        a.out.t_pad < a.env.depth0 + 1) {
        set_error("rollout: unsupported shape (n_cells=%d obs_size=%d actions=%d emb=%d hidden=%d t_pad=%d)",
                  a.env.n_cells, a.pol.obs_size, a.pol.n_actions, a.pol.emb, a.pol.hidden, a.out.t_pad);
        return TW_ERR_UNSUPPORTED;
    }
    switch (a.pol.hidden) {
        case 32:  return launch_nt<1>(a, s, blocks, threads);
        case 64:  return launch_nt<2>(a, s, blocks, threads);
        case 128: return launch_nt<4>(a, s, blocks, threads);
        case 256: return launch_nt<8>(a, s, blocks, threads);
        default:
            set_error("rollout: hidden size %d not in {32,64,128,256}", a.pol.hidden);
            return TW_ERR_UNSUPPORTED;
    }
}

}  // namespace tw
