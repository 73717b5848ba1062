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
//     gather-sum (bias + sum over cells of table[id][k], in cell order) from a 32-column chunk of
//     the table held in LDS (row stride 33 floats: bank = (id + k) % 32, so distinct tiles of one
//     cell hit distinct banks and equal tiles broadcast).
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

namespace tw {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KC      = 32;        // embedding columns per LDS chunk
constexpr int LSTR    = KC + 1;    // padded LDS row stride of the table chunk (floats)
constexpr int THREADS = 512;
constexpr int EPW     = 32;        // episodes per wave (MFMA columns)
constexpr int EPB     = 256;       // episodes per workgroup

// MFMA row i of row-tile r carries hidden unit hid(r,i) = 32r + 2g + h with
// g = (i&3) + 4*(i>>3), h = (i>>2)&1: the C/D layout (row = (g&3) + 8*(g>>2) + 4h for
// accumulator register g on lane half h) then holds hidden unit 32r + 2g + h in register g.
// (tw_api.hip builds the W1 image [k][q][i][4] = W1[k][hid(4q+c, i)] with the same formula.)

template <int NT> struct Tiles { static constexpr int NQ = (NT + 3) / 4; };

// LDS carve (floats): W[2][KC*NQ*128] | T[2][n_rows*LSTR] | b1[NT*32] | wh8[NT*32*8]
template <int NT>
__host__ __device__ inline size_t rollout_lds_floats(int obs_size)
{
    return (size_t)2 * KC * Tiles<NT>::NQ * 128 + (size_t)2 * (obs_size + 2) * LSTR + (size_t)NT * 32 * 9;
}

template <int NT, int NC>
__global__ void __launch_bounds__(THREADS, 2) rollout_f32_kernel(const RolloutArgs a)
{
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
        // W1 chunk: contiguous 4*WCHUNK bytes -> lane-linear LDS image by LDS-DMA
#pragma unroll
        for (int p = 0; p < (WPIECES + 7) / 8; ++p) {
            const int piece = wave + 8 * p;
            if (piece < WPIECES) {
                const float *src = pol.w1p + (size_t)chunk * WCHUNK + piece * 256 + lane * 4;
                __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)(lds_w + buf * WCHUNK + piece * 256),
                                                 16, 0, 0);
            }
        }
        // table chunk: rows 0..obs_size-1 (+ bias row by threads 0..7) to registers
#pragma unroll
        for (int it = 0; it < TITER; ++it) {
            const int idx = tid + it * THREADS, row = idx >> 3, q = idx & 7;
            if (row < pol.obs_size)
                tst[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    rs_emb, (row * pol.emb + q * 4) * (int)sizeof(float), chunk * KC * (int)sizeof(float), 0));
        }
        if (tid < 8)
            tsb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rs_emb, (bias_row * pol.emb + tid * 4) * (int)sizeof(float), chunk * KC * (int)sizeof(float), 0));
    };
    auto stream_commit = [&](int buf) {
        float *tb = lds_t + buf * tbuf;
#pragma unroll
        for (int it = 0; it < TITER; ++it) {
            const int idx = tid + it * THREADS, row = idx >> 3, q = idx & 7;
            if (row < pol.obs_size) {
                float *d = tb + row * LSTR + q * 4;
                d[0] = tst[it][0]; d[1] = tst[it][1]; d[2] = tst[it][2]; d[3] = tst[it][3];
            }
        }
        if (tid < 8) {
            float *d = tb + bias_row * LSTR + tid * 4;
            d[0] = tsb[0]; d[1] = tsb[1]; d[2] = tsb[2]; d[3] = tsb[3];
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
            rowoff[i] = row * LSTR + h;
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
            const float *bias_p = tb + bias_row * LSTR + h;
#pragma unroll
            for (int s = 0; s < KC / 2; ++s) {
                float b = bias_p[2 * s];
#pragma unroll
                for (int i = 0; i < NC; ++i) b = b + tb[rowoff[i] + 2 * s];
                if (pol.emb_relu) b = b > 0.0f ? b : 0.0f;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const f32x4 aw = *reinterpret_cast<const f32x4 *>(wl + (s * 2 * NQ + q) * 128);
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc)
                        if (4 * q + cc < NT)
                            acc[4 * q + cc] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[cc], b, acc[4 * q + cc], 0, 0, 0);
                }
            }
            stream_commit(cur ^ 1);
            __syncthreads();     // chunk c fully consumed by every wave; chunk c+1 (DMA + ds_write) landed
            cur ^= 1;
        }

        // ---- bias + ReLU of the common layer, then both heads (policy.rs:86-92) ---------------
        f32x16 hacc;
#pragma unroll
        for (int g = 0; g < 16; ++g) hacc[g] = 0.0f;
        // one per-lane LDS base each (kept opaque so every access is base + immediate offset)
        const float *b1_lane = lds_b1 + h;
        const float *wh_lane = lds_wh + h * 8 + (j & 7);
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
        float lg[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) lg[i] = __shfl(hacc[i], j, 64) + pol.bh8[i];
        const float value = __shfl(hacc[0], j + 32, 64) + pol.bh8[4];

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

        const float rew = puzzle_reward(st, env);
        const u32x4 gw  = rng_draw(a.seed, e_global, (uint32_t)t, STREAM_GUMBEL);
        const int action = gumbel_argmax4(lg, gw);

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

template <int NT, int NC>
static int launch_one(const RolloutArgs &a, hipStream_t s, uint32_t *blocks, uint32_t *threads)
{
    const uint64_t nb = (a.num_episodes + EPB - 1) / EPB;
    if (nb == 0 || nb > 0x7fffffffull) { set_error("rollout: bad episode count %llu", (unsigned long long)a.num_episodes); return TW_ERR_INVALID; }
    const size_t lds_bytes = rollout_lds_floats<NT>(a.pol.obs_size) * sizeof(float);
    if (lds_bytes > 159 * 1024) { set_error("rollout: %zu bytes of LDS needed, 159 KiB available", lds_bytes); return TW_ERR_UNSUPPORTED; }
    static size_t attr_bytes = 0;   // per instantiation: raise the dynamic-LDS limit above the 64 KiB default
    if (lds_bytes > attr_bytes) {
        TW_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&rollout_f32_kernel<NT, NC>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        attr_bytes = lds_bytes;
    }
    hipLaunchKernelGGL((rollout_f32_kernel<NT, NC>), dim3((unsigned)nb), dim3(THREADS), lds_bytes, s, a);
    TW_HIP(hipGetLastError());
    if (blocks) *blocks = (uint32_t)nb;
    if (threads) *threads = THREADS;
    return TW_OK;
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
        a.pol.obs_size > 256 || a.pol.n_actions != 4 || a.pol.emb % KC != 0 || a.pol.emb < KC ||
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
