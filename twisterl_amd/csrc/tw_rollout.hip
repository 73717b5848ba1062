// tw_rollout.hip -- fused PPO rollout kernel for gfx950 (MI355X), f32 "exact" arithmetic.
//
// Replaces the hot loop of PPOCollector::single_collect (reference rust/src/collector/ppo.rs:
// 54-80): per record observe + masks + reward (envs/puzzle.rs:162-185), Policy::forward_with_perm
// (nn/policy.rs:56-100: twist, EmbeddingBag, common Linear+ReLU, value/action heads, act-perm,
// -1e10 mask), sample_from_logits (policy.rs:169-172) and Env::step (puzzle.rs:135-160).
//
// Mapping to CDNA4
//   * one workgroup = 4 waves = 128 episodes, resident for the WHOLE episode (no inter-workgroup
//     communication: episodes are independent, ppo.rs:59); board state lives in registers as
//     16 packed nibbles, both lanes (j, j+32) of an MFMA column hold the same episode.
//   * the network is evaluated TRANSPOSED: h1^T[hidden x 32 episodes] = W1^T . h0^T on
//     v_mfma_f32_32x32x2_f32, episode = MFMA column = lane&31.  The B operand of k-step s is
//     ONE f32 per lane: h0[episode][2s + (lane>>5)] -- which the lane computes itself as the
//     EmbeddingBag gather-sum (bias + sum over cells of table[id][k], in cell order) from a
//     K-chunk of the table staged in LDS (row stride 33 floats: bank = (id + k) % 32, distinct
//     tiles of one cell hit distinct banks, equal tiles broadcast).
//   * an f32 MFMA chain IS a k-ordered fmaf chain, so the result is bit-equal to the oracle's
//     TWO_ARITH_CHAIN forward; rows of W1 are fed in an order (hid()) that makes the
//     accumulator registers come out in natural hidden order for the head product, which
//     consumes the accumulators directly as its B operand (no LDS round trip, no shuffles).
//   * heads: [4 logits + value] x hidden on the same MFMA shape (rows 5..31 are zero).
#include "tw_common.hpp"

namespace tw {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KC   = 32;       // embedding columns staged per LDS chunk
constexpr int LSTR = KC + 1;   // padded LDS row stride in floats
constexpr int EPW  = 32;       // episodes per wave (MFMA columns)
constexpr int EPB  = 128;      // episodes per workgroup

// MFMA row i of row-tile r carries hidden unit hid(r,i) = 32r + 2g + h with
// g = (i&3) + 4*(i>>3), h = (i>>2)&1: the C/D layout (row = (g&3) + 8*(g>>2) + 4h for
// accumulator register g on lane half h) then holds hidden unit 32r + 2g + h in register g.
__host__ __device__ inline int hid(int r, int i) { return 32 * r + 2 * ((i & 3) + 4 * (i >> 3)) + ((i >> 2) & 1); }

template <int NT, int NC>
__global__ void __launch_bounds__(256, 2) rollout_f32_kernel(const RolloutArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid  = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const PuzzleConsts env = a.env;
    const PolicyDev    pol = a.pol;
    const int n_rows   = pol.obs_size + 2;
    const int bias_row = pol.obs_size, zero_row = pol.obs_size + 1;

    // LDS carve: [n_rows][LSTR] table chunk | b1[hidden] | wh8[hidden][8]
    float *lds_b1 = lds + n_rows * LSTR;
    float *lds_wh = lds_b1 + NT * 32;
    for (int i = tid; i < NT * 32; i += 256) lds_b1[i] = pol.b1[i];
    for (int i = tid; i < NT * 32 * 8; i += 256) lds_wh[i] = pol.wh8[i];

    // weight streams go through buffer descriptors: wave-uniform base + one per-lane VGPR offset
    // + scalar offset, so no per-load 64-bit address registers are kept live across the loops
    const __amdgpu_buffer_rsrc_t rs_emb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(pol.emb_rows), 0, n_rows * pol.emb * (int)sizeof(float), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(pol.w1p), 0, pol.emb * 32 * NT * (int)sizeof(float), 0x00020000);
    const int w1_voff = (h * 32 + j) * NT * (int)sizeof(float);

    const uint64_t e_local  = (uint64_t)blockIdx.x * EPB + (uint64_t)(wave * EPW + j);
    const bool     valid    = e_local < a.num_episodes;
    const uint64_t e_global = a.episode_offset + e_local;

    PuzzleLane st;
    st.board = env.ident; st.zx = 0; st.zy = 0; st.depth = 0;
    if (valid) puzzle_reset(st, env, a.seed, e_global);

    bool     alive = valid;
    int      t = 0;
    uint32_t len = 0;
    const uint64_t rec_base = e_local * (uint64_t)a.out.t_pad;

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
        for (int kc = 0; kc < pol.emb; kc += KC) {
            __syncthreads();
            for (int idx = tid; idx < n_rows * (KC / 4); idx += 256) {
                const int row = idx >> 3, q = idx & 7;
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    rs_emb, (row * pol.emb + q * 4) * (int)sizeof(float), kc * (int)sizeof(float), 0));
                float *d = lds + row * LSTR + q * 4;
                d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
            }
            __syncthreads();
            const float *bias_p = lds + bias_row * LSTR + h;
#pragma unroll
            for (int s = 0; s < KC / 2; ++s) {
                float b = bias_p[2 * s];
#pragma unroll
                for (int i = 0; i < NC; ++i) b = b + lds[rowoff[i] + 2 * s];
                if (pol.emb_relu) b = b > 0.0f ? b : 0.0f;
                const int soff = (kc + 2 * s) * 32 * NT * (int)sizeof(float);   // k = kc + 2s (+h via w1_voff)
                if constexpr (NT % 4 == 0) {
#pragma unroll
                    for (int q = 0; q < NT / 4; ++q) {
                        const f32x4 aw = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                            rs_w1, w1_voff + q * 16, soff, 0));
                        acc[4 * q + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[0], b, acc[4 * q + 0], 0, 0, 0);
                        acc[4 * q + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[1], b, acc[4 * q + 1], 0, 0, 0);
                        acc[4 * q + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[2], b, acc[4 * q + 2], 0, 0, 0);
                        acc[4 * q + 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[3], b, acc[4 * q + 3], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < NT; ++r) {
                        const float aw = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            rs_w1, w1_voff + r * 4, soff, 0));
                        acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw, b, acc[r], 0, 0, 0);
                    }
                }
            }
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
}

template <int NT, int NC>
static int launch_one(const RolloutArgs &a, hipStream_t s, uint32_t *blocks, uint32_t *threads)
{
    const uint64_t nb = (a.num_episodes + EPB - 1) / EPB;
    if (nb == 0 || nb > 0x7fffffffull) { set_error("rollout: bad episode count %llu", (unsigned long long)a.num_episodes); return TW_ERR_INVALID; }
    const size_t lds_bytes = ((size_t)(a.pol.obs_size + 2) * LSTR + (size_t)NT * 32 * 9) * sizeof(float);
    hipLaunchKernelGGL((rollout_f32_kernel<NT, NC>), dim3((unsigned)nb), dim3(256), lds_bytes, s, a);
    TW_HIP(hipGetLastError());
    if (blocks) *blocks = (uint32_t)nb;
    if (threads) *threads = 256;
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
