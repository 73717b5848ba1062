// tw_common.hpp -- shared declarations of the HIP collector library (gfx950 only).
//
// Numeric spec shared by every kernel (and restated independently by the CPU oracle):
//  * RNG: Philox4x32-10, counter = (episode_lo, episode_hi, index, stream), key = seed.
//    Replaces rand::thread_rng() (reference puzzle.rs:124, policy.rs:72,170), which cannot
//    be seeded.
//  * "exact" forward: every Linear is a k-ordered fused-multiply-add chain from 0 with the
//    bias added last -- exactly what v_mfma_f32_32x32x2_f32 accumulates.
//  * tw_logf: fixed operation sequence (explicit fma), bit-reproducible on CPU and GPU.
// The library is compiled with -ffp-contract=off: the only FMAs are the explicit ones.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "twisterl_hip.h"

// LDS-DMA (global_load_lds_dwordx4): 16 bytes per lane from `src` + the lane's `voff` straight into LDS at the wave-uniform byte
// address in M0 (+ 16 x lane).  Inline asm on purpose: the weight streams are waited for by hand (`s_waitcnt vmcnt(N)` at the points
// the rings turn), and a load hipcc counted itself would be waited for at every use of LDS it cannot tell apart.  M0 is
// compiler-reserved and not preserved around a statement, so each statement writes it, uses it and puts the old value back
// (cdna_hip_programming.md §5.7: naming "m0" as a clobber only draws "clobber list contains reserved registers").  The s_nop 0
// is the wait state between the SALU write of M0 and the LDS-DMA reading it.
#define TW_GLDS16(voff, lds_dst, src)                                                                                         \
    do {                                                                                                                      \
        uint32_t tw_m0_keep_;                                                                                                 \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"     \
                     : "=&s"(tw_m0_keep_) : "v"(voff), "s"(lds_dst), "s"(src) : "memory");                                   \
    } while (0)
// ... with the LDS address as base + compile-time offset (one SALU add, into M0 directly)
#define TW_GLDS16_ADD(voff, lds_base, imm, src)                                                                               \
    do {                                                                                                                      \
        uint32_t tw_m0_keep_;                                                                                                 \
        asm volatile("s_mov_b32 %0, m0\n\ts_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4\n\ts_mov_b32 m0, %0" \
                     : "=&s"(tw_m0_keep_) : "v"(voff), "s"(lds_base), "i"(imm), "s"(src) : "memory", "scc");                 \
    } while (0)

// Inline-asm MFMAs remain in the two f16 engines only (tw_engine16.hpp, tw_engine16x2.hpp: the accumulator as an in/out "v" operand;
// with the builtin hipcc parks the embedding tiles in AGPRs and shuttles common-layer tiles around them: +17 % and 2.7 x,
// profiles/r04_mfma_intrinsic_vs_asm.txt).  An asm MFMA gets no hazard padding from hipcc: twisterl_amd/build.py scans the assembly of
// every build for that (scripts/scan_mfma_hazards.py).  -DTW_MFMA_INTRIN=<mask> compiles a site with the builtin instead (the
// measurement: TW_VARIANT=intrin): bit 3 Engine16, bit 4 Engine16x2.  (Engine3S and Engine3T use the builtin since round 4.)
#ifndef TW_MFMA_INTRIN
#define TW_MFMA_INTRIN 0
#endif

namespace tw {

// ---- error plumbing ------------------------------------------------------------------------
void set_error(const char *fmt, ...);
int  hip_fail(hipError_t e, const char *what, const char *file, int line);

#define TW_HIP(call)                                                              \
    do {                                                                          \
        hipError_t _e = (call);                                                   \
        if (_e != hipSuccess) return ::tw::hip_fail(_e, #call, __FILE__, __LINE__); \
    } while (0)

hipStream_t current_stream();

// Diagnostic launch overrides (tw_set_launch_option; the tests pin launch shapes with them).  Read once per collect.
struct LaunchOptions { int force_geom; int no_persist; int az_variant; int az_tree_budget; int az_tree_budget_min; int az_reuse; };
LaunchOptions launch_options();
// Raises a kernel's dynamic-LDS limit above the 64 KiB default; cached per (kernel, device), thread-safe.
int ensure_dynamic_lds(const void *kernel, size_t bytes);
// Compute units of the current device (cached per device).
int device_cus();

// ---- RNG streams (see DESIGN.md "RNG spec") -------------------------------------------------
enum : uint32_t {
    STREAM_SCRAMBLE = 0,  // Puzzle::reset scramble actions      (puzzle.rs:124-131)
    STREAM_GUMBEL   = 1,  // sample_from_logits uniforms         (policy.rs:169-172)
    STREAM_PERM     = 2,  // Policy::get_perm_id                 (policy.rs:67-77)
    STREAM_AZ_ACT   = 3,  // AZCollector root action sample      (az.rs:72)
    STREAM_MCTS     = 4,  // MCTSTree::next_sample               (search.rs:94-100)
    STREAM_SOLVE    = 5   // single_solve action sample          (solve.rs:50-54)
};

struct u32x4 { uint32_t x, y, z, w; };

__host__ __device__ inline u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1)
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0;
        const uint64_t p1 = (uint64_t)M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += W0; k1 += W1;
    }
    return u32x4{c0, c1, c2, c3};
}

__host__ __device__ inline u32x4 rng_draw(uint64_t seed, uint64_t episode, uint32_t index, uint32_t stream)
{
    return philox4x32_10((uint32_t)episode, (uint32_t)(episode >> 32), index, stream,
                         (uint32_t)seed, (uint32_t)(seed >> 32));
}

// integer in [0,n): high half of word*n
__host__ __device__ inline uint32_t u32_below(uint32_t word, uint32_t n)
{
    return (uint32_t)(((uint64_t)word * (uint64_t)n) >> 32);
}

// f32 in [0,1) with 24 random bits (granularity of rand's gen::<f32>(), policy.rs:171)
__host__ __device__ inline float u32_to_unit(uint32_t word)
{
    return (float)(word >> 8) * (1.0f / 16777216.0f);
}

// Deterministic natural log: explicit op order, same on host and device.
// Domain: positive normal floats, +0 -> -inf, +inf -> +inf.
__host__ __device__ inline float tw_logf(float x)
{
    // branch-free (selects only): the kernels call this eight times per record from a single wave per SIMD
    uint32_t ix = __builtin_bit_cast(uint32_t, x);
    int e = (int)(ix >> 23) - 127;
    float m = __builtin_bit_cast(float, (ix & 0x007fffffu) | 0x3f800000u);
    const bool big = m > 1.41421354f;
    m = big ? m * 0.5f : m;
    e = big ? e + 1 : e;
    const float f = m - 1.0f;
    const float z = f * f;
    float p = 7.0376836292e-2f;
    p = __builtin_fmaf(p, f, -1.1514610310e-1f);
    p = __builtin_fmaf(p, f,  1.1676998740e-1f);
    p = __builtin_fmaf(p, f, -1.2420140846e-1f);
    p = __builtin_fmaf(p, f,  1.4249322787e-1f);
    p = __builtin_fmaf(p, f, -1.6668057665e-1f);
    p = __builtin_fmaf(p, f,  2.0000714765e-1f);
    p = __builtin_fmaf(p, f, -2.4999993993e-1f);
    p = __builtin_fmaf(p, f,  3.3333331174e-1f);
    float y = (p * f) * z;
    const float fe = (float)e;
    y = __builtin_fmaf(fe, -2.12194440e-4f, y);
    y = __builtin_fmaf(-0.5f, z, y);
    float r = f + y;
    r = __builtin_fmaf(fe, 0.693359375f, r);
    r = x > 3.4028234e38f ? __builtin_inff() : r;
    r = x == 0.0f ? -__builtin_inff() : r;
    return r;
}

// Deterministic exp (Cephes-style, explicit op order; same sequence in the oracle): used by the
// masked softmax of predict / full_predict / the MCTS priors (policy.rs:43,118), so that UCB
// arg-max decisions are bit-reproducible on CPU and GPU.  Within 1 ulp of libm.
__host__ __device__ inline float tw_expf(float x)
{
    if (x != x) return x;
    if (x > 88.72283905206835f) return __builtin_inff();
    if (x < -103.972077083991796f) return 0.0f;
    const float fx = __builtin_fmaf(x, 1.44269504088896341f, 0.5f);
    const float fn = __builtin_floorf(fx);
    float r = __builtin_fmaf(fn, -0.693359375f, x);
    r = __builtin_fmaf(fn, 2.12194440e-4f, r);
    const float z = r * r;
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float y = __builtin_fmaf(p, z, r);
    y = y + 1.0f;
    const int n = (int)fn;
    const int n1 = n / 2, n2 = n - n1;
    const float s1 = __builtin_bit_cast(float, (uint32_t)(n1 + 127) << 23);
    const float s2 = __builtin_bit_cast(float, (uint32_t)(n2 + 127) << 23);
    y = y * s1;
    return y * s2;
}

// masked softmax without max-subtraction, eps 1e-6 (policy.rs:43-47 / 118-124), 4 actions
__host__ __device__ inline void masked_softmax4(const float l[4], uint32_t maskbits, float p[4])
{
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = ((maskbits >> i) & 1u) ? tw_expf(l[i]) : 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) sum = sum + p[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = p[i] / (sum + 0.000001f);
}

// nn::policy::sample (policy.rs:153-167) with rand 0.8.5 WeightedIndex semantics and an injected
// uniform: cumulative weights of the first n-1 entries, chosen = u*total, index = number of
// cumulative weights <= chosen; invalid weights -> 0 (the reference prints and returns 0).
__host__ __device__ inline int sample_weighted(const float *w, int n, float u)
{
    if (n <= 0) return 0;
    float total = 0.0f, cum[8];
    for (int i = 0; i < n; ++i) {
        if (!(w[i] >= 0.0f)) return 0;
        total = total + w[i];
        if (i < n - 1) cum[i] = total;
    }
    if (!(total > 0.0f)) return 0;
    const float chosen = u * total;
    int idx = 0;
    while (idx < n - 1 && cum[idx] <= chosen) ++idx;
    return idx;
}

// The same for at most four weights, without an indexed local array: `cum[idx]` with a run-time index puts the array in
// scratch (= device memory: a 1-2k-cycle round trip per access, and the tree walk of the self-play kernels calls this once
// per search).  Same operation sequence, same result.
__host__ __device__ inline int sample_weighted4(const float (&w)[4], int n, float u)
{
    if (n <= 0) return 0;
    if (n > 4) n = 4;
    float total = 0.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < n) {
            if (!(w[i] >= 0.0f)) bad = true;
            total = bad ? total : total + w[i];
            if (i == 0) c0 = total; else if (i == 1) c1 = total; else if (i == 2) c2 = total;
        }
    }
    if (bad || !(total > 0.0f)) return 0;
    const float chosen = u * total;
    int idx = 0;
    if (0 < n - 1 && c0 <= chosen) {
        idx = 1;
        if (1 < n - 1 && c1 <= chosen) {
            idx = 2;
            if (2 < n - 1 && c2 <= chosen) idx = 3;
        }
    }
    return idx;
}

// Gumbel-max over 4 masked logits with injected uniforms (policy.rs:130-151,169-172):
// argmax_i( l_i - ln(|ln(u_i)|) ), strict '>' => first max wins, NaN never wins.
__host__ __device__ inline int gumbel_argmax4(const float l[4], const u32x4 w)
{
    const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
    float g[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = tw_logf(u32_to_unit(ww[i]));
        const float b = __builtin_fabsf(a);
        g[i] = l[i] - tw_logf(b);
    }
    int best = 0; float bv = g[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) if (g[i] > bv) { bv = g[i]; best = i; }
    return best;
}

// ---- Puzzle state packed for the device: one nibble per cell (n_cells <= 16, tiles < 16) ----
struct PuzzleConsts {
    int32_t  width, height, n_cells;
    int32_t  difficulty;
    int32_t  depth0;      // depth_slope * difficulty (puzzle.rs:132)
    float    r_step;      // -0.5 / max_depth (puzzle.rs:175), computed once on the host in f32
    uint64_t ident;       // identity board: nibble i == i
};

struct PuzzleLane {
    uint64_t board;   // nibble i = tile at cell i
    int32_t  zx, zy;  // blank location (puzzle.rs:22)
    int32_t  depth;
};

__host__ __device__ inline uint32_t nib(uint64_t b, int i) { return (uint32_t)(b >> (4 * i)) & 15u; }

// Env::step (puzzle.rs:135-160): 0 left, 1 up, 2 right, 3 down; illegal = no-op; depth
// saturating_sub(1) always.
__host__ __device__ inline void puzzle_step(PuzzleLane &s, const PuzzleConsts &c, int action)
{
    // branch-free: an illegal (or unknown) action moves the blank onto itself, which leaves the board as it is
    const int dx = (action == 2 ? 1 : 0) - (action == 0 ? 1 : 0), dy = (action == 3 ? 1 : 0) - (action == 1 ? 1 : 0);
    int nx = s.zx + dx, ny = s.zy + dy;
    const bool ok = (unsigned)nx < (unsigned)c.width && (unsigned)ny < (unsigned)c.height;
    nx = ok ? nx : s.zx; ny = ok ? ny : s.zy;
    const int zi = s.zy * c.width + s.zx, ti = ny * c.width + nx;
    const uint64_t tile = (s.board >> (4 * ti)) & 15ull;                   // cell zi holds 0
    s.board = (s.board & ~(15ull << (4 * ti))) | (tile << (4 * zi));
    s.zx = nx; s.zy = ny;
    s.depth = s.depth > 0 ? s.depth - 1 : 0;
}

// Env::step for an action that is known to be legal (a tree child exists only where the mask allowed the move): the same
// board / blank / depth update without the bounds checks -- about half the scalar instructions on a wave-uniform state
__host__ __device__ inline void puzzle_step_legal(PuzzleLane &s, const PuzzleConsts &c, int action)
{
    const int zi = s.zy * c.width + s.zx;
    const int sh = 2 * action;                                  // dx + 1 = {0, 1, 2, 1}, dy + 1 = {1, 0, 1, 2} for left, up, right, down
    s.zx += ((0x64 >> sh) & 3) - 1; s.zy += ((0x91 >> sh) & 3) - 1;
    const int ti = s.zy * c.width + s.zx;
    const uint64_t tile = (s.board >> (4 * ti)) & 15ull;        // cell zi holds 0
    s.board = (s.board & ~(15ull << (4 * ti))) | (tile << (4 * zi));
    s.depth = s.depth > 0 ? s.depth - 1 : 0;
}

// Env::reset (puzzle.rs:119-133)
__host__ __device__ inline void puzzle_reset(PuzzleLane &s, const PuzzleConsts &c, uint64_t seed, uint64_t episode)
{
    s.board = c.ident; s.zx = 0; s.zy = 0; s.depth = 0;
    for (int d = 0; d < c.difficulty; ++d) {
        const u32x4 w = rng_draw(seed, episode, (uint32_t)d, STREAM_SCRAMBLE);
        puzzle_step(s, c, (int)u32_below(w.x, 4u));
    }
    s.depth = c.depth0;
}

// lowest zero nibble of a board = the blank cell (the classic zero-byte trick flags only true zeros below the first borrow)
__host__ __device__ inline int blank_cell(uint64_t b)
{
    const uint64_t m = (b - 0x1111111111111111ull) & ~b & 0x8888888888888888ull;
    return (int)(__builtin_ctzll(m) >> 2);
}

__host__ __device__ inline bool  puzzle_solved(const PuzzleLane &s, const PuzzleConsts &c) { return s.board == c.ident; }
__host__ __device__ inline bool  puzzle_final(const PuzzleLane &s, const PuzzleConsts &c) { return s.depth == 0 || s.board == c.ident; }
__host__ __device__ inline float puzzle_reward(const PuzzleLane &s, const PuzzleConsts &c)
{
    return s.board == c.ident ? 1.0f : (s.depth == 0 ? -0.5f : c.r_step);   // puzzle.rs:171-177
}
// Env::masks (puzzle.rs:162-165) as bit i = action i allowed
__host__ __device__ inline uint32_t puzzle_maskbits(const PuzzleLane &s, const PuzzleConsts &c)
{
    return (s.zx > 0 ? 1u : 0u) | (s.zy > 0 ? 2u : 0u) | (s.zx < c.width - 1 ? 4u : 0u) | (s.zy < c.height - 1 ? 8u : 0u);
}

// one Linear of a generic policy stack (EngineV, tw_engine_generic.hpp): weights in the reference's export layout
struct LayerDev {
    const float *w;      // [in][out] row-major == torch_weight.T.flatten() (layers.rs:26); out padded to a multiple of 4 (policy_eval kernels)
    const float *b;      // [nb * tb * 16]: bias, 0 beyond the layer's outputs
    int32_t in, out, relu, pad;
    // matrix-core image (EngineV): [kg * 4][nb][16][tb] floats, element (k, b, i, t) = W[k][(b * tb + t) * 16 + i], -0.0 where
    // k >= in or the output does not exist (fma(-0.0, x, acc) == acc for every finite x: padding is an exact identity)
    const float *wm;
    int32_t kg, nb, tb, pad2;     // k-groups of four inputs; blocks of tb 16-output tiles (tb = 4 from 64 outputs up, else all tiles in one block)
};

// ---- device-side policy image (built once by tw_policy_create) ------------------------------
struct PolicyDev {
    int32_t obs_size, emb, hidden, n_actions, n_perms;
    int32_t emb_relu, common_relu;
    // f32 image ("exact" mode)
    const float *emb_rows;   // [(obs_size+2)][emb]: rows 0..obs_size-1 vectors, row obs_size = bias, row obs_size+1 = 0 (evaluate kernel)
    const float *w1p;        // [emb][NQ][32][4]: W1[k][hid(4q+c, i)] (MFMA A-operand image, see tw_rollout.hip)
    const float *t_img16;    // [emb/16][21*256]: table chunk images exactly as they sit in LDS (Engine3, see tw_engine.hpp)
    const float *b1;         // [hidden] natural order
    const float *wh8;        // [hidden][8]: cols 0..3 action weights, col 4 value weight, rest 0
    const float *bh8;        // [8]: action bias 0..3, value bias 4
    // natural-layout copies for the generic evaluate kernel
    const float *w1;         // [emb][hidden]
    const float *wa;         // [hidden][n_actions]
    const float *ba;         // [n_actions]
    const float *wv;         // [hidden]
    const float *bv;         // [1]
    const uint8_t *obs_perms; // [n_perms][obs_size] (ids < 256)
    const uint8_t *act_perms; // [n_perms][n_actions]
    // f16-input image (TW_PREC_F16, Engine16 in tw_engine16.hpp); f16_nc == 0: not available for this policy
    int32_t f16_nc;           // table chunks per stage (n_cells padded to 4 / 9 / 16)
    const uint8_t *stage16;   // [emb/32][(f16_nc + 2*hidden/32) KiB]: per 32 embedding dims, table chunks then W1 chunks (MFMA A operands)
    const uint8_t *head16;    // [hidden/32][2][1 KiB]: head A operands
    const float   *ebias16;   // [emb/32][2][16]: embedding bias in accumulator-register order per lane half
    const float   *b1img16;   // [hidden/32][2][16]
    const float   *bh16;      // [8]: action bias 0..3, value bias 4
    const uint8_t *srcmap16;  // [(n_perms+1)][16]: source cell of target chunk c under twist p-1 (p = 0: identity)
    const uint8_t *vmap16;    // [(n_perms+1)][16][16]: value of the twisted id inside chunk c, 15-bit clean (< 16)
    // split-f16 image (TW_PREC_F16X2, EngineS in tw_engine16x2.hpp): operands x16, hi / lo binary16 terms
    const uint8_t *stageS;    // [2*emb/32 + 1][SP KiB]: stage 2k = [T_hi(k+1) | W1_hi(k)], 2k+1 = lo terms, last = [head_hi | head_lo]
    const uint8_t *t0S;       // [2*f16_nc KiB]: table tile 0, hi chunks then lo chunks
    // generic stacks (any Sequential depth; EngineV): layers = common | action | value; hidden == 0 marks such a policy
    int32_t generic, n_common, n_action, n_value, value_out;
    int32_t gen_rows0, gen_rows1, gen_rows2;   // rows ([unit][column]) of EngineV's three activation buffers: the widest layer each one ever holds
    const LayerDev *layers;
    const uint16_t *obs_perms16;   // [n_perms][obs_size] for obs_size > 256 (environments other than Puzzle; the evaluate kernels)
};

// padded (episode-major) trajectory workspace written by the rollout / MCTS kernels: ONE 48-byte
// record per (episode, t), so every episode has a single active cache line that the XCD's L2 merges
// completely before it is written back (six separate arrays measured 4.6x write amplification).
struct __attribute__((aligned(16))) PaddedRec {
    uint8_t obs[16];     // obs ids, zero padded to 16
    float   logits[4];   // PPO: masked logits; AZ: MCTS probs
    float   value;       // PPO only
    float   reward;      // env.reward() of the recorded state
    uint8_t action;      // PPO only
    int8_t  perm;        // -1 = None
    uint8_t pad[6];
};
static_assert(sizeof(PaddedRec) == 48, "PaddedRec must be 48 bytes");

struct PaddedTraj {
    PaddedRec *rec;      // [E][t_pad]
    uint32_t  *ep_len;   // [E]
    int32_t    t_pad;
};

__device__ inline void store_rec(PaddedRec *dst, const uint32_t (&obs4)[4], const float (&lg)[4], float value,
                                 float reward, int action, int perm)
{
    uint4 *d = reinterpret_cast<uint4 *>(dst);
    d[0] = make_uint4(obs4[0], obs4[1], obs4[2], obs4[3]);
    d[1] = make_uint4(__builtin_bit_cast(uint32_t, lg[0]), __builtin_bit_cast(uint32_t, lg[1]),
                      __builtin_bit_cast(uint32_t, lg[2]), __builtin_bit_cast(uint32_t, lg[3]));
    d[2] = make_uint4(__builtin_bit_cast(uint32_t, value), __builtin_bit_cast(uint32_t, reward),
                      (uint32_t)(action & 0xff) | ((uint32_t)(perm & 0xff) << 8), 0u);
}

// four nibbles (16 bits) -> four bytes
__host__ __device__ inline uint32_t spread_nibbles4(uint32_t t16)
{
    const uint32_t x = (t16 | (t16 << 8)) & 0x00FF00FFu;
    return (x | (x << 4)) & 0x0F0F0F0Fu;
}
// obs ids of a board as 16 bytes: byte i = i*n_cells + tile(i) for i < n_cells, else 0 (puzzle.rs:183-185).
// base[q] holds the four constants i*n_cells of word q (obs_base_words()).
__host__ __device__ inline void obs_bytes(uint64_t board, const uint32_t (&base)[4], uint32_t (&pk)[4])
{
    const uint32_t lo = (uint32_t)board, hi = (uint32_t)(board >> 32);
    pk[0] = spread_nibbles4(lo & 0xFFFFu) + base[0]; pk[1] = spread_nibbles4(lo >> 16) + base[1];
    pk[2] = spread_nibbles4(hi & 0xFFFFu) + base[2]; pk[3] = spread_nibbles4(hi >> 16) + base[3];
}
__host__ __device__ inline void obs_base_words(int n_cells, uint32_t (&base)[4])
{
    for (int q = 0; q < 4; ++q) {
        base[q] = 0u;
        for (int b = 0; b < 4; ++b) { const int i = 4 * q + b; if (i < n_cells) base[q] |= (uint32_t)(i * n_cells) << (8 * b); }
    }
}

// compact output
struct CompactTraj {
    uint8_t *obs;      // [n][n_cells]
    float   *logits;   // [n][4]
    int8_t  *perms;
    float   *values, *rewards;
    uint8_t *actions;
    float   *advs, *rets;
};

struct RolloutArgs {
    PuzzleConsts env;
    PolicyDev    pol;
    PaddedTraj   out;
    uint64_t     num_episodes, episode_offset, seed;
    // persistent-lane mode (f32 kernel, more episodes than resident lanes): a lane whose episode is over takes the next
    // one from `queue` -- ragged episode lengths then cost the MEAN length, as with the reference's work stealing
    // (ppo.rs:110-124), not the maximum of every 256-episode workgroup.  Start boards come from init_boards_kernel.
    const uint4    *init_boards;   // [num_episodes] the episodes in the order the lanes take them: {start board lo, hi, episode, 0}, or null
                                   //   (one array, one pointer: a second one for the order cost the 8-wave kernel 0.6 %)
    unsigned int   *queue;         // next unassigned entry of init_boards, or null
    int32_t         reserve_cus;   // persistent mode: CUs left without a workgroup (room for RCCL's send/recv kernels, dist.py)
};

// Waves per workgroup of the f32 engine for a batch of n columns (episodes / attempts): 8 (two per SIMD, 256 columns) is
// the throughput geometry.  Below ~40k columns the small-batch geometries win (measured crossover of the rollout,
// scripts/geom_sweep.py): 32 columns per workgroup -- shared by four waves that split the hidden units (Engine3S,
// tw_engine.hpp: geometry_for()) where the policy has >= 64 hidden units, else 64- or 32-column workgroups of this engine.
inline int waves_per_group(uint64_t n)
{
    if ((n + 255) / 256 >= 156) return 8;
    if ((n + 63) / 64 >= 192) return 2;
    return 1;
}

// kernel launchers (each returns a TW_* status)
int launch_rollout_f32(const RolloutArgs &a, hipStream_t s, uint32_t *blocks, uint32_t *threads);
uint64_t rollout_generic_resident_episodes(const PolicyDev &pol, int n_cells, int reserve_cus);   // generic stacks: episodes in flight in the persistent launch
// boards of 17 .. 25 cells (tw_rollout_big.hip): obs ids as uint16 in their own padded array [E][t_pad][n_cells]
int launch_rollout_big(const RolloutArgs &a, uint16_t *obs16, hipStream_t s, uint32_t *blocks, uint32_t *threads);
int launch_compact_obs16(const uint16_t *obs16, const uint32_t *ep_len, const uint64_t *ep_start, uint64_t E, int t_pad, int n_cells,
                         uint16_t *out, hipStream_t s);
int launch_init_boards(const PuzzleConsts &env, uint64_t seed, uint64_t episode_offset, uint64_t n, uint64_t *out, hipStream_t s, uint4 *entries = nullptr);   // entries: {board, episode i} in index order
// episodes by decreasing distance of their start board from the solved one (sum of the tiles' Manhattan distances): the order the
// self-play walkers take them in -- the episodes that are likely to run to the depth limit start first (tw_rollout.hip)
size_t episode_order_scratch_bytes(uint64_t n);
int launch_episode_order(const PuzzleConsts &env, const uint64_t *boards, uint64_t n, uint32_t *order, void *scratch, hipStream_t s, uint4 *entries = nullptr);   // order and / or entries {board, episode} in that order
uint64_t rollout_f32_resident_episodes(int reserve_cus = 0);   // episodes the f32 rollout keeps resident at once (persistent mode above that)
// Lanes the exact-f32 kernels (rollout, self-play) keep resident for a batch: episodes beyond that wait in the queue of the
// persistent-lane mode.  CUs x 256 for the 256-episode shape; between CUs x 32 and 3/4 of that the small-batch shape
// (CUs x 32 lanes, Engine3S) with the queue -- see tw_rollout.hip.
uint64_t f32_resident_episodes(uint64_t num_episodes, int hidden, bool selfplay, int reserve_cus = 0);
int launch_rollout_f16(const RolloutArgs &a, hipStream_t s, uint32_t *blocks, uint32_t *threads);
int launch_rollout_f16x2(const RolloutArgs &a, hipStream_t s, uint32_t *blocks, uint32_t *threads);
int launch_scan(const uint32_t *ep_len, uint64_t n_episodes, int merge_order, uint64_t *ep_start,
                uint64_t *total /*device*/, void *scratch, size_t scratch_bytes, hipStream_t s);
size_t scan_scratch_bytes(uint64_t n_episodes);
int launch_finalize_ppo(const PaddedTraj &in, const uint64_t *ep_start, uint64_t n_episodes, int n_cells,
                        float gamma, float lambda, const CompactTraj &out, hipStream_t s);
struct SolveArgs {
    PuzzleConsts env;
    PolicyDev    pol;
    uint64_t     num_attempts;     // episodes x num_searches
    uint64_t     episode_offset, seed;
    uint32_t     num_searches, deterministic;
    uint32_t     from_state;       // 1: every attempt starts from (start_board, start_depth) -- solve()
    uint64_t     start_board; int32_t start_zx, start_zy, start_depth;
    const uint8_t *start_cells;    // boards above 16 cells: the start state, one byte per cell (device memory)
    float       *success;          // [num_attempts] 1.0 / 0.0
    float       *total;            // [num_attempts] summed rewards (solve.rs:25-34,65-66)
    uint32_t    *n_steps;          // [num_attempts]
    uint8_t     *actions;          // [num_attempts][t_pad] or null
    int32_t      t_pad;
};
int launch_solve_f32(const SolveArgs &a, hipStream_t s);
int launch_solve_big(const SolveArgs &a, hipStream_t s);       // evaluate() of boards of 17 .. 25 cells (tw_rollout_big.hip)

struct MctsNode;   // tw_mcts.hip
// MCTS-guided inference (solve.rs:41-47): the MCTS kernel runs single_solve instead of AZ self-play when on != 0
struct MctsSolve {
    uint32_t on, deterministic, num_searches /* attempts per episode */, from_state;
    uint64_t start_board; int32_t start_zx, start_zy, start_depth;
    const uint8_t *start_cells;    // boards above 16 cells: the start state, one byte per cell (device memory)
    float   *success, *total;      // [attempts]
    uint32_t *n_steps;             // [attempts]
    uint8_t *actions;              // [attempts][act_pad] or null
    int32_t  act_pad;
};
struct MctsArgs {
    PuzzleConsts env;
    PolicyDev    pol;
    PaddedTraj   out;          // obs, logits (= MCTS probs), rewards (= env.reward() per record), ep_len
    uint64_t     num_episodes, episode_offset, seed;
    uint32_t     num_searches, max_expand_depth;
    float        C;
    MctsNode    *arena;        // [num_episodes][node_cap]
    uint32_t     node_cap;
    unsigned long long *eval_count;   // [0] policy evaluations the searches consumed (leaf + root), [1] speculative ones (deep shape),
                                      // [2] of [0]: outputs taken from the grandparent (a move taken back: same board)
                                      // [3..11] TW_OPT_AZ_REUSE = 4 diagnostics, [12] reused outputs that failed the board check (16 entries)
    MctsSolve    solve;        // on == 0: AlphaZero self-play (records into `out`)
    // persistent-lane mode (self-play with more episodes than resident lanes, see RolloutArgs): the arena is then
    // [resident lanes][node_cap], a lane reuses its arena for every episode it takes
    const uint64_t *init_boards;
    unsigned int   *queue;
    int32_t         reserve_cus;
    uint32_t        lds_nodes;     // deep shape: nodes per tree whose statistics live in LDS (set by the launcher)
    uint32_t        tree_budget;   // deep shape: cycles of tree walk per trip after which a walker stops at the next search boundary (launcher)
    uint32_t        tree_budget_min;   // ... after which it stops there as soon as another walker of the workgroup waits for a forward
    void           *tbl;               // deep shape: board-keyed output tables, [walkers][tbl_entries][32 bytes], zeroed before the launch
    uint32_t        tbl_entries;       // ... entries per walker (a power of two)
    uint32_t        reuse_mode;        // lane-per-episode kernel: how a node that takes its parent's move back finds its grandparent's output (TW_OPT_AZ_REUSE)
    const uint32_t *order;             // deep shape: the order in which the walkers take the episodes (launch_episode_order), or null = by index
    const MctsSolve *solve_dev;        // deep shape in solve mode: a copy of `solve` in device memory (null: self-play) -- the walker kernel has no registers to
                                       //   keep twenty more launch constants in; it reads them where a move or an attempt ends
    uint32_t        order_across;      // ... the first ones dealt out across the workgroups (walker w of workgroup b: number w * workgroups + b) instead of in a row
    // deep shape, split form (walkers and engine as two kernels, tw_mcts_deep.hip): one 256-byte mailbox per walker, zeroed before the launch;
    // the engine workgroup e serves the walkers [e * split_wpe, (e + 1) * split_wpe) of the split_walkers there are
    uint32_t       *mailbox;
    uint32_t        split_walkers, split_wpe;
};
size_t mcts_node_bytes();
// the deep shape of self-play (tw_mcts_deep.hip): one wave per episode, 64-byte nodes, persistent walkers + episode queue
bool     mcts_deep_applies(const MctsArgs &a);
uint64_t mcts_deep_walkers(uint64_t num_episodes, int reserve_cus, uint32_t num_searches, bool solve = false);   // tree arenas = episodes in flight
bool     mcts_deep_split(uint64_t num_episodes, int reserve_cus, uint32_t num_searches, bool solve = false);     // the split form: MctsArgs::mailbox is needed (256 bytes per walker)
void     mcts_deep_disable_split();        // this process stops taking the split form (after its watchdog fired: tw_az_collect)
size_t   mcts_deep_node_bytes();
// bytes of one walker's tree arena (72 per node, see tw_mcts_deep.hip), a multiple of 16
__host__ __device__ inline size_t mcts_deep_arena_bytes(uint64_t node_cap) { return (size_t)((node_cap * 72 + 15) / 16 * 16); }
uint32_t mcts_deep_table_entries(uint32_t num_searches, uint32_t max_expand_depth);     // per walker, 32 bytes each
int      launch_mcts_deep(const MctsArgs &a, hipStream_t s, uint32_t *blocks, uint32_t *threads);
int launch_mcts_f32(const MctsArgs &a, hipStream_t s, uint32_t *blocks, uint32_t *threads);
// self-play of boards of 17 .. 64 cells (tw_mcts_big.hip): 32-byte nodes without a board, obs ids as for launch_rollout_big
size_t mcts_big_node_bytes();
int launch_mcts_big(const MctsArgs &a, uint16_t *obs16, hipStream_t s, uint32_t *blocks, uint32_t *threads);
int launch_finalize_az(const PaddedTraj &in, const uint64_t *ep_start, uint64_t n_episodes, int n_cells,
                       uint8_t *obs_out, float *probs_out, int8_t *perms_out, float *remaining_out, hipStream_t s);
int launch_onehot(const uint8_t *obs, uint64_t row0, uint64_t rows, int n_cells, int obs_size, float *out, hipStream_t s);
int launch_ppo_pack(const float *logits, const uint8_t *actions, const int8_t *perms, const float *advs, uint64_t row0, uint64_t rows,
                    int n_actions, float mean, float denom, int normalize, float *logp_out, int64_t *acts_out, int64_t *perms_out,
                    float *advs_out, hipStream_t s);
size_t sum_scratch_doubles();
int launch_sum(const float *x, uint64_t n, double shift, int squared, double *scratch /* sum_scratch_doubles(); result in [0] */, hipStream_t s);
// device-to-device policy sync (tw_sync.hip)
struct SyncArgs {
    // torch-layout sources (device)
    const float *emb_w, *emb_b;   // [E][OS], [E]
    const float *w1, *b1;         // [H][E], [H]
    const float *wa, *ba;         // [A][H], [A]
    const float *wv, *bv;         // [1][H], [1]
    int OS, E, H, A, NT, NQ, n16, nc16, SP16, NKT;
    // destinations (the policy's arena)
    float *emb_rows, *w1p, *t_img16, *b1_d, *wh8, *bh8, *w1_nat, *wa_nat, *ba_nat, *wv_nat, *bv_nat;
    uint8_t *stage16, *head16; float *ebias16, *b1img16, *bh16;
    uint8_t *stageS, *t0S; int SPS;     // split-f16 images (tw_engine16x2.hpp); SPS = KiB per stage
    // element counts per segment (prefix sums in seg_end)
    unsigned long long seg_end[18];
};

int launch_policy_sync(const SyncArgs &a, hipStream_t s);

// the same for policies of any Sequential depth (EngineV's images): layer l of [common.. | action.. | value..]
constexpr int GEN_SYNC_MAX_LAYERS = 24;
struct GenSyncArgs {
    const float *emb_w, *emb_b;                    // torch layout: [E][OS], [E]
    float *emb_rows; int OS, E, n_layers;
    const float *w[GEN_SYNC_MAX_LAYERS], *b[GEN_SYNC_MAX_LAYERS];      // torch layout: [out][in], [out]
    float *w_nat[GEN_SYNC_MAX_LAYERS], *b_img[GEN_SYNC_MAX_LAYERS], *wm[GEN_SYNC_MAX_LAYERS];
    int in[GEN_SYNC_MAX_LAYERS], out[GEN_SYNC_MAX_LAYERS], outp[GEN_SYNC_MAX_LAYERS], kg[GEN_SYNC_MAX_LAYERS], nb[GEN_SYNC_MAX_LAYERS], tb[GEN_SYNC_MAX_LAYERS];
    unsigned long long seg_end[1 + 3 * GEN_SYNC_MAX_LAYERS];           // embedding rows | per layer: natural weights, bias image, matrix-core image
};
int launch_policy_sync_generic(const GenSyncArgs &a, hipStream_t s);

// tw_api.hip internals the exchange (tw_comm.hip) and the generic-env collector (tw_env_generic.hip) work on
int require_device();
const PolicyDev *policy_dev(const tw_policy *p);
void collected_adopt_obs_width(tw_collected *c, uint32_t obs_width);
int policy_device_image(tw_policy *p, void **image, size_t *bytes);       // the one allocation holding every weight image
int policy_restore_local_tables(tw_policy *p, hipStream_t s);             // ... and what in it is process-local (pointers)
int collected_describe(const tw_collected *c, int *is_ppo, uint32_t *n_cells, uint32_t *n_actions, uint64_t *n_records, uint64_t *n_episodes);
const void *collected_field(const tw_collected *c, int field);
// wraps device memory the caller allocated with hipMalloc into a result object (which frees it through the arena pool)
int collected_adopt(void *arena, size_t arena_bytes, int device, int is_ppo, uint32_t n_cells, uint32_t n_actions, uint64_t n_records,
                    uint64_t n_episodes, void *const (&field_ptr)[TW_F_COUNT], const size_t (&field_bytes)[TW_F_COUNT], tw_collected **out);
int launch_policy_eval(const PolicyDev &pol, int mode, const int32_t *obs_d, uint32_t n, uint32_t n_obs,
                       const uint8_t *masks_d, const int32_t *perms_d, float *out_actions_d, float *out_values_d,
                       hipStream_t s);

}  // namespace tw
