// tw_engine.hpp -- the per-workgroup policy-forward engine shared by the rollout (PPO), MCTS
// (AlphaZero) and solve/evaluate kernels: EmbeddingBag gather + common Linear on f32 MFMA + both heads,
// with both weight streams running through a ring of LDS slots.  See tw_rollout.hip for the design notes.
#pragma once
#include <type_traits>
#include "tw_common.hpp"
#include "tw_engine_generic.hpp"
#include <cstdlib>

namespace tw {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const float lds_cfloat;

constexpr int EPW = 32;            // episodes per wave (MFMA columns)

// LDS-DMA (Engine3::stream_op): one 1-KiB piece per instruction, every lane copies 16 bytes global -> LDS (destination =
// wave-uniform base in M0 + lane*16).  Issued through inline asm on purpose: the builtin form is FLAT-encoded with an LDS
// memory operand, which makes hipcc treat it as a pending flat access and degrade EVERY later LDS wait to lgkmcnt(0) until
// the DMA has been waited for.  The copy is waited for with the explicit vmcnt(0) in front of the chunk barrier.

// ReLU as ONE v_max_f32 (hipcc lowers `x > 0 ? x : 0` to a canonicalising v_max plus the v_max).
// IEEE mode: max(0, NaN) = 0, like the reference's `if x > 0.0 { x } else { 0.0 }` (layers.rs:89-91).
// The result feeds an MFMA operand; hipcc pads no hazard wait states for an asm output, so the
// VALU-write -> MFMA-read states (s_nop 1) are inside the string.
__device__ __forceinline__ float relu1(float x)
{
    float y;
    asm("v_max_f32 %0, 0, %1\n\ts_nop 1" : "=v"(y) : "v"(x));
    return y;
}
// Branch-free optional ReLU: max(lim, x) with lim = 0.0f (ReLU) or -inf (identity) in an SGPR.
__device__ __forceinline__ float relu_lim(float x, float lim)
{
    float y;
    asm("v_max_f32 %0, %1, %2\n\ts_nop 1" : "=v"(y) : "s"(lim), "v"(x));
    return y;
}

// Two independent f32 adds.  NOT v_pk_add_f32: beside MFMAs a packed f32 VALU instruction costs far more than the two plain
// adds it replaces (MI355X_MICROARCH.md, "price of one filler beside MFMAs": +13 cycles each; measured here: the rollout
// kernel is 3 % faster with plain adds) -- which is why hipcc's pre-emit peephole un-packs it in an MFMA's shadow.
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b)
{
    f32x2 r;
    r[0] = a[0] + b[0]; r[1] = a[1] + b[1];
    return r;
}

// the same max, for a result that does not feed an MFMA (no wait states)
__device__ __forceinline__ float relu_lim_v(float x, float lim)
{
    float y;
    asm("v_max_f32 %0, %1, %2" : "=v"(y) : "s"(lim), "v"(x));
    return y;
}

#ifdef TW_ABLATE   // timing-only switches of the diagnostic build, per translation unit (TW_ENG_DBG: 4 = no weight / table streams)
static __device__ int g_eng_dbg;
#endif

template <int NT> struct Tiles { static constexpr int NQ = (NT + 3) / 4; };

constexpr int MAX_LDS_PERMS = 4;   // twist tables kept in LDS (more twists fall back to global reads)

// MFMA row i of row-tile r carries hidden unit hid(r,i) = 32r + 2g + h with
// g = (i&3) + 4*(i>>3), h = (i>>2)&1: the C/D layout (row = (g&3) + 8*(g>>2) + 4h for
// accumulator register g on lane half h) then holds hidden unit 32r + 2g + h in register g.
// (tw_api.hip builds the W1 image [k][q][i][4] = W1[k][hid(4q+c, i)] with the same formula.)
//
// DBG != 0 builds are timing-only ablations (wrong results): 1 no gather, 2 no A-operand reads,
// 4 no weight streams, 8 no heads.  Never used by the product path.

// =====================================================================================================
// Engine3: the policy-forward engine.  Ring of three LDS slots, 16-column chunks; BOTH weight streams are pure LDS-DMA (the table
// chunk comes from an image that tw_policy_create lays out exactly as it sits in LDS: column
// permutation [even k | odd k], rows padded to 20 floats, slot padded to 21 KiB), so there is no
// register staging and no ds_write commit.  With three slots the chunk c+1 is complete and published
// one barrier before it is needed: the first gather group and the first A operands of chunk c+1 are
// prefetched during the last group of chunk c, so nothing is exposed behind the chunk barrier.
// LDS: W[3][4096] | T[3][5376] | b1 | wh | bh8 | twists  (~122 KB), one 512-thread workgroup per CU.
// =====================================================================================================
constexpr int R3_KC    = 16;
constexpr int R3_LSTR  = R3_KC + 4;         // 20 floats per row
constexpr int R3_TSLOT = 21 * 256;          // floats per table slot (21 KiB = 21 DMA pieces)

template <int NT>
__host__ __device__ inline size_t engine3_lds_floats(int obs_size)
{
    return (size_t)3 * R3_KC * Tiles<NT>::NQ * 128 + (size_t)3 * R3_TSLOT + (size_t)NT * 32 * 10 + 8 +
           (size_t)MAX_LDS_PERMS * ((obs_size + 3) / 4 + 1) + (size_t)NT * 32 * 5;
}

// NW = waves per workgroup (8: two per SIMD, the throughput geometry; 2 / 1: small batches, so that a few thousand
// episodes still spread over many CUs -- each workgroup streams the whole weight set either way).
template <int NT, int NC, int DBG = 0, int NW_ = 8>
struct Engine3 {
    static constexpr int NW = NW_, KC = R3_KC, THREADS = 64 * NW_, EPB = NW * EPW, LSTR = R3_LSTR;
    static constexpr int NQ     = Tiles<NT>::NQ;
    static constexpr int WSLOT  = KC * NQ * 128;            // floats per W1 slot
    static constexpr int WPIECE = WSLOT / 256;              // DMA pieces per W1 chunk
    // DMA pieces per table chunk: only the rows a board of <= NC cells can address (NC*NC ids + bias row + zero row) are
    // streamed -- 21 pieces for Puzzle-15, 7 for Puzzle-8 (the image and the LDS slot keep the full 21 KiB stride)
    static constexpr int TPIECE = ((NC * NC + 2) * R3_LSTR * 4 + 1023) / 1024;
    static_assert(TPIECE * 256 <= R3_TSLOT, "table chunk larger than its slot");
    static constexpr int NPIECE = WPIECE + TPIECE;
    static constexpr int NOPS   = (NPIECE + NW - 1) / NW;   // DMA ops per wave per chunk
    static constexpr int M      = 4 * NT;                   // MFMAs per group of four k-steps
    static constexpr bool SPLIT = false;

    __host__ __device__ static size_t lds_floats(int obs_size) { return engine3_lds_floats<NT>(obs_size); }
    __host__ __device__ static size_t lds_floats(const PolicyDev &p) { return lds_floats(p.obs_size); }
    __device__ __forceinline__ bool primary() const { return true; }           // this wave owns its episodes' stores
    __device__ __forceinline__ int  ep_lane() const { return wave * EPW + j; }  // episode index inside the workgroup
    __device__ __forceinline__ bool owns_lane() const { return true; }          // per-episode serial work (MCTS tree) of lane j runs here

    PolicyDev pol;
    int tid, lane, wave, j, h;
    uint32_t voff;                                          // lane*16: per-lane byte offset inside a DMA piece
    int bias_row, zero_row, n_chunks, rp;                   // rp: ring slot of chunk 0 of the next forward
    float emb_lim, common_lim;                              // 0 (ReLU) or -inf (none): relu_lim()
    float *lds_w, *lds_t, *lds_b1, *lds_wh, *lds_bh, *lds_wn;   // lds_wn: head weights in natural order [output 0..4][hidden]
    const uint8_t *perm_obs, *perm_act;

    // DMA op `op` of this wave for chunk `chunk` (of the flat chunk sequence) into ring slot `slot`: piece wave + NW*op of
    // [W1 pieces | table pieces].  Scalar source (SGPR base + lane*16 in a VGPR that never changes) and scalar destination (M0).
    // WPIECE is a multiple of NW, so an op's role (W1 or table) is a compile-time fact once the op loop is unrolled, and
    // everything wave-dependent is folded into dsrc_* / ddst_* at begin1: an op costs two 64-bit scalar adds, one 32-bit add
    // and the M0 write (the generic form spent 9 scalar instructions per op, which a lone wave per SIMD cannot hide).
    static_assert(WPIECE % NW == 0, "W1 pieces per chunk must be a multiple of the wave count");
#ifdef TW_ABLATE
    int eng_dbg = 0;
#endif
    const uint8_t *dsrc_w, *dsrc_t;        // image bases (+ this wave's first W1 piece)
    uint32_t ddst_w, ddst_t;               // LDS byte addresses of ring slot 0 (+ this wave's first W1 piece)
    __device__ __forceinline__ void stream_op(int chunk, int slot, int op)
    {
        if constexpr (DBG & 4) return;
#ifdef TW_ABLATE
        if (eng_dbg & 4) return;
#endif
        if (NW * op < WPIECE) {
            const uint8_t *src = dsrc_w + (size_t)chunk * (WSLOT * 4) + (size_t)op * (NW * 1024);
            const uint32_t dst = ddst_w + (uint32_t)slot * (WSLOT * 4) + (uint32_t)op * (NW * 1024);
            TW_GLDS16(voff, dst, src);
        } else {
            const int T0 = NW * op - WPIECE;                                 // table piece of wave 0
            uint32_t tp = (uint32_t)wave + (uint32_t)T0;
            if (T0 + NW - 1 >= TPIECE) tp = tp < (uint32_t)TPIECE ? tp : (uint32_t)TPIECE - 1u;   // past the end: repeat the last piece
            const uint8_t *src = dsrc_t + (size_t)chunk * (R3_TSLOT * 4) + (size_t)tp * 1024;
            const uint32_t dst = ddst_t + (uint32_t)slot * (R3_TSLOT * 4) + tp * 1024u;
            TW_GLDS16(voff, dst, src);
        }
    }

    __device__ __forceinline__ void begin1(const PolicyDev &p, float *lds)
    {
        pol = p;
#ifdef TW_ABLATE
        eng_dbg = __builtin_amdgcn_readfirstlane(g_eng_dbg);
#endif
        tid  = threadIdx.x;
        lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        j = lane & 31; h = lane >> 5;
        voff = (uint32_t)lane * 16u;
        bias_row = pol.obs_size; zero_row = pol.obs_size + 1;
        n_chunks = pol.emb / KC;
        emb_lim = pol.emb_relu ? 0.0f : -__builtin_inff();
        common_lim = pol.common_relu ? 0.0f : -__builtin_inff();
        lds_w  = lds;
        lds_t  = lds + 3 * WSLOT;
        lds_b1 = lds_t + 3 * R3_TSLOT;
        lds_wh = lds_b1 + NT * 32;
        lds_bh = lds_wh + NT * 32 * 9;
        dsrc_w = reinterpret_cast<const uint8_t *>(pol.w1p) + (size_t)wave * 1024;
        dsrc_t = reinterpret_cast<const uint8_t *>(pol.t_img16);
        ddst_w = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)lds_w) + (uint32_t)wave * 1024u;
        ddst_t = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)lds_t);
        for (int i = tid; i < NT * 32; i += THREADS) lds_b1[(i & 1) * (NT * 16) + (i >> 1)] = pol.b1[i];
        for (int i = tid; i < NT * 32 * 8; i += THREADS) {
            const int n = i >> 3, c = i & 7;
            lds_wh[(c * 2 + (n & 1)) * (NT * 16) + (n >> 1)] = pol.wh8[i];
        }
        for (int i = tid; i < NT * 32; i += THREADS) lds_wh[8 * 2 * (NT * 16) + i] = 0.0f;
        if (tid < 8) lds_bh[tid] = pol.bh8[tid];
        lds_wn = lds_bh + 8 + MAX_LDS_PERMS * ((pol.obs_size + 3) / 4 + 1);
        for (int i = tid; i < NT * 32 * 5; i += THREADS) {
            const int c = i / (NT * 32), n = i - c * (NT * 32);
            lds_wn[i] = pol.wh8[n * 8 + c];
        }
        perm_obs = pol.obs_perms; perm_act = pol.act_perms;
        if (pol.n_perms > 0 && pol.n_perms <= MAX_LDS_PERMS) {
            uint8_t *po = reinterpret_cast<uint8_t *>(lds_bh + 8);
            uint8_t *pa = po + MAX_LDS_PERMS * ((pol.obs_size + 3) / 4) * 4;
            for (int i = tid; i < pol.n_perms * pol.obs_size; i += THREADS) po[i] = pol.obs_perms[i];
            for (int i = tid; i < pol.n_perms * 4; i += THREADS) pa[i] = pol.act_perms[i];
            perm_obs = po; perm_act = pa;
        }
        // chunks 0 and 1 of the first forward into slots 0 and 1
#pragma unroll
        for (int op = 0; op < NOPS; ++op) { stream_op(0, 0, op); stream_op(n_chunks > 1 ? 1 : 0, 1, op); }
        rp = 0;
    }
    __device__ __forceinline__ void begin2() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __device__ __forceinline__ void end() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

    __device__ __forceinline__ void rows_of(uint64_t board, int n_cells, int perm, int (&rowoff)[NC]) const
    {
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            int row = zero_row;
            if (i < n_cells) {
                const int id = i * n_cells + (int)nib(board, i);
                row = perm >= 0 ? (int)perm_obs[perm * pol.obs_size + id] : id;
            }
            rowoff[i] = row * LSTR + h * (KC / 2);
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

    __device__ __forceinline__ void finish_b(f32x4 &b) const
    {
#pragma unroll
        for (int u = 0; u < 4; ++u) b[u] = relu_lim(b[u], emb_lim);
    }

    // One group of 4*NT MFMAs with B operands `bq`, while (a) the gather of the NEXT group is read
    // through the ONE set of per-row LDS pointers `ga` (+ 4*ng floats) and summed into bnext, (b) the A
    // operands of the following k-steps are fetched (after the group's last k-step: k-step kp0+4 of the
    // same slot, or k-step 0 of the next slot `wn_base` when LAST), (c) DMA ops [op0, op1) are issued.
    template <bool HAVE_NEXT_GATHER, bool LAST, bool CROSS>
    __device__ __forceinline__ void group(f32x16 (&acc)[NT], f32x4 (&aw)[NQ], const f32x4 bq, const float *w_base, int kp0,
                                          const float *wn_base, lds_cfloat *const (&ga)[NC + 1], int ng, f32x4 &bnext,
                                          int s_chunk, int s_slot, int op0, int op1)
    {
        constexpr int LAT = M >= 16 ? 4 : 1;
        auto rd_slot  = [](int q) constexpr { return M >= 16 ? (q * (M - 6)) / (NC + 1) : 0; };
        auto add_slot = [&](int q) constexpr { int v = rd_slot(q) + LAT; return v > M - 1 ? M - 1 : v; };
        f32x4 rd[NC + 1];
        f32x2 nlo, nhi;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const int u = m / NT, r = m % NT;
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[r / 4][r % 4], bq[u], acc[r], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(DBG & 2)) {
                if (r % 4 == 3 || r == NT - 1) {
                    if (u < 3 || !LAST) aw[r / 4] = *reinterpret_cast<const f32x4 *>(w_base + ((kp0 + u + 1) * 2 * NQ + r / 4) * 128);
                    else if (CROSS) aw[r / 4] = *reinterpret_cast<const f32x4 *>(wn_base + (r / 4) * 128);
                }
            }
            {   // DMA ops spread over the group
                if constexpr (NW == 8) {
                    constexpr int SP = M / 4 > 0 ? M / 4 : 1;
                    if (m % SP == SP / 2 && op0 + m / SP < op1) stream_op(s_chunk, s_slot, op0 + m / SP);
                } else {   // fewer waves, more pieces per wave: as many ops per MFMA slot as it takes
                    const int g = op1 - op0;
                    for (int o = m * g / M; o < (m + 1) * g / M; ++o) stream_op(s_chunk, s_slot, op0 + o);
                }
            }
            if constexpr (HAVE_NEXT_GATHER) {
#pragma unroll
                for (int q = 0; q <= ((DBG & 1) ? 0 : NC); ++q)
                    if (rd_slot(q) == m) rd[q] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>(ga[q] + 4 * ng);
#pragma unroll
                for (int q = 0; q <= ((DBG & 1) ? 0 : NC); ++q)
                    if (add_slot(q) == m) {
                        const f32x2 qlo = __builtin_shufflevector(rd[q], rd[q], 0, 1);
                        const f32x2 qhi = __builtin_shufflevector(rd[q], rd[q], 2, 3);
                        if (q == 0) { nlo = qlo; nhi = qhi; } else { nlo = pk_add(nlo, qlo); nhi = pk_add(nhi, qhi); }
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (HAVE_NEXT_GATHER) {
            bnext[0] = nlo[0]; bnext[1] = nlo[1]; bnext[2] = nhi[0]; bnext[3] = nhi[1];
            finish_b(bnext);
        }
    }

    __device__ __forceinline__ void forward(const int (&rowoff)[NC], float (&lg)[4], float &value)
    {
        f32x16 acc[NT];
#pragma unroll
        for (int r = 0; r < NT; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[r][g] = 0.0f;

        const int lane_w = (h * NQ * 32 + j) * 4;
        // ONE set of gather pointers (bias row + NC cells) into the ring slot being gathered from; it is
        // advanced by the slot distance once per chunk (NC+1 VALU adds per 8*NT MFMAs)
        lds_cfloat *ga[NC + 1];     // 32-bit LDS pointers (generic pointers would cost two VGPRs each)
        ga[0] = (lds_cfloat *)(lds_t + rp * R3_TSLOT + bias_row * LSTR + h * (KC / 2));
#pragma unroll
        for (int q = 0; q < NC; ++q) ga[q + 1] = (lds_cfloat *)(lds_t + rp * R3_TSLOT) + rowoff[q];

        // exposed once per forward: the first group's B operands and the first A operands
        f32x4 bq;
        {
            const f32x4 r0 = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>(ga[0]);
            f32x2 lo = __builtin_shufflevector(r0, r0, 0, 1), hi = __builtin_shufflevector(r0, r0, 2, 3);
            if constexpr (!(DBG & 1))
#pragma unroll
                for (int q = 1; q <= NC; ++q) {
                    const f32x4 rq = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>(ga[q]);
                    lo = pk_add(lo, __builtin_shufflevector(rq, rq, 0, 1));
                    hi = pk_add(hi, __builtin_shufflevector(rq, rq, 2, 3));
                }
            bq[0] = lo[0]; bq[1] = lo[1]; bq[2] = hi[0]; bq[3] = hi[1];
            finish_b(bq);
        }
        f32x4 aw[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) aw[q] = *reinterpret_cast<const f32x4 *>(lds_w + rp * WSLOT + lane_w + q * 128);

        int s0 = rp;                                         // ring slot of the current chunk
        for (int c = 0; c < n_chunks; ++c) {
            const int s1 = s0 == 2 ? 0 : s0 + 1, s2 = s1 == 2 ? 0 : s1 + 1;
            const float *wb = lds_w + s0 * WSLOT + lane_w, *wn = lds_w + s1 * WSLOT + lane_w;
            int sc = c + 2; if (sc >= n_chunks) sc -= n_chunks;      // chunk streamed now (wraps into the next forward)
            if (n_chunks == 1) sc = 0;
            constexpr int H0 = (NOPS + 1) / 2;
            f32x4 b1v, b2v;
            group<true, false, false>(acc, aw, bq, wb, 0, wn, ga, 1, b1v, sc, s2, 0, H0);
            {   // point the gather at the next slot (group 1's MFMAs only use registers)
                const int delta = (s1 - s0) * R3_TSLOT;
#pragma unroll
                for (int q = 0; q <= NC; ++q) ga[q] += delta;
            }
            // always prefetch across the boundary: after the last chunk the speculative gather (rows of the
            // NEXT timestep are not known yet) is discarded, the A operands of its first k-step are kept
            group<true, true, true>(acc, aw, b1v, wb, 4, wn, ga, 0, b2v, sc, s2, H0, NOPS);
            bq = b2v;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of chunk c+2 have landed
            __syncthreads();
            s0 = s1;
        }
        rp = s0;

        if constexpr (DBG & 8) {
            value = 0.0f;
#pragma unroll
            for (int r = 0; r < NT; ++r) value += acc[r][0];
            lg[0] = lg[1] = lg[2] = lg[3] = 0.0f;
            return;
        }
        if constexpr (NT >= 2) {
            // Heads as v_fma_f32 chains.  As MFMAs the k-ordered chain over the hidden units is 16*NT DEPENDENT 32x32x2 products
            // of which 5 rows in 32 are used: 6 % of the matrix-pipe time for 1 % of the FLOPs.  Instead: v_permlane32_swap
            // gives the lower lane half both parities of hidden units [0, H/2) of its episode and the upper half those of
            // [H/2, H); then, in phase p, the lower half runs units [0, H/2) of output p while the upper half continues
            // output p-1 over [H/2, H) -- the partial sum crosses the halves between phases.  One fma per hidden unit and
            // output in hidden order: the same chain as before, bit for bit.
            constexpr int HH = NT / 2, H = NT * 32;
            lds_cfloat *b1_lane = (lds_cfloat *)(lds_b1 + h * (NT * 16));
#pragma unroll
            for (int r = 0; r < NT; ++r)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 hb = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>(b1_lane + 16 * r + 4 * g4);
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[r][4 * g4 + g] = relu_lim_v(acc[r][4 * g4 + g] + hb[g], common_lim);
                }
#pragma unroll
            for (int r = 0; r < HH; ++r)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[r][g]), __float_as_uint(acc[r + HH][g]), false, false);
                    acc[r][g] = __uint_as_float(sw[0]); acc[r + HH][g] = __uint_as_float(sw[1]);
                }
            float a = 0.0f, res[5];
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                const int o = p - h;
                lds_cfloat *wp = (lds_cfloat *)lds_wn + (o < 0 ? 0 : (o > 4 ? 4 : o)) * H + h * (H / 2);
#pragma unroll
                for (int u4 = 0; u4 < H / 8; ++u4) {
                    const f32x4 w = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>(wp + 4 * u4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int u = 4 * u4 + k;                       // position inside the half: unit = (H/2)*h + u
                        a = __builtin_fmaf(w[k], acc[(u >> 5) + HH * (u & 1)][(u & 31) >> 1], a);
                    }
                }
                const auto sw = __builtin_amdgcn_permlane32_swap(0u, __float_as_uint(a), false, false);
                a = __uint_as_float(sw[0]);                             // lower half: 0; upper half: the lower half's partial sum
                if (p >= 1) res[p - 1] = __uint_as_float(sw[1]);        // upper half: output p-1, complete
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(res[i]), __float_as_uint(res[i]), false, false);
                const float out = __uint_as_float(sw[1]) + lds_bh[i];   // both halves: the upper half's value
                if (i < 4) lg[i] = out; else value = out;
            }
            return;
        }
        f32x16 hacc;
#pragma unroll
        for (int g = 0; g < 16; ++g) hacc[g] = 0.0f;
        lds_cfloat *b1_lane = (lds_cfloat *)(lds_b1 + h * (NT * 16));
        lds_cfloat *wh_lane = (lds_cfloat *)(lds_wh + ((j < 8 ? j : 8) * 2 + h) * (NT * 16));
        asm volatile("" : "+v"(b1_lane), "+v"(wh_lane));
        f32x4 hb[2], hw[2];
        hb[0] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>(b1_lane);
        hw[0] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>(wh_lane);
#pragma unroll
        for (int blk = 0; blk < NT * 4; ++blk) {
            const int r = blk >> 2, g0 = (blk & 3) * 4, cb = blk & 1, nb = cb ^ 1;
            if (blk + 1 < NT * 4) {
                hb[nb] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>(b1_lane + 4 * (blk + 1));
                hw[nb] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>(wh_lane + 4 * (blk + 1));
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float hv = relu_lim(acc[r][g0 + g] + hb[cb][g], common_lim);
                hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(hw[cb][g], hv, hacc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) lg[i] = __shfl(hacc[i], j, 64) + lds_bh[i];
        value = __shfl(hacc[0], j + 32, 64) + lds_bh[4];
    }
};

// =====================================================================================================
// Engine3S: the small-batch geometry.  The NS waves of a workgroup (one per SIMD) share ONE group of 32
// episodes and split the hidden units: wave w owns the row tiles [w*NT/NS, (w+1)*NT/NS) of the common
// Linear, so the 32-column forward runs on NS matrix cores instead of one.  The EmbeddingBag gather is split
// the other way: wave w sums the rows for 8/NS of the 8 k-steps of the NEXT chunk and publishes them as B
// operands through a double-buffered LDS exchange (replicating the gather in every wave makes the LDS the
// bottleneck: measured 7k cycles per chunk).  The k-ordered head chain is handed from wave to wave through
// LDS (partial accumulators are exact, so the result is still the oracle's single fma chain).
// Same weight images, ring and LDS map as Engine3 + a 2 KiB exchange area.  All NS waves carry the same
// episode state; only wave 0 (`primary()`) stores.
// =====================================================================================================
constexpr int R3S_XCHG = 256 + 1024, R3S_USER = 256;   // floats: head hand-off + B-operand exchange [2][2][64][4] | kernel use (MCTS leaf broadcast)

template <int NT, int NC, int NS>
struct Engine3S : Engine3<NT, NC, 0, NS> {
    using B = Engine3<NT, NC, 0, NS>;
    static constexpr int EPB = EPW, NTL = NT / NS, KC = B::KC, NQ = B::NQ, WSLOT = B::WSLOT;
    static constexpr int WOPS = B::WPIECE / NS, TOPS = (B::TPIECE + NS - 1) / NS, NOPS = WOPS + TOPS;   // DMA ops per wave and chunk: W1 pieces, table pieces
    static constexpr bool SPLIT = true;
    static_assert(NT % NS == 0 && (NTL == 1 || NTL == 2), "Engine3S: one or two row tiles per wave");

    float *lds_x, *lds_user;
    // The same instruction diet as the 16-column shape (docs/HISTORY.md 5.1e; a lone wave per SIMD is bound by its instruction count):
    // ring positions are compile-time facts -- a forward always starts in slot 0, its chunk sequence padded to a multiple of three
    // steps with "bubble" steps that only stream -- so the slot offsets of all LDS reads are immediates, the gather keeps one LDS
    // address per row for the whole forward, a stream op is `s_add_u32 m0, piece, literal` + the DMA instruction with a per-op
    // lane offset in a VGPR and a running source pointer, and the accumulators are pinned to VGPRs (inline-asm MFMA: with the
    // intrinsic the allocator copied the 32 loop-carried accumulator registers into AGPRs every chunk).
    const uint8_t *swp, *stp;
    int sv, n3;                            // virtual step whose data is streamed next (sv == step + 2); steps per forward (multiple of 3)
    uint32_t voffW[WOPS], voffT[TOPS], mT[TOPS];
#ifdef TW_ABLATE
    unsigned long long stq[6] = {0, 0, 0, 0, 0, 0};   // prologue | chunk loop | - | - | heads | -
#define TW_S3(var) const unsigned long long var = __builtin_readcyclecounter()
#define TW_A3(i, a, b) stq[i] += (b) - (a)
#else
#define TW_S3(var)
#define TW_A3(i, a, b)
#endif

    __host__ __device__ static size_t lds_floats(int obs_size) { return engine3_lds_floats<NT>(obs_size) + R3S_XCHG + R3S_USER; }
    __host__ __device__ static size_t lds_floats(const PolicyDev &p) { return lds_floats(p.obs_size); }
    __device__ __forceinline__ bool primary() const { return this->wave == 0; }
    __device__ __forceinline__ int  ep_lane() const { return this->j; }
    // all NS waves carry every episode's state: serial per-episode work is dealt out 32/NS episodes per wave (less divergence,
    // NS instruction streams instead of one)
    __device__ __forceinline__ bool owns_lane() const { return this->wave == this->j / (EPW / NS); }

    __device__ __forceinline__ void begin1(const PolicyDev &p, float *lds)
    {
        B::begin1(p, lds);                 // (streams chunks 0 and 1 into slots 0 and 1)
        lds_x = lds + engine3_lds_floats<NT>(p.obs_size);
        lds_user = lds_x + R3S_XCHG;
        n3 = (this->n_chunks + 2) / 3 * 3;
        sv = 2;
        {
            const int first = 2 < this->n_chunks ? 2 : 0;                        // a bubble step gets chunk 0's data (never read)
            swp = this->dsrc_w + (size_t)first * (WSLOT * 4);
            stp = this->dsrc_t + (size_t)first * (R3_TSLOT * 4);
        }
#pragma unroll
        for (int k = 0; k < WOPS; ++k) voffW[k] = this->voff + (uint32_t)k * (NS * 1024u);     // piece wave + NS*k (dsrc_w / ddst_w carry the wave)
#pragma unroll
        for (int k = 0; k < TOPS; ++k) {
            int tp = this->wave + NS * k;
            tp = tp < B::TPIECE ? tp : B::TPIECE - 1;                                           // past the end: repeat the last piece
            mT[k] = this->ddst_t + (uint32_t)tp * 1024u;
            voffT[k] = this->voff + (uint32_t)tp * 1024u;
        }
    }

    // A wave of the workgroup beyond the NS engine waves (self-play walkers, tw_mcts_deep.hip): it holds no engine state and
    // takes part in a forward only through the workgroup barriers -- one in the prologue, one per step, two in the heads.
    __device__ __forceinline__ void begin_idle(const PolicyDev &p)
    {
        this->pol = p;
        this->tid = threadIdx.x; this->lane = this->tid & 63; this->wave = __builtin_amdgcn_readfirstlane(this->tid >> 6);
        this->j = this->lane & 31; this->h = this->lane >> 5;
        this->n_chunks = this->pol.emb / KC;
        n3 = (this->n_chunks + 2) / 3 * 3;
    }
    __device__ __forceinline__ void idle_forward() const
    {
        for (int i = 0; i < n3 + 3; ++i) __builtin_amdgcn_s_barrier();
    }

    template <int S, int OP>   // DMA op OP of this wave: a piece of the chunk streamed next into ring slot S
    __device__ __forceinline__ void stream() const
    {
#ifdef TW_ABLATE
        if (this->eng_dbg & 4) return;
#endif
        if constexpr (OP < WOPS)
            TW_GLDS16_ADD(voffW[OP], this->ddst_w, S * WSLOT * 4 + OP * NS * 1024, swp);
        else
            TW_GLDS16_ADD(voffT[OP - WOPS], mT[OP - WOPS], S * R3_TSLOT * 4, stp);
    }
    // the DMA ops of MFMA slot m of M: ops [m*NOPS/M, (m+1)*NOPS/M) of this wave, into ring slot S
    template <int S, int OP = 0>
    __device__ __forceinline__ void ops_of_slot(int m, int M) const
    {
        if constexpr (OP < NOPS) {
            if (OP >= m * NOPS / M && OP < (m + 1) * NOPS / M) stream<S, OP>();     // (m is a constant after unrolling)
            ops_of_slot<S, OP + 1>(m, M);
        }
    }

    __device__ __forceinline__ void forward(const int (&rowoff)[NC], float (&lg)[4], float &value)
    {
        typedef __attribute__((address_space(3))) const f32x4 lds_cf4;
        typedef __attribute__((address_space(3))) const f32x2 lds_cf2;
        const int j = this->j, h = this->h, wave = this->wave;
        f32x16 acc[NTL];
#pragma unroll
        for (int r = 0; r < NTL; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[r][g] = 0.0f;

        const int t0    = wave * NTL;                                           // first row tile of this wave
        lds_cfloat *abase = (lds_cfloat *)this->lds_w + ((h * NQ + (t0 >> 2)) * 32 + j) * 4 + (t0 & 3);   // + slot, + ks * 2*NQ*128 for k-step ks
        // one LDS address per row for the whole forward: the ring slot is an immediate offset of the read
        constexpr int GW = 8 / NS;                                              // k-steps of a chunk this wave gathers
        lds_cfloat *ga[NC + 1];
        ga[0] = (lds_cfloat *)this->lds_t + this->bias_row * R3_LSTR + h * (KC / 2) + GW * wave;
#pragma unroll
        for (int q = 0; q < NC; ++q) ga[q + 1] = (lds_cfloat *)this->lds_t + rowoff[q] + GW * wave;

        // this wave's share (8/NS k-steps) of the B operands of a chunk: the LDS reads are spread over the MFMA slots, the sums
        // and the publication into exchange buffer `buf` follow the chunk's MFMAs
        float *xb = lds_x + 256;
        f32x2 gr2[NS == 4 ? NC + 1 : 1];
        f32x4 gr4[NS == 4 ? 1 : NC + 1];
        auto gather_finish = [&](int buf) {
            if constexpr (NS == 4) {
                f32x2 sm = gr2[0];                                                 // bias row, then the cells in order
#pragma unroll
                for (int q = 1; q <= NC; ++q) sm = pk_add(sm, gr2[q]);
                sm[0] = relu_lim(sm[0], this->emb_lim); sm[1] = relu_lim(sm[1], this->emb_lim);
                *reinterpret_cast<f32x2 *>(xb + ((buf * 2 + (wave >> 1)) * 64 + this->lane) * 4 + 2 * (wave & 1)) = sm;
            } else {
                f32x2 lo = __builtin_shufflevector(gr4[0], gr4[0], 0, 1), hi = __builtin_shufflevector(gr4[0], gr4[0], 2, 3);
#pragma unroll
                for (int q = 1; q <= NC; ++q) {
                    lo = pk_add(lo, __builtin_shufflevector(gr4[q], gr4[q], 0, 1));
                    hi = pk_add(hi, __builtin_shufflevector(gr4[q], gr4[q], 2, 3));
                }
                f32x4 b;
                b[0] = lo[0]; b[1] = lo[1]; b[2] = hi[0]; b[3] = hi[1];
                this->finish_b(b);
                *reinterpret_cast<f32x4 *>(xb + ((buf * 2 + wave) * 64 + this->lane) * 4) = b;
            }
        };
        auto read_a = [&](lds_cfloat *ap, float (&a)[NTL]) {
            if constexpr (NTL == 2) { const f32x2 a2 = *reinterpret_cast<lds_cf2 *>(ap); a[0] = a2[0]; a[1] = a2[1]; }
            else a[0] = *ap;
        };

        float aw[NTL];
        int par = 0;                                                            // B-operand buffer of the current chunk
        auto stream_advance = [&]() {
            ++sv; swp += WSLOT * 4; stp += R3_TSLOT * 4;
            if (sv == this->n_chunks || sv == n3) { swp = this->dsrc_w; stp = this->dsrc_t; }
            if (sv == n3) sv = 0;
        };
        // One step with the ring slots as compile-time facts (S0: this chunk, S1: the next one -- complete, S2: streamed now)
        auto step = [&](auto s0c, int c) {
            constexpr int S0 = decltype(s0c)::value, S1 = (S0 + 1) % 3, S2 = (S0 + 2) % 3;
            constexpr int M = 8 * NTL;
            if (c < this->n_chunks) {
                f32x4 bg[2];
                bg[0] = *reinterpret_cast<const f32x4 *>(xb + ((par * 2 + 0) * 64 + this->lane) * 4);
                bg[1] = *reinterpret_cast<const f32x4 *>(xb + ((par * 2 + 1) * 64 + this->lane) * 4);
                // 8*NTL MFMAs; in the shadow of each: its share of the DMA ops of step c+2, of the gather reads of step c+1 (which
                // sits complete in slot S1; after the last chunk that is a bubble or chunk 0 of the next forward, gathered with the
                // rows of this one -- never consumed, the next prologue rewrites exchange buffer 0) and the next A operands
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const int ks = m / NTL, r = m % NTL;
                    // (the builtin: an inline-asm MFMA here -- rounds 1 to 3 -- was 0.4 .. 1.0 % faster and outside hipcc's hazard padding;
                    //  profiles/r04_mfma_intrinsic_vs_asm.txt)
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[r], bg[ks >> 2][ks & 3], acc[r], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    ops_of_slot<S2>(m, M);
#pragma unroll
                    for (int q = m * (NC + 1) / M; q < (m + 1) * (NC + 1) / M; ++q) {
                        if constexpr (NS == 4) gr2[q] = *reinterpret_cast<lds_cf2 *>(ga[q] + S1 * R3_TSLOT);
                        else gr4[q] = *reinterpret_cast<lds_cf4 *>(ga[q] + S1 * R3_TSLOT);
                    }
                    if (r == NTL - 1) {
                        if (ks < 7) read_a(abase + S0 * WSLOT + (ks + 1) * 2 * NQ * 128, aw); else read_a(abase + S1 * WSLOT, aw);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                gather_finish(par ^ 1);
                par ^= 1;
            } else {                                                            // bubble: only the streams
                ops_of_slot<S2>(0, 1);
            }
            stream_advance();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of step c+2 have landed
            __syncthreads();
        };

        TW_S3(q_in);
        {
            read_a(abase, aw);                                                  // slot 0
#pragma unroll
            for (int q = 0; q <= NC; ++q) {
                if constexpr (NS == 4) gr2[q] = *reinterpret_cast<lds_cf2 *>(ga[q]);
                else gr4[q] = *reinterpret_cast<lds_cf4 *>(ga[q]);
            }
            gather_finish(0);
            __syncthreads();
        }
        TW_S3(q_pro);
        TW_A3(0, q_in, q_pro);
        for (int c = 0; c < n3; c += 3) {
            step(std::integral_constant<int, 0>{}, c);
            step(std::integral_constant<int, 1>{}, c + 1);
            step(std::integral_constant<int, 2>{}, c + 2);
        }
        TW_S3(q_lp);
        TW_A3(1, q_pro, q_lp);

        // heads.  The k-ordered chain over the hidden units is serial, and as a chain of dependent 32x32x2 MFMAs it costs
        // 64 cycles per two units.  Here instead: every wave writes its ReLU'd hidden units into the ring slot the last chunk
        // just freed ([unit/4][episode][4], units >= 128 in the W slot), then half-wave hw = 2*wave + h runs the chain of
        // output hw (4 logits, value) for its 32 episodes as 4-cycle v_fma_f32 steps -- the same fma chain, bit for bit.
        TW_S3(q_h0);
        {
            constexpr int fs = 2;                                                // the last step's slot: free until step 0 of the next forward streams into it
            float *hid_lo = this->lds_t + fs * R3_TSLOT, *hid_hi = this->lds_w + fs * WSLOT;
            lds_cfloat *b1_lane = (lds_cfloat *)(this->lds_b1 + h * (NT * 16)) + 16 * t0;
#pragma unroll
            for (int r = 0; r < NTL; ++r) {
                float *dst = ((t0 + r) < 4 ? hid_lo : hid_hi) + ((t0 + r) & 3) * 8 * 128 + j * 4 + h;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 hb = *reinterpret_cast<lds_cf4 *>(b1_lane + 16 * r + 4 * g4);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int gg = 4 * g4 + g;
                        dst[(gg >> 1) * 128 + 2 * (gg & 1)] = relu_lim(acc[r][gg] + hb[g], this->common_lim);
                    }
                }
            }
            __syncthreads();
            constexpr int NPASS = (5 + 2 * NS - 1) / (2 * NS);
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                const int o  = 2 * wave + h + 2 * NS * pass;
                const int oc = o < 4 ? o : 4;
                lds_cfloat *we = (lds_cfloat *)(this->lds_wh + (oc * 2) * (NT * 16));   // even units of output oc; odd units NT*16 further
                float a = 0.0f;
#pragma unroll
                for (int m = 0; m < NT * 4; ++m) {                               // 8 hidden units per trip
                    const f32x4 w0 = *reinterpret_cast<lds_cf4 *>(we + 4 * m);
                    const f32x4 w1 = *reinterpret_cast<lds_cf4 *>(we + NT * 16 + 4 * m);
                    const float *hp = (m < 16 ? hid_lo + (2 * m) * 128 : hid_hi + (2 * m - 32) * 128) + j * 4;
                    const f32x4 x0 = *reinterpret_cast<const f32x4 *>(hp);
                    const f32x4 x1 = *reinterpret_cast<const f32x4 *>(hp + 128);
                    a = __builtin_fmaf(w0[0], x0[0], a); a = __builtin_fmaf(w1[0], x0[1], a);
                    a = __builtin_fmaf(w0[1], x0[2], a); a = __builtin_fmaf(w1[1], x0[3], a);
                    a = __builtin_fmaf(w0[2], x1[0], a); a = __builtin_fmaf(w1[2], x1[1], a);
                    a = __builtin_fmaf(w0[3], x1[2], a); a = __builtin_fmaf(w1[3], x1[3], a);
                }
                if (o < 5) lds_x[o * 32 + j] = a + this->lds_bh[o];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) lg[i] = lds_x[i * 32 + j];
            value = lds_x[4 * 32 + j];
        }
        TW_S3(q_h1);
        TW_A3(4, q_h0, q_h1);
    }
};

#ifdef TW_ABLATE
#include "tw_engine_diag.hpp"          // Engine3G: a geometry measured no faster (profiles/r03_mid_rollout_two_groups_per_cu.txt); diagnostic build only
#endif

// =====================================================================================================
// Engine3T: the tiny-batch geometry.  v_mfma_f32_16x16x4_f32 is a k-ordered fma chain too (scripts/mfma_probe/
// probe_f32_16x16x4.hip: 0 mismatches in 65,536), so the exact forward also runs with SIXTEEN episodes per workgroup: four
// waves, one per SIMD, share them and split the hidden units (16-row tiles, NT/2 per wave); per 16-column chunk a wave issues
// 4 k-groups x NT/2 MFMAs of 32 cycles -- half the matrix time of Engine3S per forward, and twice as many workgroups, so
// batches of <= CUs x 16 episodes (the reference's 1,024 envs and 4,096 self-play episodes) use every CU.
// Lane l = (kq = l >> 4, jj = l & 15): A row / B and D column jj, k index kq inside a k-group; D rows 4*kq .. 4*kq+3.
//   * A operands come from the SAME W1 image as Engine3 (k-step 2g + (kq>>1), parity kq&1, 32-row position 16*half + jj):
//     one ds_read_b128 = the wave's four row tiles of a k-group;
//   * the gather is split over the waves: wave w sums the rows for k-group w (one float per lane) and publishes it through
//     the double-buffered LDS exchange, as in Engine3S;
//   * heads: v_fma_f32 chains over the hidden units written to the freed ring slot, as in Engine3S.
// In the kernels `j` is the episode column (0..15) and `h` the k index (0..3): lanes with h == 0 of wave 0 store.
// =====================================================================================================
typedef float f32x4v __attribute__((ext_vector_type(4)));

// FS ("force scalar"): the scalar operands of the DMA ops through v_readfirstlane -- for a kernel whose control flow made the compiler
// carry the engine's running source pointer in vector registers (the walker kernel in solve mode: an "s" asm operand cannot take those)
template <int NT, int NC, bool FS = false>
struct Engine3T : Engine3<NT, NC, 0, 4> {
    using B = Engine3<NT, NC, 0, 4>;
    static constexpr int NS = 4, EPB = 16, TPW = NT / 2, KC = B::KC, NQ = B::NQ, WSLOT = B::WSLOT, H = NT * 32;
    static constexpr int TOPS = (B::TPIECE + 3) / 4;                       // table DMA ops per wave and 16-column chunk
    static constexpr int NOPS = 2 * TOPS;                                  // ... per step (a step is a PAIR of chunks)
    static constexpr int GS = 4 * NQ * 128;                                // floats of the W1 image per k-group of four
    static constexpr bool SPLIT = true;
    static_assert(NT == 8 || NT == 4, "Engine3T: 128 or 256 hidden units");
    // the head buffer (hidden units of the 16 columns) sits behind the table ring when the ring area of Engine3's LDS map has
    // room for both (256 hidden units), else behind the engine's own areas
    static constexpr bool HID_IN_RING = 4 * R3_TSLOT + NT * 32 * 16 <= 3 * WSLOT + 3 * R3_TSLOT;
    static_assert(4 * R3_TSLOT <= 3 * WSLOT + 3 * R3_TSLOT, "table ring must fit the ring area of Engine3's LDS map");

    float *lds_x, *lds_user;
    // What this shape does differently from the throughput shapes, and why (docs/HISTORY.md 5.1e).  A lone wave per SIMD issues one
    // instruction every 4..5 cycles whatever its kind and cannot overlap its own stalls, so a step costs its instruction count
    // plus every round trip on its critical path (barrier -> B operands -> MFMAs -> gather adds -> exchange -> barrier):
    //   * a step is a PAIR of 16-wide chunks (32 MFMA slots between two barriers): half the barriers and B-operand round trips;
    //   * W1 never enters LDS: a lane's A operands of a k-group (its TPW tiles) are one contiguous read of the image the other
    //     shapes stream (an LDS slot is a verbatim copy of it), requested a whole step ahead into a two-deep register ring;
    //   * the table ring therefore has room for two pair-slots of two chunks (the pair being gathered, the pair being streamed;
    //     a step never touches the table of its own chunks -- they were gathered a step earlier) in the ring area of Engine3's
    //     LDS map, the head buffer behind them;
    //   * ring positions are compile-time facts: a forward starts in pair-slot 0 and has an even number of steps (a "bubble"
    //     step that only streams if the chunk-pair count is odd), so slot offsets are immediates of the LDS reads and the loop
    //     body is two straight-line steps; the gather keeps one LDS address per row for the whole forward;
    //   * a stream op is `s_add_u32 m0, piece, literal` + the DMA instruction with a per-op lane offset in a VGPR and a running
    //     source pointer (the ring runs on from one forward into the next);
    const uint8_t *stp;                    // table image of the pair streamed next
    const float *agl;                      // this lane's A operands of k-group 0 in the W1 image
    int aoff;                              // floats from there to the pair whose operands are requested next
    int sv, av, np, n2;                    // virtual pair streamed next (step + 2) / loaded next (step + 1); chunk pairs per forward; steps (even)
    uint32_t voffT[TOPS], mT[TOPS];
    float areg[2][8][TPW];
    float *tbase;                          // table ring: pair-slot s, chunk h at tbase + (2 s + h) * R3_TSLOT
    // Barrier of the FOUR ENGINE WAVES alone (forward<true>; the walker kernel's decoupled shape, tw_mcts_deep.hip: its workgroup holds
    // waves that never take part in a forward, so s_barrier -- which counts every wave of the workgroup -- cannot be used inside
    // one).  An arrival counter in LDS: every wave adds one and spins until all four of this generation have; ~150 cycles against
    // ~40 for s_barrier, 19 of them per forward.
    volatile __attribute__((address_space(3))) uint32_t *ebar_cnt = nullptr;
    uint32_t ebar_gen = 0;
    template <bool EB, bool IDLE = false>      // IDLE: a wait of unknown length (the engine waves between two forwards): sleep between polls
    __device__ __forceinline__ void fsync()
    {
        if constexpr (!EB) {
            __syncthreads();
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");            // this wave's LDS writes are complete
            ebar_gen += (uint32_t)NS;
            if (this->lane == 0) __hip_atomic_fetch_add((__attribute__((address_space(3))) uint32_t *)ebar_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            while ((int32_t)((uint32_t)__builtin_amdgcn_readfirstlane((int)*ebar_cnt) - ebar_gen) < 0) { if constexpr (IDLE) __builtin_amdgcn_s_sleep(4); }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
#ifdef TW_ABLATE
    unsigned long long stq[6] = {0, 0, 0, 0, 0, 0};   // prologue | step loop | - | - | heads | -
#endif

    __host__ __device__ static size_t lds_floats(int obs_size) { return engine3_lds_floats<NT>(obs_size) + R3S_XCHG + R3S_USER + (HID_IN_RING ? 0 : NT * 32 * 16); }
    __host__ __device__ static size_t lds_floats(const PolicyDev &p) { return lds_floats(p.obs_size); }
    __device__ __forceinline__ bool primary() const { return this->wave == 0; }
    __device__ __forceinline__ int  ep_lane() const { return this->j; }
    __device__ __forceinline__ bool owns_lane() const { return this->wave == this->j / (EPB / NS); }

    __device__ __forceinline__ static void load_a(const float *p, float (&a)[TPW])
    {
        typedef const __attribute__((address_space(1))) f32x4 gl4;
        typedef const __attribute__((address_space(1))) f32x2 gl2;
        if constexpr (TPW == 4) { const f32x4 v = *(gl4 *)p; a[0] = v[0]; a[1] = v[1]; a[2] = v[2]; a[3] = v[3]; }
        else { const f32x2 v = *(gl2 *)p; a[0] = v[0]; a[1] = v[1]; }
    }

    template <int SLOT, int OP>   // DMA op OP of this wave: a table piece of the pair streamed next into pair-slot SLOT (OP >= TOPS: its second chunk)
    __device__ __forceinline__ void stream() const
    {
        constexpr int HALF = OP / TOPS, K = OP % TOPS;
#ifdef TW_KNOCK
        if (TW_KNOCK & 4) return;                 // timing-only knock-outs (a variant build with -DTW_KNOCK=bits): 1 no row reads, 2 no A-operand loads, 4 no table streams, 8 no MFMAs, 16 no closing wait
#endif
        const uint8_t *src = stp; uint32_t mk = mT[K];
        if constexpr (FS) {
            const uint64_t u = (uint64_t)(uintptr_t)stp;
            src = reinterpret_cast<const uint8_t *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u));
            mk = (uint32_t)__builtin_amdgcn_readfirstlane((int)mT[K]);
        }
        TW_GLDS16_ADD(voffT[K] + (uint32_t)(HALF * R3_TSLOT * 4), mk, (2 * SLOT + HALF) * R3_TSLOT * 4, src);
    }
    // the DMA ops of MFMA slot m of M: ops [m*NOPS/M, (m+1)*NOPS/M) of this wave
    template <int SLOT, int OP = 0>
    __device__ __forceinline__ void ops_of_slot(int m, int M) const
    {
        if constexpr (OP < NOPS) {
            if (OP >= m * NOPS / M && OP < (m + 1) * NOPS / M) stream<SLOT, OP>();     // (m is a constant after unrolling: one op survives)
            ops_of_slot<SLOT, OP + 1>(m, M);
        }
    }

    __device__ __forceinline__ void begin1(const PolicyDev &p, float *lds)
    {
        B::begin1(p, lds);                 // (LDS constants; its first streams land in the ring area and are overwritten below)
        this->j = this->lane & 15; this->h = this->lane >> 4;
        lds_x = lds + engine3_lds_floats<NT>(p.obs_size);
        lds_user = lds_x + R3S_XCHG;
        tbase = this->lds_w;
        np = (this->n_chunks + 1) / 2;     // (tw_policy_create: the embedding is a multiple of 32, so chunks come in pairs)
        n2 = (np + 1) / 2 * 2;
        const int kq = this->h, jj = this->j, wave = this->wave;
        const int half = wave >> 1, q = NT == 8 ? (wave & 1) : 0, cc0 = NT == 8 ? 0 : 2 * (wave & 1);
        agl = p.w1p + (kq * NQ + q) * 128 + (16 * half + jj) * 4 + cc0;
        const uint32_t dt = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)tbase);
#pragma unroll
        for (int k = 0; k < TOPS; ++k) {
            int tp = wave + 4 * k;
            tp = tp < B::TPIECE ? tp : B::TPIECE - 1;                                           // past the end: repeat the last piece
            mT[k] = dt + (uint32_t)tp * 1024u;
            voffT[k] = this->voff + (uint32_t)tp * 1024u;
        }
        // pairs 0 and 1 of the first forward into pair-slots 0 and 1 (after B's own first streams have landed: same LDS area),
        // the A operands of pair 0 into register slot 0
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stp = this->dsrc_t;
        stream_pair<0>();
        stp = this->dsrc_t + (size_t)(np > 1 ? 1 : 0) * (2 * R3_TSLOT * 4);
        stream_pair<1>();
#pragma unroll
        for (int g = 0; g < 8; ++g) load_a(agl + g * GS, areg[0][g]);
        sv = 2 < n2 ? 2 : 0; av = 1;       // (virtual pair n2 + k is pair k of the next forward)
        set_pointers();
    }
    // A wave of the workgroup beyond the four engine waves (self-play walkers, tw_mcts_deep.hip): it takes part in begin1 and in a
    // forward only through the workgroup barriers -- one in begin1, one in the prologue, one per step, two in the heads.
    __device__ __forceinline__ void begin_idle(const PolicyDev &p)
    {
        this->pol = p;
        this->tid = threadIdx.x; this->lane = this->tid & 63; this->wave = __builtin_amdgcn_readfirstlane(this->tid >> 6);
        this->j = this->lane & 15; this->h = this->lane >> 4;
        this->n_chunks = this->pol.emb / KC;
        np = (this->n_chunks + 1) / 2;
        n2 = (np + 1) / 2 * 2;
        __builtin_amdgcn_s_barrier();
    }
    __device__ __forceinline__ void idle_forward() const
    {
        for (int i = 0; i < n2 + 3; ++i) __builtin_amdgcn_s_barrier();
    }
    template <int SLOT, int OP = 0>
    __device__ __forceinline__ void stream_pair() const
    {
        if constexpr (OP < NOPS) { stream<SLOT, OP>(); stream_pair<SLOT, OP + 1>(); }
    }
    // virtual pair v: a real pair of this forward (v < np), a bubble (np <= v < n2: pair 0's data, never read), or pair v - n2 of the next forward
    __device__ __forceinline__ int real_pair(int v) const { return v < np ? v : (v < n2 ? 0 : v - n2); }
    __device__ __forceinline__ void set_pointers()
    {
        stp  = this->dsrc_t + (size_t)real_pair(sv) * (2 * R3_TSLOT * 4);
        aoff = real_pair(av) * (2 * WSLOT);
    }
    __device__ __forceinline__ void advance()
    {
        ++sv; ++av; stp += 2 * R3_TSLOT * 4; aoff += 2 * WSLOT;
        if (sv == np || sv == n2) stp = this->dsrc_t;
        if (av == np || av == n2) aoff = 0;
        if (sv == n2) sv = 0;
        if (av == n2) av = 0;
    }

    __device__ __forceinline__ void rows_of(uint64_t board, int n_cells, int perm, int (&rowoff)[NC]) const
    {
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            int row = this->zero_row;
            if (i < n_cells) {
                const int id = i * n_cells + (int)nib(board, i);
                row = perm >= 0 ? (int)this->perm_obs[perm * this->pol.obs_size + id] : id;
            }
            rowoff[i] = row * R3_LSTR;
        }
    }

    template <bool EB = false>
    __device__ __forceinline__ void forward(const int (&rowoff)[NC], float (&lg)[4], float &value)
    {
        typedef __attribute__((address_space(3))) const f32x4 lds_cf4;
        const int jj = this->j, kq = this->h, wave = this->wave;
        f32x4v acc[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};

        // gather: k-group `wave` of a chunk, this lane's k = 4*wave + kq of it -> position in the [even k | odd k] row image.
        // One LDS address per row for the whole forward: pair-slot and chunk are immediate offsets of the read.
        lds_cfloat *gb = (lds_cfloat *)tbase + ((kq & 1) * 8 + 2 * wave + (kq >> 1));
        lds_cfloat *ga[NC + 1];
        ga[0] = gb + this->bias_row * R3_LSTR;
#pragma unroll
        for (int c = 0; c < NC; ++c) ga[c + 1] = gb + rowoff[c];

        float *xb = lds_x + 256;                                                  // [2 buffers][64 lanes][8 k-groups of a pair]
        float gr[2][NC + 1];
        auto gather_finish = [&](int buf) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                float sm = gr[hf][0];                                             // bias row, then the cells in order
#ifdef TW_KNOCK
                if (!(TW_KNOCK & 64))                                             // 64: no add chains
#endif
#pragma unroll
                for (int c = 1; c <= NC; ++c) sm = sm + gr[hf][c];
                xb[(buf * 64 + this->lane) * 8 + hf * 4 + wave] = relu_lim_v(sm, this->emb_lim);
            }
        };

        int par = 0;                                                              // B-operand buffer of the current pair
        // One step = one pair of chunks; P: register slot of its A operands = pair-slot streamed into; P^1: pair-slot gathered from
        auto step = [&](auto pc, int p) {
            constexpr int P = decltype(pc)::value, Q = P ^ 1;
            constexpr int M = 8 * TPW;
            // A operands of the NEXT pair, requested first: they land during this step (the closing wait covers them)
#ifdef TW_KNOCK
            if (!(TW_KNOCK & 2))
#endif
            {
                const float *ap = agl + aoff;
#pragma unroll
                for (int g = 0; g < 8; ++g) load_a(ap + g * GS, areg[Q][g]);
            }
            if (p < np) {
                const f32x4 bq0 = *reinterpret_cast<const f32x4 *>(xb + (par * 64 + this->lane) * 8);
                const f32x4 bq1 = *reinterpret_cast<const f32x4 *>(xb + (par * 64 + this->lane) * 8 + 4);
                // (one visible use of the LAST of this step's operand loads before any of this step's DMA ops exist: the compiler
                //  waits here once, with only the eight loads issued a moment ago in flight, instead of once per k-group with
                //  counts that -- not knowing the inline-asm streams -- would wait for those as well)
                asm volatile("" :: "v"(areg[P][7][TPW - 1]));
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    ops_of_slot<P>(m, M);
                    const int g = m / TPW, t = m % TPW;
                    // (the builtin: the inline-asm form of rounds 1 to 3 measured no faster in round 4 -- config 1 1.052 ms against 1.072,
                    //  self-play equal, profiles/r04_mfma_intrinsic_vs_asm.txt -- and sat outside hipcc's hazard padding)
#ifdef TW_KNOCK
                    if (!(TW_KNOCK & 8))
#endif
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[P][g][t], g < 4 ? bq0[g & 3] : bq1[g & 3], acc[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#ifdef TW_KNOCK
                    if (!(TW_KNOCK & 1))
#endif
#pragma unroll
                    for (int r = m * 2 * (NC + 1) / M; r < (m + 1) * 2 * (NC + 1) / M; ++r)         // the next pair, complete in pair-slot Q
                        gr[r / (NC + 1)][r % (NC + 1)] = ga[r % (NC + 1)][(2 * Q + r / (NC + 1)) * R3_TSLOT];
                    __builtin_amdgcn_sched_barrier(0);
                }
                gather_finish(par ^ 1);
                par ^= 1;
            } else {                                                              // bubble: only the streams
                stream_pair<P>();
            }
            advance();
#ifdef TW_KNOCK
            if (!(TW_KNOCK & 16))
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef TW_KNOCK
            if (!(TW_KNOCK & 32))                                                 // 32: no barrier at the end of a step
#endif
            this->template fsync<EB>();
        };

        TW_S3(q_in);
        {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int c = 0; c <= NC; ++c) gr[hf][c] = ga[c][hf * R3_TSLOT];   // pair 0 in pair-slot 0
            gather_finish(0);
            this->template fsync<EB>();
        }
        TW_S3(q_pro);
        TW_A3(0, q_in, q_pro);
        for (int p = 0; p < n2; p += 2) {
            step(std::integral_constant<int, 0>{}, p);
            step(std::integral_constant<int, 1>{}, p + 1);
        }
        TW_S3(q_lp);
        TW_A3(1, q_pro, q_lp);

        // heads: hidden units -> the buffer behind the table ring as [unit/4][16 episodes][4], then one v_fma_f32 chain per (episode, output)
        TW_S3(q_h0);
        {
            const int half = wave >> 1, q = NT == 8 ? (wave & 1) : 0, cc0 = NT == 8 ? 0 : 2 * (wave & 1);
            float *hid = HID_IN_RING ? tbase + 4 * R3_TSLOT : lds_user + R3S_USER;
            // D row 4*kq + r of local tile t is hidden unit 32*(4q + cc0 + t) + 2*(r + 8*half + 4*(kq>>1)) + (kq&1)
            const int ublk = 8 * half + 4 * (kq >> 1);                            // the lane-dependent part of g'
            float *dst = hid + (8 * (4 * q + cc0) + 2 * (2 * half + (kq >> 1))) * 64 + jj * 4 + (kq & 1);
            lds_cfloat *b1p = (lds_cfloat *)(this->lds_b1 + (kq & 1) * (NT * 16)) + 16 * (4 * q + cc0) + ublk;
#pragma unroll
            for (int t = 0; t < TPW; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    dst[t * 8 * 64 + (r >> 1) * 64 + 2 * (r & 1)] = relu_lim_v(acc[t][r] + b1p[16 * t + r], this->common_lim);
            this->template fsync<EB>();
            const int o  = wave + 4 * kq;                                          // outputs 0..3: lanes kq == 0 of wave o; value: kq == 1 of wave 0
            const int oc = o < 4 ? o : 4;
            lds_cfloat *wp = (lds_cfloat *)this->lds_wn + oc * H;
            float a = 0.0f;
            // (EB -- the decoupled walker shapes, workgroups of 384 .. 768 threads, 256 registers per lane at most: fully unrolled, the
            //  scheduler requests all 2 x 64 quads of the chain up front and the kernel spills ~450 bytes per lane to scratch)
#pragma unroll EB ? 8 : H / 4
            for (int m = 0; m < H / 4; ++m) {
                const f32x4 w = *reinterpret_cast<lds_cf4 *>(wp + 4 * m);
                const f32x4 x = *reinterpret_cast<const f32x4 *>(hid + m * 64 + jj * 4);
                a = __builtin_fmaf(w[0], x[0], a); a = __builtin_fmaf(w[1], x[1], a);
                a = __builtin_fmaf(w[2], x[2], a); a = __builtin_fmaf(w[3], x[3], a);
            }
            if (o < 5) lds_x[o * 16 + jj] = a + this->lds_bh[o];
            this->template fsync<EB>();
#pragma unroll
            for (int i = 0; i < 4; ++i) lg[i] = lds_x[i * 16 + jj];
            value = lds_x[4 * 16 + jj];
        }
        TW_S3(q_h1);
        TW_A3(4, q_h0, q_h1);
    }
};

// launch geometry code -> engine: NW > 0 = NW independent waves of 32 episodes (Engine3); NW < 0 = -NW waves sharing 32 (Engine3S)
template <int NT, int NC, int DBG, int NW> struct Geom { using Eng = Engine3<NT, NC, DBG, NW>; static constexpr int WAVES = NW; };
template <int NT, int NC, int DBG> struct Geom<NT, NC, DBG, -4> { using Eng = Engine3S<NT, NC, 4>; static constexpr int WAVES = 4; };
template <int NT, int NC, int DBG> struct Geom<NT, NC, DBG, -2> { using Eng = Engine3S<NT, NC, 2>; static constexpr int WAVES = 2; };
template <int NT, int NC, int DBG> struct Geom<NT, NC, DBG, -16> { using Eng = Engine3T<NT, NC>; static constexpr int WAVES = 4; };   // 16 episodes per workgroup
template <int NT, int NC, int DBG> struct Geom<NT, NC, DBG, -17> { using Eng = Engine3T<NT, NC, true>; static constexpr int WAVES = 4; };   // ... its scalar DMA operands forced (walker kernel, solve mode)
#ifdef TW_ABLATE
template <int NT, int NC, int DBG> struct Geom<NT, NC, DBG, -5> { using Eng = Engine3G<NT, NC>; static constexpr int WAVES = 4; };    // four waves share 32, two workgroups per CU (tw_engine_diag.hpp)
#endif
template <int NT, int NC, int DBG> struct Geom<NT, NC, DBG, -64> { using Eng = EngineV<NC, true>; static constexpr int WAVES = 4; };   // generic stacks, inline-asm MFMAs: not used any more (tw_engine_generic.hpp)
template <int NT, int NC, int DBG> struct Geom<NT, NC, DBG, -65> { using Eng = EngineV<NC, false>; static constexpr int WAVES = 4; }; // generic stacks (any Sequential depth)

// geometry for n episodes: 8 = the throughput shape; below ~3/4 of a chip of 256-episode workgroups the split shape
template <int NT> inline int geometry_for(uint64_t n)
{
    int nw = waves_per_group(n);
    const int force = launch_options().force_geom;                 // diagnostic (tw_set_launch_option): 8 = throughput shape, else small-batch
    if (force) nw = force == 8 ? 8 : 1;
    if (nw == 8 || NT < 2) return nw;
    // up to one 16-episode workgroup per CU (4,096 episodes on an MI355X): the tiny-batch shape; force_geom 32 keeps the 32-episode one
    if (NT >= 4 && n <= rollout_f32_resident_episodes() / 16 && force != 32) return -16;
    return NT >= 4 ? -4 : -2;
}

}  // namespace tw
