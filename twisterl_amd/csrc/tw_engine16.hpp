// tw_engine16.hpp -- f16-input policy-forward engine (TW_PREC_F16) for gfx950.
//
// Numeric spec of the mode (restated by the oracle's TWO_ARITH_F16): table rows, W1 and the head weights are
// rounded to f16 (RNE) once at tw_policy_create; h0 = relu(bias + sum of table rows) and h1 = relu(b1 + W1.h0)
// are accumulated in f32 and rounded to f16 (RNE) where they feed the next product; biases, logits and the
// value stay f32.  Products of two f16 numbers are exact in f32, so the only freedom left to the hardware is
// the order of the f32 accumulation inside v_mfma_f32_32x32x16_f16: results agree with the oracle to f32
// rounding (tests: 1e-5), not bit for bit.  Environment transitions, masks, rewards stay bit-exact.
//
// Mapping to CDNA4
//   * one workgroup = 4 waves (one per SIMD, 512 registers) = 256 episodes; a wave owns TWO MFMA column
//     tiles (64 episodes), so every weight operand read from LDS feeds two MFMAs.  Lanes j and j+32 hold
//     the same two episodes (tile 0 / tile 1 of column j); lane half h post-processes tile h.
//   * the EmbeddingBag is evaluated ON THE MATRIX CORE as (table^T) x (one-hot): chunk c of 16 ids is
//     cell c of the board, so the one-hot B operand of a lane half is "1.0 at position tile^8h, if < 8":
//     one ds_read_b128 from a 9-entry LDS table, no VALU work.  With f16 inputs the matrix core is 16x
//     faster than the f32 path while the VALU gather would stay where it is (and its LDS traffic, 16 KB per
//     record, would be the limiter); the one-hot product costs as many MFMAs as the common layer but reads
//     the table chunk once per 64 episodes.
//   * the embedding accumulators ARE the next B operands: accumulator register r of lane half h holds row
//     8(r>>2)+4h+(r&3) of the 32-row tile, so registers 8m..8m+7 (bias added, v_cvt_pk_f16_f32) are the B
//     fragment of k-step m of that tile; tw_policy_create orders the k index of the W1 image (and the
//     hidden index of the head image) accordingly.  No LDS round trip between the layers.
//   * software pipeline: iteration k issues the embedding MFMAs of tile k+1 and then the common-layer MFMAs
//     of tile k, converting tile k+1's accumulators in the shadow of the latter; operand reads run D pairs
//     ahead of the MFMAs, across iteration boundaries; issue order pinned with sched_barrier.
//   * weights stream through a ring of three LDS slots; one stage = NC table chunks (of tile k+1) + 2*NHT W1
//     chunks (of tile k) of 1 KiB each, lane-linear (lane l reads bytes [16l,16l+16) of a chunk: conflict-free
//     ds_read_b128), filled by LDS-DMA two stages ahead (L2 -> LDS, 32 KiB per stage for Puzzle-15).  Table
//     tile 0 stays resident in LDS for the prologue of every forward.
#pragma once
#include "tw_engine.hpp"

#include <type_traits>

namespace tw {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const uint8_t lds_cu8;

constexpr int E16_MAXP   = 4;      // twists supported by the f16 engine (tables in LDS)
constexpr int E16_MAX_KT = 32;     // embedding size <= 1024

template <int NHT, int NC>
__host__ __device__ constexpr size_t engine16_lds_bytes()
{
    return 160 + (size_t)(E16_MAXP + 1) * (16 + 256) + 32 + (size_t)2 * (E16_MAXP + 1) * 256 + 64 + (size_t)NHT * 128 + (size_t)E16_MAX_KT * 128 +
           (size_t)NHT * 2048 + (size_t)NC * 1024 + (size_t)3 * ((NC + 2 * NHT + 3) / 4) * 4096;
}

template <int NHT, int NC>
struct Engine16 {
    static constexpr int NW = 4, THREADS = 256, EPB = 256, D = 3;
    static constexpr int SP     = NC + 2 * NHT;              // 1-KiB pieces per stage
    static constexpr int NOPS   = (SP + NW - 1) / NW;        // DMA ops per wave per stage (one piece each)
    static constexpr int SBYTES = NOPS * NW * 1024;          // stage image / ring slot, padded to whole rounds of NW pieces
    static constexpr float OUT_SCALE = 1.0f;                 // head accumulator -> logit - bias
    // LDS map (small tables first: their offsets fit the 16-bit ds offset field)
    static constexpr uint32_t O_OH = 0, O_SRC = 160, O_VMAP = O_SRC + (E16_MAXP + 1) * 16, O_ACT = O_VMAP + (E16_MAXP + 1) * 256,
                              O_OHB = O_ACT + 32, O_BH = O_OHB + 2 * (E16_MAXP + 1) * 256, O_B1 = O_BH + 64, O_EBIAS = O_B1 + NHT * 128, O_HEAD = O_EBIAS + E16_MAX_KT * 128,
                              O_T0 = O_HEAD + NHT * 2048, O_RING = O_T0 + NC * 1024;
    static_assert(O_RING + 3 * SBYTES == engine16_lds_bytes<NHT, NC>(), "LDS map");

    struct Pipe { h16x8 a[D], x0[D], x1[D]; f32x16 eb; };    // operand reads in flight for the next D MFMA pairs; next embedding bias
    struct OneHots { uint32_t a0[NC], a1[NC]; };             // LDS address of the one-hot fragment of chunk c (tile 0 / 1)

    PolicyDev pol;
    int tid, lane, wave, j, hh, n_kt, rp;
#ifdef TW_ABLATE   // diagnostic build: per-wave cycle stamps (s_memtime), summed into tw::g_stamps16 at kernel end
    unsigned long long st[8];      // 0 pre 1 prologue 2 stage bodies 3 vmcnt wait 4 stage barrier 5 heads 6 post 7 step barrier
#define TW_STAMP(var) const unsigned long long var = __builtin_readcyclecounter()
#define TW_ACC(i, a, b) st[i] += (b) - (a)
#else
#define TW_STAMP(var)
#define TW_ACC(i, a, b)
#endif
    uint8_t *lg_;                  // LDS base, generic
    uint32_t lds_u32, voff;        // LDS base as an M0 value; per-lane global byte offset of this wave's DMA piece
    lds_cu8 *L;                    // LDS base, address space 3
    h16x2 emb_lim, common_lim;

    // One DMA op = this wave's 1-KiB piece (wave + NW*op) of a stage: global (SGPR base + per-lane VGPR offset) -> LDS
    // (M0 = wave-uniform destination, the hardware adds lane*16).  Inline asm: see the LDS-DMA note in tw_engine.hpp.
    __device__ __forceinline__ void stream_op(const uint8_t *stage_base, uint32_t slot_m0, int op) const
    {
        TW_GLDS16(voff, slot_m0 + (uint32_t)op * (NW * 1024u), stage_base + (size_t)op * (NW * 1024));      // (M0 saved and restored: tw_common.hpp)
    }
    __device__ __forceinline__ const uint8_t *stage_ptr(int stage) const { return pol.stage16 + (size_t)stage * SBYTES; }
    __device__ __forceinline__ uint32_t slot_m0(int slot) const { return lds_u32 + O_RING + (uint32_t)slot * SBYTES + (uint32_t)wave * 1024u; }

    __device__ __forceinline__ void begin1(const PolicyDev &p, uint8_t *lds)
    {
        pol = p;
        tid = threadIdx.x; lane = tid & 63; wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        j = lane & 31; hh = lane >> 5;
        n_kt = pol.emb / 32;
        lg_ = lds; L = (lds_cu8 *)lds;
        lds_u32 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)lds);
        voff = (uint32_t)wave * 1024u + (uint32_t)lane * 16u;
        const _Float16 zl = (_Float16)0.0f, ml = (_Float16)(-__builtin_inff());
        emb_lim    = pol.emb_relu ? h16x2{zl, zl} : h16x2{ml, ml};
        common_lim = pol.common_relu ? h16x2{zl, zl} : h16x2{ml, ml};
        // constants
        for (int i = tid; i < NHT * 128; i += THREADS)
            reinterpret_cast<uint4 *>(lds + O_HEAD)[i] = reinterpret_cast<const uint4 *>(pol.head16)[i];
        for (int i = tid; i < NC * 64; i += THREADS)          // table tile 0 = table part of the LAST stage image
            reinterpret_cast<uint4 *>(lds + O_T0)[i] = reinterpret_cast<const uint4 *>(pol.stage16 + (size_t)(n_kt - 1) * SBYTES)[i];
        for (int i = tid; i < NHT * 32; i += THREADS) reinterpret_cast<float *>(lds + O_B1)[i] = pol.b1img16[i];
        if (tid < 16) reinterpret_cast<float *>(lds + O_BH)[tid] = tid < 8 ? pol.bh16[tid] : 0.0f;
        if (tid < 40) {   // one-hot fragments: entry p < 8 has 1.0 (0x3C00) in half p, entry 8 is zero
            const int e = tid >> 2, w = tid & 3;
            uint32_t v = 0;
            if (e < 8 && (e >> 1) == w) v = 0x3C00u << (16 * (e & 1));
            reinterpret_cast<uint32_t *>(lds + O_OH)[tid] = v;
        }
        const int np1 = pol.n_perms + 1;
        for (int i = tid; i < np1 * 16; i += THREADS) lds[O_SRC + i] = pol.srcmap16[i];
        for (int i = tid; i < np1 * 256; i += THREADS) lds[O_VMAP + i] = pol.vmap16[i];
        // one-hot fragment address of (lane half, twist, chunk, tile value): 16 * min(value' ^ 8h, 8), value' = twisted value
        for (int i = tid; i < 2 * np1 * 256; i += THREADS) {
            const int h2 = i / (np1 * 256), r = i - h2 * (np1 * 256);
            const uint32_t pos = (uint32_t)pol.vmap16[r] ^ ((uint32_t)h2 << 3);
            lds[O_OHB + h2 * ((E16_MAXP + 1) * 256) + r] = (uint8_t)((pos < 8u ? pos : 8u) * 16u);
        }
        if (tid < 4) lds[O_ACT + tid] = (uint8_t)tid;
        for (int i = tid; i < pol.n_perms * 4; i += THREADS) lds[O_ACT + 4 + i] = pol.act_perms[i];
        for (int i = tid; i < n_kt * 32; i += THREADS) reinterpret_cast<float *>(lds + O_EBIAS)[i] = pol.ebias16[i];
        // stages 0 and 1 of the first forward into slots 0 and 1
#pragma unroll
        for (int op = 0; op < NOPS; ++op) { stream_op(stage_ptr(0), slot_m0(0), op); stream_op(stage_ptr(n_kt > 1 ? 1 : 0), slot_m0(1), op); }
        rp = 0;
    }
    __device__ __forceinline__ void begin2() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __device__ __forceinline__ void end() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

    // LDS addresses of the one-hot B fragments of one episode under twist perm (-1 = none): 2*NC registers per
    // lane, computed once per timestep, so that a fragment read in the MFMA loop is ONE instruction.  Per chunk:
    // source cell (one 16-byte row read per twist), its tile, then one byte lookup that yields the address.
    __device__ __forceinline__ void onehots(uint64_t board, int perm, uint32_t (&w)[NC]) const
    {
        static_assert(O_OH == 0, "the byte table holds LDS addresses");
        const int pi = perm + 1;
        typedef uint32_t u32v4 __attribute__((ext_vector_type(4)));
        const u32v4 sr = *(const __attribute__((address_space(3))) u32v4 *)(L + O_SRC + pi * 16);
        const uint32_t srw[4] = {sr[0], sr[1], sr[2], sr[3]};
        const uint32_t tb = O_OHB + (uint32_t)(hh * (E16_MAXP + 1) + pi) * 256u;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const uint32_t src = (srw[c >> 2] >> (8 * (c & 3))) & 0xffu;
            const uint32_t v   = nib(board, (int)src);
            w[c] = L[tb + c * 16 + v];
        }
    }

    __device__ __forceinline__ void act_perm(int perm, float (&lg)[4]) const
    {
        typedef __attribute__((address_space(3))) const uint32_t lu1;
        const uint32_t ap = *(const lu1 *)(L + O_ACT + (perm + 1) * 4);      // perm -1: identity row
        const float l0 = lg[0], l1 = lg[1], l2 = lg[2], l3 = lg[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t src = (ap >> (8 * i)) & 0xffu;
            lg[i] = src == 0 ? l0 : (src == 1 ? l1 : (src == 2 ? l2 : l3));
        }
    }

    __device__ __forceinline__ f32x16 ld16(uint32_t off) const
    {
        typedef __attribute__((address_space(3))) const f32x4 lf4;
        const lf4 *p = (const lf4 *)(L + off);
        const f32x4 a = p[0], b = p[1], c = p[2], d = p[3];
        f32x16 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) { r[i] = a[i]; r[4 + i] = b[i]; r[8 + i] = c[i]; r[12 + i] = d[i]; }
        return r;
    }
    __device__ __forceinline__ h16x8 ld8(uint32_t off) const
    {
        return *(const __attribute__((address_space(3))) h16x8 *)(L + off);
    }
    // Embedding MFMA with the accumulator in ARCHITECTURAL VGPRs (inline asm, "v" constraints): the 2*NHT*16
    // common-layer accumulators fill the AGPR file; given the choice hipcc parks these two tiles there as well
    // and swaps common-layer tiles out and back every stage.  hipcc pads no hazard states around inline asm:
    // the chain e0 -> e0 is always separated by the other tile's MFMA (>= 32 cycles, more than any XDL->SrcC
    // requirement), operands come from ds_read (s_waitcnt is inserted for asm operands), and the first VALU
    // read of the result is at least one MFMA pair later (phase()) or behind explicit s_nops (prologue).
    // The first MFMA of a chain takes the embedding bias (f32, accumulator-register order) as its C operand.
    static __device__ __forceinline__ void mfma_v(f32x16 &d, const h16x8 a, const h16x8 b, bool first, const f32x16 &c0)
    {
        if constexpr (TW_MFMA_INTRIN & 8) d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, first ? c0 : d, 0, 0, 0);
        else if (first) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c0));
        else       asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
    }
    // one B register: accumulator registers (2q, 2q+1) of fragment m (+ bias), optional ReLU, to f16
    template <bool BIAS>
    static __device__ __forceinline__ void cvt_unit(const f32x16 &e, const f32x16 &bias, int m, int q, h16x2 lim, h16x8 &dst)
    {
        f32x2 v = {e[8 * m + 2 * q], e[8 * m + 2 * q + 1]};
        if (BIAS) v = pk_add(v, f32x2{bias[8 * m + 2 * q], bias[8 * m + 2 * q + 1]});
        h16x2 p = __builtin_convertvector(v, h16x2);
        p = __builtin_elementwise_max(p, lim);
        dst[2 * q] = p[0]; dst[2 * q + 1] = p[1];
    }

    // Pair order of a phase: two embedding pairs (three LDS reads each) per common-layer pair (one read) until
    // the embedding pairs are used up -- an even LDS load instead of a burst -- keeping the last common-layer
    // pairs for the shadow of the accumulator conversion.  code(i) = 2*index + (1 if embedding pair).
    template <bool E_P, bool M_P> struct Sched {
        static constexpr int NE = E_P ? NC : 0, NM = M_P ? 2 * NHT : 0, NP = NE + NM;
        static constexpr int KEEP = NM < 8 ? NM : 8;
        static constexpr int code(int i)
        {
            int e = 0, m = 0;
            for (int k = 0; k <= i; ++k) {
                bool take_e = false;
                if (e >= NE) take_e = false;
                else if (m >= NM || NM - m <= KEEP) take_e = true;
                else take_e = (k % 3) != 2;
                if (k == i) return take_e ? e * 2 + 1 : m * 2;
                if (take_e) ++e; else ++m;
            }
            return 0;
        }
        static constexpr int last_e()   // position of the last embedding pair (-1: none)
        {
            int r = -1;
            for (int i = 0; i < NP; ++i) if (code(i) & 1) r = i;
            return r;
        }
    };

    // One phase of the software pipeline.
    //   E_P: NC embedding MFMA pairs (A operands at baseE + 1 KiB * c, B = one-hots, C of the first = pp.eb), results
    //        converted into (Bn0, Bn1) -- in the shadow of the trailing M pairs when M_P, else right after the loop;
    //   M_P: 2*NHT common-layer MFMA pairs with (Bc0, Bc1) (A operands at baseM + 1 KiB * q);
    //   NEXT: prefetch the operands of the first D pairs of the following stage (pair order Sched<true, true>,
    //        embedding A operands at nbase, common-layer ones at nbase + NC KiB) and the embedding bias of tile ke_next;
    //   STREAM: issue this wave's DMA ops of stage sg into slot s2, then wait for them and pass the stage barrier.
    template <bool E_P, bool M_P, bool FIRST, int NEXT, bool STREAM>
    __device__ __forceinline__ void phase(int ke_next, uint32_t baseE, uint32_t baseM, uint32_t nbase, int sg, int s2,
                                          const OneHots &oh, f32x16 (&acc0)[NHT], f32x16 (&acc1)[NHT],
                                          const h16x8 (&Bc0)[2], const h16x8 (&Bc1)[2], h16x8 (&Bn0)[2], h16x8 (&Bn1)[2], Pipe &pp)
    {
        TW_STAMP(t_in);
        using S = Sched<E_P, M_P>;
        using SN = Sched<true, true>;

        constexpr int NP = S::NP, LE = S::last_e();
        constexpr int MAFTER = NP - 1 - LE;                                     // common-layer pairs behind the last embedding pair
        constexpr int UPP = !M_P ? 16 : (MAFTER > 1 ? (16 + MAFTER - 2) / (MAFTER - 1) : 16);
        const uint8_t *g_stage = stage_ptr(sg);
        const uint32_t m0_slot = slot_m0(s2);
        h16x8 A[NP + D], X0[NP + D], X1[NP + D];                                // indexed by POSITION in the pair order
#pragma unroll
        for (int d = 0; d < D; ++d) { A[d] = pp.a[d]; X0[d] = pp.x0[d]; X1[d] = pp.x1[d]; }
        const f32x16 eb = pp.eb;
        f32x16 zero16;
#pragma unroll
        for (int g = 0; g < 16; ++g) zero16[g] = 0.0f;
        f32x16 e0, e1;
        f32x16 cbv[2];                                                           // FIRST: b1 of hidden tile ht (parity ht&1) as the C operand
        auto unit = [&](int u) {      // u in [0,16): tile u>>3, fragment (u>>2)&1, register u&3
            if ((u >> 3) == 0) cvt_unit<false>(e0, zero16, (u >> 2) & 1, u & 3, emb_lim, Bn0[(u >> 2) & 1]);
            else               cvt_unit<false>(e1, zero16, (u >> 2) & 1, u & 3, emb_lim, Bn1[(u >> 2) & 1]);
        };
        if (FIRST) {                                                            // b1 for common-layer pairs at positions 0 / 1 (no lookahead room)
#pragma unroll
            for (int p0 = 0; p0 < 2 && p0 < NP; ++p0) {
                const int c0 = S::code(p0);
                if (!(c0 & 1) && ((c0 >> 1) & 1) == 0) cbv[((c0 >> 1) >> 1) & 1] = ld16(O_B1 + (uint32_t)(((c0 >> 1) >> 1) * 2 + hh) * 64u);
            }
        }
        int op = 0, udone = 0;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            // ---- operand reads of position p + D (one-hots first: the A operand's arrival then covers all three)
            const int pn = p + D;
            if (pn < NP) {
                const int cd = S::code(pn), ix = cd >> 1;
                if (cd & 1) {
                    X0[pn] = ld8(oh.a0[ix]); X1[pn] = ld8(oh.a1[ix]);
                    A[pn] = ld8(baseE + ix * 1024);
                } else A[pn] = ld8(baseM + ix * 1024);
            } else if (NEXT != 0) {
                const int cd = SN::code(pn - NP), ix = cd >> 1;
                if (cd & 1) { X0[pn] = ld8(oh.a0[ix]); X1[pn] = ld8(oh.a1[ix]); A[pn] = ld8(nbase + ix * 1024); }
                else A[pn] = ld8(nbase + (NC + ix) * 1024);
            }
            if (NEXT != 0 && p == NP - 1) pp.eb = ld16(O_EBIAS + (uint32_t)(ke_next * 2 + hh) * 64u);
            if (FIRST && p + 2 < NP) {                                          // b1 of the hidden tile whose first pair is two positions ahead
                const int c2 = S::code(p + 2);
                if (!(c2 & 1) && ((c2 >> 1) & 1) == 0) cbv[((c2 >> 1) >> 1) & 1] = ld16(O_B1 + (uint32_t)(((c2 >> 1) >> 1) * 2 + hh) * 64u);
            }
            __builtin_amdgcn_sched_barrier(0);
            const int cd = S::code(p), ix = cd >> 1;
            const bool is_e = cd & 1;
            // ---- first MFMA of the pair
            if (is_e) mfma_v(e0, A[p], X0[p], ix == 0, eb);
            else {
                const int ht = ix >> 1, m = ix & 1;
                acc0[ht] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[p], Bc0[m], (FIRST && m == 0) ? cbv[ht & 1] : acc0[ht], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- side work A: DMA
            if (STREAM) {
                constexpr int SPREAD = NP * 5 / 8 > 0 ? NP * 5 / 8 : 1;          // all ops within the first 5/8 of the phase
                while (op < NOPS && op * SPREAD / NOPS <= p) { stream_op(g_stage, m0_slot, op); ++op; }
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- second MFMA of the pair
            if (is_e) mfma_v(e1, A[p], X1[p], ix == 0, eb);
            else {
                const int ht = ix >> 1, m = ix & 1;
                acc1[ht] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[p], Bc1[m], (FIRST && m == 0) ? cbv[ht & 1] : acc1[ht], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- side work B: conversion of the embedding tile computed in this phase
            if (E_P && M_P && p > LE + 1) {
                const int lim = (p - LE - 1) * UPP < 16 ? (p - LE - 1) * UPP : 16;
                // One hidden tile: only two common-layer pairs follow the last embedding pair, and hipcc moves these conversions in
                // front of them (they do not depend on each other) -- straight behind the asm MFMA of e1, 8 wait states where the
                // XDL write -> VALU read needs 12 (scripts/scan_mfma_hazards.py found it; with more tiles the pairs stay in between
                // and the build's scan keeps watching).  The pad is part of the dependency chain of e0 / e1, so it cannot move.
                if (NHT < 2 && udone == 0 && lim > 0) asm volatile("s_nop 11" : "+v"(e0), "+v"(e1));
                for (; udone < lim; ++udone) unit(udone);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (E_P) {
            // XDL write -> VALU read of e0/e1 (asm MFMA: no automatic padding).  The accumulators are operands of
            // the asm so that the conversions below cannot be scheduled in front of the wait states.
            if (!M_P || MAFTER < 1) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(e0), "+v"(e1));
#pragma unroll
            for (int u = 0; u < 16; ++u) if (u >= udone) unit(u);
        }
        if (NEXT != 0) {
#pragma unroll
            for (int d = 0; d < D; ++d) { pp.a[d] = A[NP + d]; pp.x0[d] = X0[NP + d]; pp.x1[d] = X1[NP + d]; }
        }
        if (STREAM) {
#pragma unroll
            for (; op < NOPS; ++op) stream_op(g_stage, m0_slot, op);
            TW_STAMP(t_b);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of stage sg have landed
            TW_STAMP(t_w);
            __syncthreads();
            TW_STAMP(t_s);
            TW_ACC(2, t_in, t_b); TW_ACC(3, t_b, t_w); TW_ACC(4, t_w, t_s);
        }
    }

    // policy forward for the wave's two column tiles; out0/out1: rows (registers) 0..3 logits, 4 value (no head bias)
    __device__ __forceinline__ void forward(const OneHots &oh, f32x16 &out0, f32x16 &out1)
    {
        f32x16 acc0[NHT], acc1[NHT];
        h16x8 Ba0[2], Ba1[2], Bb0[2], Bb1[2];                   // B fragments of the current / next embedding tile (ping-pong)
        Pipe pp;
        const uint32_t lo = (uint32_t)lane * 16u;
        auto slot_base = [&](int s) { return O_RING + (uint32_t)s * SBYTES + lo; };
        auto stage_of = [&](int kt) { int sg = kt + 2; if (sg >= n_kt) sg -= n_kt; if (sg >= n_kt) sg -= n_kt; return sg; };
        auto tile_of  = [&](int kt) { return kt < n_kt ? kt : 0; };   // embedding tile computed in stage kt-1 (the last stage's is discarded)
        int s0 = rp;
        TW_STAMP(t_p0);
        // prologue: embedding tile 0 from the resident copy (its bias and first D operand reads are exposed)
        pp.eb = ld16(O_EBIAS + (uint32_t)hh * 64u);
#pragma unroll
        for (int d = 0; d < D; ++d) { pp.x0[d] = ld8(oh.a0[d]); pp.x1[d] = ld8(oh.a1[d]); pp.a[d] = ld8(O_T0 + lo + d * 1024); }
        phase<true, false, false, 1, false>(tile_of(1), O_T0 + lo, 0, slot_base(s0), 0, 0, oh, acc0, acc1, Ba0, Ba1, Ba0, Ba1, pp);
        TW_STAMP(t_p1);
        TW_ACC(1, t_p0, t_p1);
        // every stage but the last: embedding MFMAs of tile kt+1 interleaved with the common-layer MFMAs of tile kt;
        // two stages per loop trip: the B fragments ping-pong between (Ba) and (Bb) without copies.  The last stage
        // has only the common-layer part; it fetches its first operands itself (the stage before it prefetched for
        // the order of a full stage).
        auto stage = [&](auto first, int kt, const h16x8 (&c0)[2], const h16x8 (&c1)[2], h16x8 (&n0)[2], h16x8 (&n1)[2]) {
            const int s1 = s0 == 2 ? 0 : s0 + 1, s2 = s0 == 0 ? 2 : s0 - 1;
            phase<true, true, decltype(first)::value, 1, true>(tile_of(kt + 2), slot_base(s0), slot_base(s0) + NC * 1024, slot_base(s1),
                                                               stage_of(kt), s2, oh, acc0, acc1, c0, c1, n0, n1, pp);
            s0 = s1;
        };
        auto last = [&](auto first, int kt, const h16x8 (&c0)[2], const h16x8 (&c1)[2]) {
            const int s1 = s0 == 2 ? 0 : s0 + 1, s2 = s0 == 0 ? 2 : s0 - 1;
            const uint32_t bM = slot_base(s0) + NC * 1024;
#pragma unroll
            for (int d = 0; d < D; ++d) pp.a[d] = ld8(bM + d * 1024);
            h16x8 dummy0[2], dummy1[2];
            phase<false, true, decltype(first)::value, 0, true>(0, 0, bM, 0, stage_of(kt), s2, oh, acc0, acc1, c0, c1, dummy0, dummy1, pp);
            s0 = s1;
        };
        if (n_kt == 1) last(std::true_type{}, 0, Ba0, Ba1);
        else {
            stage(std::true_type{}, 0, Ba0, Ba1, Bb0, Bb1);
            int kt = 1;
            for (; kt + 2 < n_kt; kt += 2) {
                stage(std::false_type{}, kt, Bb0, Bb1, Ba0, Ba1);
                stage(std::false_type{}, kt + 1, Ba0, Ba1, Bb0, Bb1);
            }
            if (kt + 1 < n_kt) {                 // odd number of stages left: one more, then move its fragments (16 registers)
                stage(std::false_type{}, kt, Bb0, Bb1, Ba0, Ba1);
#pragma unroll
                for (int m = 0; m < 2; ++m) { Bb0[m] = Ba0[m]; Bb1[m] = Ba1[m]; }
                ++kt;
            }
            last(std::false_type{}, kt, Bb0, Bb1);
        }
        rp = s0;
        TW_STAMP(t_h0);
        // heads: h1 = relu(acc) in f16 is the B operand (b1 went in as the C operand of the first stage), hidden
        // index in accumulator-register order
        f32x16 h0, h1;
#pragma unroll
        for (int g = 0; g < 16; ++g) { h0[g] = 0.0f; h1[g] = 0.0f; }
#pragma unroll
        for (int ht = 0; ht < NHT; ++ht) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const h16x8 a = ld8(O_HEAD + (uint32_t)(ht * 2 + m) * 1024u + lo);
                h16x8 b0, b1;
#pragma unroll
                for (int q = 0; q < 4; ++q) { cvt_unit<false>(acc0[ht], h0, m, q, common_lim, b0); cvt_unit<false>(acc1[ht], h0, m, q, common_lim, b1); }
                h0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b0, h0, 0, 0, 0);
                h1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b1, h1, 0, 0, 0);
            }
        }
        out0 = h0; out1 = h1;
#ifdef TW_ABLATE
        asm volatile("" :: "v"(h0), "v"(h1));
#endif
        TW_STAMP(t_h1);
        TW_ACC(5, t_h0, t_h1);
    }
};

}  // namespace tw
