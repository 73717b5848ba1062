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

namespace tw {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const uint8_t lds_cu8;

constexpr int E16_MAXP   = 4;      // twists supported by the f16 engine (tables in LDS)
constexpr int E16_MAX_KT = 32;     // embedding size <= 1024

template <int NHT, int NC>
__host__ __device__ constexpr size_t engine16_lds_bytes()
{
    return 160 + (size_t)(E16_MAXP + 1) * (16 + 256) + 32 + 64 + (size_t)NHT * 128 + (size_t)E16_MAX_KT * 128 +
           (size_t)NHT * 2048 + (size_t)NC * 1024 + (size_t)3 * (NC + 2 * NHT) * 1024;
}

template <int NHT, int NC>
struct Engine16 {
    static constexpr int NW = 4, THREADS = 256, EPB = 256, D = 3;
    static constexpr int SP     = NC + 2 * NHT;              // 1-KiB pieces per stage
    static constexpr int SBYTES = SP * 1024;
    static constexpr int NOPS   = (SP + NW - 1) / NW;        // DMA ops per wave per stage
    // LDS map (small tables first: their offsets fit the 16-bit ds offset field)
    static constexpr uint32_t O_OH = 0, O_SRC = 160, O_VMAP = O_SRC + (E16_MAXP + 1) * 16, O_ACT = O_VMAP + (E16_MAXP + 1) * 256,
                              O_BH = O_ACT + 32, O_B1 = O_BH + 64, O_EBIAS = O_B1 + NHT * 128, O_HEAD = O_EBIAS + E16_MAX_KT * 128,
                              O_T0 = O_HEAD + NHT * 2048, O_RING = O_T0 + NC * 1024;
    static_assert(O_RING + 3 * SBYTES == engine16_lds_bytes<NHT, NC>(), "LDS map");

    struct Pipe { h16x8 a[D], x0[D], x1[D]; };               // operand reads in flight for the next D MFMA pairs
    struct OneHots { uint32_t w0[(NC + 3) / 4], w1[(NC + 3) / 4]; };   // byte c = 16 * one-hot table index of chunk c (tile 0 / 1)

    PolicyDev pol;
    int tid, lane, wave, j, hh, n_kt, rp;
    uint8_t *lg_;                  // LDS base, generic
    lds_cu8 *L;                    // LDS base, address space 3
    h16x2 emb_lim, common_lim;

    __device__ __forceinline__ void stream_op(int stage, int slot, int op)
    {
        int piece = wave + NW * op;
        piece = piece < SP ? piece : SP - 1;                 // branch-free: past the end repeat the last piece
        const uint8_t *src = pol.stage16 + (size_t)stage * SBYTES + piece * 1024 + lane * 16;
        glds16(reinterpret_cast<const float *>(src), reinterpret_cast<float *>(lg_ + O_RING + slot * SBYTES + piece * 1024));
    }

    __device__ __forceinline__ void begin1(const PolicyDev &p, uint8_t *lds)
    {
        pol = p;
        tid = threadIdx.x; lane = tid & 63; wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        j = lane & 31; hh = lane >> 5;
        n_kt = pol.emb / 32;
        lg_ = lds; L = (lds_cu8 *)lds;
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
        if (tid < 4) lds[O_ACT + tid] = (uint8_t)tid;
        for (int i = tid; i < pol.n_perms * 4; i += THREADS) lds[O_ACT + 4 + i] = pol.act_perms[i];
        for (int i = tid; i < n_kt * 32; i += THREADS) reinterpret_cast<float *>(lds + O_EBIAS)[i] = pol.ebias16[i];
        // stages 0 and 1 of the first forward into slots 0 and 1
#pragma unroll
        for (int op = 0; op < NOPS; ++op) { stream_op(0, 0, op); stream_op(n_kt > 1 ? 1 : 0, 1, op); }
        rp = 0;
    }
    __device__ __forceinline__ void begin2() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __device__ __forceinline__ void end() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

    // one-hot table indices (x16, one byte per chunk) of one episode under twist perm (-1 = none)
    __device__ __forceinline__ void onehots(uint64_t board, int perm, uint32_t (&w)[(NC + 3) / 4]) const
    {
        const int pi = perm + 1;
#pragma unroll
        for (int q = 0; q < (NC + 3) / 4; ++q) w[q] = 0u;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const uint32_t src = L[O_SRC + pi * 16 + c];
            const uint32_t v   = nib(board, (int)src);
            const uint32_t v2  = L[O_VMAP + (pi * 16 + c) * 16 + v];
            const uint32_t pos = v2 ^ ((uint32_t)hh << 3);
            w[c >> 2] |= ((pos < 8u ? pos : 8u) * 16u) << (8 * (c & 3));
        }
    }

    __device__ __forceinline__ void act_perm(int perm, float (&lg)[4]) const
    {
        if (perm < 0) return;
        const float l0 = lg[0], l1 = lg[1], l2 = lg[2], l3 = lg[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int src = L[O_ACT + (perm + 1) * 4 + i];
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
    __device__ __forceinline__ h16x8 ldoh(const uint32_t (&w)[(NC + 3) / 4], int c) const
    {
        return ld8(O_OH + ((w[c >> 2] >> (8 * (c & 3))) & 0xffu));
    }
    // Embedding MFMA with the accumulator in ARCHITECTURAL VGPRs (inline asm, "v" constraints): the 2*NHT*16
    // common-layer accumulators fill the AGPR file; given the choice hipcc parks these two tiles there as well
    // and swaps common-layer tiles out and back every stage.  hipcc pads no hazard states around inline asm:
    // the chain e0 -> e0 is always separated by the other tile's MFMA (>= 32 cycles, more than any XDL->SrcC
    // requirement), operands come from ds_read (s_waitcnt is inserted for asm operands), and the first VALU
    // read of the result is at least one MFMA pair later (phase()) or behind explicit s_nops (prologue).
    static __device__ __forceinline__ void mfma_v(f32x16 &d, const h16x8 a, const h16x8 b, bool first)
    {
        if (first) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
        else       asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
    }
    // one B register: accumulator registers (2q, 2q+1) of fragment m, plus bias, optional ReLU, to f16
    static __device__ __forceinline__ void cvt_unit(const f32x16 &e, const f32x16 &bias, int m, int q, h16x2 lim, h16x8 &dst)
    {
        const f32x2 v = {e[8 * m + 2 * q] + bias[8 * m + 2 * q], e[8 * m + 2 * q + 1] + bias[8 * m + 2 * q + 1]};
        h16x2 p = __builtin_convertvector(v, h16x2);
        p = __builtin_elementwise_max(p, lim);
        dst[2 * q] = p[0]; dst[2 * q + 1] = p[1];
    }

    // One phase of the software pipeline.
    //   E_P: NC embedding MFMA pairs of tile `ke` (A operands at baseE + 1 KiB * c, B = one-hots), results
    //        converted into (B0, B1) -- in the shadow of the M pairs when M_P, else right after the loop;
    //   M_P: 2*NHT common-layer MFMA pairs with the CURRENT (B0, B1) (A operands at baseM + 1 KiB * q);
    //   NEXT: which operands to prefetch for the first D pairs of the following phase: 0 none, 1 its E pairs
    //        (A at nbase, one-hots), 2 its M pairs (A at nbase);  STREAM: issue this wave's DMA ops of stage sg
    //        into slot s2, then wait for them and pass the stage barrier.
    template <bool E_P, bool M_P, bool FIRST, int NEXT, bool STREAM>
    __device__ __forceinline__ void phase(int ke, uint32_t baseE, uint32_t baseM, uint32_t nbase, int sg, int s2,
                                          const OneHots &oh_in, f32x16 (&acc0)[NHT], f32x16 (&acc1)[NHT],
                                          h16x8 (&B0)[2], h16x8 (&B1)[2], Pipe &pp)
    {
        constexpr int NE = E_P ? NC : 0, NM = M_P ? 2 * NHT : 0, NP = NE + NM;
        // the one-hot bytes are loop invariant over the stages: keep hipcc from hoisting the 2*NC byte extracts
        // (and their registers) out of the stage loop
        OneHots oh = oh_in;
#pragma unroll
        for (int q = 0; q < (NC + 3) / 4; ++q) asm volatile("" : "+v"(oh.w0[q]), "+v"(oh.w1[q]));
        constexpr int UPP = !M_P ? 16 : (NM > 1 ? (16 + NM - 2) / (NM - 1) : 16);   // conversion units per M pair (from pair 1)
        h16x8 A[NP + D], X0[NE + D], X1[NE + D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            A[d] = pp.a[d];
            if (E_P) { X0[d] = pp.x0[d]; X1[d] = pp.x1[d]; }
        }
        f32x16 zero16;
#pragma unroll
        for (int g = 0; g < 16; ++g) zero16[g] = 0.0f;
        f32x16 e0, e1, ebv = zero16;
        h16x8 Bn0[2], Bn1[2];
        auto unit = [&](int u) {      // u in [0,16): tile u>>3, fragment (u>>2)&1, register u&3
            if ((u >> 3) == 0) cvt_unit(e0, ebv, (u >> 2) & 1, u & 3, emb_lim, Bn0[(u >> 2) & 1]);
            else               cvt_unit(e1, ebv, (u >> 2) & 1, u & 3, emb_lim, Bn1[(u >> 2) & 1]);
        };
        int op = 0, udone = 0;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            // ---- operand reads of pair p + D
            const int pn = p + D;
            if (pn < NP) {
                A[pn] = ld8((pn < NE ? baseE + pn * 1024 : baseM + (pn - NE) * 1024));
                if (pn < NE) { X0[pn] = ldoh(oh.w0, pn); X1[pn] = ldoh(oh.w1, pn); }
            } else if (NEXT != 0) {
                const int q = pn - NP;
                A[pn] = ld8(nbase + q * 1024);
                if (NEXT == 1) { X0[NE + q] = ldoh(oh.w0, q); X1[NE + q] = ldoh(oh.w1, q); }
            }
            if (E_P && M_P && p == NE) ebv = ld16(O_EBIAS + (uint32_t)(ke * 2 + hh) * 64u);
            __builtin_amdgcn_sched_barrier(0);
            // ---- first MFMA of the pair
            if (p < NE) mfma_v(e0, A[p], X0[p], p == 0);
            else {
                const int q = p - NE, ht = q >> 1, m = q & 1;
                acc0[ht] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[p], B0[m], (FIRST && m == 0) ? zero16 : acc0[ht], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- side work A: DMA
            if (STREAM) {
                constexpr int SPREAD = NP * 5 / 8 > 0 ? NP * 5 / 8 : 1;          // all ops within the first 5/8 of the phase
                while (op < NOPS && op * SPREAD / NOPS <= p) { stream_op(sg, s2, op); ++op; }
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- second MFMA of the pair
            if (p < NE) mfma_v(e1, A[p], X1[p], p == 0);
            else {
                const int q = p - NE, ht = q >> 1, m = q & 1;
                acc1[ht] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[p], B1[m], (FIRST && m == 0) ? zero16 : acc1[ht], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- side work B: conversion of the embedding tile computed in this phase
            if (E_P && M_P && p > NE) {
                const int lim = (p - NE) * UPP < 16 ? (p - NE) * UPP : 16;
                for (; udone < lim; ++udone) unit(udone);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (E_P) {
            if (!M_P) {
                ebv = ld16(O_EBIAS + (uint32_t)(ke * 2 + hh) * 64u);
                asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // XDL write -> VALU read of e0/e1 (asm MFMA: no automatic padding)
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) if (u >= udone) unit(u);
#pragma unroll
            for (int m = 0; m < 2; ++m) { B0[m] = Bn0[m]; B1[m] = Bn1[m]; }
        }
        if (NEXT != 0) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                pp.a[d] = A[NP + d];
                if (NEXT == 1) { pp.x0[d] = X0[NE + d]; pp.x1[d] = X1[NE + d]; }
            }
        }
        if (STREAM) {
#pragma unroll
            for (; op < NOPS; ++op) stream_op(sg, s2, op);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of stage sg have landed
            __syncthreads();
        }
    }

    // policy forward for the wave's two column tiles; out0/out1: rows (registers) 0..3 logits, 4 value (no head bias)
    __device__ __forceinline__ void forward(const OneHots &oh, f32x16 &out0, f32x16 &out1)
    {
        f32x16 acc0[NHT], acc1[NHT];
        h16x8 B0[2], B1[2];
        Pipe pp;
        const uint32_t lo = (uint32_t)lane * 16u;
        auto slot_base = [&](int s) { return O_RING + (uint32_t)s * SBYTES + lo; };
        auto stage_of = [&](int kt) { int sg = kt + 2; if (sg >= n_kt) sg -= n_kt; if (sg >= n_kt) sg -= n_kt; return sg; };
        int s0 = rp;
        // prologue: embedding tile 0 from the resident copy (its first D operand reads are exposed)
#pragma unroll
        for (int d = 0; d < D; ++d) { pp.a[d] = ld8(O_T0 + lo + d * 1024); pp.x0[d] = ldoh(oh.w0, d); pp.x1[d] = ldoh(oh.w1, d); }
        phase<true, false, false, 1, false>(0, O_T0 + lo, 0, slot_base(s0), 0, 0, oh, acc0, acc1, B0, B1, pp);
        // every iteration: embedding MFMAs of tile kt+1, then common-layer MFMAs of tile kt.  (In the last
        // iteration the embedding part runs on tile 0 again and is discarded: one uniform loop body keeps
        // the register allocation of the 512-register kernel simple; 3 % of the MFMAs.)
        {
            const int s1 = s0 == 2 ? 0 : s0 + 1, s2 = s0 == 0 ? 2 : s0 - 1;
            phase<true, true, true, 1, true>(n_kt > 1 ? 1 : 0, slot_base(s0), slot_base(s0) + NC * 1024, slot_base(s1), stage_of(0), s2,
                                             oh, acc0, acc1, B0, B1, pp);
            s0 = s1;
        }
        for (int kt = 1; kt < n_kt; ++kt) {
            const int s1 = s0 == 2 ? 0 : s0 + 1, s2 = s0 == 0 ? 2 : s0 - 1;
            phase<true, true, false, 1, true>(kt + 1 < n_kt ? kt + 1 : 0, slot_base(s0), slot_base(s0) + NC * 1024, slot_base(s1),
                                              stage_of(kt), s2, oh, acc0, acc1, B0, B1, pp);
            s0 = s1;
        }
        rp = s0;
        // heads: h1 = relu(acc + b1) in f16 is the B operand, hidden index in accumulator-register order
        f32x16 h0, h1;
#pragma unroll
        for (int g = 0; g < 16; ++g) { h0[g] = 0.0f; h1[g] = 0.0f; }
#pragma unroll
        for (int ht = 0; ht < NHT; ++ht) {
            const f32x16 cb = ld16(O_B1 + (uint32_t)(ht * 2 + hh) * 64u);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const h16x8 a = ld8(O_HEAD + (uint32_t)(ht * 2 + m) * 1024u + lo);
                h16x8 b0, b1;
#pragma unroll
                for (int q = 0; q < 4; ++q) { cvt_unit(acc0[ht], cb, m, q, common_lim, b0); cvt_unit(acc1[ht], cb, m, q, common_lim, b1); }
                h0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b0, h0, 0, 0, 0);
                h1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b1, h1, 0, 0, 0);
            }
        }
        out0 = h0; out1 = h1;
    }
};

}  // namespace tw
