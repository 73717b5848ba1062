// tw_engine16x2.hpp -- f32-equivalent policy forward on the f16 matrix core (TW_PREC_F16X2) for gfx950.
//
// Every f32 operand is split into two binary16 terms, x = x_hi + x_lo (x_hi = f16(x), x_lo = f16(x - x_hi): 22
// significant bits).  Products of f16 numbers are exact in the f32 accumulator of v_mfma_f32_32x32x16_f16, so
//   embedding    bias + sum of table rows      = one-hot x T_hi  +  one-hot x T_lo                 (2 MFMAs per chunk)
//   common layer W1 . h0                        = W_hi.h_hi + W_hi.h_lo + W_lo.h_hi  (+ W_lo.h_lo ~ 2^-22, dropped)
//   heads        likewise, three products
// reproduce the f32 result to accumulation rounding (logits within ~1e-6 of the reference's f32 arithmetic; tests
// allow 1e-5, the tolerance BASELINE.json states).  All operands are pre-scaled by exact powers of two so that the
// low terms stay normal f16 numbers and no multiplies are needed on the way: table, embedding bias, W1 and the head
// weights x16, b1 x256; then relu(e) of the embedding accumulator IS 16*h0, relu(acc) of the common accumulator IS
// 256*h1, and the head accumulator is 4096*(logit - bias).
//
// Same mapping as tw_engine16.hpp (one wave per SIMD, two column tiles per wave, embedding on the matrix core,
// accumulators converted in registers, ring of three LDS slots fed by LDS-DMA).  The hi and lo images are ALTERNATE
// ring stages: stage 2k holds [T_hi(k+1) | W_hi(k)], stage 2k+1 holds [T_lo(k+1) | W_lo(k)]; the embedding
// accumulators of tile k+1 are carried across the pair; stage 2*n_kt holds the head images [hi | lo].
// 160 MFMAs per embedding tile instead of 64 -- against 16x fewer cycles per MFMA than the f32 path.
#pragma once
#include "tw_engine16.hpp"

#include <utility>

namespace tw {

template <int NHT, int NC>
__host__ __device__ constexpr int engineS_pieces() { return (NC + 2 * NHT > 4 * NHT) ? NC + 2 * NHT : 4 * NHT; }
template <int NHT, int NC>
__host__ __device__ constexpr size_t engineS_lds_bytes()
{
    return 160 + (size_t)(E16_MAXP + 1) * (16 + 256) + 32 + (size_t)2 * (E16_MAXP + 1) * 256 + 64 + (size_t)NHT * 128 +
           (size_t)E16_MAX_KT * 128 + (size_t)2 * NC * 1024 + (size_t)3 * ((engineS_pieces<NHT, NC>() + 3) / 4) * 4096;
}

template <int NHT, int NC>
struct EngineS {
    static constexpr int NW = 4, THREADS = 256, EPB = 256, D = 3;
    static constexpr int NM0    = 2 * NHT;                   // W1 (or head) chunks per image
    static constexpr int SP     = engineS_pieces<NHT, NC>();
    static constexpr int NOPS   = (SP + NW - 1) / NW;
    static constexpr int SBYTES = NOPS * NW * 1024;
    static constexpr float OUT_SCALE = 1.0f / 4096.0f;       // head accumulator -> logit - bias
    static constexpr uint32_t O_OH = 0, O_SRC = 160, O_VMAP = O_SRC + (E16_MAXP + 1) * 16, O_ACT = O_VMAP + (E16_MAXP + 1) * 256,
                              O_OHB = O_ACT + 32, O_BH = O_OHB + 2 * (E16_MAXP + 1) * 256, O_B1 = O_BH + 64, O_EBIAS = O_B1 + NHT * 128,
                              O_T0 = O_EBIAS + E16_MAX_KT * 128, O_RING = O_T0 + 2 * NC * 1024;
    static_assert(O_RING + 3 * SBYTES == engineS_lds_bytes<NHT, NC>(), "LDS map");

    struct Pipe { h16x8 a[D], x0[D], x1[D]; f32x16 eb; };
    struct OneHots { uint32_t a0[(NC + 3) / 4], a1[(NC + 3) / 4]; };   // byte c = LDS address of the one-hot fragment of chunk c (tile 0 / 1)
    struct Frags { h16x8 h0[2], h1[2], l0[2], l1[2]; };      // hi / lo B fragments (k-steps 0,1) of tile 0 / tile 1

    PolicyDev pol;
    int tid, lane, wave, j, hh, n_kt, rp, n_stages;
#ifdef TW_ABLATE
    unsigned long long st[8];      // (stamps are only taken around the forward in this engine)
#endif
    uint8_t *lg_;
    uint32_t lds_u32, voff;
    lds_cu8 *L;
    float emb_lim, common_lim;                               // 0 (ReLU) or -inf (none)

    __device__ __forceinline__ void stream_op(const uint8_t *stage_base, uint32_t slot_m0, int op) const
    {
        TW_GLDS16(voff, slot_m0 + (uint32_t)op * (NW * 1024u), stage_base + (size_t)op * (NW * 1024));      // (M0 saved and restored: tw_common.hpp)
    }
    __device__ __forceinline__ const uint8_t *stage_ptr(int stage) const { return pol.stageS + (size_t)stage * SBYTES; }
    __device__ __forceinline__ uint32_t slot_m0(int slot) const { return lds_u32 + O_RING + (uint32_t)slot * SBYTES + (uint32_t)wave * 1024u; }
    __device__ __forceinline__ int wrap(int q) const { return q >= n_stages ? q - n_stages : q; }

    __device__ __forceinline__ void begin1(const PolicyDev &p, uint8_t *lds)
    {
        pol = p;
        tid = threadIdx.x; lane = tid & 63; wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        j = lane & 31; hh = lane >> 5;
        n_kt = pol.emb / 32; n_stages = 2 * n_kt + 1;
        lg_ = lds; L = (lds_cu8 *)lds;
        lds_u32 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)lds);
        voff = (uint32_t)wave * 1024u + (uint32_t)lane * 16u;
        emb_lim    = pol.emb_relu ? 0.0f : -__builtin_inff();
        common_lim = pol.common_relu ? 0.0f : -__builtin_inff();
        for (int i = tid; i < 2 * NC * 64; i += THREADS)      // table tile 0, hi chunks then lo chunks
            reinterpret_cast<uint4 *>(lds + O_T0)[i] = reinterpret_cast<const uint4 *>(pol.t0S)[i];
        for (int i = tid; i < NHT * 32; i += THREADS) reinterpret_cast<float *>(lds + O_B1)[i] = pol.b1img16[i] * 256.0f;
        if (tid < 16) reinterpret_cast<float *>(lds + O_BH)[tid] = tid < 8 ? pol.bh16[tid] : 0.0f;
        if (tid < 40) {
            const int e = tid >> 2, w = tid & 3;
            uint32_t v = 0;
            if (e < 8 && (e >> 1) == w) v = 0x3C00u << (16 * (e & 1));
            reinterpret_cast<uint32_t *>(lds + O_OH)[tid] = v;
        }
        const int np1 = pol.n_perms + 1;
        for (int i = tid; i < np1 * 16; i += THREADS) lds[O_SRC + i] = pol.srcmap16[i];
        for (int i = tid; i < np1 * 256; i += THREADS) lds[O_VMAP + i] = pol.vmap16[i];
        for (int i = tid; i < 2 * np1 * 256; i += THREADS) {
            const int h2 = i / (np1 * 256), r = i - h2 * (np1 * 256);
            const uint32_t pos = (uint32_t)pol.vmap16[r] ^ ((uint32_t)h2 << 3);
            lds[O_OHB + h2 * ((E16_MAXP + 1) * 256) + r] = (uint8_t)((pos < 8u ? pos : 8u) * 16u);
        }
        if (tid < 4) lds[O_ACT + tid] = (uint8_t)tid;
        for (int i = tid; i < pol.n_perms * 4; i += THREADS) lds[O_ACT + 4 + i] = pol.act_perms[i];
        for (int i = tid; i < n_kt * 32; i += THREADS) reinterpret_cast<float *>(lds + O_EBIAS)[i] = pol.ebias16[i] * 16.0f;
#pragma unroll
        for (int op = 0; op < NOPS; ++op) { stream_op(stage_ptr(0), slot_m0(0), op); stream_op(stage_ptr(1), slot_m0(1), op); }
        rp = 0;
    }
    __device__ __forceinline__ void begin2() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __device__ __forceinline__ void end() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

    // one-hot fragment addresses packed four to a register (this mode is matrix-bound: the byte extract next to each
    // read is free, 2*NC address registers are not)
    __device__ __forceinline__ void onehots(uint64_t board, int perm, uint32_t (&w)[(NC + 3) / 4]) const
    {
        static_assert(O_OH == 0, "the byte table holds LDS addresses");
        const int pi = perm + 1;
        typedef uint32_t u32v4 __attribute__((ext_vector_type(4)));
        const u32v4 sr = *(const __attribute__((address_space(3))) u32v4 *)(L + O_SRC + pi * 16);
        const uint32_t srw[4] = {sr[0], sr[1], sr[2], sr[3]};
        const uint32_t tb = O_OHB + (uint32_t)(hh * (E16_MAXP + 1) + pi) * 256u;
#pragma unroll
        for (int q = 0; q < (NC + 3) / 4; ++q) w[q] = 0u;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const uint32_t src = (srw[c >> 2] >> (8 * (c & 3))) & 0xffu;
            w[c >> 2] |= (uint32_t)L[tb + c * 16 + nib(board, (int)src)] << (8 * (c & 3));
        }
    }
    __device__ __forceinline__ h16x8 ldoh(const uint32_t (&w)[(NC + 3) / 4], int c) const
    {
        return ld8(O_OH + ((w[c >> 2] >> (8 * (c & 3))) & 0xffu));
    }
    __device__ __forceinline__ void act_perm(int perm, float (&lg)[4]) const
    {
        typedef __attribute__((address_space(3))) const uint32_t lu1;
        const uint32_t ap = *(const lu1 *)(L + O_ACT + (perm + 1) * 4);
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
    __device__ __forceinline__ h16x8 ld8(uint32_t off) const { return *(const __attribute__((address_space(3))) h16x8 *)(L + off); }
    static __device__ __forceinline__ void mfma_v(f32x16 &d, const h16x8 a, const h16x8 b, bool first, const f32x16 &c0)
    {   // accumulator in architectural VGPRs (see tw_engine16.hpp)
        if constexpr (TW_MFMA_INTRIN & 16) d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, first ? c0 : d, 0, 0, 0);
        else if (first) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c0));
        else       asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
    }
    // accumulator registers (2q, 2q+1) of fragment m -> one register of the hi fragment and one of the lo fragment:
    // u = max(x, lim); hi = f16(u); lo = f16(u - f32(hi))
    static __device__ __forceinline__ void split_unit(const f32x16 &e, int m, int q, float lim, h16x8 &hi, h16x8 &lo)
    {
        const f32x2 u = {__builtin_fmaxf(e[8 * m + 2 * q], lim), __builtin_fmaxf(e[8 * m + 2 * q + 1], lim)};
        const h16x2 a = __builtin_convertvector(u, h16x2);
        const f32x2 back = __builtin_convertvector(a, f32x2);
        const h16x2 b = __builtin_convertvector(u - back, h16x2);
        hi[2 * q] = a[0]; hi[2 * q + 1] = a[1];
        lo[2 * q] = b[0]; lo[2 * q + 1] = b[1];
    }

    // pair order: two embedding pairs per common-layer pair until the embedding pairs are used up, keeping the last
    // common-layer pairs for the shadow of the conversion; code(i) = 2*index + (1 if embedding pair)
    template <int NE, int NM, int KEEP_ = 8> struct Sched {
        static constexpr int NP = NE + NM;
        static constexpr int KEEP = NM < KEEP_ ? NM : KEEP_;
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
        static constexpr int last_e() { int r = -1; for (int i = 0; i < NP; ++i) if (code(i) & 1) r = i; return r; }
    };
    // KIND 0 (hi stage): embedding pairs with T_hi (start the chain: C = embedding bias), common pairs (chunk q, h_hi) and
    //                    (chunk q, h_lo) with W_hi;      M index i -> chunk i>>1, operand i&1
    // KIND 1 (lo stage): embedding pairs with T_lo (continue the chain), common pairs (chunk q, h_hi) with W_lo; then the
    //                    finished embedding tile is split into the next fragments
    // KIND 2 (prologue): embedding pairs with T_hi(0) and T_lo(0) from the resident copy, then the split (exposed)
    template <int KIND, bool HAVE_E> struct Shape {
        static constexpr int NE = KIND == 2 ? 2 * NC : (HAVE_E ? NC : 0);
        static constexpr int NM = KIND == 0 ? 2 * NM0 : (KIND == 1 ? NM0 : 0);
        using S = Sched<NE, NM, KIND == 1 ? 12 : 8>;      // lo stage: more trailing pairs for the split of the finished tile
    };

    // compile-time position loop: every index below is a constant expression for the FRONT END (the pinned schedule must
    // not depend on the optimizer folding constexpr schedule functions inside a runtime loop)
    template <class Fn, int... Is>
    static __device__ __forceinline__ void sfor_impl(Fn &&f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
    template <int N, class Fn>
    static __device__ __forceinline__ void sfor(Fn &&f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

    // NEXT: 0 no prefetch for the following stage, 1 it has embedding pairs, 2 it has none.  The following stage of a hi
    // stage is the lo stage of the same tile; of a lo stage (and of the prologue) the next hi stage.
    template <int KIND, bool HAVE_E, bool FIRST, int NEXT, bool STREAM>
    __device__ __forceinline__ void phase(int ke_next, uint32_t baseE, uint32_t baseM, uint32_t nbase, int sg, int s2, const OneHots &oh_in,
                                          f32x16 (&acc0)[NHT], f32x16 (&acc1)[NHT], f32x16 &e0, f32x16 &e1, const Frags &Bc, Frags &Bn, Pipe &pp)
    {
        TW_STAMP(t_in);
        using SH = Shape<KIND, HAVE_E>;
        using S  = typename SH::S;
        using SN = typename Shape<KIND == 0 ? 1 : 0, NEXT == 1>::S;
        constexpr int NP = S::NP, LE = S::last_e();
        constexpr int MAFTER = NP - 1 - LE;
        constexpr bool SPLIT = (KIND == 1 && HAVE_E) || KIND == 2;                // this phase finishes an embedding tile
        constexpr int UPP = (KIND == 2 || MAFTER <= 1) ? 16 : (16 + MAFTER - 2) / (MAFTER - 1);
        constexpr int SPREAD = NP * 5 / 8 > 0 ? NP * 5 / 8 : 1;                  // all DMA ops within the first 5/8 of the phase
        const uint8_t *g_stage = stage_ptr(sg);
        const uint32_t m0_slot = slot_m0(s2);
        OneHots oh = oh_in;
#pragma unroll
        for (int q = 0; q < (NC + 3) / 4; ++q) asm volatile("" : "+v"(oh.a0[q]), "+v"(oh.a1[q]));
        h16x8 A[NP + D], X0[NP + D], X1[NP + D];
#pragma unroll
        for (int d = 0; d < D; ++d) { A[d] = pp.a[d]; X0[d] = pp.x0[d]; X1[d] = pp.x1[d]; }
        const f32x16 eb = pp.eb;
        f32x16 cbv[2];
        auto unit = [&](auto uc) {    // u in [0,16): tile u>>3, fragment (u>>2)&1, register u&3
            constexpr int u = decltype(uc)::value, m = (u >> 2) & 1, q = u & 3;
            if constexpr ((u >> 3) == 0) split_unit(e0, m, q, emb_lim, Bn.h0[m], Bn.l0[m]);
            else                         split_unit(e1, m, q, emb_lim, Bn.h1[m], Bn.l1[m]);
        };
        auto load_cb = [&](auto cc) {  // b1 (x256) of the hidden tile whose first pair has schedule code cc
            constexpr int cd = decltype(cc)::value;
            if constexpr (!(cd & 1) && ((cd >> 1) & 3) == 0) cbv[((cd >> 1) >> 2) & 1] = ld16(O_B1 + (uint32_t)(((cd >> 1) >> 2) * 2 + hh) * 64u);
        };
        if constexpr (FIRST) {
            if constexpr (NP > 0) load_cb(std::integral_constant<int, S::code(0)>{});
            if constexpr (NP > 1) load_cb(std::integral_constant<int, S::code(1)>{});
        }
        sfor<NP>([&](auto pc) {
            constexpr int p = decltype(pc)::value, pn = p + D;
            // ---- operand reads of position p + D
            if constexpr (pn < NP) {
                constexpr int cd = S::code(pn), ix = cd >> 1;
                if constexpr (cd & 1) { X0[pn] = ldoh(oh.a0, ix % NC); X1[pn] = ldoh(oh.a1, ix % NC); A[pn] = ld8(baseE + ix * 1024); }
                else A[pn] = ld8(baseM + (KIND == 0 ? ix >> 1 : ix) * 1024);
            } else if constexpr (NEXT != 0) {
                constexpr int cd = SN::code(pn - NP), ix = cd >> 1;
                if constexpr (cd & 1) { X0[pn] = ldoh(oh.a0, ix); X1[pn] = ldoh(oh.a1, ix); A[pn] = ld8(nbase + ix * 1024); }
                else A[pn] = ld8(nbase + (NC + (KIND == 0 ? ix : ix >> 1)) * 1024);     // the stage after a hi stage is a lo stage (M index = chunk)
            }
            if constexpr (NEXT != 0 && KIND != 0 && p == NP - 1) pp.eb = ld16(O_EBIAS + (uint32_t)(ke_next * 2 + hh) * 64u);
            if constexpr (FIRST && p + 2 < NP) load_cb(std::integral_constant<int, S::code(p + 2 < NP ? p + 2 : 0)>{});
            __builtin_amdgcn_sched_barrier(0);
            constexpr int cd = S::code(p), ix = cd >> 1;
            constexpr bool is_e = cd & 1;
            constexpr int chunk = KIND == 0 ? ix >> 1 : ix, ht = (chunk >> 1) < NHT ? (chunk >> 1) : 0, m = chunk & 1;
            constexpr bool use_lo = KIND == 0 && (ix & 1);
            constexpr bool c_b1 = FIRST && KIND == 0 && (ix & 3) == 0;
            if constexpr (is_e) mfma_v(e0, A[p], X0[p], KIND != 1 && ix == 0, eb);
            else acc0[ht] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[p], use_lo ? Bc.l0[m] : Bc.h0[m], c_b1 ? cbv[ht & 1] : acc0[ht], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (STREAM) {
                constexpr int o_lo = dma_upto(p - 1, SPREAD), o_hi = dma_upto(p, SPREAD);
                sfor<o_hi - o_lo>([&](auto oc) { stream_op(g_stage, m0_slot, o_lo + decltype(oc)::value); });
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (is_e) mfma_v(e1, A[p], X1[p], KIND != 1 && ix == 0, eb);
            else acc1[ht] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[p], use_lo ? Bc.l1[m] : Bc.h1[m], c_b1 ? cbv[ht & 1] : acc1[ht], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (SPLIT && KIND == 1 && p > LE + 1) {
                constexpr int u_lo = units_upto(p - 1, LE, UPP), u_hi = units_upto(p, LE, UPP);
                sfor<u_hi - u_lo>([&](auto uc) { unit(std::integral_constant<int, u_lo + decltype(uc)::value>{}); });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (SPLIT) {
            constexpr int done = (KIND == 1 && NP > 0) ? units_upto(NP - 1, LE, UPP) : 0;
            if constexpr (KIND == 2 || MAFTER < 1) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(e0), "+v"(e1));   // XDL write -> VALU read (asm MFMA)
            sfor<16 - done>([&](auto uc) { unit(std::integral_constant<int, done + decltype(uc)::value>{}); });
        }
        if constexpr (NEXT != 0) {
#pragma unroll
            for (int d = 0; d < D; ++d) { pp.a[d] = A[NP + d]; pp.x0[d] = X0[NP + d]; pp.x1[d] = X1[NP + d]; }
        }
        if constexpr (STREAM) {
            constexpr int issued = NP > 0 ? dma_upto(NP - 1, SPREAD) : 0;
            sfor<NOPS - issued>([&](auto oc) { stream_op(g_stage, m0_slot, issued + decltype(oc)::value); });
            TW_STAMP(t_b);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            TW_STAMP(t_w);
            __syncthreads();
            TW_STAMP(t_s);
            TW_ACC(2, t_in, t_b); TW_ACC(3, t_b, t_w); TW_ACC(4, t_w, t_s);
        }
    }
    // DMA ops issued up to and including position p (op o goes to the first position p with o*SPREAD/NOPS <= p)
    static constexpr int dma_upto(int p, int spread)
    {
        int n = 0;
        for (int o = 0; o < NOPS; ++o) if (o * spread / NOPS <= p) ++n;
        return p < 0 ? 0 : n;
    }
    // conversion units done up to and including position p (UPP per position from LE + 2 on)
    static constexpr int units_upto(int p, int le, int upp)
    {
        if (p <= le + 1) return 0;
        const int v = (p - le - 1) * upp;
        return v < 16 ? v : 16;
    }

    // out0/out1: registers 0..3 logits, 4 value, scaled by 4096 (no head bias)
    __device__ __forceinline__ void forward(const OneHots &oh, f32x16 &out0, f32x16 &out1)
    {
        f32x16 acc0[NHT], acc1[NHT], e0, e1;
        Frags Fa, Fb;
        Pipe pp;
        const uint32_t lo = (uint32_t)lane * 16u;
        auto slot_base = [&](int s) { return O_RING + (uint32_t)s * SBYTES + lo; };
        auto tile_of  = [&](int kt) { return kt < n_kt ? kt : 0; };
        int s0 = rp, q = 0;                                       // ring slot / stage number of the stage being executed
        // prologue: embedding tile 0 (hi then lo chunks of the resident copy)
        pp.eb = ld16(O_EBIAS + (uint32_t)hh * 64u);
#pragma unroll
        for (int d = 0; d < D; ++d) { pp.x0[d] = ldoh(oh.a0, d); pp.x1[d] = ldoh(oh.a1, d); pp.a[d] = ld8(O_T0 + lo + d * 1024); }
        phase<2, true, false, 1, false>(tile_of(1), O_T0 + lo, 0, slot_base(s0), 0, 0, oh, acc0, acc1, e0, e1, Fa, Fa, pp);
        // one embedding tile = a hi stage and a lo stage; `have_e`: the embedding part computes tile kt+1.  ONE steady
        // instance (plus the first and the last tile): the whole per-timestep code has to stay inside the 64 KiB
        // instruction cache -- with parity / look-ahead variants of the tile it did not, and ran 40x slower.  The next
        // fragments are therefore copied (32 registers per tile) instead of ping-ponged, and every lo stage prefetches
        // as if the next tile had an embedding part (the last tile fetches its own first operands).
        auto tile = [&](auto first, auto have_e, int kt, const Frags &c, Frags &n) {
            constexpr bool HE = decltype(have_e)::value;
            {
                const int s1 = s0 == 2 ? 0 : s0 + 1, s2 = s0 == 0 ? 2 : s0 - 1;
                phase<0, HE, decltype(first)::value, HE ? 1 : 2, true>(0, slot_base(s0), slot_base(s0) + NC * 1024, slot_base(s1), wrap(q + 2), s2,
                                                                         oh, acc0, acc1, e0, e1, c, n, pp);
                s0 = s1; ++q;
            }
            {
                const int s1 = s0 == 2 ? 0 : s0 + 1, s2 = s0 == 0 ? 2 : s0 - 1;
                phase<1, HE, false, HE ? 1 : 0, true>(tile_of(kt + 2), slot_base(s0), slot_base(s0) + NC * 1024, slot_base(s1), wrap(q + 2), s2,
                                                       oh, acc0, acc1, e0, e1, c, n, pp);
                s0 = s1; ++q;
            }
        };
        using T = std::true_type; using F = std::false_type;
        tile(T{}, T{}, 0, Fa, Fb);
        Fa = Fb;
        for (int kt = 1; kt + 1 < n_kt; ++kt) {
            tile(F{}, T{}, kt, Fa, Fb);
            Fa = Fb;
        }
        {   // last tile: common layer only (n_kt >= 2 is checked on the host)
            const uint32_t bM = slot_base(s0) + NC * 1024;
#pragma unroll
            for (int d = 0; d < D; ++d) pp.a[d] = ld8(bM + (d >> 1) * 1024);
            tile(F{}, F{}, n_kt - 1, Fa, Fb);
        }
        // heads stage: [head_hi chunks | head_lo chunks]; per hidden tile: split relu(acc) and three products
        {
            TW_STAMP(t_h0);
            const int s1 = s0 == 2 ? 0 : s0 + 1, s2 = s0 == 0 ? 2 : s0 - 1;
            const uint32_t bH = slot_base(s0);
            const uint8_t *g_stage = stage_ptr(wrap(q + 2));
            const uint32_t m0_slot = slot_m0(s2);
#pragma unroll
            for (int op = 0; op < NOPS; ++op) stream_op(g_stage, m0_slot, op);
            f32x16 h0, h1;
#pragma unroll
            for (int g = 0; g < 16; ++g) { h0[g] = 0.0f; h1[g] = 0.0f; }
#pragma unroll
            for (int ht = 0; ht < NHT; ++ht) {
                Frags f;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    split_unit(acc0[ht], u >> 2, u & 3, common_lim, f.h0[u >> 2], f.l0[u >> 2]);
                    split_unit(acc1[ht], u >> 2, u & 3, common_lim, f.h1[u >> 2], f.l1[u >> 2]);
                }
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const h16x8 ahi = ld8(bH + (uint32_t)(ht * 2 + m) * 1024u), alo = ld8(bH + (uint32_t)(NM0 + ht * 2 + m) * 1024u);
                    h0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, f.h0[m], h0, 0, 0, 0);
                    h1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, f.h1[m], h1, 0, 0, 0);
                    h0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, f.l0[m], h0, 0, 0, 0);
                    h1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, f.l1[m], h1, 0, 0, 0);
                    h0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, f.h0[m], h0, 0, 0, 0);
                    h1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, f.h1[m], h1, 0, 0, 0);
                }
            }
            out0 = h0; out1 = h1;
#ifdef TW_ABLATE
            asm volatile("" :: "v"(h0), "v"(h1));
#endif
            TW_STAMP(t_hb);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            TW_STAMP(t_hs);
            TW_ACC(5, t_h0, t_hb); TW_ACC(4, t_hb, t_hs);
            s0 = s1;
        }
        rp = s0;
    }
};

}  // namespace tw
