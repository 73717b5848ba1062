// tw_engine.hpp -- the per-workgroup policy-forward engine shared by the rollout (PPO) and MCTS
// (AlphaZero) kernels: EmbeddingBag gather + common Linear on f32 MFMA + both heads, with the two
// weight streams double-buffered in LDS.  See tw_rollout.hip for the design notes.
#pragma once
#include "tw_common.hpp"

namespace tw {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const float lds_cfloat;

constexpr int EPW = 32;            // episodes per wave (MFMA columns)

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

// Two f32 adds in one VALU instruction (IEEE-exact per element).  Inline asm because hipcc's
// pre-emit peephole un-packs v_pk_add_f32 that sits in the shadow of an MFMA.
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b)
{
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int NT> struct Tiles { static constexpr int NQ = (NT + 3) / 4; };

// LDS carve (floats): W[2][KC*NQ*128] | T[2][n_rows*LSTR] | b1[2][NT*16] | wh[9][2][NT*16] | bh8[8] |
//                     obs_perms u8[n_perms][obs_size] | act_perms u8[n_perms][4]
constexpr int MAX_LDS_PERMS = 4;   // twist tables kept in LDS (more twists fall back to global reads)
template <int NT, int KC>
__host__ __device__ inline size_t engine_lds_floats(int obs_size)
{
    return (size_t)2 * KC * Tiles<NT>::NQ * 128 + (size_t)2 * (obs_size + 2) * (KC + 4) + (size_t)NT * 32 * 10 + 8 +
           (size_t)MAX_LDS_PERMS * ((obs_size + 3) / 4 + 1);
}

// MFMA row i of row-tile r carries hidden unit hid(r,i) = 32r + 2g + h with
// g = (i&3) + 4*(i>>3), h = (i>>2)&1: the C/D layout (row = (g&3) + 8*(g>>2) + 4h for
// accumulator register g on lane half h) then holds hidden unit 32r + 2g + h in register g.
// (tw_api.hip builds the W1 image [k][q][i][4] = W1[k][hid(4q+c, i)] with the same formula.)
//
// DBG != 0 builds are timing-only ablations (wrong results): 1 no gather, 2 no A-operand reads,
// 4 no weight streams, 8 no heads.  Never used by the product path.
template <int NT, int NC, int NW, int KC, int DBG = 0>
struct Engine {
    static constexpr int THREADS = NW * 64;
    static constexpr int EPB     = NW * EPW;
    static constexpr int LSTR    = KC + 4;      // padded row stride (floats), 16-B aligned rows
    static constexpr int NG      = KC / 8;      // groups of four k-steps per chunk
    static constexpr int NQ      = Tiles<NT>::NQ;
    static constexpr int WCHUNK  = KC * NQ * 128;       // floats per W1 chunk
    static constexpr int WPIECES = WCHUNK / 256;        // 1-KiB LDS-DMA pieces per chunk
    static constexpr int TITER   = (NC * NC * (KC / 4) + THREADS - 1) / THREADS;   // float4 loads per thread per table chunk

    PolicyDev pol;
    int tid, lane, wave, j, h;
    int n_rows, bias_row, zero_row, n_chunks, tbuf, cur;
    float *lds_w, *lds_t, *lds_b1, *lds_wh, *lds_bh;
    const uint8_t *perm_obs, *perm_act;   // twist tables: LDS copies when n_perms <= MAX_LDS_PERMS
    __amdgpu_buffer_rsrc_t rs_emb;
    f32x4 tst[TITER];   // table chunk in flight (registers)
    f32x4 tsb;          // bias-row piece (threads 0..KC/4-1)
    // DBG & 16: s_memtime sums [0] stream issue, [1] chunk prologue, [2] MFMA groups, [3] commit+wait, [4] barrier, [5] heads
    unsigned long long stamp[6];
    __device__ __forceinline__ unsigned long long now() const
    {
        unsigned long long t;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
        __builtin_amdgcn_sched_barrier(0);
        return t;
    }

    __device__ __forceinline__ void stream_issue(int chunk, int buf)
    {
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
        // table chunk: rows 0..obs_size-1 (+ bias row) to registers.  The loads are UNCONDITIONAL on
        // purpose: a predicated load makes hipcc branch around it and wait vmcnt(0) per load (one L2
        // round trip each); rows past the table are dropped by the descriptor's bounds check (read 0)
        // and never committed.
#pragma unroll
        for (int it = 0; it < TITER; ++it) {
            const int idx = tid + it * THREADS, row = idx / (KC / 4), q = idx % (KC / 4);
            tst[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rs_emb, (row * pol.emb + q * 4) * (int)sizeof(float), chunk * KC * (int)sizeof(float), 0));
        }
        tsb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            rs_emb, (bias_row * pol.emb + (tid % (KC / 4)) * 4) * (int)sizeof(float), chunk * KC * (int)sizeof(float), 0));
    }

    // One stream operation of the next chunk (op < NOPS): DMA pieces first, then table loads, then
    // the bias piece.  Issued one at a time between MFMAs so the VMEM queue never sees a burst.
    static constexpr int NDMA = (WPIECES + NW - 1) / NW;
    static constexpr int NOPS = NDMA + TITER + 1;
    __device__ __forceinline__ void stream_op(int chunk, int buf, int op)
    {
        if constexpr (DBG & 4) return;
        if (op < NDMA) {
            const int piece = wave + NW * op;
            if (piece < WPIECES) {
                const float *src = pol.w1p + (size_t)chunk * WCHUNK + piece * 256 + lane * 4;
                glds16(src, lds_w + buf * WCHUNK + piece * 256);
            }
        } else if (op < NDMA + TITER) {
            const int it = op - NDMA;
            const int idx = tid + it * THREADS, row = idx / (KC / 4), q = idx % (KC / 4);
            tst[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rs_emb, (row * pol.emb + q * 4) * (int)sizeof(float), chunk * KC * (int)sizeof(float), 0));
        } else {
            tsb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rs_emb, (bias_row * pol.emb + (tid % (KC / 4)) * 4) * (int)sizeof(float), chunk * KC * (int)sizeof(float), 0));
        }
    }

    __device__ __forceinline__ void stream_commit(int buf)
    {
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
    }

    // Part 1 of the set-up: LDS constants and the first chunk's loads (the caller overlaps its own
    // start-up work, e.g. the scramble, before calling begin2()).
    __device__ __forceinline__ void begin1(const PolicyDev &p, float *lds)
    {
        pol = p;
        tid  = threadIdx.x;
        lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        j = lane & 31; h = lane >> 5;
        n_rows = pol.obs_size + 2; bias_row = pol.obs_size; zero_row = pol.obs_size + 1;
        n_chunks = pol.emb / KC;
        lds_w  = lds;                                  // [2][WCHUNK]
        lds_t  = lds + 2 * WCHUNK;                     // [2][n_rows*LSTR]
        lds_b1 = lds_t + 2 * n_rows * LSTR;
        lds_wh = lds_b1 + NT * 32;
        lds_bh = lds_wh + NT * 32 * 9;
        tbuf = n_rows * LSTR;
        // head operands re-laid so that a lane's next four k-steps are one 16-byte word:
        //   b1[h][m] = b1[2m + h];  wh[c][h][m] = wh8[2m + h][c] for c < 8;  wh[8][.][.] = 0 (lanes j >= 8)
        for (int i = tid; i < NT * 32; i += THREADS) lds_b1[(i & 1) * (NT * 16) + (i >> 1)] = pol.b1[i];
        for (int i = tid; i < NT * 32 * 8; i += THREADS) {
            const int n = i >> 3, c = i & 7;
            lds_wh[(c * 2 + (n & 1)) * (NT * 16) + (n >> 1)] = pol.wh8[i];
        }
        for (int i = tid; i < NT * 32; i += THREADS) lds_wh[8 * 2 * (NT * 16) + i] = 0.0f;
        if (tid < 8) lds_bh[tid] = pol.bh8[tid];
        perm_obs = pol.obs_perms; perm_act = pol.act_perms;
        if (pol.n_perms > 0 && pol.n_perms <= MAX_LDS_PERMS) {
            uint8_t *po = reinterpret_cast<uint8_t *>(lds_bh + 8);
            uint8_t *pa = po + MAX_LDS_PERMS * ((pol.obs_size + 3) / 4) * 4;
            for (int i = tid; i < pol.n_perms * pol.obs_size; i += THREADS) po[i] = pol.obs_perms[i];
            for (int i = tid; i < pol.n_perms * 4; i += THREADS) pa[i] = pol.act_perms[i];
            perm_obs = po; perm_act = pa;
        }
        if (tid < 2 * LSTR) lds_t[(tid / LSTR) * tbuf + zero_row * LSTR + (tid % LSTR)] = 0.0f;   // zero rows, never restaged
        rs_emb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(pol.emb_rows), 0,
                                                   n_rows * pol.emb * (int)sizeof(float), 0x00020000);
        if constexpr (DBG & 16) for (int i = 0; i < 6; ++i) stamp[i] = 0;
        stream_issue(0, 0);
    }
    // Part 2: commit chunk 0.  The caller's next workgroup barrier publishes it.
    __device__ __forceinline__ void begin2()
    {
        stream_commit(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        cur = 0;
    }
    // drain the stream that ran ahead of the last forward before the LDS is released
    __device__ __forceinline__ void end() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

    // LDS float offsets of the obs rows of one board (observe, puzzle.rs:183-185) under twist `perm`
    // (policy.rs:81-83); cells beyond n_cells point at the zero row.
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

    // Policy::_raw_predict before the act-perm (policy.rs:86-92) for the 32 episodes of this wave:
    // raw action logits (+bias) and value (+bias), valid in BOTH lanes (j, j+32) of an episode.
    // Must be called by every thread of the workgroup (it contains the chunk barriers).
    __device__ __forceinline__ void forward(const int (&rowoff)[NC], float (&lg)[4], float &value)
    {
        f32x16 acc[NT];
#pragma unroll
        for (int r = 0; r < NT; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[r][g] = 0.0f;

        // ---- EmbeddingBag (layers.rs:56-62,82-84) fused into common Linear (layers.rs:31-37) --
        for (int c = 0; c < n_chunks; ++c) {
            // run the streams one chunk ahead (wrapping to chunk 0 of the next forward)
            unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
            if constexpr (DBG & 16) t0 = now();
            const int next_chunk = c + 1 == n_chunks ? 0 : c + 1;
            if constexpr (NG < 2) stream_issue(next_chunk, cur ^ 1);      // no room to spread
            if constexpr (DBG & 16) t1 = now();

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
            // (the f32 MFMA executes on the SIMD's f32 FMA lanes, so VALU work does not overlap it:
            //  the four add chains go through v_pk_add_f32, two chains per instruction, IEEE-exact)
            f32x4 bq;
            {
                const f32x4 r0 = *gather_ptr(0, 0);
                f32x2 lo = __builtin_shufflevector(r0, r0, 0, 1), hi = __builtin_shufflevector(r0, r0, 2, 3);
                if constexpr (!(DBG & 1))
#pragma unroll
                    for (int q = 1; q <= NC; ++q) {
                        const f32x4 rq = *gather_ptr(q, 0);
                        lo = pk_add(lo, __builtin_shufflevector(rq, rq, 0, 1));
                        hi = pk_add(hi, __builtin_shufflevector(rq, rq, 2, 3));
                    }
                bq[0] = lo[0]; bq[1] = lo[1]; bq[2] = hi[0]; bq[3] = hi[1];
            }
            if (pol.emb_relu) {
#pragma unroll
                for (int u = 0; u < 4; ++u) bq[u] = relu1(bq[u]);
            } else {
                asm volatile("s_nop 1" : "+v"(bq));      // asm VALU result -> MFMA operand wait states
            }
            f32x4 aw[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) aw[q] = *a_ptr(0, q);
            if constexpr (DBG & 16) t2 = now();

#pragma unroll
            for (int g = 0; g < NG; ++g) {
                f32x4 rd[NC + 1];
                f32x2 nlo, nhi;
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const int u = m / NT, r = m % NT;               // k-step 4g+u, row-tile r
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[r / 4][r % 4], bq[u], acc[r], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    // A operands of the next k-step, as soon as their registers are free
                    if constexpr (!(DBG & 2))
                        if ((r % 4 == 3 || r == NT - 1) && !(g == NG - 1 && u == 3))
                            aw[r / 4] = *a_ptr(4 * g + u + 1, r / 4);
                    // stream of the next chunk: one VMEM op every SPACING slots over the first NG-1 groups
                    if constexpr (NG >= 2) {
                        constexpr int SPAN = (NG - 1) * M, SPACING = SPAN / NOPS > 0 ? SPAN / NOPS : 1;
                        const int slot = g * M + m;
                        if (slot % SPACING == 0 && slot / SPACING < NOPS) stream_op(next_chunk, cur ^ 1, slot / SPACING);
                        if (SPAN < NOPS && slot == SPAN - 1)                   // more ops than slots: flush the rest
                            for (int op = SPAN; op < NOPS; ++op) stream_op(next_chunk, cur ^ 1, op);
                    }
                    // gather of the next group: reads, then (LAT slots later) four independent adds
                    if (g + 1 < NG) {
#pragma unroll
                        for (int q = 0; q <= ((DBG & 1) ? 0 : NC); ++q)
                            if (rd_slot(q) == m) rd[q] = *gather_ptr(q, g + 1);
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
                if (g + 1 < NG) {
                    bq[0] = nlo[0]; bq[1] = nlo[1]; bq[2] = nhi[0]; bq[3] = nhi[1];
                    if (pol.emb_relu) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) bq[u] = relu1(bq[u]);
                    } else {
                        asm volatile("s_nop 1" : "+v"(bq));
                    }
                }
            }
            if constexpr (DBG & 16) t3 = now();
            stream_commit(cur ^ 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces have landed
            if constexpr (DBG & 16) t4 = now();
            __syncthreads();     // chunk c fully consumed by every wave; chunk c+1 (DMA + ds_write) landed
            cur ^= 1;
            if constexpr (DBG & 16) {
                const unsigned long long t5 = now();
                stamp[0] += t1 - t0; stamp[1] += t2 - t1; stamp[2] += t3 - t2; stamp[3] += t4 - t3; stamp[4] += t5 - t4;
            }
        }
        unsigned long long th0 = 0;
        if constexpr (DBG & 16) th0 = now();

        if constexpr (DBG & 8) {
            value = 0.0f;
#pragma unroll
            for (int r = 0; r < NT; ++r) value += acc[r][0];   // keeps the GEMM alive
            lg[0] = lg[1] = lg[2] = lg[3] = 0.0f;
            return;
        }
        // ---- bias + ReLU of the common layer, then both heads (policy.rs:86-92) ---------------
        f32x16 hacc;
#pragma unroll
        for (int g = 0; g < 16; ++g) hacc[g] = 0.0f;
        // per lane: b1 of its half, and its head-weight row (lanes j >= 8 read the zero row)
        lds_cfloat *b1_lane = (lds_cfloat *)(lds_b1 + h * (NT * 16));
        lds_cfloat *wh_lane = (lds_cfloat *)(lds_wh + ((j < 8 ? j : 8) * 2 + h) * (NT * 16));
        asm volatile("" : "+v"(b1_lane), "+v"(wh_lane));   // opaque: every access = base + immediate
        // register g of row-tile r holds hidden unit 32r + 2g + h = element 16r + g of this lane's half
        f32x4 hb[2], hw[2];
        hb[0] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>(b1_lane);
        hw[0] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>(wh_lane);
#pragma unroll
        for (int blk = 0; blk < NT * 4; ++blk) {           // 4 accumulator registers per block
            const int r = blk >> 2, g0 = (blk & 3) * 4, cb = blk & 1, nb = cb ^ 1;
            if (blk + 1 < NT * 4) {
                hb[nb] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>(b1_lane + 4 * (blk + 1));
                hw[nb] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>(wh_lane + 4 * (blk + 1));
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float hv = acc[r][g0 + g] + hb[cb][g];
                if (pol.common_relu) hv = relu1(hv);
                hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(hw[cb][g], hv, hacc, 0, 0, 0);
            }
        }
        // rows 0..3 (logits) sit in registers 0..3 of lane (j,0); row 4 (value) in register 0 of lane (j,1)
#pragma unroll
        for (int i = 0; i < 4; ++i) lg[i] = __shfl(hacc[i], j, 64) + lds_bh[i];
        value = __shfl(hacc[0], j + 32, 64) + lds_bh[4];
        if constexpr (DBG & 16) stamp[5] += now() - th0;
    }

    // logits'[i] = logits[act_perm[i]]  (policy.rs:95-97)
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
};

}  // namespace tw
