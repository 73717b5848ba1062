// tw_engine_diag.hpp -- engines that exist for MEASUREMENTS only.  Included by tw_engine.hpp under TW_ABLATE (the diagnostic library,
// twisterl_amd/lib/ablate/): the product library neither parses nor launches anything in here.
// (textually included INSIDE namespace tw of tw_engine.hpp, behind Engine3S whose helpers it uses: no namespace, no guard of its own)

// =====================================================================================================
// Engine3G: Engine3S's geometry (four waves share 32 episodes and split the hidden units, same lane mapping, same arithmetic)
// with HALF its LDS, so that TWO such workgroups share a CU and each one's bubbles -- the barrier and the B-operand round trip of
// every chunk, the heads, the kernel's own per-step work -- are the other one's matrix time (round 3; Engine3S alone keeps the
// matrix cores 54 % busy on a 16,384-env rollout).  What had to go for that:
//   * W1 never enters LDS: a lane's A operands of a chunk are eight 8-byte reads of the image the other shapes stream (an LDS
//     slot is a verbatim copy of it), requested one chunk ahead into a two-deep register ring -- as in Engine3T;
//   * the table ring has two slots, not three: the gather of chunk c+1 runs during chunk c (its sums are chunk c+1's B
//     operands, published through the exchange buffer), so a chunk's table is dead once its step is over, and chunk c+2
//     streams into the slot chunk c's table just left;
//   * nothing is streamed ahead across forwards (the heads park the hidden units in the ring: both slots are free then): a
//     forward's prologue waits for its first table chunk -- with a second workgroup on the CU that wait is not idle time.
// LDS: T[2][5376] | b1 | wh | bh8 | twists | wn | exchange  (~79 KB for the 512 / 256 policy).
// =====================================================================================================
template <int NT>
__host__ __device__ inline size_t engine3g_lds_floats(int obs_size)
{
    return (size_t)2 * R3_TSLOT + (size_t)NT * 32 * 10 + 8 + (size_t)MAX_LDS_PERMS * ((obs_size + 3) / 4 + 1) + (size_t)NT * 32 * 5 + R3S_XCHG + R3S_USER;
}

template <int NT, int NC>
struct Engine3G : Engine3<NT, NC, 0, 4> {
    using B = Engine3<NT, NC, 0, 4>;
    static constexpr int NS = 4, EPB = EPW, NTL = NT / NS, KC = B::KC, NQ = B::NQ, WSLOT = B::WSLOT;
    static constexpr int TOPS = (B::TPIECE + NS - 1) / NS;                  // DMA ops per wave and chunk (table pieces only)
    static constexpr bool SPLIT = true;
    static_assert(NT % NS == 0 && (NTL == 1 || NTL == 2), "Engine3G: one or two row tiles per wave");
    static_assert((size_t)NT * 32 * 32 <= (size_t)2 * R3_TSLOT, "the hidden units of 32 episodes must fit the two table slots");

    float *lds_x, *lds_user;
    const float *agl;                      // this lane's A operands: + chunk * WSLOT + k-step * 2*NQ*128
    uint32_t voffT[TOPS], mT[TOPS];
#ifdef TW_ABLATE
    unsigned long long stq[6] = {0, 0, 0, 0, 0, 0};   // prologue | chunk loop | - | - | heads | -
#endif

    __host__ __device__ static size_t lds_floats(int obs_size) { return engine3g_lds_floats<NT>(obs_size); }
    __host__ __device__ static size_t lds_floats(const PolicyDev &p) { return lds_floats(p.obs_size); }
    __device__ __forceinline__ bool primary() const { return this->wave == 0; }
    __device__ __forceinline__ int  ep_lane() const { return this->j; }
    __device__ __forceinline__ bool owns_lane() const { return this->wave == this->j / (EPW / NS); }

    __device__ __forceinline__ void begin1(const PolicyDev &p, float *lds)
    {
        // (Engine3::begin1 with this engine's LDS map and without its first streams)
        this->pol = p;
#ifdef TW_ABLATE
        this->eng_dbg = __builtin_amdgcn_readfirstlane(g_eng_dbg);     // timing-only knock-outs: 1 no gather reads, 2 no A-operand loads, 4 no table streams
#endif
        this->tid  = threadIdx.x;
        this->lane = this->tid & 63;
        this->wave = __builtin_amdgcn_readfirstlane(this->tid >> 6);
        this->j = this->lane & 31; this->h = this->lane >> 5;
        this->voff = (uint32_t)this->lane * 16u;
        this->bias_row = p.obs_size; this->zero_row = p.obs_size + 1;
        this->n_chunks = p.emb / KC;
        this->emb_lim = p.emb_relu ? 0.0f : -__builtin_inff();
        this->common_lim = p.common_relu ? 0.0f : -__builtin_inff();
        this->lds_w  = lds;                                   // (no W slots)
        this->lds_t  = lds;
        this->lds_b1 = this->lds_t + 2 * R3_TSLOT;
        this->lds_wh = this->lds_b1 + NT * 32;
        this->lds_bh = this->lds_wh + NT * 32 * 9;
        this->dsrc_t = reinterpret_cast<const uint8_t *>(p.t_img16);
        this->ddst_t = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)this->lds_t);
        constexpr int THREADS = 256;
        for (int i = this->tid; i < NT * 32; i += THREADS) this->lds_b1[(i & 1) * (NT * 16) + (i >> 1)] = p.b1[i];
        for (int i = this->tid; i < NT * 32 * 8; i += THREADS) {
            const int n = i >> 3, c = i & 7;
            this->lds_wh[(c * 2 + (n & 1)) * (NT * 16) + (n >> 1)] = p.wh8[i];
        }
        for (int i = this->tid; i < NT * 32; i += THREADS) this->lds_wh[8 * 2 * (NT * 16) + i] = 0.0f;
        if (this->tid < 8) this->lds_bh[this->tid] = p.bh8[this->tid];
        this->lds_wn = this->lds_bh + 8 + MAX_LDS_PERMS * ((p.obs_size + 3) / 4 + 1);
        for (int i = this->tid; i < NT * 32 * 5; i += THREADS) {
            const int c = i / (NT * 32), n = i - c * (NT * 32);
            this->lds_wn[i] = p.wh8[n * 8 + c];
        }
        this->perm_obs = p.obs_perms; this->perm_act = p.act_perms;
        if (p.n_perms > 0 && p.n_perms <= MAX_LDS_PERMS) {
            uint8_t *po = reinterpret_cast<uint8_t *>(this->lds_bh + 8);
            uint8_t *pa = po + MAX_LDS_PERMS * ((p.obs_size + 3) / 4) * 4;
            for (int i = this->tid; i < p.n_perms * p.obs_size; i += THREADS) po[i] = p.obs_perms[i];
            for (int i = this->tid; i < p.n_perms * 4; i += THREADS) pa[i] = p.act_perms[i];
            this->perm_obs = po; this->perm_act = pa;
        }
        lds_x = this->lds_wn + NT * 32 * 5;
        lds_user = lds_x + R3S_XCHG;
        const int t0 = this->wave * NTL;
        agl = p.w1p + ((this->h * NQ + (t0 >> 2)) * 32 + this->j) * 4 + (t0 & 3);
#pragma unroll
        for (int k = 0; k < TOPS; ++k) {
            int tp = this->wave + NS * k;
            tp = tp < B::TPIECE ? tp : B::TPIECE - 1;                                           // past the end: repeat the last piece
            mT[k] = this->ddst_t + (uint32_t)tp * 1024u;
            voffT[k] = this->voff + (uint32_t)tp * 1024u;
        }
    }
    __device__ __forceinline__ void begin2() { __syncthreads(); }          // (publishes the LDS constants)
    __device__ __forceinline__ void end() {}

    // table piece K of this wave, of the chunk at `src`, into ring slot S
    template <int S, int K>
    __device__ __forceinline__ void stream_op(const uint8_t *src) const
    {
#ifdef TW_ABLATE
        if (this->eng_dbg & 4) return;
#endif
        TW_GLDS16_ADD(voffT[K], mT[K], S * R3_TSLOT * 4, src);
    }
    template <int S, int K = 0>
    __device__ __forceinline__ void stream_table(const uint8_t *src) const
    {
        if constexpr (K < TOPS) { stream_op<S, K>(src); stream_table<S, K + 1>(src); }
    }
    template <int S, int K = 0>   // the ops of MFMA slot m: op m (m is a constant after unrolling)
    __device__ __forceinline__ void ops_of_slot(int m, const uint8_t *src) const
    {
        if constexpr (K < TOPS) { if (K == m) stream_op<S, K>(src); ops_of_slot<S, K + 1>(m, src); }
    }

    __device__ __forceinline__ void forward(const int (&rowoff)[NC], float (&lg)[4], float &value)
    {
        typedef __attribute__((address_space(3))) const f32x4 lds_cf4;
        typedef __attribute__((address_space(3))) const f32x2 lds_cf2;
        typedef const __attribute__((address_space(1))) f32x2 g_cf2;
        typedef const __attribute__((address_space(1))) float g_cf1;
        const int j = this->j, h = this->h, wave = this->wave;
        const int nch = this->n_chunks;
        f32x16 acc[NTL];
#pragma unroll
        for (int r = 0; r < NTL; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[r][g] = 0.0f;

        // one LDS address per row for the whole forward: the ring slot is an immediate offset of the read
        constexpr int GW = 8 / NS;                                              // k-steps of a chunk this wave gathers
        lds_cfloat *ga[NC + 1];
        ga[0] = (lds_cfloat *)this->lds_t + this->bias_row * R3_LSTR + h * (KC / 2) + GW * wave;
#pragma unroll
        for (int q = 0; q < NC; ++q) ga[q + 1] = (lds_cfloat *)this->lds_t + rowoff[q] + GW * wave;

        float *xb = lds_x + 256;
        f32x2 gr2[NC + 1];
        auto gather_finish = [&](int buf) {
            f32x2 sm = gr2[0];                                                     // bias row, then the cells in order
#pragma unroll
            for (int q = 1; q <= NC; ++q) sm = pk_add(sm, gr2[q]);
            sm[0] = relu_lim(sm[0], this->emb_lim); sm[1] = relu_lim(sm[1], this->emb_lim);
            *reinterpret_cast<f32x2 *>(xb + ((buf * 2 + (wave >> 1)) * 64 + this->lane) * 4 + 2 * (wave & 1)) = sm;
        };
        float areg[2][8][NTL];                                                   // A operands: this chunk | the next one
        auto load_a = [&](int chunk, float (&a)[8][NTL]) {
            const float *ap = agl + (size_t)chunk * WSLOT;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if constexpr (NTL == 2) { const f32x2 v = *(g_cf2 *)(ap + ks * 2 * NQ * 128); a[ks][0] = v[0]; a[ks][1] = v[1]; }
                else a[ks][0] = *(g_cf1 *)(ap + ks * 2 * NQ * 128);
            }
        };

        // prologue: table chunks 0 and 1 -> slots 0 and 1, the A operands of chunk 0; the gather of chunk 0 as soon as IT has landed
        TW_S3(q_in);
        stream_table<0>(this->dsrc_t);
        load_a(0, areg[0]);
        if (nch > 1) stream_table<1>(this->dsrc_t + R3_TSLOT * 4);
        if (nch > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(TOPS) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int q = 0; q <= NC; ++q) gr2[q] = *reinterpret_cast<lds_cf2 *>(ga[q]);
        gather_finish(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        TW_S3(q_pro);
        TW_A3(0, q_in, q_pro);
        int par = 0;
        const uint8_t *stp = this->dsrc_t + 2 * (R3_TSLOT * 4);                 // the chunk streamed next (c + 2)
        // One step: chunk c with the A operands in register slot P = c & 1; the table of chunk c+1 sits complete in ring slot
        // Q = P ^ 1 (gathered now), chunk c+2 streams into ring slot P (chunk c's table was gathered a step ago)
        auto step = [&](auto pc, int c) {
            constexpr int P = decltype(pc)::value, Q = P ^ 1;
            constexpr int M = 8 * NTL;
            static_assert(TOPS <= M, "one DMA op per MFMA slot");
            const bool more = c + 1 < nch, stream = c + 2 < nch;
#ifdef TW_ABLATE
            if (more && !(this->eng_dbg & 2))
#else
            if (more)
#endif
                load_a(c + 1, areg[Q]);                                         // requested first: they land during this step
            const f32x4 bg0 = *reinterpret_cast<const f32x4 *>(xb + ((par * 2 + 0) * 64 + this->lane) * 4);
            const f32x4 bg1 = *reinterpret_cast<const f32x4 *>(xb + ((par * 2 + 1) * 64 + this->lane) * 4);
            // (one visible use of the LAST of this step's operand loads before any of this step's DMA ops exist: the compiler waits
            //  here once, with only the loads issued a moment ago in flight -- see Engine3T)
            asm volatile("" :: "v"(areg[P][7][NTL - 1]));
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const int ks = m / NTL, r = m % NTL;
                if (stream) ops_of_slot<P>(m, stp);
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[r]) : "v"(areg[P][ks][r]), "v"(ks < 4 ? bg0[ks & 3] : bg1[ks & 3]));
                __builtin_amdgcn_sched_barrier(0);
#ifdef TW_ABLATE
                if (more && !(this->eng_dbg & 1)) {
#else
                if (more) {
#endif
#pragma unroll
                    for (int q = m * (NC + 1) / M; q < (m + 1) * (NC + 1) / M; ++q) gr2[q] = *reinterpret_cast<lds_cf2 *>(ga[q] + Q * R3_TSLOT);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) { gather_finish(par ^ 1); par ^= 1; }
            stp += R3_TSLOT * 4;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of chunk c+2 and the A operands of chunk c+1 have landed
            __syncthreads();
        };
        for (int c = 0; c < nch; c += 2) {
            step(std::integral_constant<int, 0>{}, c);
            if (c + 1 < nch) step(std::integral_constant<int, 1>{}, c + 1);
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // MFMA results -> vector ALU
        TW_S3(q_lp);
        TW_A3(1, q_pro, q_lp);

        // heads: as Engine3S (hidden units into the ring -- both table slots are free now --, one v_fma_f32 chain per half-wave and output)
        {
            const int t0 = wave * NTL;
            float *hid_lo = this->lds_t, *hid_hi = this->lds_t + 4 * 8 * 128;   // units 0..127 | 128..255 as [unit/4... see Engine3S]
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
            TW_S3(q_h1);
            TW_A3(4, q_lp, q_h1);
        }
    }
};

