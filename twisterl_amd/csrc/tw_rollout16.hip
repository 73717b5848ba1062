// tw_rollout16.hip -- fused PPO rollout kernel, f16-input MFMA policy forward (TW_PREC_F16), gfx950.
//
// Same path as tw_rollout.hip (reference rust/src/collector/ppo.rs:54-80 with envs/puzzle.rs and
// nn/policy.rs:56-100,169-172), same RNG keys, same record format; only the policy arithmetic differs
// (see tw_engine16.hpp for the numeric spec and the MFMA mapping).  One workgroup = 4 waves = 256
// episodes; each lane carries the state of TWO episodes (column j of tile 0 and of tile 1); lane half h
// samples, records and draws the twist for tile h and hands action and twist to the other half with one
// cross-half shuffle each.
#include "tw_engine16x2.hpp"

#include <cstdio>
#include <cstdlib>

namespace tw {

#ifdef TW_ABLATE
__device__ unsigned long long g_stamps16[8];
#endif

template <class Eng, int NC, bool PERSIST = false>
__global__ void __launch_bounds__(256, 1) rollout_f16_kernel(const RolloutArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds16[];
#ifdef TW_ABLATE
    const unsigned long long t_kernel0 = __builtin_readcyclecounter();
#endif
    Eng eng;
    eng.begin1(a.pol, lds16);

    const PuzzleConsts env = a.env;
    const int j = eng.j, hh = eng.hh;
    // this lane half owns tile hh: episode (wave, hh, j)
    uint64_t   e_own  = (uint64_t)blockIdx.x * Eng::EPB + (uint64_t)(eng.wave * 64 + hh * 32 + j);
    const bool v_own  = e_own < a.num_episodes;
    uint64_t   eg_own = a.episode_offset + e_own;

    auto from_board = [&](uint64_t b) {                    // persistent mode: start state from a pre-scrambled board
        PuzzleLane s; s.board = b;
        const int z = blank_cell(b);
        s.zx = z % env.width; s.zy = z / env.width; s.depth = env.depth0;
        return s;
    };
    PuzzleLane own;
    own.board = env.ident; own.zx = 0; own.zy = 0; own.depth = 0;
    if (v_own) {
        if constexpr (PERSIST) {             // (RolloutArgs::init_boards: the episodes in the order the lanes take them, the longest-looking ones first)
            const uint4 ib = a.init_boards[e_own];
            e_own = ib.z; eg_own = a.episode_offset + e_own;
            own = from_board(((uint64_t)ib.y << 32) | ib.x);
        } else puzzle_reset(own, env, a.seed, eg_own);
    }
    // both halves keep both episodes' state (the one-hot operands of both tiles are built on every lane)
    PuzzleLane st0, st1;
    {
        PuzzleLane oth;
        oth.board = ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(own.board >> 32), 32, 64) << 32) |
                    (uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)own.board, 32, 64);
        oth.zx = __shfl_xor(own.zx, 32, 64); oth.zy = __shfl_xor(own.zy, 32, 64); oth.depth = __shfl_xor(own.depth, 32, 64);
        st0 = hh ? oth : own; st1 = hh ? own : oth;
    }
    const bool v_oth = __shfl_xor(v_own ? 1 : 0, 32, 64) != 0;
    bool alive0 = hh ? v_oth : v_own, alive1 = hh ? v_own : v_oth;
    int      t = 0;                                        // timestep of the OWN episode (the other half keeps the other tile's)
    uint32_t len_own = 0;
    uint64_t rec_base = e_own * (uint64_t)a.out.t_pad;
    bool     more = PERSIST;                               // the episode queue may still hold work

    uint32_t obs_base[4];
    obs_base_words(env.n_cells, obs_base);
    float bh[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) bh[i] = a.pol.bh16[i];          // uniform: scalar loads, live in SGPRs

    eng.begin2();
#ifdef TW_ABLATE
    for (int i = 0; i < 8; ++i) eng.st[i] = 0;
    unsigned long long t_prev = __builtin_readcyclecounter();
#endif
    while (__syncthreads_or((alive0 || alive1) ? 1 : 0)) {
#ifdef TW_ABLATE
        const unsigned long long t_top = __builtin_readcyclecounter();
        eng.st[7] += t_top - t_prev;
#endif
        // ---- twist draw for the own tile (policy.rs:67-77), exchanged with the other half --------
        int perm_own = -1;
        if (eng.pol.n_perms > 0) {
            const u32x4 w = rng_draw(a.seed, eg_own, (uint32_t)t, STREAM_PERM);
            perm_own = (int)u32_below(w.x, (uint32_t)eng.pol.n_perms);
        }
        const int perm_oth = __shfl_xor(perm_own, 32, 64);
        const int perm0 = hh ? perm_oth : perm_own, perm1 = hh ? perm_own : perm_oth;
        typename Eng::OneHots oh;
        eng.onehots(st0.board, perm0, oh.a0);
        eng.onehots(st1.board, perm1, oh.a1);

#ifdef TW_ABLATE
        asm volatile("" :: "v"(oh.a0[0]), "v"(oh.a1[0]));
        const unsigned long long t_fw = __builtin_readcyclecounter();
        eng.st[0] += t_fw - t_top;
#endif
        f32x16 out0, out1;
        eng.forward(oh, out0, out1);
#ifdef TW_ABLATE
        const unsigned long long t_po = __builtin_readcyclecounter();
#endif

        // ---- own tile: head bias, act-perm, mask, reward, Gumbel-max (policy.rs:56-65,169-172) ---
        // (everything below is select-based: one wave per SIMD has nothing to hide a branch behind)
        float lg[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) lg[i] = (hh ? out1[i] : out0[i]) * Eng::OUT_SCALE + bh[i];
        const float value = (hh ? out1[4] : out0[4]) * Eng::OUT_SCALE + bh[4];
        PuzzleLane mine;
        mine.board = hh ? st1.board : st0.board; mine.zx = hh ? st1.zx : st0.zx; mine.zy = hh ? st1.zy : st0.zy;
        mine.depth = hh ? st1.depth : st0.depth;
        eng.act_perm(perm_own, lg);
        const uint32_t mb = puzzle_maskbits(mine, env);
#pragma unroll
        for (int i = 0; i < 4; ++i) lg[i] = ((mb >> i) & 1u) ? lg[i] : -1e10f;
        const float rew = puzzle_reward(mine, env);
        const u32x4 gw = rng_draw(a.seed, eg_own, (uint32_t)t, STREAM_GUMBEL);
        const int act_own = gumbel_argmax4(lg, gw);
        const bool alive_own = hh ? alive1 : alive0;
        if (alive_own) {
            uint32_t pk[4];
            obs_bytes(mine.board, obs_base, pk);
            store_rec(a.out.rec + rec_base + (uint64_t)t, pk, lg, value, rew, act_own, perm_own);
        }
        const int act_oth = __shfl_xor(act_own, 32, 64);
        const int act0 = hh ? act_oth : act_own, act1 = hh ? act_own : act_oth;
        // ---- is_final / step for both tiles (ppo.rs:78-79) ----------------------------------------
        {
            const bool fin0 = puzzle_final(st0, env), fin1 = puzzle_final(st1, env);
            PuzzleLane n0 = st0, n1 = st1;
            puzzle_step(n0, env, act0); puzzle_step(n1, env, act1);
            const bool go0 = alive0 && !fin0, go1 = alive1 && !fin1;
            st0.board = go0 ? n0.board : st0.board; st0.zx = go0 ? n0.zx : st0.zx; st0.zy = go0 ? n0.zy : st0.zy; st0.depth = go0 ? n0.depth : st0.depth;
            st1.board = go1 ? n1.board : st1.board; st1.zx = go1 ? n1.zx : st1.zx; st1.zy = go1 ? n1.zy : st1.zy; st1.depth = go1 ? n1.depth : st1.depth;
            const bool died_own = hh ? (alive1 && fin1) : (alive0 && fin0);
            len_own = died_own ? (uint32_t)t + 1u : len_own;
            alive0 = go0; alive1 = go1;
            if constexpr (PERSIST) { if (died_own) a.out.ep_len[e_own] = len_own; }
        }
        ++t;
        if constexpr (PERSIST) {
            // a half whose own episode is over takes the next one off the queue; the other half of the lane pair
            // needs that tile's new board too (one-hot operands of both tiles are built on every lane)
            const bool idle_own = !(hh ? alive1 : alive0);
            unsigned got = 0xffffffffu;
            if (idle_own && more) got = atomicAdd(a.queue, 1u);
            const bool took = (uint64_t)got < a.num_episodes;
            more = more && !(idle_own && !took);
            uint64_t nb = 0;
            if (took) { const uint4 ib = a.init_boards[got]; e_own = ib.z; eg_own = a.episode_offset + e_own; rec_base = e_own * (uint64_t)a.out.t_pad; t = 0; nb = ((uint64_t)ib.y << 32) | ib.x; }
            const bool took_oth = __shfl_xor(took ? 1 : 0, 32, 64) != 0;
            const uint64_t nb_oth = ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(nb >> 32), 32, 64) << 32) |
                                    (uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)nb, 32, 64);
            const PuzzleLane n_own = from_board(nb), n_oth = from_board(nb_oth);
            const bool t0 = hh ? took_oth : took, t1 = hh ? took : took_oth;
            const PuzzleLane &n0 = hh ? n_oth : n_own, &n1 = hh ? n_own : n_oth;
            if (t0) { st0 = n0; alive0 = true; }
            if (t1) { st1 = n1; alive1 = true; }
        }
#ifdef TW_ABLATE
        t_prev = __builtin_readcyclecounter();
        eng.st[6] += t_prev - t_po;
#endif
    }
#ifdef TW_ABLATE
    eng.st[3] = __builtin_readcyclecounter() - t_kernel0;      // (slot 3 reused: whole wave lifetime)
    if (eng.lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&g_stamps16[i], eng.st[i]);
#endif
    if constexpr (!PERSIST) { if (v_own) a.out.ep_len[e_own] = len_own; }
    eng.end();
}

template <class Eng, int NC, bool PERSIST>
static int launch16p(const RolloutArgs &a, hipStream_t s, uint32_t *blocks, uint32_t *threads, size_t lds_bytes)
{
    const uint64_t nb = PERSIST ? rollout_f32_resident_episodes(a.reserve_cus) / Eng::EPB : (a.num_episodes + Eng::EPB - 1) / Eng::EPB;
    if (nb == 0 || nb > 0x7fffffffull) { set_error("rollout16: bad episode count %llu", (unsigned long long)a.num_episodes); return TW_ERR_INVALID; }
    if (lds_bytes > 159 * 1024) { set_error("rollout16: %zu bytes of LDS needed, 159 KiB available", lds_bytes); return TW_ERR_UNSUPPORTED; }
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&rollout_f16_kernel<Eng, NC, PERSIST>), lds_bytes)) return rc;
#ifdef TW_ABLATE
    static const unsigned long long zeros[8] = {};
    if (getenv("TW_STAMPS")) TW_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps16), zeros, sizeof(zeros)));
#endif
    hipLaunchKernelGGL((rollout_f16_kernel<Eng, NC, PERSIST>), dim3((unsigned)nb), dim3(Eng::THREADS), lds_bytes, s, a);
    TW_HIP(hipGetLastError());
#ifdef TW_ABLATE
    if (getenv("TW_STAMPS")) {
        unsigned long long h[8];
        TW_HIP(hipStreamSynchronize(s));
        TW_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps16), sizeof(h)));
        const double nw = (double)nb * 4.0;
        fprintf(stderr, "[stamps16] per wave (cycles x100MHz ticks): pre %.0f prologue %.0f stage-body %.0f vmcnt %.0f stage-barrier %.0f heads %.0f post %.0f step-barrier %.0f\n",
                h[0] / nw, h[1] / nw, h[2] / nw, h[3] / nw, h[4] / nw, h[5] / nw, h[6] / nw, h[7] / nw);
    }
#endif
    if (blocks) *blocks = (uint32_t)nb;
    if (threads) *threads = Eng::THREADS;
    return TW_OK;
}

template <class Eng, int NC>
static int launch16(const RolloutArgs &a, hipStream_t s, uint32_t *blocks, uint32_t *threads, size_t lds_bytes)
{
    // more episodes than resident lanes: persistent lanes that take the next episode from a queue (tw_rollout.hip)
    if (a.queue && a.init_boards && a.num_episodes > rollout_f32_resident_episodes(a.reserve_cus)) return launch16p<Eng, NC, true>(a, s, blocks, threads, lds_bytes);
    return launch16p<Eng, NC, false>(a, s, blocks, threads, lds_bytes);
}

template <int NHT, bool SPLIT>
static int launch16_nc(const RolloutArgs &a, hipStream_t s, uint32_t *blocks, uint32_t *threads)
{
    if constexpr (SPLIT) {
        switch (a.pol.f16_nc) {
            case 4:  return launch16<EngineS<NHT, 4>, 4>(a, s, blocks, threads, engineS_lds_bytes<NHT, 4>());
            case 9:  return launch16<EngineS<NHT, 9>, 9>(a, s, blocks, threads, engineS_lds_bytes<NHT, 9>());
            case 16: return launch16<EngineS<NHT, 16>, 16>(a, s, blocks, threads, engineS_lds_bytes<NHT, 16>());
            default: break;
        }
    } else {
        switch (a.pol.f16_nc) {
            case 4:  return launch16<Engine16<NHT, 4>, 4>(a, s, blocks, threads, engine16_lds_bytes<NHT, 4>());
            case 9:  return launch16<Engine16<NHT, 9>, 9>(a, s, blocks, threads, engine16_lds_bytes<NHT, 9>());
            case 16: return launch16<Engine16<NHT, 16>, 16>(a, s, blocks, threads, engine16_lds_bytes<NHT, 16>());
            default: break;
        }
    }
    set_error("rollout16: bad chunk count %d", a.pol.f16_nc);
    return TW_ERR_UNSUPPORTED;
}

template <bool SPLIT>
static int launch_rollout_16(const RolloutArgs &a, hipStream_t s, uint32_t *blocks, uint32_t *threads)
{
    if (a.pol.f16_nc == 0) {
        set_error("precision f16: this policy has no f16 image (needs obs ids of the form cell*n+tile with n <= 16, at most %d "
                  "twists, and twists that map cells to cells)", E16_MAXP);
        return TW_ERR_UNSUPPORTED;
    }
    if (a.env.n_cells < 1 || a.env.n_cells > 16 || a.pol.obs_size != a.env.n_cells * a.env.n_cells || a.pol.n_actions != 4 ||
        a.pol.emb % 32 != 0 || a.pol.emb < (SPLIT ? 64 : 32) || a.pol.emb > 32 * E16_MAX_KT || a.env.n_cells > a.pol.f16_nc ||
        a.out.t_pad < a.env.depth0 + 1 || (SPLIT && (!a.pol.stageS || !a.pol.t0S))) {
        set_error("rollout16: unsupported shape (n_cells=%d obs_size=%d actions=%d emb=%d hidden=%d t_pad=%d split=%d)",
                  a.env.n_cells, a.pol.obs_size, a.pol.n_actions, a.pol.emb, a.pol.hidden, a.out.t_pad, (int)SPLIT);
        return TW_ERR_UNSUPPORTED;
    }
    switch (a.pol.hidden) {
        case 32:  return launch16_nc<1, SPLIT>(a, s, blocks, threads);
        case 64:  return launch16_nc<2, SPLIT>(a, s, blocks, threads);
        case 128: return launch16_nc<4, SPLIT>(a, s, blocks, threads);
        case 256: return launch16_nc<8, SPLIT>(a, s, blocks, threads);
        default:
            set_error("rollout16: hidden size %d not in {32,64,128,256}", a.pol.hidden);
            return TW_ERR_UNSUPPORTED;
    }
}

int launch_rollout_f16(const RolloutArgs &a, hipStream_t s, uint32_t *blocks, uint32_t *threads) { return launch_rollout_16<false>(a, s, blocks, threads); }
int launch_rollout_f16x2(const RolloutArgs &a, hipStream_t s, uint32_t *blocks, uint32_t *threads) { return launch_rollout_16<true>(a, s, blocks, threads); }

}  // namespace tw
