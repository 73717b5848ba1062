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
//     gather-sum (bias + sum over cells of table[id][k], in cell order) from a 16-column chunk of
//     the table held in LDS.  The chunk image is column-permuted ([even k | odd k]) so the four
//     k-steps a lane handles next sit in ONE 16-byte word: one ds_read_b128 per cell feeds four add
//     chains.  Row stride 20 floats: (16i+tile)*20 mod 64 takes 16 distinct 16-byte bank slots for
//     the 16 tiles of a cell (conflict-free; equal tiles broadcast).
//   * an f32 MFMA chain IS a k-ordered fmaf chain, so the result is bit-equal to the oracle's
//     TWO_ARITH_CHAIN forward; rows of W1 are fed in an order (hid()) that makes the accumulator
//     registers come out in natural hidden order for the head product.
//   * both weight streams are pure LDS-DMA (global_load_lds_dwordx4 via inline asm) into a ring of
//     three slots, two chunks ahead, spread between the MFMAs; the inner loop is software-pipelined
//     by hand across chunk boundaries (issue order pinned with sched_barrier, chunk body branch-free),
//     so nothing but the barrier itself sits between the last MFMA of a chunk and the first of the next.
//   * heads: [4 logits + value] x hidden as v_fma_f32 chains in hidden order over v_permlane32_swap'd accumulators
//     (tw_engine.hpp; the dependent-MFMA form cost 6 % of the matrix-pipe time for 1 % of the FLOPs).
//   * the f32-input MFMA executes on the SIMD's f32 FMA lanes, so the gather's VALU adds do NOT overlap
//     it: the practical ceiling is MFMA cycles + add cycles (~0.85 of the MFMA-only peak).
#include "tw_engine.hpp"

#include <cstdlib>
#include <cstring>

namespace tw {

// scrambled start boards of episodes [offset, offset+n) (Env::reset, puzzle.rs:119-133) for the persistent-lane mode
__global__ void __launch_bounds__(256) init_boards_kernel(const PuzzleConsts env, uint64_t seed, uint64_t episode_offset, uint64_t n, uint64_t *out, uint4 *entries)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    PuzzleLane st;
    puzzle_reset(st, env, seed, episode_offset + i);
    if (out) out[i] = st.board;
    if (entries) entries[i] = make_uint4((uint32_t)st.board, (uint32_t)(st.board >> 32), (uint32_t)i, 0u);
}

int launch_init_boards(const PuzzleConsts &env, uint64_t seed, uint64_t episode_offset, uint64_t n, uint64_t *out, hipStream_t s, uint4 *entries)
{
    if (n == 0) return TW_OK;
    hipLaunchKernelGGL(init_boards_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, env, seed, episode_offset, n, out, entries);
    TW_HIP(hipGetLastError());
    return TW_OK;
}

// Longest-first order for the self-play episode queue.  A collect is as long as its longest episode's chain of searches plus
// the time that episode waited for a walker; episodes that are not solved run to the depth limit (17 moves against a mean of 5 at
// difficulty 8), and the boards far from the solved one are the ones that are not solved.  Key = sum over the tiles of the Manhattan
// distance to their place (a lower bound of the moves needed), counting sort by decreasing key (below), ~0.1 ms for 262,144 episodes.
// Which walker runs which episode never changes a bit of the result (every episode is keyed by its own global index).
__device__ __forceinline__ bool rank_is_leader(unsigned long long peers, int lane) { return (peers & ((1ull << lane) - 1ull)) == 0ull; }

// A STABLE counting sort by decreasing key (equal keys stay in index order: the schedule, and with it the collect's time, is the same
// every run), in three small launches so that a quarter of a million episodes cost ~0.1 ms, not 1.8: the indices are cut into segments
// of ORD_SPAN (one wave each, 16 per workgroup); (1) every wave counts its segment's keys, (2) one workgroup turns the counts into every
// segment's first output position per key -- all larger keys first, then the segments in index order --, (3) every wave scatters its
// segment in order from those cursors.
constexpr uint32_t ORD_WAVES = 16;
struct OrdKey {
    uint64_t lutx, luty; int n_cells, width;
    __device__ explicit OrdKey(const PuzzleConsts &env) : lutx(0), luty(0), n_cells(env.n_cells), width(env.width)
    {
        for (int i = 0; i < env.n_cells; ++i) {                 // place of tile v in the solved board, one nibble each
            const uint64_t v = nib(env.ident, i);
            lutx |= (uint64_t)(i % env.width) << (4 * v); luty |= (uint64_t)(i / env.width) << (4 * v);
        }
    }
    __device__ uint32_t operator()(uint64_t b) const
    {
        int d = 0;
        for (int i = 0; i < n_cells; ++i) {
            const uint64_t v = nib(b, i);
            if (v == 0) continue;                               // the blank
            const int dx = i % width - (int)((lutx >> (4 * v)) & 15), dy = i / width - (int)((luty >> (4 * v)) & 15);
            d += (dx < 0 ? -dx : dx) + (dy < 0 ? -dy : dy);
        }
        return (uint32_t)(d < 63 ? d : 63);
    }
};

__global__ void __launch_bounds__(1024) episode_order_count_kernel(const PuzzleConsts env, const uint64_t *boards, uint32_t n, uint32_t span, uint32_t *cnt)
{
    __shared__ uint32_t c[ORD_WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < (int)ORD_WAVES * 64; i += 1024) (&c[0][0])[i] = 0;
    const OrdKey key(env);
    const uint32_t seg = blockIdx.x * ORD_WAVES + (uint32_t)wave;
    const uint64_t lo64 = (uint64_t)seg * span;
    const uint32_t lo = lo64 < n ? (uint32_t)lo64 : n, hi = lo64 + span < n ? (uint32_t)(lo64 + span) : n;
    __syncthreads();
    for (uint32_t i = lo + lane; i < hi; i += 64) atomicAdd(&c[wave][key(boards[i])], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < (int)ORD_WAVES * 64; i += 1024) cnt[(size_t)blockIdx.x * ORD_WAVES * 64 + i] = (&c[0][0])[i];
}

__global__ void __launch_bounds__(64) episode_order_prefix_kernel(uint32_t *cnt, uint32_t n_seg)      // cnt[seg][key] -> first output position
{
    __shared__ uint32_t total[64];
    const int k = threadIdx.x;
    uint32_t t = 0;
    for (uint32_t sgm = 0; sgm < n_seg; ++sgm) t += cnt[(size_t)sgm * 64 + k];
    total[k] = t;
    __syncthreads();
    uint32_t pos = 0;
    for (int kk = 63; kk > k; --kk) pos += total[kk];
    for (uint32_t sgm = 0; sgm < n_seg; ++sgm) { const uint32_t c = cnt[(size_t)sgm * 64 + k]; cnt[(size_t)sgm * 64 + k] = pos; pos += c; }
}

__global__ void __launch_bounds__(1024) episode_order_scatter_kernel(const PuzzleConsts env, const uint64_t *boards, uint32_t n, uint32_t span, const uint32_t *base,
                                                                      uint32_t *order, uint4 *entries)
{
    __shared__ uint32_t cur[ORD_WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < (int)ORD_WAVES * 64; i += 1024) (&cur[0][0])[i] = base[(size_t)blockIdx.x * ORD_WAVES * 64 + i];
    const OrdKey key(env);
    const uint32_t seg = blockIdx.x * ORD_WAVES + (uint32_t)wave;
    const uint64_t lo64 = (uint64_t)seg * span;
    const uint32_t lo = lo64 < n ? (uint32_t)lo64 : n, hi = lo64 + span < n ? (uint32_t)(lo64 + span) : n;
    __syncthreads();
    for (uint32_t i0 = lo; i0 < hi; i0 += 64) {                 // (wave-uniform trip count)
        const uint32_t i = i0 + lane;
        const bool on = i < hi;
        const uint64_t bd = on ? boards[i] : 0ull;
        const uint32_t k = on ? key(bd) : 64u;
        unsigned long long peers = __builtin_amdgcn_ballot_w64(on);
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(on && ((k >> b) & 1u));
            peers &= ((k >> b) & 1u) ? m : ~m;
        }
        if (on) {
            const uint32_t rank = (uint32_t)__builtin_popcountll(peers & ((1ull << lane) - 1ull));
            const uint32_t pos = cur[wave][k] + rank;
            if (order) order[pos] = i;
            if (entries) entries[pos] = make_uint4((uint32_t)bd, (uint32_t)(bd >> 32), i, 0u);
        }
        __builtin_amdgcn_wave_barrier();
        if (on && rank_is_leader(peers, lane)) cur[wave][k] += (uint32_t)__builtin_popcountll(peers);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// segments of at least 1,024 indices, at most 64 workgroups
static void episode_order_shape(uint64_t n, uint32_t *span, uint32_t *blocks)
{
    uint64_t sp = 1024;
    while ((n + sp * ORD_WAVES - 1) / (sp * ORD_WAVES) > 64) sp *= 2;
    *span = (uint32_t)sp;
    *blocks = (uint32_t)((n + sp * ORD_WAVES - 1) / (sp * ORD_WAVES));
}
size_t episode_order_scratch_bytes(uint64_t n)
{
    uint32_t span, blocks; episode_order_shape(n, &span, &blocks);
    return (size_t)blocks * ORD_WAVES * 64 * sizeof(uint32_t);
}

int launch_episode_order(const PuzzleConsts &env, const uint64_t *boards, uint64_t n, uint32_t *order, void *scratch, hipStream_t s, uint4 *entries)
{
    if (n == 0) return TW_OK;
    if (n > 0xffffffffull || env.n_cells < 1 || env.n_cells > 16 || !scratch) { set_error("episode order: unsupported shape"); return TW_ERR_INVALID; }
    uint32_t span, blocks; episode_order_shape(n, &span, &blocks);
    uint32_t *cnt = reinterpret_cast<uint32_t *>(scratch);
    hipLaunchKernelGGL(episode_order_count_kernel, dim3(blocks), dim3(1024), 0, s, env, boards, (uint32_t)n, span, cnt);
    hipLaunchKernelGGL(episode_order_prefix_kernel, dim3(1), dim3(64), 0, s, cnt, blocks * ORD_WAVES);
    hipLaunchKernelGGL(episode_order_scatter_kernel, dim3(blocks), dim3(1024), 0, s, env, boards, (uint32_t)n, span, (const uint32_t *)cnt, order, entries);
    TW_HIP(hipGetLastError());
    return TW_OK;
}

#ifdef TW_ABLATE   // diagnostic build: cycle stamps of the generic engine (TW_STAMPS=1 prints them per launch)
__device__ unsigned long long g_gen_stamps[8];
#endif

// (the generic engine's workgroups are small -- 256 threads, exact-size activation buffers: two of them share a CU, and while one
//  gathers its embeddings from L2 the other one keeps the matrix cores busy)
template <int NT, int NC, int DBG = 0, int NW = 8, bool PERSIST = false>
__global__ void __launch_bounds__((Geom<NT, NC, DBG, NW>::WAVES * 64), ((NW == 8 || NW == -65 || NW == -5) ? 2 : 1)) rollout_f32_kernel(const RolloutArgs a)
{
    using Eng = typename Geom<NT, NC, DBG, NW>::Eng;       // NW < 0: -NW waves share 32 episodes (Engine3S); all carry the same state
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Eng eng;
    eng.begin1(a.pol, lds);                               // first weight chunks stream in while the scramble runs

    const PuzzleConsts env = a.env;
    const int j = eng.j, h = eng.h;
    // Loop-carried per-episode state is kept to five registers (board, depth, episode, t): everything derivable (blank position,
    // global episode index, record address, obs-id constants) is recomputed after the forward -- held across it, those were the
    // registers the 8-wave x 256-accumulator shape spilled to scratch.
    const uint64_t e_first = (uint64_t)blockIdx.x * Eng::EPB + (uint64_t)eng.ep_lane();
    const bool valid  = e_first < a.num_episodes;
    const bool writer = h == 0 && eng.primary();          // the lane that stores the episode's records
    uint32_t e_local = (uint32_t)e_first;                 // (launch_geom bounds the episode count of a launch to 2^31)
    uint64_t board = env.ident; int depth = 0;
    if (valid) {
        if constexpr (PERSIST) {              // the episode and its start board from the pre-pass (RolloutArgs::init_boards: in the order the lanes take them)
            const uint4 ib = a.init_boards[e_local];
            e_local = ib.z; board = ((uint64_t)ib.y << 32) | ib.x; depth = env.depth0;
        }
        else { PuzzleLane s0; puzzle_reset(s0, env, a.seed, a.episode_offset + e_first); board = s0.board; depth = s0.depth; }
    }

    bool     alive = valid;
    bool     more  = PERSIST;                             // the queue may still hold episodes
    int      t = 0;                                       // (stays at the last record once the episode is over: records = t + 1)

    eng.begin2();
    // (the barrier inside __syncthreads_or publishes the first two ring slots and the LDS constants)
#ifdef TW_ABLATE
    const unsigned long long q_loop = __builtin_readcyclecounter();
    unsigned long long n_fwd = 0;
#endif

    while (__syncthreads_or(alive ? 1 : 0)) {
#ifdef TW_ABLATE
        ++n_fwd;
#endif
        // ---- observe (puzzle.rs:183-185) + twist of the obs ids (policy.rs:67-83) -------------
        int perm = -1;
        if (eng.pol.n_perms > 0) {
            const u32x4 w = rng_draw(a.seed, a.episode_offset + (uint64_t)e_local, (uint32_t)t, STREAM_PERM);
            perm = (int)u32_below(w.x, (uint32_t)eng.pol.n_perms);
        }
        int rowoff[NC];
        eng.rows_of(board, env.n_cells, perm, rowoff);

        float lg[4]; float value;
        eng.forward(rowoff, lg, value);

        PuzzleLane st; st.board = board; st.depth = depth;
        { const int z = blank_cell(board); st.zx = z % env.width; st.zy = z / env.width; }
        float rew = 0.0f; int action = t & 3;
        if constexpr (!(DBG & 8)) {
            eng.act_perm(perm, lg);
            const uint32_t mb = puzzle_maskbits(st, env);
#pragma unroll
            for (int i = 0; i < 4; ++i) lg[i] = ((mb >> i) & 1u) ? lg[i] : -1e10f;   // policy.rs:62
            rew = puzzle_reward(st, env);
            const u32x4 gw = rng_draw(a.seed, a.episode_offset + (uint64_t)e_local, (uint32_t)t, STREAM_GUMBEL);
            action = gumbel_argmax4(lg, gw);
        }
        // ---- push the record (ppo.rs:71-76), then is_final / step (ppo.rs:78-79) --------------
        if (alive) {
            if (writer) {
                const uint64_t rec = (uint64_t)e_local * (uint64_t)a.out.t_pad + (uint64_t)t;
                uint32_t obs_base[4], pk[4];
                obs_base_words(env.n_cells, obs_base);
                obs_bytes(board, obs_base, pk);
                store_rec(a.out.rec + rec, pk, lg, value, rew, action, perm);
            }
            if (puzzle_final(st, env)) {
                alive = false;
                if constexpr (PERSIST) { if (writer) a.out.ep_len[e_local] = (uint32_t)t + 1u; }
            } else { puzzle_step(st, env, action); board = st.board; depth = st.depth; ++t; }
        }
        if constexpr (PERSIST) {
            const bool want = !alive && more;             // take the next episode off the queue (every lane of the episode the same one)
            unsigned got = 0xffffffffu;
            if (want && writer) got = atomicAdd(a.queue, 1u);
            if constexpr (Eng::SPLIT) {                   // all waves of the workgroup carry this episode: hand the index round in LDS
                unsigned *bc = reinterpret_cast<unsigned *>(eng.lds_user);
                if (writer) bc[j] = got;
                __syncthreads();
                got = bc[j];
            } else got = (unsigned)__shfl((int)got, j, 64);
            if (want) {
                if ((uint64_t)got < a.num_episodes) {
                    const uint4 ib = a.init_boards[got];
                    e_local = ib.z; board = ((uint64_t)ib.y << 32) | ib.x; depth = env.depth0;
                    alive = true; t = 0;
                } else more = false;
            }
        }
    }
    if constexpr (!PERSIST) { if (valid && writer) a.out.ep_len[e_local] = (uint32_t)t + 1u; }
    eng.end();
#ifdef TW_ABLATE
    if constexpr (NW == -64 || NW == -65) {
        if (eng.lane == 0 && (eng.wave == 0 || eng.wave == 3)) for (int i = 0; i < 4; ++i) atomicAdd(&g_gen_stamps[(eng.wave ? 4 : 0) + i], eng.stq[i]);
    }
    if constexpr (NW == -16 || NW == -4 || NW == -5) {      // wave 0: prologue | chunk loop | - | - | heads; forwards; cycles in the step loop
        if (eng.lane == 0 && eng.wave == 0) for (int i = 0; i < 5; ++i) atomicAdd(&g_gen_stamps[i], eng.stq[i]);
        if (eng.lane == 0 && eng.wave == 0) atomicAdd(&g_gen_stamps[5], n_fwd);
        if (eng.lane == 0 && eng.wave == 0) atomicAdd(&g_gen_stamps[6], __builtin_readcyclecounter() - q_loop);
    }
#endif
}

// persistent mode: one 256-episode workgroup per CU of the current device (256 on an MI355X; every rollout kernel needs
// most of a CU's LDS, so one workgroup is resident per CU)
static uint64_t persist_blocks(int reserve_cus)
{
    const int cus = device_cus();
    const int r = reserve_cus < 0 ? 0 : (reserve_cus > cus - 1 ? cus - 1 : reserve_cus);   // reserve_cus CUs stay free (RCCL beside the persistent grid)
    return (uint64_t)(cus - r);
}

uint64_t rollout_f32_resident_episodes(int reserve_cus) { return persist_blocks(reserve_cus) * 8 * EPW; }

// generic policy stacks (EngineV): workgroups of 16 episodes, as many per CU as their LDS allows (at most two: the registers)
static uint64_t generic_groups_per_cu(const PolicyDev &pol, int n_cells)
{
    const size_t lds_bytes = (n_cells <= 4 ? EngineV<4>::lds_floats(pol) : (n_cells <= 9 ? EngineV<9>::lds_floats(pol) : EngineV<16>::lds_floats(pol))) * sizeof(float);
    return lds_bytes * 2 <= 159 * 1024 ? 2 : 1;
}
uint64_t rollout_generic_resident_episodes(const PolicyDev &pol, int n_cells, int reserve_cus) { return persist_blocks(reserve_cus) * generic_groups_per_cu(pol, n_cells) * 16; }

// Episodes are ragged (a solved puzzle ends its episode), and a lane whose episode is over can only be refilled when there
// are more episodes than lanes.  Between CUs x 32 episodes and 3/4 of CUs x 256 the small-batch shape with the episode
// queue (CUs x 32 lanes, every one busy until the queue is empty) beats both running the 32-episode workgroups one after
// the other and leaving most of the lanes of the 256-episode shape idle behind the longest episodes (self-play, 32,768
// episodes: 1.8x); from there on the 256-episode shape wins.
// Upper end of that range: where the 256-episode shape overtakes with episodes of equal length (the rollout crossover of
// waves_per_group()); self-play episodes are always ragged, there the range extends to 3/4 of CUs x 256.
uint64_t f32_resident_episodes(uint64_t num_episodes, int hidden, bool selfplay, int reserve_cus)
{
    const uint64_t full = rollout_f32_resident_episodes(reserve_cus), small = full / 8;
    const bool in_range = selfplay ? num_episodes * 4 <= full * 3 : waves_per_group(num_episodes) != 8;
    if (hidden >= 128 && num_episodes > small && in_range && !launch_options().force_geom) {
#ifdef TW_ABLATE   // diagnostic build, TW_MID_G=1: two 32-episode workgroups per CU (Engine3G; measured no faster -- profiles/r03_mid_rollout_two_groups_per_cu.txt)
        if (!selfplay && num_episodes >= 2 * small && getenv("TW_MID_G")) return 2 * small;
#endif
        return small;
    }
    return full;
}

template <int NT, int NC, int DBG = 0, int NW = 8, bool PERSIST = false>
static int launch_geom(const RolloutArgs &a, hipStream_t s, uint32_t *blocks, uint32_t *threads)
{
    using G = Geom<NT, NC, DBG, NW>;
    constexpr int EPB = G::Eng::EPB, THREADS = 64 * G::WAVES;
    const uint64_t nb = PERSIST ? persist_blocks(a.reserve_cus) * (NW == -65 ? generic_groups_per_cu(a.pol, a.env.n_cells) : (NW == -5 ? 2 : 1)) : (a.num_episodes + EPB - 1) / EPB;
    if (nb == 0 || nb > 0x7fffffffull) { set_error("rollout: bad episode count %llu", (unsigned long long)a.num_episodes); return TW_ERR_INVALID; }
    const size_t lds_bytes = G::Eng::lds_floats(a.pol) * sizeof(float);
    if (lds_bytes > 159 * 1024) { set_error("rollout: %zu bytes of LDS needed, 159 KiB available", lds_bytes); return TW_ERR_UNSUPPORTED; }
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&rollout_f32_kernel<NT, NC, DBG, NW, PERSIST>), lds_bytes)) return rc;
#ifdef TW_ABLATE
    const bool stamps = (NW == -64 || NW == -65 || NW == -16 || NW == -4 || NW == -5) && getenv("TW_STAMPS");
    if (stamps) { unsigned long long z[8] = {0}; TW_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_gen_stamps), z, sizeof(z))); }
    { const char *d = getenv("TW_ENG_DBG"); const int v = d ? atoi(d) : 0; TW_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_eng_dbg), &v, sizeof(v))); }
#endif
    hipLaunchKernelGGL((rollout_f32_kernel<NT, NC, DBG, NW, PERSIST>), dim3((unsigned)nb), dim3(THREADS), lds_bytes, s, a);
#ifdef TW_ABLATE
    if (stamps) {
        unsigned long long g[8];
        TW_HIP(hipStreamSynchronize(s));
        TW_HIP(hipMemcpyFromSymbol(g, HIP_SYMBOL(g_gen_stamps), sizeof(g)));
        if (NW == -64 || NW == -65)
            fprintf(stderr, "[generic engine stamps, cycles summed over %llu workgroups] wave0: embed %llu common %llu value %llu action %llu | wave3: %llu %llu %llu %llu\n",
                    (unsigned long long)nb, g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7]);
        else
            fprintf(stderr, "[geometry %d, wave 0, cycles per forward] prologue %.0f chunk loop %.0f heads %.0f | whole step %.0f (%llu forwards)\n",
                    NW, (double)g[0] / g[5], (double)g[1] / g[5], (double)g[4] / g[5], (double)g[6] / g[5], g[5]);
    }
#endif
    TW_HIP(hipGetLastError());
    if (blocks) *blocks = (uint32_t)nb;
    if (threads) *threads = THREADS;
    return TW_OK;
}

template <int NT, int NC>
static int launch_one(const RolloutArgs &a, hipStream_t s, uint32_t *blocks, uint32_t *threads)
{
#ifdef TW_ABLATE   // timing-only ablation builds: TW_ROLLOUT_DBG = 1 | 2 | 4 | 8 (see tw_engine.hpp)
    if constexpr (NT == 8 && NC == 16) {
        const char *d = getenv("TW_ROLLOUT_DBG");
        switch (d ? atoi(d) : 0) {
            case 1: return launch_geom<NT, NC, 1>(a, s, blocks, threads);
            case 2: return launch_geom<NT, NC, 2>(a, s, blocks, threads);
            case 4: return launch_geom<NT, NC, 4>(a, s, blocks, threads);
            case 8: return launch_geom<NT, NC, 8>(a, s, blocks, threads);
            default: break;
        }
    }
#endif
    // small batches: fewer waves per workgroup, so that the episodes spread over more CUs
    const uint64_t resident = f32_resident_episodes(a.num_episodes, a.pol.hidden, false, a.reserve_cus);
    if (a.queue && a.init_boards && a.num_episodes > resident) {
        if constexpr (NT >= 4) {
            const uint64_t full = rollout_f32_resident_episodes(a.reserve_cus);
#ifdef TW_ABLATE
            if (resident == full / 4) return launch_geom<NT, NC, 0, -5, true>(a, s, blocks, threads);
#endif
            if (resident < full) return launch_geom<NT, NC, 0, -4, true>(a, s, blocks, threads);
        }
        return launch_geom<NT, NC, 0, 8, true>(a, s, blocks, threads);
    }
    const int nw = geometry_for<NT>(a.num_episodes);
    if constexpr (NT >= 4) { if (nw == -16) return launch_geom<NT, NC, 0, -16>(a, s, blocks, threads); }
    if constexpr (NT >= 4) { if (nw == -4) return launch_geom<NT, NC, 0, -4>(a, s, blocks, threads); }
    else if constexpr (NT == 2) { if (nw == -2) return launch_geom<NT, NC, 0, -2>(a, s, blocks, threads); }
    else {
        if (nw == 1) return launch_geom<NT, NC, 0, 1>(a, s, blocks, threads);
        if (nw == 2) return launch_geom<NT, NC, 0, 2>(a, s, blocks, threads);
    }
    return launch_geom<NT, NC>(a, s, blocks, threads);
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
    if (a.pol.generic) {          // any Sequential depth: the vector-ALU engine (tw_engine_generic.hpp); shapes validated by tw_policy_create
        if (a.env.n_cells < 1 || a.env.n_cells > 16 || a.pol.obs_size != a.env.n_cells * a.env.n_cells || a.pol.n_actions != 4 ||
            a.out.t_pad < a.env.depth0 + 1 || (a.queue == nullptr) != (a.init_boards == nullptr)) {
            set_error("rollout: unsupported shape for a generic policy (n_cells=%d obs_size=%d actions=%d)", a.env.n_cells, a.pol.obs_size, a.pol.n_actions);
            return TW_ERR_UNSUPPORTED;
        }
        const int nc = a.env.n_cells;
        if (a.queue) {            // more episodes than one 16-episode workgroup per CU: persistent lanes + episode queue
            if (nc <= 4) return launch_geom<0, 4, 0, -65, true>(a, s, blocks, threads);
            if (nc <= 9) return launch_geom<0, 9, 0, -65, true>(a, s, blocks, threads);
            return launch_geom<0, 16, 0, -65, true>(a, s, blocks, threads);
        }
        if (nc <= 4) return launch_geom<0, 4, 0, -65>(a, s, blocks, threads);
        if (nc <= 9) return launch_geom<0, 9, 0, -65>(a, s, blocks, threads);
        return launch_geom<0, 16, 0, -65>(a, s, blocks, threads);
    }
    // host-side shape checks: everything the kernel indexes with is validated here
    if (a.env.n_cells < 1 || a.env.n_cells > 16 || a.pol.obs_size != a.env.n_cells * a.env.n_cells ||
        a.pol.obs_size > 256 || a.pol.n_actions != 4 || a.pol.emb % 32 != 0 || a.pol.emb < 32 ||
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
