// tw_rollout_big.hip -- the fused PPO rollout for Puzzle boards of 17 .. 64 cells (5 x 5, 6 x 4, 6 x 6, 8 x 8, ..) on the device.
//
// Puzzle::new takes any width x height (reference rust/src/envs/puzzle.rs:34-42); the rollout kernels of tw_rollout.hip pack a
// board as 16 nibbles in one 64-bit register.  Here a board of up to 25 cells is 25 x 5 bits in a 128-bit integer, a board of up
// to 36 / 64 cells one byte per cell in 9 / 16 registers (Board8); obs ids (cell * n_cells + tile, puzzle.rs:183-185) run up to
// 4,095 and are stored as uint16, and the policy -- any Sequential stack over an obs_size above 256 is
// a "generic" policy (tw_policy_create) -- runs on EngineV (tw_engine_generic.hpp: every Linear on the matrix cores, the
// EmbeddingBag gathered from global memory).  Same path otherwise: PPOCollector::single_collect (collector/ppo.rs:54-80),
// Policy::forward_with_perm (nn/policy.rs:56-100), sample_from_logits (policy.rs:169-172), Env::step / masks / reward / is_final
// (puzzle.rs:135-181), same RNG streams, same arithmetic: bit-equal to the oracle.  evaluate() without MCTS runs here too
// (solve_big_kernel).  Boards above 64 cells, self-play, solve() from a given state and MCTS-guided evaluate of boards above 16
// cells stay on the host-stepped path (tw_env_generic.hip).
#include "tw_engine_generic.hpp"
#include "tw_big_board.hpp"

namespace tw {

template <int NC>
__global__ void __launch_bounds__(256, 1) rollout_big_kernel(const RolloutArgs a, uint16_t *obs16)
{
    using Eng = EngineV<NC, false>;      // (the MFMAs as the intrinsic: tw_engine_generic.hpp, ASM_MFMA)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Eng eng;
    eng.begin1(a.pol, lds);
    const PuzzleConsts env = a.env;
    const int j = eng.j;
    const int nc = env.n_cells;
    using Board = typename BoardOf<NC>::T;
    const Board ident = Board::ident(nc);
    // every lane group of every wave carries the state of column j (the engine's mapping); lanes 0-15 of wave 0 store
    const uint64_t e_local = (uint64_t)blockIdx.x * Eng::EPB + (uint64_t)j;
    const bool valid  = e_local < a.num_episodes;
    const bool writer = eng.h == 0 && eng.primary();
    const uint64_t e_global = a.episode_offset + e_local;
    BigLaneT<Board> st; st.board = ident; st.zx = 0; st.zy = 0; st.depth = 0;
    if (valid) {                                                             // Env::reset (puzzle.rs:119-133)
        for (int d = 0; d < env.difficulty; ++d) {
            const u32x4 w = rng_draw(a.seed, e_global, (uint32_t)d, STREAM_SCRAMBLE);
            big_step(st, env, (int)u32_below(w.x, 4u));
        }
        st.depth = env.depth0;
    }
    bool alive = valid;
    int  t = 0;
    eng.begin2();

    while (__syncthreads_or(alive ? 1 : 0)) {
        // ---- observe (puzzle.rs:183-185) + twist of the obs ids (policy.rs:67-83) -------------
        int perm = -1;
        if (eng.pol.n_perms > 0) {
            const u32x4 w = rng_draw(a.seed, e_global, (uint32_t)t, STREAM_PERM);
            perm = (int)u32_below(w.x, (uint32_t)eng.pol.n_perms);
        }
        int rowoff[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            int row = -1;
            if (i < nc) {
                const int id = i * nc + (int)st.board.cell(i);
                row = perm >= 0 ? (int)eng.pol.obs_perms16[(size_t)perm * eng.pol.obs_size + id] : id;
            }
            rowoff[i] = row;
        }
        float lg[4]; float value;
        eng.forward(rowoff, lg, value);
        eng.act_perm(perm, lg);
        const uint32_t mb = (st.zx > 0 ? 1u : 0u) | (st.zy > 0 ? 2u : 0u) | (st.zx < env.width - 1 ? 4u : 0u) | (st.zy < env.height - 1 ? 8u : 0u);   // puzzle.rs:162-165
#pragma unroll
        for (int i = 0; i < 4; ++i) lg[i] = ((mb >> i) & 1u) ? lg[i] : -1e10f;       // policy.rs:62
        const bool solved = st.board == ident;
        const float rew = solved ? 1.0f : (st.depth == 0 ? -0.5f : env.r_step);      // puzzle.rs:171-177
        const u32x4 gw = rng_draw(a.seed, e_global, (uint32_t)t, STREAM_GUMBEL);
        const int action = gumbel_argmax4(lg, gw);
        // ---- push the record (ppo.rs:71-76), then is_final / step (ppo.rs:78-79) --------------
        if (alive) {
            if (writer) {
                const uint64_t rec = e_local * (uint64_t)a.out.t_pad + (uint64_t)t;
                const uint32_t zero4[4] = {0u, 0u, 0u, 0u};
                store_rec(a.out.rec + rec, zero4, lg, value, rew, action, perm);
                uint16_t *o = obs16 + rec * (uint64_t)nc;
#pragma unroll
                for (int i = 0; i < NC; ++i) if (i < nc) o[i] = (uint16_t)(i * nc + (int)st.board.cell(i));
            }
            if (st.depth == 0 || solved) alive = false;                      // puzzle.rs:167-169
            else { big_step(st, env, action); ++t; }
        }
    }
    if (valid && writer) a.out.ep_len[e_local] = (uint32_t)t + 1u;
    eng.end();
}

// evaluate() of such a board (rust/src/rl/evaluate.rs:22-89 over single_solve, rl/solve.rs:17-71; tw_solve.hip for boards up to 16
// cells): one column = one ATTEMPT (episode e, search a): reset, then while !is_final { total += reward; probs = Policy::predict
// (masked softmax, random twist); action = argmax | weighted sample; step }.  Best-of-N and the means are reduced on the host.
template <int NC>
__global__ void __launch_bounds__(256, 1) solve_big_kernel(const SolveArgs a)
{
    using Eng = EngineV<NC, false>;      // (the MFMAs as the intrinsic: tw_engine_generic.hpp, ASM_MFMA)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Eng eng;
    eng.begin1(a.pol, lds);
    const PuzzleConsts env = a.env;
    const int nc = env.n_cells;
    using Board = typename BoardOf<NC>::T;
    const Board ident = Board::ident(nc);
    const uint64_t att = (uint64_t)blockIdx.x * Eng::EPB + (uint64_t)eng.j;
    const bool valid = att < a.num_attempts, writer = eng.h == 0 && eng.primary();
    const uint64_t ep  = a.episode_offset + att / a.num_searches;                    // episode: keys the start state
    const uint64_t key = ep * (uint64_t)a.num_searches + att % a.num_searches;       // keys this attempt's draws
    BigLaneT<Board> st; st.board = ident; st.zx = 0; st.zy = 0; st.depth = 0;
    if (valid && a.from_state) {                                                      // solve(): every attempt clones the caller's env (solve.rs:85)
        st.board = big_board_from_cells(a.start_cells, nc, ident); st.zx = a.start_zx; st.zy = a.start_zy; st.depth = a.start_depth;
    } else if (valid) {                                                               // env.reset() per episode (evaluate.rs:39,65)
        for (int d = 0; d < env.difficulty; ++d) {
            const u32x4 w = rng_draw(a.seed, ep, (uint32_t)d, STREAM_SCRAMBLE);
            big_step(st, env, (int)u32_below(w.x, 4u));
        }
        st.depth = env.depth0;
    }
    bool  alive = valid && !(st.depth == 0 || st.board == ident);
    float total = 0.0f;
    int   t = 0;
    eng.begin2();
    while (__syncthreads_or(alive ? 1 : 0)) {
        int perm = -1;
        if (eng.pol.n_perms > 0) {
            const u32x4 w = rng_draw(a.seed, key, (uint32_t)t, STREAM_PERM);
            perm = (int)u32_below(w.x, (uint32_t)eng.pol.n_perms);
        }
        int rowoff[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            int row = -1;
            if (i < nc) {
                const int id = i * nc + (int)st.board.cell(i);
                row = perm >= 0 ? (int)eng.pol.obs_perms16[(size_t)perm * eng.pol.obs_size + id] : id;
            }
            rowoff[i] = row;
        }
        float lg[4], value;
        eng.forward(rowoff, lg, value);
        eng.act_perm(perm, lg);
        const uint32_t mb = (st.zx > 0 ? 1u : 0u) | (st.zy > 0 ? 2u : 0u) | (st.zx < env.width - 1 ? 4u : 0u) | (st.zy < env.height - 1 ? 8u : 0u);
        float probs[4];
        masked_softmax4(lg, mb, probs);                                               // policy.rs:43-47
        if (alive) {
            total = total + (st.board == ident ? 1.0f : (st.depth == 0 ? -0.5f : env.r_step));      // solve.rs:31,34
            int action = 0;
            if (a.deterministic) {
                float bv = probs[0];
#pragma unroll
                for (int i = 1; i < 4; ++i) if (probs[i] > bv) { bv = probs[i]; action = i; }
            } else {
                const u32x4 w = rng_draw(a.seed, key, (uint32_t)t, STREAM_SOLVE);
                action = sample_weighted4(probs, 4, u32_to_unit(w.x));
            }
            if (a.actions && writer) a.actions[att * (uint64_t)a.t_pad + (uint64_t)t] = (uint8_t)action;
            big_step(st, env, action);                                                // solve.rs:56
            ++t;
            if (st.depth == 0 || st.board == ident) alive = false;
        }
    }
    if (valid && writer) {
        total = total + (st.board == ident ? 1.0f : (st.depth == 0 ? -0.5f : env.r_step));          // solve.rs:65-66
        a.success[att] = st.board == ident ? 1.0f : 0.0f;                             // solve.rs:68
        a.total[att]   = total;
        a.n_steps[att] = (uint32_t)t;
    }
    eng.end();
}

template <int NC>
static int launch_solve_big_nc(const SolveArgs &a, hipStream_t s)
{
    using Eng = EngineV<NC, false>;      // (the MFMAs as the intrinsic: tw_engine_generic.hpp, ASM_MFMA)
    const uint64_t nb = (a.num_attempts + Eng::EPB - 1) / Eng::EPB;
    if (nb == 0 || nb > 0x7fffffffull) { set_error("solve: bad attempt count %llu", (unsigned long long)a.num_attempts); return TW_ERR_INVALID; }
    const size_t lds_bytes = Eng::lds_floats(a.pol) * sizeof(float);
    if (lds_bytes > 159 * 1024) { set_error("solve: %zu bytes of LDS needed, 159 KiB available", lds_bytes); return TW_ERR_UNSUPPORTED; }
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&solve_big_kernel<NC>), lds_bytes)) return rc;
    hipLaunchKernelGGL((solve_big_kernel<NC>), dim3((unsigned)nb), dim3(Eng::THREADS), lds_bytes, s, a);
    TW_HIP(hipGetLastError());
    return TW_OK;
}

int launch_solve_big(const SolveArgs &a, hipStream_t s)
{
    if (a.env.n_cells <= 16 || a.env.n_cells > 64 || !a.pol.generic || a.pol.obs_size != a.env.n_cells * a.env.n_cells || a.pol.n_actions != 4 ||
        (a.pol.n_perms > 0 && !a.pol.obs_perms16) || a.num_searches == 0 || !a.success || !a.total || !a.n_steps || (a.actions && a.t_pad < 1) ||
        (a.from_state && !a.start_cells)) {
        set_error("evaluate (boards above 16 cells): unsupported shape (n_cells=%d obs_size=%d actions=%d)", a.env.n_cells, a.pol.obs_size, a.pol.n_actions);
        return TW_ERR_UNSUPPORTED;
    }
    if (a.env.n_cells <= BIG_NC) return launch_solve_big_nc<BIG_NC>(a, s);
    if (a.env.n_cells <= 36) return launch_solve_big_nc<36>(a, s);
    return launch_solve_big_nc<64>(a, s);
}

// obs ids of the padded trajectories -> the compact result (one wave per episode, grid-stride)
__global__ void __launch_bounds__(256) compact_obs16_kernel(const uint16_t *obs16, const uint32_t *ep_len, const uint64_t *ep_start, uint64_t E,
                                                            int t_pad, int n_cells, uint16_t *out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint64_t e = (uint64_t)blockIdx.x * 4 + wave; e < E; e += (uint64_t)gridDim.x * 4) {
        const uint64_t n = (uint64_t)ep_len[e] * (uint64_t)n_cells;
        const uint16_t *src = obs16 + e * (uint64_t)t_pad * (uint64_t)n_cells;
        uint16_t *dst = out + ep_start[e] * (uint64_t)n_cells;
        for (uint64_t i = lane; i < n; i += 64) dst[i] = src[i];
    }
}

template <int NC>
static int launch_rollout_big_nc(const RolloutArgs &a, uint16_t *obs16, hipStream_t s, uint32_t *blocks, uint32_t *threads)
{
    using Eng = EngineV<NC, false>;      // (the MFMAs as the intrinsic: tw_engine_generic.hpp, ASM_MFMA)
    const uint64_t nb = (a.num_episodes + Eng::EPB - 1) / Eng::EPB;
    if (nb == 0 || nb > 0x7fffffffull) { set_error("rollout: bad episode count %llu", (unsigned long long)a.num_episodes); return TW_ERR_INVALID; }
    const size_t lds_bytes = Eng::lds_floats(a.pol) * sizeof(float);
    if (lds_bytes > 159 * 1024) { set_error("rollout: %zu bytes of LDS needed, 159 KiB available", lds_bytes); return TW_ERR_UNSUPPORTED; }
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&rollout_big_kernel<NC>), lds_bytes)) return rc;
    hipLaunchKernelGGL((rollout_big_kernel<NC>), dim3((unsigned)nb), dim3(Eng::THREADS), lds_bytes, s, a, obs16);
    TW_HIP(hipGetLastError());
    if (blocks) *blocks = (uint32_t)nb;
    if (threads) *threads = Eng::THREADS;
    return TW_OK;
}

int launch_rollout_big(const RolloutArgs &a, uint16_t *obs16, hipStream_t s, uint32_t *blocks, uint32_t *threads)
{
    // host-side shape checks: everything the kernel indexes with is validated here
    if (a.env.n_cells <= 16 || a.env.n_cells > 64 || a.env.width * a.env.height != a.env.n_cells || !a.pol.generic ||
        a.pol.obs_size != a.env.n_cells * a.env.n_cells || a.pol.n_actions != 4 || (a.pol.n_perms > 0 && !a.pol.obs_perms16) ||
        a.out.t_pad < a.env.depth0 + 1 || !obs16 || a.queue || a.init_boards) {
        set_error("rollout (boards above 16 cells): unsupported shape (n_cells=%d obs_size=%d actions=%d generic=%d)", a.env.n_cells, a.pol.obs_size,
                  a.pol.n_actions, a.pol.generic);
        return TW_ERR_UNSUPPORTED;
    }
    if (a.env.n_cells <= BIG_NC) return launch_rollout_big_nc<BIG_NC>(a, obs16, s, blocks, threads);
    if (a.env.n_cells <= 36) return launch_rollout_big_nc<36>(a, obs16, s, blocks, threads);
    return launch_rollout_big_nc<64>(a, obs16, s, blocks, threads);
}

int launch_compact_obs16(const uint16_t *obs16, const uint32_t *ep_len, const uint64_t *ep_start, uint64_t E, int t_pad, int n_cells,
                         uint16_t *out, hipStream_t s)
{
    if (E == 0) return TW_OK;
    uint64_t blocks = (E + 3) / 4;
    if (blocks > 256ull * 16) blocks = 256ull * 16;
    hipLaunchKernelGGL(compact_obs16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, obs16, ep_len, ep_start, E, t_pad, n_cells, out);
    TW_HIP(hipGetLastError());
    return TW_OK;
}

}  // namespace tw
