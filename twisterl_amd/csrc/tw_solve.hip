// tw_solve.hip -- batched single_solve (reference rust/src/rl/solve.rs:17-71) on the shared MFMA
// policy engine: the inference loop behind `solve` (best of num_searches attempts, solve.rs:73-101)
// and `evaluate` (success rate / mean reward over episodes, rust/src/rl/evaluate.rs:22-89).
//
// One lane pair = one ATTEMPT (episode e, search a): while !is_final { total += reward; probs =
// Policy::predict (masked softmax, random twist; nn/policy.rs:34-49); action = argmax | weighted
// sample; step }.  The best-of-N selection and the episode means are reduced on the host from the
// per-attempt (success, total) pairs in the reference's serial order.
#include "tw_engine.hpp"

namespace tw {

template <int NT, int NC, int NW>
__global__ void __launch_bounds__((Geom<NT, NC, 0, NW>::WAVES * 64), (NW == 8 ? 2 : 1)) solve_f32_kernel(const SolveArgs a)
{
    using Eng = typename Geom<NT, NC, 0, NW>::Eng;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Eng eng;
    eng.begin1(a.pol, lds);

    const PuzzleConsts env = a.env;
    const int h = eng.h;
    const uint64_t att   = (uint64_t)blockIdx.x * Eng::EPB + (uint64_t)eng.ep_lane();   // attempt index
    const bool     valid = att < a.num_attempts;
    const bool     writer = h == 0 && eng.primary();
    const uint64_t ep    = a.episode_offset + att / a.num_searches;       // episode: keys the start state
    const uint64_t key   = ep * (uint64_t)a.num_searches + att % a.num_searches;   // keys this attempt's draws

    PuzzleLane st;
    st.board = env.ident; st.zx = 0; st.zy = 0; st.depth = 0;
    if (valid) {
        if (a.from_state) {           // solve(): every attempt clones the caller's env (solve.rs:85)
            st.board = a.start_board; st.zx = a.start_zx; st.zy = a.start_zy; st.depth = a.start_depth;
        } else {                      // evaluate(): env.reset() per episode (evaluate.rs:39,65)
            puzzle_reset(st, env, a.seed, ep);
        }
    }
    bool  alive = valid && !puzzle_final(st, env);
    float total = 0.0f;
    int   t = 0;

    eng.begin2();
    while (__syncthreads_or(alive ? 1 : 0)) {
        int perm = -1;
        if (eng.pol.n_perms > 0) {    // predict -> predict_with_perm -> get_perm_id (policy.rs:34-40,67-77)
            const u32x4 w = rng_draw(a.seed, key, (uint32_t)t, STREAM_PERM);
            perm = (int)u32_below(w.x, (uint32_t)eng.pol.n_perms);
        }
        int rowoff[NC];
        eng.rows_of(st.board, env.n_cells, perm, rowoff);
        float lg[4], value;
        eng.forward(rowoff, lg, value);
        eng.act_perm(perm, lg);
        float probs[4];
        masked_softmax4(lg, puzzle_maskbits(st, env), probs);        // policy.rs:43-47
        if (alive) {
            total = total + puzzle_reward(st, env);                   // solve.rs:31,34
            int action = 0;
            if (a.deterministic) {                                    // argmax (policy.rs:130-151)
                float bv = probs[0];
#pragma unroll
                for (int i = 1; i < 4; ++i) if (probs[i] > bv) { bv = probs[i]; action = i; }
            } else {                                                  // sample (policy.rs:153-167)
                const u32x4 w = rng_draw(a.seed, key, (uint32_t)t, STREAM_SOLVE);
                action = sample_weighted4(probs, 4, u32_to_unit(w.x));
            }
            if (a.actions && writer) a.actions[att * (uint64_t)a.t_pad + (uint64_t)t] = (uint8_t)action;
            puzzle_step(st, env, action);                             // solve.rs:56
            ++t;
            if (puzzle_final(st, env)) alive = false;
        }
    }
    if (valid && writer) {
        total = total + puzzle_reward(st, env);                       // solve.rs:65-66
        a.success[att] = puzzle_solved(st, env) ? 1.0f : 0.0f;        // solve.rs:68
        a.total[att]   = total;
        a.n_steps[att] = (uint32_t)t;
    }
    eng.end();
}

template <int NT, int NC, int NW>
static int launch_solve_geom(const SolveArgs &a, hipStream_t s)
{
    using G = Geom<NT, NC, 0, NW>;
    constexpr int EPB = G::Eng::EPB;
    const uint64_t nb = (a.num_attempts + EPB - 1) / EPB;
    if (nb == 0 || nb > 0x7fffffffull) { set_error("solve: bad attempt count %llu", (unsigned long long)a.num_attempts); return TW_ERR_INVALID; }
    const size_t lds_bytes = G::Eng::lds_floats(a.pol) * sizeof(float);
    if (lds_bytes > 159 * 1024) { set_error("solve: %zu bytes of LDS needed, 159 KiB available", lds_bytes); return TW_ERR_UNSUPPORTED; }
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&solve_f32_kernel<NT, NC, NW>), lds_bytes)) return rc;
    hipLaunchKernelGGL((solve_f32_kernel<NT, NC, NW>), dim3((unsigned)nb), dim3(64 * G::WAVES), lds_bytes, s, a);
    TW_HIP(hipGetLastError());
    return TW_OK;
}

template <int NT, int NC>
static int launch_solve_one(const SolveArgs &a, hipStream_t s)
{
    const int nw = geometry_for<NT>(a.num_attempts);
    if constexpr (NT >= 4) { if (nw == -16) return launch_solve_geom<NT, NC, -16>(a, s); }
    if constexpr (NT >= 4) { if (nw == -4) return launch_solve_geom<NT, NC, -4>(a, s); }
    else if constexpr (NT == 2) { if (nw == -2) return launch_solve_geom<NT, NC, -2>(a, s); }
    else {
        if (nw == 1) return launch_solve_geom<NT, NC, 1>(a, s);
        if (nw == 2) return launch_solve_geom<NT, NC, 2>(a, s);
    }
    return launch_solve_geom<NT, NC, 8>(a, s);
}

template <int NT>
static int launch_solve_nt(const SolveArgs &a, hipStream_t s)
{
    const int nc = a.env.n_cells;
    if (nc <= 4) return launch_solve_one<NT, 4>(a, s);
    if (nc <= 9) return launch_solve_one<NT, 9>(a, s);
    return launch_solve_one<NT, 16>(a, s);
}

int launch_solve_f32(const SolveArgs &a, hipStream_t s)
{
    if (a.pol.generic) {          // any Sequential depth: the vector-ALU engine (tw_engine_generic.hpp)
        if (a.env.n_cells < 1 || a.env.n_cells > 16 || a.pol.obs_size != a.env.n_cells * a.env.n_cells || a.pol.n_actions != 4 ||
            a.num_searches == 0 || !a.success || !a.total || !a.n_steps || (a.actions && a.t_pad < 1)) {
            set_error("solve: unsupported shape for a generic policy (n_cells=%d obs_size=%d actions=%d)", a.env.n_cells, a.pol.obs_size, a.pol.n_actions);
            return TW_ERR_UNSUPPORTED;
        }
        const int nc = a.env.n_cells;
        if (nc <= 4) return launch_solve_geom<0, 4, -65>(a, s);
        if (nc <= 9) return launch_solve_geom<0, 9, -65>(a, s);
        return launch_solve_geom<0, 16, -65>(a, s);
    }
    if (a.env.n_cells < 1 || a.env.n_cells > 16 || a.pol.obs_size != a.env.n_cells * a.env.n_cells ||
        a.pol.obs_size > 256 || a.pol.n_actions != 4 || a.pol.emb % 32 != 0 || a.pol.emb < 32 || a.num_searches == 0 ||
        !a.success || !a.total || !a.n_steps || (a.actions && a.t_pad < 1)) {
        set_error("solve: unsupported shape (n_cells=%d obs_size=%d actions=%d emb=%d hidden=%d searches=%u)",
                  a.env.n_cells, a.pol.obs_size, a.pol.n_actions, a.pol.emb, a.pol.hidden, a.num_searches);
        return TW_ERR_UNSUPPORTED;
    }
    switch (a.pol.hidden) {
        case 32:  return launch_solve_nt<1>(a, s);
        case 64:  return launch_solve_nt<2>(a, s);
        case 128: return launch_solve_nt<4>(a, s);
        case 256: return launch_solve_nt<8>(a, s);
        default:
            set_error("solve: hidden size %d not in {32,64,128,256}", a.pol.hidden);
            return TW_ERR_UNSUPPORTED;
    }
}

}  // namespace tw
