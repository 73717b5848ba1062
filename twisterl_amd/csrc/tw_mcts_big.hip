// tw_mcts_big.hip -- AlphaZero self-play for Puzzle boards of 17 .. 64 cells on the device.
//
// The same algorithm, RNG keys and arithmetic as tw_mcts.hip (AZCollector::single_collect, rust/src/collector/az.rs:51-109, over
// predict_probs_mcts, rust/src/rl/search.rs:104-189: ucb :29-39, backpropagate :45-53, expand :56-75, next :77-91, next_sample
// :94-100) -- one column of the policy engine = one episode, one batched Policy::full_predict per search step across the 16
// episodes of a workgroup, the per-episode tree in an HBM arena -- for boards the nibble packing of that kernel does not hold:
//   * the board is a Board5 / Board8 (tw_big_board.hpp); a node stores NO board -- the state follows the chosen actions down
//     the tree (a child's state IS step(parent state, action)), so a node stays 32 bytes whatever the board size;
//   * the policy is a generic one (obs_size > 256): EngineV<NC>, obs ids through the two-byte twist table;
//   * obs ids of the records go to their own array [E][t_pad][n_cells] (compact_obs16_kernel, as the PPO rollout of such boards);
//   * plain form of the search: back-propagation through the parent links, no stored outputs, no persistent lanes -- this path is
//     about being on the device at all (the host-stepped collectors of tw_env_generic.hip step the environment on the CPU).
// Bit-equal to the oracle's native collector.  MctsArgs::solve.on: MCTS-guided inference instead (single_solve over predict_probs_mcts,
// rust/src/rl/solve.rs:17-71 -- evaluate() and solve() with num_mcts_searches > 0), one column = one ATTEMPT, as in tw_mcts.hip.
#include "tw_engine_generic.hpp"
#include "tw_big_board.hpp"

namespace tw {

struct __attribute__((aligned(16))) BigNode {     // MCTSNode + Node<T> (search.rs:20-26, tree.rs:18-23) without the state
    float    value_sum;
    uint32_t visit;
    float    prior;
    uint32_t parent;       // 0xffffffff = None
    uint32_t child_base;   // children are contiguous in the arena (expand adds them together)
    uint32_t meta;         // n_children | action_taken << 8 (0xff = None) | depth << 16
    uint32_t pad[2];
};
static_assert(sizeof(BigNode) == 32, "BigNode must be 32 bytes");
size_t mcts_big_node_bytes() { return sizeof(BigNode); }

constexpr uint32_t BN_NONE = 0xffffffffu;
enum { BP_ROOT = 0, BP_LEAF = 1, BP_DONE = 2 };

template <int NC>
__global__ void __launch_bounds__(256, 1) mcts_big_kernel(const MctsArgs a, uint16_t *obs16)
{
    // (the MFMAs as the intrinsic: this kernel parks registers in AGPRs, and a reload the allocator puts right in front of an
    //  inline-asm MFMA comes without the wait states an MFMA needs after a VALU write -- tw_engine_generic.hpp, ASM_MFMA)
    using Eng = EngineV<NC, false>;
    using Board = typename BoardOf<NC>::T;
    using Lane = BigLaneT<Board>;
    constexpr int BW = (int)(sizeof(Board) / 4);                      // dwords of a board
    constexpr int MW = BW + 4;                                         // mirror: board | zx, zy (the forward's rows come from wave 0)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Eng eng;
    eng.begin1(a.pol, lds);
    const PuzzleConsts env = a.env;
    const int nc = env.n_cells;
    const Board ident = Board::ident(nc);
    const int j = eng.j;
    const uint64_t e_local = (uint64_t)blockIdx.x * Eng::EPB + (uint64_t)j;
    const bool valid = e_local < a.num_episodes;
    const bool owner = valid && eng.h == 0 && eng.owns_lane();       // the lane that walks / mutates this column's tree
    const MctsSolve sv = a.solve;
    // solve mode: column = ATTEMPT (episode, search); its draws are keyed like single_solve's (tw_solve.hip)
    const uint64_t sv_ep = sv.on ? a.episode_offset + e_local / sv.num_searches : 0;
    const uint64_t e_global = sv.on ? sv_ep * (uint64_t)sv.num_searches + e_local % sv.num_searches : a.episode_offset + e_local;
    BigNode *nodes = reinterpret_cast<BigNode *>(a.arena) + (valid ? e_local : 0) * (uint64_t)a.node_cap;
    const uint32_t S = a.num_searches, MED = a.max_expand_depth;
    const uint64_t rec_base = e_local * (uint64_t)a.out.t_pad;
    uint32_t *mir = reinterpret_cast<uint32_t *>(lds + Eng::lds_floats(a.pol)) + j * MW;     // the pending leaf of column j

    Lane st; st.board = ident; st.zx = 0; st.zy = 0; st.depth = 0;   // the episode's env (az.rs:56-57)
    if (owner && sv.on && sv.from_state) {                            // solve(): every attempt clones the caller's env (solve.rs:85)
        st.board = big_board_from_cells(sv.start_cells, nc, ident); st.zx = sv.start_zx; st.zy = sv.start_zy; st.depth = sv.start_depth;
    } else if (owner) {                                               // Env::reset (puzzle.rs:119-133)
        for (int d = 0; d < env.difficulty; ++d) {                    // (solve mode: env.reset() per episode, evaluate.rs:39,65)
            const u32x4 w = rng_draw(a.seed, sv.on ? sv_ep : e_global, (uint32_t)d, STREAM_SCRAMBLE);
            big_step(st, env, (int)u32_below(w.x, 4u));
        }
        st.depth = env.depth0;
    }
    Lane leaf = st, cur = st;                                         // state whose evaluation is pending | state of `node`
    int      phase = owner ? BP_ROOT : BP_DONE;
    if (sv.on && owner && big_final(st, ident)) phase = BP_DONE;      // `while !env.is_final()` (solve.rs:30)
    float    total = 0.0f;                                            // solve mode: summed rewards (solve.rs:25-34)
    uint32_t it = 0, expanded = 0, node = 0, n_nodes = 0;
    int      t = 0;
    float    value = 0.0f;
    unsigned long long evals = 0;

    const bool pub = eng.h == 0 && eng.owns_lane();                   // (columns without an episode publish the solved board)
    auto publish = [&]() {                                            // the pending leaf -> the lanes that feed the forward
        if (pub) {
            const uint32_t *bw = reinterpret_cast<const uint32_t *>(&leaf.board);
#pragma unroll
            for (int k = 0; k < BW; ++k) mir[k] = bw[k];
            mir[BW] = (uint32_t)leaf.zx; mir[BW + 1] = (uint32_t)leaf.zy;
        }
    };
    publish();
    eng.begin2();

    for (;;) {
        if (!__syncthreads_or(phase != BP_DONE ? 1 : 0)) break;       // (the barrier publishes the leaves)
        // ---- Policy::full_predict of the pending leaves (policy.rs:102-126); the engine takes a column's rows from wave 0 ----
        Board lb = ident;
        {
            uint32_t *bw = reinterpret_cast<uint32_t *>(&lb);
#pragma unroll
            for (int k = 0; k < BW; ++k) bw[k] = mir[k];
        }
        float lsum[4] = {0.0f, 0.0f, 0.0f, 0.0f}, vsum = 0.0f;
        const int n_pass = eng.pol.n_perms > 0 ? eng.pol.n_perms : 1;
        const float np = (float)eng.pol.n_perms;
        for (int pass = 0; pass < n_pass; ++pass) {
            const int perm = eng.pol.n_perms > 0 ? pass : -1;
            int rowoff[NC];
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                int row = -1;
                if (i < nc) {
                    const int id = i * nc + (int)lb.cell(i);
                    row = perm >= 0 ? (int)eng.pol.obs_perms16[(size_t)perm * eng.pol.obs_size + id] : id;
                }
                rowoff[i] = row;
            }
            float lg[4], v;
            eng.forward(rowoff, lg, v);
            eng.act_perm(perm, lg);
            if (eng.pol.n_perms > 0) {
                vsum = vsum + v / np;                                            // policy.rs:111
#pragma unroll
                for (int i = 0; i < 4; ++i) lsum[i] = lsum[i] + lg[i] / np;      // policy.rs:112-114
            } else {
                vsum = v;
#pragma unroll
                for (int i = 0; i < 4; ++i) lsum[i] = lg[i];
            }
        }

        // ---- per-episode tree work on the owner lane -------------------------------------------------------------------------
        if (owner && phase != BP_DONE) {
            float probs[4];
            masked_softmax4(lsum, big_maskbits(leaf, env), probs);
            const float nn_value = vsum;
            ++evals;
            float pri[4] = {0.0f, 0.0f, 0.0f, 0.0f};                  // priors of the children just created, in child order
            uint32_t acts = 0;                                        // ... and their actions, 2 bits each
            // expand (search.rs:56-75): one child per action with prior > 0
            auto expand = [&](uint32_t idx, const Lane &s) -> uint32_t {
                uint32_t cnt = 0;
                const int cdepth = s.depth > 0 ? s.depth - 1 : 0;     // (Env::step: depth.saturating_sub(1))
                acts = 0;
#pragma unroll
                for (int act = 0; act < 4; ++act) {
                    if (!(probs[act] > 0.0f)) continue;
                    if (cnt == 0) pri[0] = probs[act]; else if (cnt == 1) pri[1] = probs[act];
                    else if (cnt == 2) pri[2] = probs[act]; else pri[3] = probs[act];
                    acts |= (uint32_t)act << (2 * cnt);
                    BigNode nn;
                    nn.value_sum = 0.0f; nn.visit = 0; nn.prior = probs[act]; nn.parent = idx; nn.child_base = 0;
                    nn.meta = 0u | ((uint32_t)act << 8) | ((uint32_t)cdepth << 16);
                    nn.pad[0] = 0; nn.pad[1] = 0;
                    nodes[n_nodes + cnt] = nn;
                    ++cnt;
                }
                nodes[idx].child_base = n_nodes;
                nodes[idx].meta = (nodes[idx].meta & ~0xffu) | cnt;
                n_nodes += cnt;
                return cnt;
            };
            // backpropagate (search.rs:45-53): value_sum += v, visit_count += 1 from the node up to the root
            auto backprop = [&](uint32_t idx, float val) {
                while (idx != BN_NONE) {
                    const BigNode n = nodes[idx];
                    uint2 w; w.x = __float_as_uint(n.value_sum + val); w.y = n.visit + 1u;
                    *reinterpret_cast<uint2 *>(&nodes[idx].value_sum) = w;
                    idx = n.parent;
                }
            };
            // next_sample (search.rs:94-100) among the children just created; the state follows
            auto sample_child = [&](uint32_t cb, uint32_t nch, Lane &s) {
                const u32x4 w = rng_draw(a.seed, e_global, it * MED + expanded, STREAM_MCTS | ((uint32_t)t << 8));
                const int k = sample_weighted4(pri, (int)nch, u32_to_unit(w.x));
                node = cb + (uint32_t)k;
                big_step(s, env, (int)((acts >> (2 * k)) & 3u));
            };

            if (phase == BP_ROOT) {
                // root node (search.rs:120-129): visit_count 1, expanded with the root priors
                BigNode r;
                r.value_sum = 0.0f; r.visit = 1; r.prior = 0.0f; r.parent = BN_NONE; r.child_base = 0;
                r.meta = 0u | (0xffu << 8) | ((uint32_t)st.depth << 16); r.pad[0] = 0; r.pad[1] = 0;
                nodes[0] = r; n_nodes = 1;
                expand(0u, st);
                it = 0;
            } else {
                // the leaf just evaluated (search.rs:154-159): expand, sample a child by the priors
                const uint32_t cb = n_nodes;
                const uint32_t nch = expand(node, leaf);
                cur = leaf;
                if (nch > 0) sample_child(cb, nch, cur);
                value = nn_value;
                ++expanded;
            }
            bool resume = (phase == BP_LEAF);
            for (;;) {
                if (!resume) {
                    if (it == S) {
                        // ---- move finished: visit counts -> probs (search.rs:166-188) -----------------------------------------
                        float mp[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                        const BigNode root = nodes[0];
                        const uint32_t rnc = root.meta & 0xffu;
                        for (uint32_t c = 0; c < rnc; ++c) {
                            const BigNode ch = nodes[root.child_base + c];
                            const int act = (int)((ch.meta >> 8) & 3u);
                            const float vis = (float)ch.visit;
                            mp[0] = act == 0 ? vis : mp[0]; mp[1] = act == 1 ? vis : mp[1];
                            mp[2] = act == 2 ? vis : mp[2]; mp[3] = act == 3 ? vis : mp[3];
                        }
                        float sum = 0.0f;
#pragma unroll
                        for (int i = 0; i < 4; ++i) sum = sum + mp[i];
                        if (sum > 0.0f) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) mp[i] = mp[i] / sum;
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i) mp[i] = 1.0f / 4.0f;
                        }
                        if (sv.on) {
                            // solve.rs:31-58: total += reward; action = argmax | sample of the MCTS probs; step
                            total = total + big_reward(st, ident, env);
                            int action = 0;
                            if (sv.deterministic) {
                                float bv = mp[0];
#pragma unroll
                                for (int i = 1; i < 4; ++i) if (mp[i] > bv) { bv = mp[i]; action = i; }
                            } else {
                                const u32x4 w = rng_draw(a.seed, e_global, (uint32_t)t, STREAM_SOLVE);
                                action = sample_weighted4(mp, 4, u32_to_unit(w.x));
                            }
                            if (sv.actions) sv.actions[e_local * (uint64_t)sv.act_pad + (uint64_t)t] = (uint8_t)action;
                            big_step(st, env, action);
                            ++t;
                            if (big_final(st, ident)) { phase = BP_DONE; break; }
                            phase = BP_ROOT; leaf = st;
                            break;
                        }
                        // az.rs:72-81: action = sample(mcts_probs); val = env.reward(); store the record
                        const u32x4 w = rng_draw(a.seed, e_global, (uint32_t)t, STREAM_AZ_ACT);
                        const int action = sample_weighted4(mp, 4, u32_to_unit(w.x));
                        const uint64_t rec = rec_base + (uint64_t)t;
                        const uint32_t zero4[4] = {0u, 0u, 0u, 0u};
                        store_rec(a.out.rec + rec, zero4, mp, 0.0f, big_reward(st, ident, env), 0, -1);
                        uint16_t *o = obs16 + rec * (uint64_t)nc;
#pragma unroll
                        for (int i = 0; i < NC; ++i) if (i < nc) o[i] = (uint16_t)(i * nc + (int)st.board.cell(i));
                        if (big_final(st, ident)) {                                                      // az.rs:84
                            a.out.ep_len[e_local] = (uint32_t)t + 1u;
                            phase = BP_DONE;
                            break;
                        }
                        big_step(st, env, action);                                                       // az.rs:89
                        ++t;
                        phase = BP_ROOT; leaf = st;
                        break;
                    }
                    // ---- descend to a leaf by UCB (search.rs:133-138, next :77-91, ucb :29-39); the state follows the actions ----
                    node = 0; cur = st;
                    BigNode cn = nodes[0];
                    for (;;) {
                        const uint32_t nch = cn.meta & 0xffu, cb = cn.child_base;
                        if (nch == 0) break;
                        uint32_t best = BN_NONE; float best_ucb = -__builtin_inff();
                        BigNode bestn = cn;
                        const float sq = sqrtf((float)cn.visit);
                        BigNode chs[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) chs[c] = nodes[cb + ((uint32_t)c < nch ? (uint32_t)c : nch - 1u)];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const BigNode &ch = chs[c];
                            const float q = ch.visit == 0 ? 0.0f : ch.value_sum / (float)ch.visit;
                            float d = sq / ((float)ch.visit + 1.0f);
                            d = a.C * d;
                            d = d * ch.prior;
                            const float u = q + d;
                            if ((uint32_t)c < nch && u > best_ucb) { best = cb + (uint32_t)c; best_ucb = u; bestn = ch; }
                        }
                        if (best == BN_NONE) break;                   // all-NaN UCB: the reference panics here
                        node = best; cn = bestn;
                        big_step(cur, env, (int)((bestn.meta >> 8) & 3u));
                    }
                    value = 0.0f; expanded = 0;
                }
                resume = false;
                // leaf phase (search.rs:143-160)
                bool need_nn = false;
                while (expanded < MED) {
                    value = big_reward(cur, ident, env);                              // :146
                    if (big_final(cur, ident)) break;                                 // :149
                    phase = BP_LEAF; leaf = cur; need_nn = true;                      // :154 needs the network
                    break;
                }
                if (need_nn) break;
                backprop(node, value);                                                // :163
                ++it;
            }
        }
        publish();                                                    // (the mirror was read before the forward's first barrier)
    }
    if (owner) {
        if (sv.on) {
            total = total + big_reward(st, ident, env);                   // solve.rs:65-66
            sv.success[e_local] = st.board == ident ? 1.0f : 0.0f;        // solve.rs:68
            sv.total[e_local]   = total;
            sv.n_steps[e_local] = (uint32_t)t;
        }
        atomicAdd(a.eval_count, evals);
    }
    eng.end();
}

template <int NC>
static int launch_mcts_big_nc(const MctsArgs &a, uint16_t *obs16, hipStream_t s, uint32_t *blocks, uint32_t *threads)
{
    using Eng = EngineV<NC, false>;
    using Board = typename BoardOf<NC>::T;
    const uint64_t nb = (a.num_episodes + Eng::EPB - 1) / Eng::EPB;
    if (nb == 0 || nb > 0x7fffffffull) { set_error("mcts (boards above 16 cells): bad episode count %llu", (unsigned long long)a.num_episodes); return TW_ERR_INVALID; }
    const size_t lds_bytes = (Eng::lds_floats(a.pol) + (size_t)Eng::EPB * (sizeof(Board) / 4 + 4)) * sizeof(float);
    if (lds_bytes > 159 * 1024) { set_error("mcts: %zu bytes of LDS needed, 159 KiB available", lds_bytes); return TW_ERR_UNSUPPORTED; }
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&mcts_big_kernel<NC>), lds_bytes)) return rc;
    hipLaunchKernelGGL((mcts_big_kernel<NC>), dim3((unsigned)nb), dim3(Eng::THREADS), lds_bytes, s, a, obs16);
    TW_HIP(hipGetLastError());
    if (blocks) *blocks = (uint32_t)nb;
    if (threads) *threads = Eng::THREADS;
    return TW_OK;
}

int launch_mcts_big(const MctsArgs &a, uint16_t *obs16, hipStream_t s, uint32_t *blocks, uint32_t *threads)
{
    const uint64_t need = 5ull + 4ull * a.num_searches * (a.max_expand_depth ? a.max_expand_depth : 1u);
    if (a.env.n_cells <= 16 || a.env.n_cells > 64 || a.env.width * a.env.height != a.env.n_cells || !a.pol.generic ||
        a.pol.obs_size != a.env.n_cells * a.env.n_cells || a.pol.n_actions != 4 || (a.pol.n_perms > 0 && !a.pol.obs_perms16) ||
        (!a.solve.on && (a.out.t_pad < a.env.depth0 + 1 || !obs16)) || !a.arena || !a.eval_count || a.node_cap < need || a.queue || a.init_boards ||
        (a.solve.on && (!a.solve.success || !a.solve.total || !a.solve.n_steps || a.solve.num_searches == 0 || (a.solve.from_state && !a.solve.start_cells)))) {
        set_error("mcts (boards above 16 cells): unsupported shape (n_cells=%d obs_size=%d actions=%d generic=%d node_cap=%u need=%llu)", a.env.n_cells,
                  a.pol.obs_size, a.pol.n_actions, a.pol.generic, a.node_cap, (unsigned long long)need);
        return TW_ERR_UNSUPPORTED;
    }
    if (a.env.n_cells <= BIG_NC) return launch_mcts_big_nc<BIG_NC>(a, obs16, s, blocks, threads);
    if (a.env.n_cells <= 36) return launch_mcts_big_nc<36>(a, obs16, s, blocks, threads);
    return launch_mcts_big_nc<64>(a, obs16, s, blocks, threads);
}

}  // namespace tw
