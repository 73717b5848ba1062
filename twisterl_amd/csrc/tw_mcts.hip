// tw_mcts.hip -- AlphaZero self-play kernel: MCTS over per-episode trees with the leaf evaluations
// of all episodes batched on the shared MFMA policy engine (gfx950).
//
// Replaces AZCollector::single_collect (reference rust/src/collector/az.rs:51-109) and
// predict_probs_mcts (rust/src/rl/search.rs:104-189: MCTSNode::ucb :29-39, backpropagate :45-53,
// expand :56-75, next :77-91, next_sample :94-100) over the arena tree of rust/src/rl/tree.rs.
//
// Mapping: one lane pair = one episode, exactly as in the rollout kernel.  Every episode owns an
// arena of 32-byte nodes in HBM.  The workgroup alternates between
//   (1) a per-lane phase: tree walk (UCB descent), expansion of the node evaluated last, weighted
//       child sampling, back-propagation -- repeated until the lane reaches a leaf that needs the
//       network (leaves that are final states need none and are consumed on the spot), or the
//       search budget of the move is spent (then: visit counts -> probs, action, record, env.step);
//   (2) ONE collective policy evaluation (Policy::full_predict, nn/policy.rs:102-126) of the 32
//       leaves of each wave on the MFMA engine, i.e. the reference's one-forward-per-search becomes
//       one batched forward per search step across all resident episodes.
// All arithmetic follows the numeric spec of DESIGN.md, so MCTS probs are bit-equal to the oracle.
#include "tw_engine.hpp"

namespace tw {

struct __attribute__((aligned(16))) MctsNode {   // MCTSNode + Node<T> (search.rs:20-26, tree.rs:18-23)
    uint64_t board;        // state (nibble-packed); the blank position is recovered from it
    float    value_sum;
    uint32_t visit;
    float    prior;
    uint32_t parent;       // 0xffffffff = None
    uint32_t child_base;   // children are contiguous in the arena (expand adds them together)
    uint8_t  n_children;
    uint8_t  action;       // action_taken (0xff = None); | ACT_UNDO: this move takes the parent's move back
    uint16_t depth;
};
static_assert(sizeof(MctsNode) == 32, "MctsNode must be 32 bytes");
// An episode's arena: MctsNode nodes[node_cap] | uint4 outs[node_cap][2] -- the network output every expanded node was expanded
// with (probs[4] | value).  A node whose move takes its parent's move back holds its GRANDPARENT's board, and the output is a
// function of the board alone: it takes the grandparent's stored output instead of asking for a forward (same bits as the
// evaluation the reference repeats; see tw_mcts_deep.hip).
constexpr uint8_t ACT_UNDO = 4;
size_t mcts_node_bytes() { return sizeof(MctsNode) + 32; }

constexpr uint32_t NONE = 0xffffffffu;
constexpr int PATH_DEPTH = 8;     // levels of the search path kept in LDS per episode (deeper paths fall back to parent chasing)
#ifdef TW_ABLATE   // diagnostic build: per-wave cycle accounting (forward | tree phase | loop trips | searches consumed | max trips)
__device__ unsigned long long g_mcts_stamps[24];
#endif
enum { PH_ROOT = 0, PH_LEAF = 1, PH_DONE = 2 };

__device__ inline PuzzleLane lane_of(const MctsNode &n, const PuzzleConsts &c)
{
    PuzzleLane s; s.board = n.board; const int z = blank_cell(n.board);
    s.zx = z % c.width; s.zy = z / c.width; s.depth = n.depth;
    return s;
}

template <int NT, int NC, int NW, bool PERSIST = false>
__global__ void __launch_bounds__((Geom<NT, NC, 0, NW>::WAVES * 64), (NW == 8 ? 2 : 1)) mcts_f32_kernel(const MctsArgs a)
{
    using Eng = typename Geom<NT, NC, 0, NW>::Eng;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Eng eng;
    eng.begin1(a.pol, lds);

    const PuzzleConsts env = a.env;
    const int j = eng.j, h = eng.h;
    const uint64_t slot     = (uint64_t)blockIdx.x * Eng::EPB + (uint64_t)eng.ep_lane();           // lane pair (arena) index
    uint64_t       e_local  = slot;                                                                // episode being played
    const bool     valid    = e_local < a.num_episodes;
    const MctsSolve sv      = a.solve;
    // solve mode: lane pair = ATTEMPT (episode, search); its draws are keyed like single_solve's (tw_solve.hip)
    const uint64_t sv_ep    = sv.on ? a.episode_offset + e_local / sv.num_searches : 0;
    uint64_t       e_global = sv.on ? sv_ep * (uint64_t)sv.num_searches + e_local % sv.num_searches : a.episode_offset + e_local;
    const bool     owner    = valid && h == 0 && eng.owns_lane();   // the lane that walks / mutates the tree
    MctsNode *nodes = reinterpret_cast<MctsNode *>(reinterpret_cast<uint8_t *>(a.arena) + (valid ? slot : 0) * (uint64_t)a.node_cap * (sizeof(MctsNode) + 32));
    uint4    *outs  = reinterpret_cast<uint4 *>(nodes + a.node_cap);
    const uint32_t S = a.num_searches, MED = a.max_expand_depth;
    uint64_t rec_base = e_local * (uint64_t)a.out.t_pad;

    PuzzleLane st;                                        // the episode's env (az.rs:56-57)
    st.board = env.ident; st.zx = 0; st.zy = 0; st.depth = 0;
    auto take = [&](uint64_t e) {                         // persistent mode: start board of episode e from the pre-pass
        st.board = a.init_boards[e];
        const int z = blank_cell(st.board);
        st.zx = z % env.width; st.zy = z / env.width; st.depth = env.depth0;
    };
    if (valid) {
        if constexpr (PERSIST) take(e_local);
        else if (!sv.on) puzzle_reset(st, env, a.seed, e_global);
        else if (sv.from_state) { st.board = sv.start_board; st.zx = sv.start_zx; st.zy = sv.start_zy; st.depth = sv.start_depth; }
        else puzzle_reset(st, env, a.seed, sv_ep);
    }
    float total = 0.0f;                                   // solve mode: summed rewards (solve.rs:25-34)

    // per-episode search state (meaningful on the owner lane; phase/leaf are mirrored to the partner)
    int      phase = valid ? PH_ROOT : PH_DONE;
    if (sv.on && valid && puzzle_final(st, env)) phase = PH_DONE;      // `while !env.is_final()` (solve.rs:30)
    uint32_t it = 0, expanded = 0, node = 0, n_nodes = 0;
    int      t = 0;
    uint32_t len = 0;
    float    value = 0.0f;
    PuzzleLane leaf = st;                                 // state whose evaluation is pending
    unsigned long long evals = 0, reused = 0;
    bool more = PERSIST;                                  // the episode queue may still hold work

    uint32_t obs_base[4];
    obs_base_words(env.n_cells, obs_base);

    // The search path (root .. current node) with the value_sum / visit_count read on the way down, per owner lane in LDS
    // ([level][episode], three planes): back-propagation then needs no reads at all -- it stores value_sum + v and visit + 1
    // for every level at once instead of chasing parent pointers through HBM (one dependent round trip per level).
    float    *p_vs  = lds + Eng::lds_floats(a.pol) + eng.ep_lane();
    uint32_t *p_idx = reinterpret_cast<uint32_t *>(p_vs + PATH_DEPTH * Eng::EPB);
    uint32_t *p_vis = p_idx + PATH_DEPTH * Eng::EPB;
    int      plen = 0;
    bool     overflow = false;
    float    root_vs = 0.0f;                              // the root's record lives in registers between searches
    uint32_t root_visit = 0, root_cb = 0, root_nc = 0;
    auto st_stats = [&](uint32_t i, float vs, uint32_t vis) {          // value_sum, visit_count: one 8-byte store
        uint2 w; w.x = __float_as_uint(vs); w.y = vis;
        *reinterpret_cast<uint2 *>(&nodes[i].value_sum) = w;
    };
    auto push = [&](uint32_t idx, float vs, uint32_t vis) {
        if (plen < PATH_DEPTH) { p_idx[plen * Eng::EPB] = idx; p_vs[plen * Eng::EPB] = vs; p_vis[plen * Eng::EPB] = vis; ++plen; }
        else overflow = true;
    };

    eng.begin2();

#ifdef TW_ABLATE
    unsigned long long c_fwd = 0, c_tree = 0, c_trips = 0, c_inner = 0, c_desc = 0, c_pre = 0, c_dsc = 0, c_bp = 0, c_wl = 0;
#define TW_MS(var) const unsigned long long var = __builtin_readcyclecounter()
#define TW_MA(acc, a, b) acc += (b) - (a)
#else
#define TW_MS(var)
#define TW_MA(acc, a, b)
#endif
#ifdef TW_ABLATE
    unsigned long long x_or = 0, x_rows = 0, x_eng = 0, x_soft = 0, x_own = 0, x_mir = 0;
#endif
    for (;;) {
        TW_MS(z0);
        if (!__syncthreads_or(phase != PH_DONE ? 1 : 0)) break;
#ifdef TW_ABLATE
        const unsigned long long s0 = __builtin_readcyclecounter();
        x_or += s0 - z0;
#endif
        // ---- (2) Policy::full_predict of the pending leaf (policy.rs:102-126) ------------------
        float lsum[4] = {0.0f, 0.0f, 0.0f, 0.0f}, vsum = 0.0f;
        const int n_pass = eng.pol.n_perms > 0 ? eng.pol.n_perms : 1;
        const float np = (float)eng.pol.n_perms;
        for (int pass = 0; pass < n_pass; ++pass) {
            const int perm = eng.pol.n_perms > 0 ? pass : -1;
            int rowoff[NC];
            TW_MS(z1);
            eng.rows_of(leaf.board, env.n_cells, perm, rowoff);
            float lg[4], v;
            TW_MS(z2);
            eng.forward(rowoff, lg, v);
            TW_MS(z3);
            TW_MA(x_rows, z1, z2); TW_MA(x_eng, z2, z3);
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
        float probs[4];
        masked_softmax4(lsum, puzzle_maskbits(leaf, env), probs);
        const float nn_value = vsum;

#ifdef TW_ABLATE
        const unsigned long long s1 = __builtin_readcyclecounter();
        c_fwd += s1 - s0; ++c_trips;
#endif
        TW_MS(z4);
        // ---- (1) per-episode tree work on the owner lane ---------------------------------------
        if (owner && phase != PH_DONE) {
            ++evals;
            // expand (search.rs:56-75): one child per action with prior > 0, state = clone + step
            float pri[4] = {0.0f, 0.0f, 0.0f, 0.0f};        // priors of the children just created, in child order
            auto expand = [&](uint32_t idx, const PuzzleLane &s, uint32_t pact) -> uint32_t {
                uint32_t cnt = 0;
#pragma unroll
                for (int act = 0; act < 4; ++act) {
                    if (!(probs[act] > 0.0f)) continue;
                    // (cnt <= act: a compile-time chain of selects instead of a dynamically indexed register array)
                    if (cnt == 0) pri[0] = probs[act]; else if (cnt == 1) pri[1] = probs[act];
                    else if (cnt == 2) pri[2] = probs[act]; else pri[3] = probs[act];
                    PuzzleLane c = s;
                    puzzle_step(c, env, act);
                    MctsNode nn;
                    nn.board = c.board; nn.value_sum = 0.0f; nn.visit = 0; nn.prior = probs[act];
                    nn.parent = idx; nn.child_base = 0; nn.n_children = 0;
                    nn.action = (uint8_t)(act | ((idx != 0u && (uint32_t)act == ((pact & 3u) ^ 2u)) ? ACT_UNDO : 0));   // 0 left, 1 up, 2 right, 3 down
                    nn.depth = (uint16_t)c.depth;
                    nodes[n_nodes + cnt] = nn;
                    ++cnt;
                }
                nodes[idx].child_base = n_nodes; nodes[idx].n_children = (uint8_t)cnt;
                n_nodes += cnt;
                return cnt;
            };
            auto keep_output = [&](uint32_t idx, float v) {       // the output node idx is expanded with (probs[], v)
                outs[2 * idx] = make_uint4(__float_as_uint(probs[0]), __float_as_uint(probs[1]), __float_as_uint(probs[2]), __float_as_uint(probs[3]));
                outs[2 * idx + 1] = make_uint4(__float_as_uint(v), 0u, 0u, 0u);
            };
            // backpropagate (search.rs:45-53): value_sum += v, visit_count += 1 on every node of the path
            auto backprop = [&](uint32_t idx, float val) {
                if (!overflow) {
                    for (int l = 0; l < plen; ++l)
                        st_stats(p_idx[l * Eng::EPB], p_vs[l * Eng::EPB] + val, p_vis[l * Eng::EPB] + 1u);
                } else {
                    while (idx != NONE) {
                        const MctsNode n = nodes[idx];
                        st_stats(idx, n.value_sum + val, n.visit + 1u);
                        idx = n.parent;
                    }
                }
                root_vs = root_vs + val; root_visit += 1u;
            };

            TW_MS(m_pre0);
            if (phase == PH_ROOT) {
                // root node (search.rs:120-129): visit_count 1, then expand with the root priors
                MctsNode r;
                r.board = st.board; r.value_sum = 0.0f; r.visit = 1; r.prior = 0.0f; r.parent = NONE;
                r.child_base = 0; r.n_children = 0; r.action = 0xff; r.depth = (uint16_t)st.depth;
                nodes[0] = r; n_nodes = 1;
                root_vs = 0.0f; root_visit = 1u; root_cb = 1u;
                root_nc = expand(0, st, 0xffu);
                keep_output(0u, nn_value);
                it = 0;
            } else {
                // the leaf just evaluated (search.rs:154-159): expand, sample a child by the priors (the children were
                // written a moment ago: their priors are still in registers)
                const uint32_t cb = n_nodes;
                const uint32_t nch = expand(node, leaf, (uint32_t)nodes[node].action);
                keep_output(node, nn_value);
                if (nch > 0) {
                    const u32x4 w = rng_draw(a.seed, e_global, it * MED + expanded, STREAM_MCTS | ((uint32_t)t << 8));
                    node = cb + (uint32_t)sample_weighted4(pri, (int)nch, u32_to_unit(w.x));
                    push(node, 0.0f, 0u);
                }
                value = nn_value;
                ++expanded;
            }

            TW_MS(m_pre1);
            TW_MA(c_pre, m_pre0, m_pre1);
            // run the search loop until a leaf needs the network or the move is finished
            bool resume_expand = (phase == PH_LEAF);
            MctsNode cur; bool have_cur = false;      // record of `node` when it was just read by the descent
            cur.board = 0; cur.value_sum = 0.0f; cur.visit = 0; cur.prior = 0.0f; cur.parent = NONE; cur.child_base = 0;
            cur.n_children = 0; cur.action = 0; cur.depth = 0;
            for (;;) {
#ifdef TW_ABLATE
                ++c_inner;
#endif
                if (!resume_expand) {
                    if (it == S) {
                        // ---- move finished: visit counts -> probs (search.rs:166-188) --------------
                        float mp[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                        if (root_nc > 0) {
                            MctsNode chs[4];
#pragma unroll
                            for (int c = 0; c < 4; ++c) chs[c] = nodes[root_cb + ((uint32_t)c < root_nc ? (uint32_t)c : root_nc - 1u)];
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                if ((uint32_t)c >= root_nc) continue;
                                const int act = chs[c].action & 3; // (a chain of selects instead of a dynamically indexed register array)
                                const float vis = (float)chs[c].visit;
                                mp[0] = act == 0 ? vis : mp[0]; mp[1] = act == 1 ? vis : mp[1];
                                mp[2] = act == 2 ? vis : mp[2]; mp[3] = act == 3 ? vis : mp[3];
                            }
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
                            total = total + puzzle_reward(st, env);
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
                            puzzle_step(st, env, action);
                            ++t;
                            if (puzzle_final(st, env)) { phase = PH_DONE; break; }
                            phase = PH_ROOT; leaf = st;
                            break;
                        }
                        // az.rs:72-81: action = sample(mcts_probs); val = env.reward(); store record
                        const u32x4 w = rng_draw(a.seed, e_global, (uint32_t)t, STREAM_AZ_ACT);
                        const int action = sample_weighted4(mp, 4, u32_to_unit(w.x));
                        const uint64_t rec = rec_base + (uint64_t)t;
                        uint32_t pk[4];
                        obs_bytes(st.board, obs_base, pk);
                        store_rec(a.out.rec + rec, pk, mp, 0.0f, puzzle_reward(st, env), 0, -1);
                        if (puzzle_final(st, env)) {                                                     // az.rs:84
                            phase = PH_DONE; len = (uint32_t)t + 1u;
                            if constexpr (PERSIST) {       // episode over: record its length, take the next one off the queue
                                a.out.ep_len[e_local] = len;
                                const unsigned got = more ? atomicAdd(a.queue, 1u) : 0xffffffffu;
                                if ((uint64_t)got < a.num_episodes) {
                                    e_local = got; e_global = a.episode_offset + e_local; rec_base = e_local * (uint64_t)a.out.t_pad;
                                    take(e_local);
                                    t = 0; phase = PH_ROOT; leaf = st;
                                } else more = false;
                            }
                            break;
                        }
                        puzzle_step(st, env, action);                                                   // az.rs:89
                        ++t;
                        phase = PH_ROOT; leaf = st;
                        break;
                    }
                    // descend to a leaf by UCB (search.rs:133-138, next :77-91, ucb :29-39).  The chosen child's record is
                    // kept in registers: one dependent HBM round trip per level (its children) instead of two
                    TW_MS(m_d0);
                    node = 0;
                    cur.board = st.board; cur.value_sum = root_vs; cur.visit = root_visit; cur.prior = 0.0f; cur.parent = NONE;
                    cur.child_base = root_cb; cur.n_children = (uint8_t)root_nc; cur.action = 0xff; cur.depth = (uint16_t)st.depth;
                    plen = 0; overflow = false;
                    push(0u, root_vs, root_visit);
                    for (;;) {
                        if (cur.n_children == 0) break;
                        uint32_t best = NONE; float best_ucb = -__builtin_inff();
                        MctsNode bestn = cur;
                        const float sq = sqrtf((float)cur.visit);
                        // all (up to four, contiguous) children are fetched at once: ONE dependent round trip per level, not one
                        // per child (slots past n_children re-read the last child and are ignored)
                        const uint32_t nch = cur.n_children, cb = cur.child_base;
                        MctsNode chs[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) chs[c] = nodes[cb + ((uint32_t)c < nch ? (uint32_t)c : nch - 1u)];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const MctsNode &ch = chs[c];
                            const float q = ch.visit == 0 ? 0.0f : ch.value_sum / (float)ch.visit;
                            float d = sq / ((float)ch.visit + 1.0f);
                            d = a.C * d;
                            d = d * ch.prior;
                            const float u = q + d;
                            if ((uint32_t)c < nch && u > best_ucb) { best = cb + (uint32_t)c; best_ucb = u; bestn = ch; }
                        }
                        if (best == NONE) break;        // all-NaN UCB: the reference panics here
                        node = best; cur = bestn;
                        push(best, bestn.value_sum, bestn.visit);
#ifdef TW_ABLATE
                        ++c_desc;
#endif
                    }
                    value = 0.0f; expanded = 0;
                    have_cur = true;
                    TW_MS(m_d1);
                    TW_MA(c_dsc, m_d0, m_d1);
                }
                resume_expand = false;
                // leaf phase (search.rs:143-160)
                bool need_nn = false;
                while (expanded < MED) {
                    const MctsNode n = have_cur ? cur : nodes[node];
                    have_cur = false;
                    const PuzzleLane s = lane_of(n, env);
                    value = puzzle_reward(s, env);                                   // :146
                    if (puzzle_final(s, env)) break;                                 // :149
                    // a node whose move took its parent's move back holds the board of its grandparent, whose output is stored:
                    // expand with it, sample a child, go on (search.rs:154-159).  The grandparent is level plen-3 of the search path
                    // (level plen-1 is `node`) while the path fits its PATH_DEPTH levels; below that `push` stops recording -- plen
                    // stays at PATH_DEPTH and level plen-3 is an ancestor further up -- so it is found through the parent links.
                    // a.reuse_mode (TW_OPT_AZ_REUSE, diagnostic): 0 as described, 1 no reuse, 2 the path level whatever the depth
                    // (round 2's first form: WRONG below PATH_DEPTH levels, kept to show it), 3 parent links only, 4 counts how the
                    // two ways of finding the grandparent disagree (eval_count[3..11]) and goes by the parent links
                    uint32_t g = NONE;
                    if (n.action != 0xffu && (n.action & ACT_UNDO) && a.reuse_mode != 1u) {
                        uint32_t g_path = NONE, g_link = NONE;
                        const bool deep_path = overflow;
                        if (a.reuse_mode == 2u || a.reuse_mode == 4u || (a.reuse_mode == 0u && !deep_path))
                            if (plen >= 3) g_path = p_idx[(plen - 3) * Eng::EPB];
                        if (a.reuse_mode == 3u || a.reuse_mode == 4u || (a.reuse_mode == 0u && deep_path))
                            if (n.parent != NONE && n.parent != 0u) g_link = nodes[n.parent].parent;
                        if (a.reuse_mode == 4u) {
                            atomicAdd(a.eval_count + 3, 1ull);
                            if (deep_path) atomicAdd(a.eval_count + 4, 1ull);
                            if (plen < 3) atomicAdd(a.eval_count + 5, 1ull);
                            if (g_path != g_link) atomicAdd(a.eval_count + (deep_path ? 7 : 6), 1ull);
                            if (g_link == NONE || nodes[g_link].board != n.board || nodes[g_link].n_children == 0) atomicAdd(a.eval_count + 8, 1ull);
                            if (g_path == NONE || nodes[g_path].board != n.board || nodes[g_path].n_children == 0) atomicAdd(a.eval_count + 9, 1ull);
                            if (!deep_path && (plen < 1 || p_idx[(plen - 1) * Eng::EPB] != node)) atomicAdd(a.eval_count + 10, 1ull);
                            if (!deep_path && (plen < 2 || p_idx[(plen - 2) * Eng::EPB] != n.parent)) atomicAdd(a.eval_count + 11, 1ull);
                        }
                        g = (a.reuse_mode == 2u || (a.reuse_mode == 0u && !deep_path)) ? g_path : g_link;
                        // tripwire, not a filter: the grandparent's board IS this node's board (every child is a legal move, so the
                        // move back restores it) and it was expanded on the way down; a failure is counted and fails the collect
                        if (g != NONE && a.reuse_mode != 2u && (nodes[g].board != n.board || nodes[g].n_children == 0)) { atomicAdd(a.eval_count + 12, 1ull); g = NONE; }
                    }
                    if (g == NONE) { phase = PH_LEAF; leaf = s; need_nn = true; break; }       // :154 needs the network
                    const uint4 o2 = outs[2 * g], o3 = outs[2 * g + 1];
                    probs[0] = __uint_as_float(o2.x); probs[1] = __uint_as_float(o2.y); probs[2] = __uint_as_float(o2.z); probs[3] = __uint_as_float(o2.w);
                    const float gv = __uint_as_float(o3.x);
                    ++evals; ++reused;
                    const uint32_t cb = n_nodes;
                    const uint32_t nch = expand(node, s, n.action);
                    keep_output(node, gv);
                    if (nch > 0) {
                        const u32x4 w = rng_draw(a.seed, e_global, it * MED + expanded, STREAM_MCTS | ((uint32_t)t << 8));
                        node = cb + (uint32_t)sample_weighted4(pri, (int)nch, u32_to_unit(w.x));
                        push(node, 0.0f, 0u);
                    }
                    value = gv;
                    ++expanded;
                }
                if (need_nn) break;
                TW_MS(m_b0);
                backprop(node, value);                                               // :163
                ++it;
                TW_MS(m_b1);
                TW_MA(c_bp, m_b0, m_b1);
#ifdef TW_ABLATE
                ++c_wl;
#endif
            }
        }
#ifdef TW_ABLATE
        c_tree += __builtin_readcyclecounter() - s1;
#endif
        TW_MS(z5);
        TW_MA(x_own, z4, z5);
        // mirror what the other lanes of the episode need for the next collective evaluation
        if constexpr (Eng::SPLIT) {        // lanes j / j+32 of every wave of the workgroup: through LDS
            uint32_t *bc = reinterpret_cast<uint32_t *>(eng.lds_user) + j * 8;
            if (h == 0 && eng.owns_lane()) {
                bc[0] = (uint32_t)phase; bc[1] = (uint32_t)leaf.board; bc[2] = (uint32_t)(leaf.board >> 32);
                bc[3] = (uint32_t)leaf.zx; bc[4] = (uint32_t)leaf.zy; bc[5] = (uint32_t)leaf.depth;
            }
            __syncthreads();
            phase = (int)bc[0];
            leaf.board = ((uint64_t)bc[2] << 32) | (uint64_t)bc[1];
            leaf.zx = (int)bc[3]; leaf.zy = (int)bc[4]; leaf.depth = (int)bc[5];
        } else {
            phase      = __shfl(phase, j, 64);
            leaf.board = ((uint64_t)__shfl((uint32_t)(leaf.board >> 32), j, 64) << 32) | (uint64_t)__shfl((uint32_t)leaf.board, j, 64);
            leaf.zx    = __shfl(leaf.zx, j, 64);
            leaf.zy    = __shfl(leaf.zy, j, 64);
            leaf.depth = __shfl(leaf.depth, j, 64);
        }
        TW_MS(z6);
        TW_MA(x_mir, z5, z6);
    }
    if (owner) {
        if (sv.on) {
            total = total + puzzle_reward(st, env);                       // solve.rs:65-66
            sv.success[e_local] = puzzle_solved(st, env) ? 1.0f : 0.0f;   // solve.rs:68
            sv.total[e_local]   = total;
            sv.n_steps[e_local] = (uint32_t)t;
        } else if constexpr (!PERSIST) a.out.ep_len[e_local] = len;
        atomicAdd(a.eval_count, evals);
        if (!sv.on && reused) atomicAdd(a.eval_count + 2, reused);
    }
#ifdef TW_ABLATE
    {   // wave-level: cycles from lane 0; per-lane counters: sum and max over the wave's owners
        unsigned long long mx = c_inner, sm = owner ? c_inner : 0, ds = owner ? c_desc : 0;
        unsigned long long own_work = c_pre + c_dsc + c_bp;      // wave-uniform clock: the max over the lanes is the wave's own tree work
        for (int o = 32; o; o >>= 1) {
            const unsigned long long w2 = ((unsigned long long)__shfl_xor((unsigned)(own_work >> 32), o, 64) << 32) | __shfl_xor((unsigned)own_work, o, 64);
            own_work = w2 > own_work ? w2 : own_work;
        }
        for (int o = 32; o; o >>= 1) {
            const unsigned long long m2 = ((unsigned long long)__shfl_xor((unsigned)(mx >> 32), o, 64) << 32) | __shfl_xor((unsigned)mx, o, 64);
            mx = m2 > mx ? m2 : mx;
            sm += ((unsigned long long)__shfl_xor((unsigned)(sm >> 32), o, 64) << 32) | __shfl_xor((unsigned)sm, o, 64);
            ds += ((unsigned long long)__shfl_xor((unsigned)(ds >> 32), o, 64) << 32) | __shfl_xor((unsigned)ds, o, 64);
        }
        if (eng.lane == 0) {
            atomicAdd(&g_mcts_stamps[0], c_fwd); atomicAdd(&g_mcts_stamps[1], c_tree); atomicAdd(&g_mcts_stamps[2], c_trips);
            if constexpr (Eng::SPLIT) { for (int i = 0; i < 5; ++i) atomicAdd(&g_mcts_stamps[8 + i], eng.stq[i]); }
            atomicAdd(&g_mcts_stamps[13], c_pre); atomicAdd(&g_mcts_stamps[14], c_dsc); atomicAdd(&g_mcts_stamps[15], c_bp);
            atomicAdd(&g_mcts_stamps[3], sm); atomicAdd(&g_mcts_stamps[4], mx); atomicAdd(&g_mcts_stamps[5], ds); atomicAdd(&g_mcts_stamps[6], 1ull);
            atomicAdd(&g_mcts_stamps[16], x_or); atomicAdd(&g_mcts_stamps[17], x_rows); atomicAdd(&g_mcts_stamps[18], x_eng);
            atomicAdd(&g_mcts_stamps[19], x_own); atomicAdd(&g_mcts_stamps[20], x_mir); atomicAdd(&g_mcts_stamps[21], own_work);
        }
    }
#endif
    eng.end();
}

template <int NT, int NC, int NW, bool PERSIST = false>
static int launch_mcts_geom(const MctsArgs &a, hipStream_t s, uint32_t *blocks, uint32_t *threads)
{
    using G = Geom<NT, NC, 0, NW>;
    constexpr int EPB = G::Eng::EPB;
    const uint64_t nb = PERSIST ? rollout_f32_resident_episodes(a.reserve_cus) / (8 * EPW) : (a.num_episodes + EPB - 1) / EPB;   // persistent: one workgroup per CU
    if (nb == 0 || nb > 0x7fffffffull) { set_error("mcts: bad episode count %llu", (unsigned long long)a.num_episodes); return TW_ERR_INVALID; }
    const size_t lds_bytes = (G::Eng::lds_floats(a.pol) + (size_t)3 * PATH_DEPTH * EPB) * sizeof(float);
    if (lds_bytes > 159 * 1024) { set_error("mcts: %zu bytes of LDS needed, 159 KiB available", lds_bytes); return TW_ERR_UNSUPPORTED; }
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&mcts_f32_kernel<NT, NC, NW, PERSIST>), lds_bytes)) return rc;
#ifdef TW_ABLATE
    unsigned long long zeros[24] = {0};
    if (getenv("TW_STAMPS")) TW_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_mcts_stamps), zeros, sizeof(zeros)));
#endif
    hipLaunchKernelGGL((mcts_f32_kernel<NT, NC, NW, PERSIST>), dim3((unsigned)nb), dim3(64 * G::WAVES), lds_bytes, s, a);
    TW_HIP(hipGetLastError());
#ifdef TW_ABLATE
    if (getenv("TW_STAMPS")) {
        unsigned long long h[24];
        TW_HIP(hipStreamSynchronize(s));
        TW_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_mcts_stamps), sizeof(h)));
        const double w = (double)h[6];
        fprintf(stderr, "mcts stamps: waves %.0f | per wave: fwd %.0f cyc, tree %.0f cyc, trips %.1f | per trip: fwd %.0f, tree %.0f | inner per lane-trip %.2f, max-lane inner per trip %.2f, descents per inner %.2f\n",
                w, h[0] / w, h[1] / w, h[2] / w, (double)h[0] / h[2], (double)h[1] / h[2], (double)h[3] / (32.0 * h[2]), (double)h[4] / h[2], (double)h[5] / (double)h[3]);
        fprintf(stderr, "  tree phase per trip: expand+sample %.0f, descents %.0f, backprops %.0f\n", (double)h[13] / h[2], (double)h[14] / h[2], (double)h[15] / h[2]);
        fprintf(stderr, "  per trip: loop-top barrier %.0f, rows_of %.0f, engine forward %.0f, (softmax = fwd - these), owner block %.0f (of which expand/descend/backprop, busiest lane %.0f), mirror + barrier %.0f\n",
                (double)h[16] / h[2], (double)h[17] / h[2], (double)h[18] / h[2], (double)h[19] / h[2], (double)h[21] / h[2], (double)h[20] / h[2]);
        fprintf(stderr, "  split engine per trip: prologue %.0f, chunk compute %.0f, vmcnt wait %.0f, barrier wait %.0f, heads %.0f\n",
                (double)h[8] / h[2], (double)h[9] / h[2], (double)h[10] / h[2], (double)h[11] / h[2], (double)h[12] / h[2]);
    }
#endif
    if (blocks) *blocks = (uint32_t)nb;
    if (threads) *threads = 64 * G::WAVES;
    return TW_OK;
}

template <int NT, int NC>
static int launch_mcts_one(const MctsArgs &a, hipStream_t s, uint32_t *blocks, uint32_t *threads)
{
    const uint64_t resident = f32_resident_episodes(a.num_episodes, a.pol.hidden, true, a.reserve_cus);
    if (!a.solve.on && a.queue && a.init_boards && a.num_episodes > resident) {
        if constexpr (NT >= 4) { if (resident < rollout_f32_resident_episodes(a.reserve_cus)) return launch_mcts_geom<NT, NC, -4, true>(a, s, blocks, threads); }
        return launch_mcts_geom<NT, NC, 8, true>(a, s, blocks, threads);
    }
    const int nw = geometry_for<NT>(a.num_episodes);
    if constexpr (NT >= 4) { if (nw == -16) return launch_mcts_geom<NT, NC, -16>(a, s, blocks, threads); }
    if constexpr (NT >= 4) { if (nw == -4) return launch_mcts_geom<NT, NC, -4>(a, s, blocks, threads); }
    else if constexpr (NT == 2) { if (nw == -2) return launch_mcts_geom<NT, NC, -2>(a, s, blocks, threads); }
    else {
        if (nw == 1) return launch_mcts_geom<NT, NC, 1>(a, s, blocks, threads);
        if (nw == 2) return launch_mcts_geom<NT, NC, 2>(a, s, blocks, threads);
    }
    return launch_mcts_geom<NT, NC, 8>(a, s, blocks, threads);
}

template <int NT>
static int launch_mcts_nt(const MctsArgs &a, hipStream_t s, uint32_t *blocks, uint32_t *threads)
{
    const int nc = a.env.n_cells;
    if (nc <= 4) return launch_mcts_one<NT, 4>(a, s, blocks, threads);
    if (nc <= 9) return launch_mcts_one<NT, 9>(a, s, blocks, threads);
    return launch_mcts_one<NT, 16>(a, s, blocks, threads);
}

int launch_mcts_f32(const MctsArgs &a, hipStream_t s, uint32_t *blocks, uint32_t *threads)
{
    // host-side shape checks: everything the kernel indexes with is validated here
    const uint64_t need = 5ull + 4ull * a.num_searches * (a.max_expand_depth ? a.max_expand_depth : 1u);
    if (a.pol.generic) {          // any Sequential depth: the vector-ALU engine (tw_engine_generic.hpp)
        if (a.env.n_cells < 1 || a.env.n_cells > 16 || a.pol.obs_size != a.env.n_cells * a.env.n_cells || a.pol.n_actions != 4 ||
            (!a.solve.on && a.out.t_pad < a.env.depth0 + 1) || a.node_cap < need || !a.arena || !a.eval_count || a.queue || a.init_boards ||
            (a.solve.on && (!a.solve.success || !a.solve.total || !a.solve.n_steps || a.solve.num_searches == 0))) {
            set_error("mcts: unsupported shape for a generic policy (n_cells=%d obs_size=%d actions=%d)", a.env.n_cells, a.pol.obs_size, a.pol.n_actions);
            return TW_ERR_UNSUPPORTED;
        }
        const int nc = a.env.n_cells;
        if (nc <= 4) return launch_mcts_geom<0, 4, -65>(a, s, blocks, threads);
        if (nc <= 9) return launch_mcts_geom<0, 9, -65>(a, s, blocks, threads);
        return launch_mcts_geom<0, 16, -65>(a, s, blocks, threads);
    }
    if (a.env.n_cells < 1 || a.env.n_cells > 16 || a.pol.obs_size != a.env.n_cells * a.env.n_cells ||
        a.pol.obs_size > 256 || a.pol.n_actions != 4 || a.pol.emb % 32 != 0 || a.pol.emb < 32 ||
        (!a.solve.on && a.out.t_pad < a.env.depth0 + 1) || a.node_cap < need || !a.arena || !a.eval_count ||
        (a.solve.on && (!a.solve.success || !a.solve.total || !a.solve.n_steps || a.solve.num_searches == 0))) {
        set_error("mcts: unsupported shape (n_cells=%d obs_size=%d actions=%d emb=%d hidden=%d t_pad=%d node_cap=%u need=%llu)",
                  a.env.n_cells, a.pol.obs_size, a.pol.n_actions, a.pol.emb, a.pol.hidden, a.out.t_pad, a.node_cap,
                  (unsigned long long)need);
        return TW_ERR_UNSUPPORTED;
    }
    switch (a.pol.hidden) {
        case 32:  return launch_mcts_nt<1>(a, s, blocks, threads);
        case 64:  return launch_mcts_nt<2>(a, s, blocks, threads);
        case 128: return launch_mcts_nt<4>(a, s, blocks, threads);
        case 256: return launch_mcts_nt<8>(a, s, blocks, threads);
        default:
            set_error("mcts: hidden size %d not in {32,64,128,256}", a.pol.hidden);
            return TW_ERR_UNSUPPORTED;
    }
}

// ---- AZ finalize: remaining_values + compaction (az.rs:93-106) -------------------------------
constexpr int AZF_WAVES = 4;

__global__ void __launch_bounds__(AZF_WAVES * 64) finalize_az_kernel(const PaddedTraj in, const uint64_t *ep_start, uint64_t E,
                                                                     int n_cells, uint8_t *obs_out, float *probs_out,
                                                                     int8_t *perms_out, float *remaining_out)
{
    extern __shared__ __attribute__((aligned(16))) float az_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t_pad = in.t_pad;
    float *sr = az_lds + (size_t)wave * 2 * t_pad;   // rewards | prefix sums (total_vals, az.rs:74)
    float *sp = sr + t_pad;
    for (uint64_t e = (uint64_t)blockIdx.x * AZF_WAVES + wave; e < E; e += (uint64_t)gridDim.x * AZF_WAVES) {
        const int      n   = (int)in.ep_len[e];
        const uint64_t src = e * (uint64_t)t_pad;
        const uint64_t dst = ep_start[e];
        for (int t = lane; t < n; t += 64) sr[t] = in.rec[src + t].reward;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float total = 0.0f;                              // az.rs:64,74-76: total_vals.push(total); total += val
        for (int t = 0; t < n; ++t) {
            if (lane == 0) sp[t] = total;
            total = total + sr[t];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int t = lane; t < n; t += 64) {
            remaining_out[dst + t] = total - sp[t];      // az.rs:93
            perms_out[dst + t] = (int8_t)-1;             // az.rs:95
            reinterpret_cast<uint4 *>(probs_out)[dst + t] = reinterpret_cast<const uint4 *>(in.rec + src + t)[1];
        }
        if (n_cells == 16) {
            for (int t = lane; t < n; t += 64)
                reinterpret_cast<uint4 *>(obs_out)[dst + t] = reinterpret_cast<const uint4 *>(in.rec + src + t)[0];
        } else {
            const int nb = n * n_cells;
            for (int i = lane; i < nb; i += 64) {
                const int t = i / n_cells, c = i - t * n_cells;
                obs_out[dst * n_cells + i] = in.rec[src + t].obs[c];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

int launch_finalize_az(const PaddedTraj &in, const uint64_t *ep_start, uint64_t E, int n_cells, uint8_t *obs_out,
                       float *probs_out, int8_t *perms_out, float *remaining_out, hipStream_t s)
{
    if (E == 0) return TW_OK;
    const size_t lds_bytes = (size_t)AZF_WAVES * 2 * in.t_pad * sizeof(float);
    if (lds_bytes > 64 * 1024) { set_error("finalize_az: t_pad %d too large for the LDS tile", in.t_pad); return TW_ERR_UNSUPPORTED; }
    uint64_t blocks = (E + AZF_WAVES - 1) / AZF_WAVES;
    if (blocks > 256ull * 16) blocks = 256ull * 16;
    hipLaunchKernelGGL(finalize_az_kernel, dim3((unsigned)blocks), dim3(AZF_WAVES * 64), lds_bytes, s, in, ep_start, E, n_cells,
                       obs_out, probs_out, perms_out, remaining_out);
    TW_HIP(hipGetLastError());
    return TW_OK;
}

}  // namespace tw
