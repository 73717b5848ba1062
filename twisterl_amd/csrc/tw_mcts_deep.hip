// tw_mcts_deep.hip -- AlphaZero self-play for FEW, DEEP searches: one wave = one episode ("walker"), every policy
// forward evaluates the leaf each walker is waiting for PLUS frontier nodes of its tree that no search has asked for yet.
//
// Same reference semantics as tw_mcts.hip (AZCollector::single_collect, rust/src/collector/az.rs:51-109, over
// predict_probs_mcts, rust/src/rl/search.rs:104-189), same RNG keys, same arithmetic: bit-identical results.
//
// Why a second kernel.  At the reference's batch (4,096 episodes per GPU x 100..1,000 searches per move) the collect is as
// long as its LONGEST episode's chain of searches: every search needs the value of the previous one (search.rs:132-164), so
// an episode of 17 moves x 101 evaluations is 1,717 dependent (tree walk -> policy forward) steps however many episodes run
// beside it.  The lane-per-episode kernel pays one whole forward (~45k cycles for the 16 MFMA columns of a workgroup) plus
// the slowest lane's tree walk for each of them -- while 2/3 of the columns belong to episodes that are already over.
// Here a workgroup runs only ONE to EIGHT episodes at a time (deep_shape below) on the 16 or 32 columns of a forward; with four:
//   * column 0 of a walker's share carries the leaf its search is blocked on (the "demand"), the other 3 (7) columns
//     carry nodes of its tree that exist but have not been evaluated, in creation order -- UCB with an untrained
//     (flat) prior visits nodes nearly breadth-first, so 60 % (77 %) of the later demands find their network output already
//     stored in the node and the search goes on without waiting for a forward (measured: scripts/spec_sim.py);
//   * a policy forward is a pure function of the board, evaluated per MFMA column as one k-ordered fma chain: which column
//     and which batch evaluates a node does not change a bit of its output.  Every node is evaluated at most once and its
//     output consumed at most once -- nothing is cached across demands (the root of every move is evaluated again, as the
//     reference does);
//   * the tree walk is wave-parallel instead of lane-serial: the (up to four) children of a node are scored by four lanes,
//     expanded by four lanes, the search path lives one level per lane (back-propagation is ONE store instruction);
//   * workgroups are persistent: a walker whose episode is over takes the next one from the queue (start boards from
//     init_boards_kernel), its tree arena is reused;
//   * the network output is a function of the BOARD alone, and a search meets the same boards over and over: a child whose move
//     takes its parent's move back holds its grandparent's board, that child's children hold the boards of the grandparent's
//     children, the tree of the next move is the tree of this one seen from a neighbouring root ...  Every output a search
//     expands a node with is therefore kept in a per-walker table keyed by the packed board (direct-mapped, overwritten on
//     collision, kept across the walker's episodes: MctsArgs::tbl), and a leaf that holds no output asks the table before it
//     asks for a forward -- the same bits the reference computes again (CPU count, 100 searches: 31 % of the leaf evaluations
//     repeat the grandparent's board, 27 % another board of the same tree, 28 % a board of an earlier move's tree).
// 64 bytes per node in the walker's arena, as separate arrays: statistics (touched only beyond the nodes kept in LDS), boards,
// and the stored network outputs (written once and read at most once -- non-temporal): what a walk re-reads stays small enough
// for the weights and the trees of the 32 CUs of an XCD to share its 4 MB L2 (eight walkers per CU: 105 M L2 misses per
// 4,096 x 100 collect with 64-byte node records, scripts/pmc_az_l2.sh).
#include "tw_engine.hpp"

#include <atomic>
#include <cstdlib>
#include <mutex>

namespace tw {

// arena of one walker: uint4 hot[node_cap] | uint4 brd[node_cap] | uint4 out[node_cap][2] | uint2 hot2[node_cap]
//   hot: q = value_sum / visit_count (f32 bits, 0 while unvisited), visit_count, prior (f32 bits), link = child_base | n_children << 24 | action << 27 | has_output << 29
//        -- for the first `lds_nodes` nodes of a tree this quad lives in LDS instead (same layout), its arena slot is never touched
//   brd: board.lo, board.hi, parent, depth
//   out: masked-softmax probs[4] of full_predict (f32 bits) | network value (f32 bits), 0, 0, 0 -- once evaluated ahead of the search
//   hot2: value_sum (f32 bits) | best = the child `next` (search.rs:77-91) would choose, as child index | action << 27 (DNONE: no
//        children, or no score beats -inf).  q and best are functions of statistics that only a
//        back-propagation changes, and a back-propagation touches only the nodes of ITS search path: so the lane that updates
//        path level l (one lane per level) also re-derives q of its node and, from the children's stored statistics, its best
//        child -- the very operations of search.rs:29-39 on the very operands, computed when the operands change instead of
//        every time they are read.  A descent is then a chain of `best` links.  Like `hot`, in LDS for the first `lds_nodes` nodes
constexpr size_t DEEP_NODE_BYTES = 72;
size_t mcts_deep_node_bytes() { return DEEP_NODE_BYTES; }
// Entries of a walker's board-keyed output table: sixteen times the nodes one move's searches expand (an episode expands ~0.15 x
// searches NEW boards per move, CPU count; measured at 1 / 2 / 4 / 8 / 16 x: 4,096 x 1,000 138 / 133 / 134 / 129 / 127 ms, 64 x 1,000 69.4 /
// 65.9 / 63.6 / 63.0 / 63.7), between 1,024 and 32,768 -- 32 KiB to 1 MiB per walker.  Direct-mapped and overwritten on collision:
// whatever the table forgets costs a column of a forward, never a bit of the result.
uint32_t mcts_deep_table_entries(uint32_t num_searches, uint32_t max_expand_depth)
{
    const uint64_t want = 16ull * num_searches * (max_expand_depth ? max_expand_depth : 1u);
    uint32_t t = 1024;
    while (t < want && t < 32768u) t <<= 1;
    return t;
}

constexpr uint32_t DNONE = 0xffffffffu;
constexpr int DEEP_WAVES = 4;            // waves that run the forward; the first NWK (1, 2, 4) of them are walkers -- or, with NWK = 8 on the
                                         // 32-column engine, four more waves that only walk (eight waves per workgroup, two per SIMD)
// outputs evaluated ahead and not consumed yet, kept in LDS per walker (older ones: the arena); eight walkers have three columns
// of look-ahead each and little LDS: a shorter pool leaves room for a third more tree statistics (40 / 24 / 12 entries: 32.7 /
// 31.9 / 32.1 ms at 4,096 x 100)
__host__ __device__ constexpr int deep_pool(int walkers) { return walkers > 4 ? 24 : 40; }
constexpr uint32_t LK_CB = 0x00ffffffu, LK_OUT = 1u << 29, LK_UNDO = 1u << 30;   // LK_UNDO: the node's move takes its parent's move back
enum { DP_ROOT = 0, DP_LEAF = 1, DP_DEAD = 2 };
// MctsArgs::tree_budget_min / tree_budget: cycles of tree walk per trip after which a walker stops at the next search boundary
// -- once another walker of the workgroup waits for a forward (its demand, or the root of a new move) / unconditionally; all
// its columns then carry frontier nodes.  A waiting walker is not kept waiting for long streaks of stored outputs, and as long
// as nobody waits no forward is run for look-ahead alone: the unconditional budget is off by default (4.29e9 cycles).  It used
// to be 300,000 cycles (round 2, before the board-keyed table: scripts/az_budget_sweep.sh, profiles/r02_az_budget_sweep.txt); with the
// table a streak of searches served from it runs for hundreds of thousands of cycles, and stopping it bought a forward whose
// columns the next demand's forward would have carried anyway (a lone walker: 53 % of its trips were such stops).  Round 3, ms,
// 300,000 / 1,000,000 / 3,000,000 / never: 64 x 1,000 62.5 / 61.5 / 59.3 / 60.6, 512 x 1,000 73.9 / 70.8 / 70.4 / 70.1; 300,000 / never:
// 4,096 x 1,000 128.4 / 120.0, 1,024 x 1,000 81.7 / 79.8, 4,096 x 100 19.5 / 19.2, 1,024 x 100 11.6 / 11.5, 512 x 100 9.23 / 8.96, 256 x 100 7.66 / 7.56.
// The minimum (round 2: fixed budget of 48,000 / 72,000 cycles against min 48,000 + max 300,000: 4,096 x 100 33.0 / 33.3 / 32.3,
// 256 x 100 10.7 / 9.7 / 9.3, 4,096 x 1,000 244 / 222 / 202; a minimum of 16,000: 35.4, 8,000: 38.5 at 4,096 x 100) stays: the forward
// costs more than a tree phase, long tree phases amortise it.
static void deep_tree_budgets(uint32_t *min_cycles, uint32_t *max_cycles)
{
    const LaunchOptions o = launch_options();                       // diagnostic: TW_OPT_AZ_TREE_BUDGET(_MIN)
    *max_cycles = o.az_tree_budget > 0 ? (uint32_t)o.az_tree_budget : 0xffffffffu;
    *min_cycles = o.az_tree_budget_min > 0 ? (uint32_t)o.az_tree_budget_min : 48000u;
    if (*min_cycles > *max_cycles) *min_cycles = *max_cycles;
}

__device__ __forceinline__ uint32_t rdl(uint32_t v, int lane_uniform) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane_uniform); }
__device__ __forceinline__ float    rdlf(float v, int lane_uniform) { return __uint_as_float(rdl(__float_as_uint(v), lane_uniform)); }
__device__ __forceinline__ int      uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float    unif(float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v))); }
__device__ __forceinline__ uint32_t uniu(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t lk_nch(uint32_t link) { return (link >> 24) & 7u; }
__device__ __forceinline__ int      lk_act(uint32_t link) { return (int)((link >> 27) & 3u); }

// floats of LDS beyond the engine's area: request boards [C][2] | results [C][8] | per walker: hot table [lds_nodes][4] | q / sqrt table [lds_nodes][2] | pool idx [POOL] | pool outputs [POOL][8]
__host__ __device__ inline size_t deep_extra_floats(int columns, uint32_t lds_nodes, int walkers, bool dec = false)
{
    // (+ 24: the walkers' alive flags and waiting flags -- decoupled shape: request / done sequence numbers and dead flags, + 16 for the
    //  engine waves' barrier counter, batch word and snapshot; eight walkers, coupled: + 64 dwords each, where the wave-uniform walker
    //  state waits during a forward)
    return (size_t)columns * 10 + 24 + (dec ? 16 : 0) + ((!dec && walkers > DEEP_WAVES) ? (size_t)walkers * 64 : 0) +
           (size_t)walkers * ((size_t)lds_nodes * 6 + deep_pool(walkers) + deep_pool(walkers) * 8);
}

#ifdef TW_ABLATE
__device__ unsigned long long g_deep_stamps[16];
__device__ unsigned long long g_deep_extra[4];
#endif

// DEC, the decoupled shape (round 4): the four engine waves ONLY run forwards, NWK more waves ONLY walk (4 + NWK waves per workgroup).
// A walker that needs an output posts its columns and waits for THAT forward; a walker that needs none is never stopped -- in the
// coupled shapes every forward is a workgroup barrier, and 68 % of the walker trips of a 4,096 x 1,000 collect were walkers stopped in
// the middle of a streak because a neighbour needed a forward (profiles/r03_az_walker_stamps.txt: "yielded").  The engine waves
// synchronise among themselves through an LDS counter (Engine3T::fsync<true>): the hardware barrier counts every wave of the workgroup.
// Requests / completions are sequence numbers in LDS; every spin is bounded (a watchdog sets MctsArgs::eval_count[13] and everybody leaves).
//
// SPL, the split shape (round 4; implies DEC): the walkers are a KERNEL OF THEIR OWN -- NWK walker waves per workgroup, no engine
// wave, no engine LDS -- beside `mcts_engine_kernel` (below), whose workgroups only run forwards.  All waves of a workgroup share ONE
// register allocation, and the walker code needs 67 registers per lane where the engine wants 370: in its own kernel a walker costs a
// sixth of a wave slot and a CU holds 16 of them (an XCD's L2 latency is what a search mostly waits for: more waves in flight hide
// it), with the whole 159 KB of LDS for tree statistics.  Requests and results travel through a 256-byte mailbox per walker in
// device memory, written and polled with relaxed agent-scope atomics (coherent across XCDs without an L2 write-back) and ordered with
// s_waitcnt; every spin is bounded (watchdog -> MctsArgs::eval_count[13], the collect fails).
template <int NT, int NC, int NW, int NWK, bool SOLVE = false, bool DEC = false, bool SPL = false>
__global__ void __launch_bounds__((SPL ? 64 * NWK : DEC ? 64 * (4 + NWK) : (NWK > 4 ? 512 : 256)), 1) mcts_deep_kernel(const MctsArgs a)
{
    static_assert(!SPL || (DEC && !SOLVE), "the split shape is a decoupled self-play shape");
    using Eng = typename Geom<NT, NC, 0, NW>::Eng;
    typedef unsigned int ux4 __attribute__((ext_vector_type(4)));
    typedef unsigned int ux2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) ux4 lds_u4;
    typedef __attribute__((address_space(3))) ux2 lds_u2;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    typedef __attribute__((address_space(3))) float lds_f32;
    constexpr int CPW = SPL ? 4 : Eng::EPB / NWK, C = SPL ? 4 * NWK : Eng::EPB;   // columns per walker (split: of a request), request columns of the workgroup
    constexpr int TWV = SPL ? NWK : DEC ? DEEP_WAVES + NWK : (NWK > DEEP_WAVES ? NWK : DEEP_WAVES);   // waves per workgroup
    constexpr bool PARK = !DEC && TWV > DEEP_WAVES;            // coupled eight-walker shape: the walker state waits in LDS during a forward
    constexpr int DEEP_POOL = deep_pool(NWK);
    static_assert(Geom<NT, NC, 0, NW>::WAVES == DEEP_WAVES && NWK >= 1 && NWK <= (SPL ? 16 : 8) && CPW >= 2 && CPW <= 64, "one walker per wave");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Eng eng;
    bool engw = true;                                          // this wave runs the forward
    if constexpr (SPL) {
        engw = false;
        eng.begin_idle(a.pol);                                 // (the engine's constants only: its barrier is matched by every wave here)
    } else if constexpr (TWV > DEEP_WAVES) {
        engw = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) < DEEP_WAVES;
        if (engw) eng.begin1(a.pol, lds); else eng.begin_idle(a.pol);
    } else {
        eng.begin1(a.pol, lds);
    }

    const PuzzleConsts env = a.env;
    const int lane = eng.lane, wave = eng.wave;
    const int col  = eng.ep_lane();                           // this lane's MFMA column in the engine's mapping
    const uint32_t NL = a.lds_nodes;                          // nodes of a tree whose hot quad lives in LDS
    float *xbase = SPL ? lds : lds + Eng::lds_floats(a.pol);
    uint2 *req = reinterpret_cast<uint2 *>(xbase);                               // request boards [C]
    float *res = xbase + 2 * C;                                                  // results [C][8 floats: probs, value, -]
    const bool walker = SPL ? true : DEC ? wave >= DEEP_WAVES : wave < NWK;      // (coupled: waves NWK..3 only run the forward; decoupled: waves 0..3; split: none)
    const int  wk = walker ? ((DEC && !SPL) ? wave - DEEP_WAVES : wave) : 0;     // this wave's walker number inside the workgroup
    int *alive_f = reinterpret_cast<int *>(res + 8 * C);                         // [2 trips][TWV waves]: walker still has an episode
    // [8]: walker w has stopped in front of a forward it needs (demand / new root).  An LDS-typed pointer: through a generic one the
    // eight polls of a search pass were eight serialised flat loads (hundreds of cycles each); now two 16-byte LDS reads
    volatile lds_u32 *wait_f = (volatile lds_u32 *)(res + 8 * C + 16);
    // decoupled shape: the same 24 dwords hold rq_seq[8] (walker w has posted request number ..) | dn_seq[8] (.. has been served up to ..) |
    // dead_f[8], and 16 more: the engine waves' barrier counter, the batch word, the watchdog flag, the snapshot of rq_seq of a batch
    volatile lds_u32 *ctl = (volatile lds_u32 *)(res + 8 * C);
    float *park_base = res + 8 * C + 24 + (DEC ? 16 : 0);                        // [TWV][64] parked walker state (coupled eight-walker shape)
    float *wbase = park_base + (PARK ? TWV * 64 : 0) + (size_t)wk * ((size_t)NL * 6 + DEEP_POOL + DEEP_POOL * 8);
    lds_u4  *tbl  = (lds_u4 *)wbase;                                             // hot quads of nodes 0 .. NL-1
    lds_u2  *tq   = (lds_u2 *)(wbase + (size_t)NL * 4);                          // q, sqrt(visit) of nodes 0 .. NL-1
    lds_u32 *pidx = (lds_u32 *)(wbase + (size_t)NL * 6);                         // pool: node index (DNONE = free)
    lds_f32 *pout = (lds_f32 *)(wbase + (size_t)NL * 6 + DEEP_POOL);             // pool: probs[4], value, - - -

    const uint32_t S = a.num_searches, MED = a.max_expand_depth;
    const uint64_t E = a.num_episodes;
    const uint64_t slot = (uint64_t)blockIdx.x * NWK + (uint64_t)wk;                      // walker = tree arena index
    uint8_t *arena_w = reinterpret_cast<uint8_t *>(a.arena) + slot * (uint64_t)mcts_deep_arena_bytes(a.node_cap);
    ux4   *hotq = reinterpret_cast<ux4 *>(arena_w);                                   // [node_cap] statistics
    uint4 *brdq = reinterpret_cast<uint4 *>(arena_w + (size_t)a.node_cap * 16);       // [node_cap] boards
    ux4   *outs = reinterpret_cast<ux4 *>(arena_w + (size_t)a.node_cap * 32);         // [node_cap][2] outputs evaluated ahead of the search
    ux2   *hot2 = reinterpret_cast<ux2 *>(arena_w + (size_t)a.node_cap * 64);         // [node_cap] q, sqrt(visit)
    // board-keyed output table of this walker: entry = two quads {board.lo, board.hi, probs[0], probs[1]} {probs[2], probs[3], value, 0};
    // zeroed by the host before the launch (no board is 0), written and read by this wave alone (program order: no races)
    ux4   *tblq = reinterpret_cast<ux4 *>(a.tbl) + slot * (uint64_t)a.tbl_entries * 2;
    const uint32_t tmask = a.tbl_entries - 1u;
    auto tbl_slot = [&](uint64_t b) -> uint32_t { return (uint32_t)((b * 0x9E3779B97F4A7C15ull) >> 36) & tmask; };
    auto tbl_get = [&](uint64_t b, float (&pb)[4], float &v) -> bool {               // b wave-uniform: one line, every lane the same answer
        const uint32_t hs = tbl_slot(b);
        const ux4 q0 = tblq[2 * (size_t)hs], q1 = tblq[2 * (size_t)hs + 1];
        pb[0] = unif(__uint_as_float(q0.z)); pb[1] = unif(__uint_as_float(q0.w)); pb[2] = unif(__uint_as_float(q1.x)); pb[3] = unif(__uint_as_float(q1.y));
        v = unif(__uint_as_float(q1.z));
        return uniu((q0.x == (uint32_t)b && q0.y == (uint32_t)(b >> 32)) ? 1u : 0u) != 0u;
    };
    auto tbl_put = [&](uint64_t b, const float (&pb)[4], float v) {                  // (every lane stores the same 32 bytes)
        const uint32_t hs = tbl_slot(b);
        tblq[2 * (size_t)hs]     = ux4{(uint32_t)b, (uint32_t)(b >> 32), __float_as_uint(pb[0]), __float_as_uint(pb[1])};
        tblq[2 * (size_t)hs + 1] = ux4{__float_as_uint(pb[2]), __float_as_uint(pb[3]), __float_as_uint(v), 0u};
    };

    // hot quad of node idx: LDS for the first NL nodes of the tree, the arena beyond
    auto hot_ld = [&](uint32_t idx) -> ux4 { if (idx < NL) return tbl[idx]; return hotq[idx]; };
    auto hot_st = [&](uint32_t idx, ux4 v) { if (idx < NL) tbl[idx] = v; else hotq[idx] = v; };
    auto hot_ld_link = [&](uint32_t idx) -> uint32_t { if (idx < NL) return reinterpret_cast<lds_u32 *>(tbl + idx)[3]; return reinterpret_cast<const uint32_t *>(hotq + idx)[3]; };
    auto hot2_ld_vs = [&](uint32_t idx) -> uint32_t { if (idx < NL) return reinterpret_cast<lds_u32 *>(tq + idx)[0]; return reinterpret_cast<const uint32_t *>(hot2 + idx)[0]; };
    auto hot2_ld_best = [&](uint32_t idx) -> uint32_t { if (idx < NL) return reinterpret_cast<lds_u32 *>(tq + idx)[1]; return reinterpret_cast<const uint32_t *>(hot2 + idx)[1]; };
    auto hot2_st = [&](uint32_t idx, float vs, uint32_t best) {
        ux2 w; w.x = __float_as_uint(vs); w.y = best;
        if (idx < NL) tq[idx] = w; else hot2[idx] = w;
    };
    auto hot_st_link = [&](uint32_t idx, uint32_t link) {
        if (idx < NL) reinterpret_cast<lds_u32 *>(tbl + idx)[3] = link;
        else reinterpret_cast<uint32_t *>(hotq + idx)[3] = link;
    };

    // ---- walker state (wave-uniform) ----------------------------------------------------------------------------------
    PuzzleLane st;  st.board = env.ident; st.zx = 0; st.zy = 0; st.depth = 0;      // the episode's env (az.rs:56-57)
    PuzzleLane cur = st;                                                            // state of `node` (the leaf being worked on)
    uint64_t e_local = slot, e_global = a.episode_offset + slot, rec_base = 0;
    int      phase = DP_DEAD;
    int      t = 0;
    uint32_t it = 0, expanded = 0, node = 0, n_nodes = 0, cursor = 1, cur_link = 0;
    float    value = 0.0f;
    float    root_vs = 0.0f; uint32_t root_visit = 0, root_cb = 0, root_nc = 0, root_best = DNONE;   // (the root's record lives in registers)
    uint32_t dem_idx = 0;
    unsigned long long evals = 0, spec_evals = 0, reused = 0;      // outputs consumed | columns evaluated ahead | outputs taken from a grandparent
    bool more = true;
    // search path, one level per lane: node index
    uint32_t p_idx = 0; int plen = 0; bool overflow = false;
    // next_sample's draws, 64 at a time: lane i holds word 0 of the draw with index rng_base + i of the current move's stream
    uint32_t rng_buf = 0, rng_base = 0xffffffffu;
    // the request this lane issued ahead of the search in the last assembly (stored into its node when the forward is done)
    bool my_take = false; uint32_t my_idx = 0; int my_rank = 0, my_slot = 0;
    uint32_t pool_head = 0; int n_spec = 0;
    bool yielded = false;                                                           // stopped at a search boundary without a demand
    // this walker's columns of the forward being assembled / evaluated: the C columns are dealt out to the walkers that still
    // have an episode (4, 5, 8 or 16 each) -- at the tail of a collect the long episodes get the look-ahead of the finished ones
    int my_base = 0, my_share = 0; uint32_t trip = 0;

    // MCTS-guided inference (MctsArgs::solve.on; single_solve over predict_probs_mcts, rust/src/rl/solve.rs:17-71): an "episode" of the
    // queue is an ATTEMPT (episode, search) -- its draws keyed like single_solve's (tw_solve.hip), its start the episode's reset or
    // the caller's state --, a move is argmax | sample of the MCTS probs, and what is kept is (success, summed rewards, steps, actions)
    // (a kernel instantiation of its own, SOLVE: the self-play instantiations stay what they were; the launch constants of this mode
    //  are read from device memory where a move or an attempt ends -- MctsArgs::solve_dev)
    const MctsSolve *svp = SOLVE ? a.solve_dev : nullptr;
    constexpr bool sv_on = SOLVE;
    float total = 0.0f;                                                             // solve mode: summed rewards (solve.rs:25-34)
    auto take1 = [&](uint64_t e) {
        e_local = e;
        uint64_t b0;
        int depth_start = env.depth0;
        if (sv_on) {
            const uint32_t ns = uniu(svp->num_searches);
            const uint64_t ep = e / ns;
            e_global = (a.episode_offset + ep) * (uint64_t)ns + e % ns; rec_base = 0;
            if (uniu(svp->from_state)) { b0 = svp->start_board; depth_start = uni(svp->start_depth); }
            else b0 = a.init_boards[ep];
        } else {
            e_global = a.episode_offset + e; rec_base = e * (uint64_t)a.out.t_pad;
            b0 = a.init_boards[e];
        }
        // (wave-uniform values the search branches on are made provably uniform where they come from memory: the walk
        //  then runs on scalar branches instead of exec masks)
        st.board = ((uint64_t)uniu((uint32_t)(b0 >> 32)) << 32) | uniu((uint32_t)b0);
        const int z = blank_cell(st.board);
        st.zx = z % env.width; st.zy = z / env.width; st.depth = depth_start;
        t = 0; phase = DP_ROOT; total = 0.0f;
    };
    // solve mode: the attempt is over (solve.rs:65-68); AZ mode: nothing to write here
    auto finish_attempt = [&]() {
        total = total + puzzle_reward(st, env);
        if (lane == 0) {
            svp->success[e_local] = puzzle_solved(st, env) ? 1.0f : 0.0f;
            svp->total[e_local] = total; svp->n_steps[e_local] = (uint32_t)t;
        }
    };
    auto take = [&](uint64_t e) {
        take1(e);
        // `while !env.is_final()` (solve.rs:30): an attempt that starts in a final state is over at once -- on to the next one
        while (sv_on && phase != DP_DEAD && puzzle_final(st, env)) {
            finish_attempt();
            unsigned got = 0xffffffffu;
            if (more) { if (lane == 0) got = atomicAdd(a.queue, 1u); got = (unsigned)uni((int)got); }
            if ((uint64_t)got < E) take1((uint64_t)got); else { more = false; phase = DP_DEAD; }
        }
    };
    // The walkers take the episodes in MctsArgs::order, the ones that look longest first; MctsArgs::order_across: dealt out ACROSS
    // the workgroups at launch (walker w of workgroup b starts with number w * workgroups + b), from the queue afterwards.
    auto nth = [&](uint64_t q) -> uint64_t { return a.order ? (uint64_t)uniu(a.order[q]) : q; };
    const uint64_t first = (a.order && a.order_across) ? (uint64_t)wk * gridDim.x + blockIdx.x : slot;
    if (walker && first < E) take(nth(first));
    if (walker && lane < DEEP_POOL) pidx[lane] = DNONE;
    if (!walker) more = false;
    if constexpr (DEC) {
        if (threadIdx.x < 40) ctl[threadIdx.x] = 0u;                               // sequence numbers, dead flags, barrier counter, batch, watchdog
        if (threadIdx.x < C) req[threadIdx.x] = make_uint2((uint32_t)env.ident, (uint32_t)(env.ident >> 32));   // (a column nobody has asked for yet holds a valid board)
        if constexpr (DEC && !SPL) eng.ebar_cnt = ctl + 24;
    } else {
        if (lane == 0) { alive_f[wave] = phase != DP_DEAD ? 1 : 0; alive_f[TWV + wave] = phase != DP_DEAD ? 1 : 0; wait_f[wave] = 0; if (wave + 4 < 8) wait_f[wave + 4] = 0; }
    }
    __syncthreads();

    uint32_t obs_base[4];
    obs_base_words(env.n_cells, obs_base);

    auto push = [&](uint32_t idx) {
        if (plen < 64) { if (lane == plen) p_idx = idx; ++plen; }
        else overflow = true;
    };

    // request columns of this walker for the next forward: the demand + up to CPW-1 unevaluated frontier nodes in creation order
    auto assemble = [&]() {
        my_take = false; n_spec = 0;
        if constexpr (DEC) {
            // decoupled shape: walker w owns columns [w * CPW, (w + 1) * CPW) for good (a finished walker's columns idle)
            if (!walker || phase == DP_DEAD) { my_share = 0; return; }
            my_share = CPW; my_base = wk * CPW;
            ++trip;
        } else {
        // publish whether this walker still has an episode (read by everybody one trip later: a walker that ran out THIS trip
        // keeps its columns for one more forward and fills them with the identity board)
        alive_f[(trip & 1u) * TWV + wave] = (walker && phase != DP_DEAD) ? 1 : 0;      // (every lane stores the same value)
        const int *af = alive_f + ((trip & 1u) ^ 1u) * TWV;
        ++trip;
        int n_alive = 0, rank_me = 0;
#pragma unroll
        for (int w = 0; w < TWV; ++w) { const int f = uni(af[w]); rank_me += (w < wave) ? f : 0; n_alive += f; }
        const bool mine = walker && uni(af[wave]) != 0;
        if (!mine) { my_share = 0; return; }
        my_share = C / n_alive; my_base = rank_me * my_share;
        if (rank_me == n_alive - 1) my_share = C - my_base;                  // the last one takes the remainder (16 = 5 + 5 + 6)
        }
        const uint64_t ident = env.ident;
        if (phase == DP_DEAD) {
            if (lane < my_share) req[my_base + lane] = make_uint2((uint32_t)ident, (uint32_t)(ident >> 32));
            return;
        }
        uint64_t db = cur.board;
        if (phase == DP_ROOT) db = st.board;
        const int col0 = yielded ? 0 : 1;                                // a walker that stopped between two searches has no demand
        if (lane == 0 && !yielded) req[my_base] = make_uint2((uint32_t)db, (uint32_t)(db >> 32));
        // 64 candidates in one round trip: nodes cursor .. cursor+63
        const uint32_t idx = cursor + (uint32_t)lane;
        const bool in_tree = phase == DP_LEAF && idx < n_nodes;         // (a new move's tree does not exist yet)
        uint4 c1 = make_uint4(0, 0, 0, 0); ux4 ch = {0u, 0u, 0u, 0u};
        if (in_tree) { c1 = brdq[idx]; ch = hot_ld(idx); }
        const uint64_t cb64 = ((uint64_t)c1.y << 32) | c1.x;
        // a node needs the network if it is not final (search.rs:149), not expanded and holds no output yet
        bool valid = in_tree && !(ch.w & (LK_OUT | LK_UNDO)) && lk_nch(ch.w) == 0u && !(c1.w == 0u || cb64 == ident) && (yielded || idx != dem_idx);
        if (valid) {                                                                   // ... and its board's output is not in the table
            const ux4 k0 = tblq[2 * (size_t)tbl_slot(cb64)];
            valid = !(k0.x == c1.x && k0.y == c1.y);
        }
        const int quota = my_share - col0;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(valid);
        const int rank = __builtin_popcountll(m & ((1ull << lane) - 1ull));
        const int n_valid = __builtin_popcountll(m);
        if (valid && rank < quota) {
            req[my_base + col0 + rank] = make_uint2(c1.x, c1.y);
            my_take = true; my_idx = idx; my_rank = col0 + rank; my_slot = rank;     // (my_rank: column inside the walker's share)
        }
        const int n_take = n_valid < quota ? n_valid : quota;
        if (lane >= col0 + n_take && lane < my_share) req[my_base + lane] = make_uint2((uint32_t)ident, (uint32_t)(ident >> 32));
        // everything below the cursor is evaluated, expanded, final or being evaluated now
        uint32_t nc2;
        if (n_valid >= quota) {
            const unsigned long long last = __builtin_amdgcn_ballot_w64(valid && rank == quota - 1);
            nc2 = cursor + (uint32_t)__builtin_ctzll(last) + 1u;
        } else {
            nc2 = cursor + 64u < n_nodes ? cursor + 64u : (phase == DP_LEAF ? n_nodes : cursor);
        }
        cursor = nc2;
        n_spec = n_take;
        spec_evals += (unsigned long long)n_take;
    };

    // Eight-walker shape (256 registers per lane): the wave-uniform walker state -- about 45 values that would otherwise be
    // spilled to scratch as 64-lane registers around every forward (368 bytes per lane: the scratch of the eight waves of each
    // of an XCD's 32 CUs, the trees and the weights do not fit its L2 together; 64 M L2 misses per collect) -- waits in LDS
    // while the forward runs: one dword per value and wave.
    volatile lds_u32 *pkw = (volatile lds_u32 *)(park_base + wave * 64);
    auto park = [&]() {
        if constexpr (PARK) {
            {   // (every lane stores the same dword: no lane-0 branch in the walker's scalar control flow)
                pkw[0] = (uint32_t)st.board; pkw[1] = (uint32_t)(st.board >> 32); pkw[2] = (uint32_t)st.zx; pkw[3] = (uint32_t)st.zy; pkw[4] = (uint32_t)st.depth;
                pkw[5] = (uint32_t)cur.board; pkw[6] = (uint32_t)(cur.board >> 32); pkw[7] = (uint32_t)cur.zx; pkw[8] = (uint32_t)cur.zy; pkw[9] = (uint32_t)cur.depth;
                pkw[10] = (uint32_t)e_local; pkw[11] = (uint32_t)(e_local >> 32);
                pkw[12] = (uint32_t)phase; pkw[13] = (uint32_t)t; pkw[14] = it; pkw[15] = expanded; pkw[16] = node; pkw[17] = n_nodes; pkw[18] = cursor; pkw[19] = cur_link;
                pkw[20] = __float_as_uint(value); pkw[21] = __float_as_uint(root_vs); pkw[22] = root_visit; pkw[23] = root_cb; pkw[24] = root_nc; pkw[25] = dem_idx;
                pkw[26] = (uint32_t)evals; pkw[27] = (uint32_t)(evals >> 32); pkw[28] = (uint32_t)spec_evals; pkw[29] = (uint32_t)(spec_evals >> 32);
                pkw[30] = (more ? 1u : 0u) | (overflow ? 2u : 0u) | (yielded ? 4u : 0u);
                pkw[31] = (uint32_t)plen; pkw[32] = pool_head; pkw[33] = (uint32_t)n_spec; pkw[34] = (uint32_t)my_base; pkw[35] = (uint32_t)my_share; pkw[36] = trip;
                pkw[37] = (uint32_t)reused; pkw[38] = (uint32_t)(reused >> 32); pkw[39] = root_best; pkw[40] = __float_as_uint(total);
            }
        }
    };
    auto unpark = [&]() {
        if constexpr (PARK) {
            st.board = ((uint64_t)uniu(pkw[1]) << 32) | uniu(pkw[0]); st.zx = (int)uniu(pkw[2]); st.zy = (int)uniu(pkw[3]); st.depth = (int)uniu(pkw[4]);
            cur.board = ((uint64_t)uniu(pkw[6]) << 32) | uniu(pkw[5]); cur.zx = (int)uniu(pkw[7]); cur.zy = (int)uniu(pkw[8]); cur.depth = (int)uniu(pkw[9]);
            e_local = ((uint64_t)uniu(pkw[11]) << 32) | uniu(pkw[10]);
            if (sv_on) { const uint32_t ns = uniu(svp->num_searches); e_global = (a.episode_offset + e_local / ns) * (uint64_t)ns + e_local % ns; rec_base = 0; }
            else { e_global = a.episode_offset + e_local; rec_base = e_local * (uint64_t)a.out.t_pad; }
            phase = (int)uniu(pkw[12]); t = (int)uniu(pkw[13]); it = uniu(pkw[14]); expanded = uniu(pkw[15]); node = uniu(pkw[16]); n_nodes = uniu(pkw[17]);
            cursor = uniu(pkw[18]); cur_link = uniu(pkw[19]);
            value = __uint_as_float(uniu(pkw[20])); root_vs = __uint_as_float(uniu(pkw[21])); root_visit = uniu(pkw[22]); root_cb = uniu(pkw[23]);
            root_nc = uniu(pkw[24]); dem_idx = uniu(pkw[25]);
            evals = ((unsigned long long)uniu(pkw[27]) << 32) | uniu(pkw[26]); spec_evals = ((unsigned long long)uniu(pkw[29]) << 32) | uniu(pkw[28]);
            const uint32_t fl = uniu(pkw[30]);
            more = (fl & 1u) != 0; overflow = (fl & 2u) != 0; yielded = (fl & 4u) != 0;
            plen = (int)uniu(pkw[31]); pool_head = uniu(pkw[32]); n_spec = (int)uniu(pkw[33]); my_base = (int)uniu(pkw[34]); my_share = (int)uniu(pkw[35]);
            trip = uniu(pkw[36]);
            reused = ((unsigned long long)uniu(pkw[38]) << 32) | uniu(pkw[37]);
            root_best = uniu(pkw[39]); total = __uint_as_float(uniu(pkw[40]));
            rng_base = 0xffffffffu;                                      // (the buffered draws are not kept across a forward in this shape)
        }
    };

    // ---- Policy::full_predict of the C requested boards (policy.rs:102-126): the engine waves; `batch` (decoupled shape): bit w = walker w is served
    auto run_forward = [&](uint32_t batch) {
        (void)batch;
        const uint2 rb = req[col];
        const uint64_t board = ((uint64_t)rb.y << 32) | rb.x;
        float lsum[4] = {0.0f, 0.0f, 0.0f, 0.0f}, vsum = 0.0f;
        const int n_pass = eng.pol.n_perms > 0 ? eng.pol.n_perms : 1;
        const float np = (float)eng.pol.n_perms;
        for (int pass = 0; pass < n_pass; ++pass) {
            const int perm = eng.pol.n_perms > 0 ? pass : -1;
            int rowoff[NC];
            eng.rows_of(board, env.n_cells, perm, rowoff);
            float lg[4], v;
            if constexpr (DEC) eng.template forward<true>(rowoff, lg, v); else eng.forward(rowoff, lg, v);     // (DEC: Engine3T with its engine-only barrier)
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
        PuzzleLane lf; lf.board = board; lf.depth = 0;
        const int z = blank_cell(board);
        lf.zx = z % env.width; lf.zy = z / env.width;
        float probs[4];
        masked_softmax4(lsum, puzzle_maskbits(lf, env), probs);
        // every engine wave holds every column's output: wave w publishes the columns of walker w, for itself -- or, when
        // there are walkers beyond the engine waves, every fourth column for everybody
        // (decoupled shape: only the columns of the walkers this forward serves -- the others may still be reading their last results)
        const bool pub = DEC ? ((col & 3) == wave && ((batch >> (col / CPW)) & 1u) != 0u) : (TWV > DEEP_WAVES ? (col & 3) == wave : (col >= my_base && col < my_base + my_share));
        if (eng.h == 0 && pub) {
            float4 *dst = reinterpret_cast<float4 *>(res + col * 8);
            dst[0] = make_float4(probs[0], probs[1], probs[2], probs[3]);
            dst[1] = make_float4(vsum, 0.0f, 0.0f, 0.0f);
        }
    };

    // ---- decoupled shape: request / completion hand-shake between the walkers and the engine waves (LDS, sequence numbers) -------------
    uint32_t my_seq = 0;                                        // walker: requests posted so far
    bool     aborted = false;
    constexpr uint32_t SPIN_LIMIT = SPL ? 1u << 22 : 1u << 25;  // x >= ~100 cycles (split: ~1 us) per poll: seconds -- a hand-shake bug ends the kernel instead of hanging the GPU
    auto watchdog = [&]() { aborted = true; ctl[26] = 1u; if (lane == 0) atomicAdd(a.eval_count + 13, 1ull); };
    // split shape: this walker's mailbox (MctsArgs::mailbox, 64 dwords: [0] request number (0xffffffff: the walker is done) | [2..9] up to four
    // boards, the demand first | [32] number of the request served last | [33] how many of its boards were evaluated | [36..55] their outputs,
    // five floats each).  Relaxed agent-scope atomics only: coherent across XCDs without the L2 write-back an agent-scope fence would do.
    uint32_t *mbx = SPL ? a.mailbox + slot * 64 : nullptr;
    auto post = [&]() {                                         // this walker's columns are in `req`: ask for a forward
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        ++my_seq;
        if constexpr (SPL) {
            if (lane < CPW) {
                const uint2 b = req[my_base + lane];
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(mbx + 2) + lane, ((unsigned long long)b.y << 32) | b.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // (s_waitcnt: the boards are out before the number goes up)
            if (lane == 0) __hip_atomic_store(mbx, my_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            ctl[wk] = my_seq;
        }
    };
    auto mark_dead = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if constexpr (SPL) { if (lane == 0) __hip_atomic_store(mbx, 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else ctl[16 + wk] = 1u;
    };
    auto wait_result = [&]() {
        uint32_t spins = 0;
        if constexpr (SPL) {
            while (uniu(__hip_atomic_load(mbx + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != my_seq) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > SPIN_LIMIT || uniu(ctl[26]) != 0u) { if (spins > SPIN_LIMIT) watchdog(); aborted = true; break; }
            }
            // the outputs into the workgroup's staging area, where the tree phase reads them (as the engine waves leave them in the other shapes)
            const uint32_t n_done = uniu(__hip_atomic_load(mbx + 33, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (lane < CPW * 5) {
                const int c = lane / 5, i = lane - 5 * c;
                res[(my_base + c) * 8 + i] = __uint_as_float(__hip_atomic_load(mbx + 36 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
            if (my_take && (uint32_t)my_rank >= n_done) my_take = false;      // (a busy engine evaluates fewer of the look-ahead boards: they stay what they were)
        } else {
            while (uniu(ctl[8 + wk]) != my_seq) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > SPIN_LIMIT || uniu(ctl[26]) != 0u) { if (spins > SPIN_LIMIT) watchdog(); aborted = true; break; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };

    if (engw) eng.begin2();
    assemble();
    int live = phase != DP_DEAD ? 1 : 0;
    park();
    if constexpr (DEC) {
        if (walker) { if (phase != DP_DEAD) post(); else mark_dead(); }
        if constexpr (!SPL) if (engw) {
            // ---- the engine waves: serve forwards until every walker is done.  Wave 0 polls the walkers' request numbers and publishes
            //      the batch (which walkers this forward serves) -- the other three wait for it in the engine barrier.
            uint32_t served[NWK];
#pragma unroll
            for (int w = 0; w < NWK; ++w) served[w] = 0u;
            for (;;) {
                uint32_t mask = 0u;
                if (wave == 0) {
                    uint32_t spins = 0, ndead = 0;
                    for (;;) {
                        mask = 0u; ndead = 0u;
#pragma unroll
                        for (int w = 0; w < NWK; ++w) {
                            const uint32_t r = uniu(ctl[w]);
                            ndead += uniu(ctl[16 + w]);
                            if (r != served[w]) { mask |= 1u << w; ctl[28 + w] = r; }
                        }
                        if (mask != 0u || ndead == (uint32_t)NWK || uniu(ctl[26]) != 0u) break;
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > SPIN_LIMIT) { watchdog(); break; }
                    }
                    // (a walker posts before it can be dead: no request is pending once all are dead)
                    ctl[25] = (mask != 0u && uniu(ctl[26]) == 0u) ? mask : 0x80000000u;
                }
                if constexpr (DEC) eng.template fsync<true, true>();    // (waves 1 .. 3 idle here, asleep between polls: they share their SIMDs with the walkers)
                const uint32_t batch = uniu(ctl[25]);
                if (batch & 0x80000000u) break;
                // (the forward is on the critical path of every walker waiting for it; the walkers it shares its SIMDs with are bound by
                //  latency, not by issue slots)
                __builtin_amdgcn_s_setprio(2);
                run_forward(batch);
                __builtin_amdgcn_s_setprio(0);
                if constexpr (DEC) eng.template fsync<true>();    // every served column's result is in `res`
                if (wave == 0) {
#pragma unroll
                    for (int w = 0; w < NWK; ++w) if ((batch >> w) & 1u) { served[w] = uniu(ctl[28 + w]); ctl[8 + w] = served[w]; }
                }
            }
        }
    }

#ifdef TW_ABLATE
    unsigned long long c_fwd = 0, c_tree = 0, c_bar = 0, c_trips = 0, c_search = 0, c_hits = 0, c_asm = 0, c_yield = 0, c_root = 0, c_dead = 0, c_nspec = 0;
    unsigned long long c_pre = 0, c_desc = 0, c_leaf = 0, c_bp = 0, c_fin = 0, c_res = 0, c_lvl = 0;
#define TW_DS(var) const unsigned long long var = __builtin_readcyclecounter()
#define TW_DA(acc, x, y) acc += (y) - (x)
#else
#define TW_DS(var)
#define TW_DA(acc, x, y)
#endif
    // (decoupled shape: only the walker waves run this loop -- the engine waves have served their last forward above)
    if (!DEC || walker) for (;;) {
        TW_DS(z0);
        if constexpr (DEC) {
            if (phase == DP_DEAD) break;                       // (marked dead where the last episode ended)
            wait_result();                                     // the forward that carries this walker's columns
            if (aborted) break;
        } else {
            if (!__syncthreads_or(live)) break;
        }
        TW_DS(z1);
        TW_DA(c_bar, z0, z1);
        if constexpr (!DEC) {
            // ---- Policy::full_predict of the C requested boards (policy.rs:102-126) -------------------------------------
            if (engw) {
                run_forward(0u);
            } else if constexpr (TWV > DEEP_WAVES) {
                // a wave that only walks: the barriers of the forwards the engine waves run (nothing else synchronises in there)
                const int n_pass = eng.pol.n_perms > 0 ? eng.pol.n_perms : 1;
                for (int pass = 0; pass < n_pass; ++pass) eng.idle_forward();
            }
            if constexpr (TWV > DEEP_WAVES) {
                __syncthreads();
                unpark();
            } else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        TW_DS(z2);
        TW_DA(c_fwd, z1, z2);
#ifdef TW_ABLATE
        ++c_trips;
        if (phase == DP_DEAD) { if (walker) ++c_dead; } else if (yielded) ++c_yield; else if (phase == DP_ROOT) ++c_root;
        c_nspec += (unsigned long long)n_spec;
#endif

        // ---- tree phase of this wave's walker ------------------------------------------------------------------------
        if (phase != DP_DEAD) {
            TW_DS(y0);
            if constexpr (!DEC) wait_f[wave] = 0;              // (every lane stores: a lane-0 branch here costs the walk its scalar branches)
            // outputs evaluated ahead of the search -> their nodes (arena) and the LDS pool; the node's hot quad gets the flag
            if (my_take) {
                const float4 *src = reinterpret_cast<const float4 *>(res + (my_base + my_rank) * 8);
                const float4 pr = src[0]; const float4 vv = src[1];
                __builtin_nontemporal_store(ux4{__float_as_uint(pr.x), __float_as_uint(pr.y), __float_as_uint(pr.z), __float_as_uint(pr.w)}, outs + 2 * my_idx);
                __builtin_nontemporal_store(ux4{__float_as_uint(vv.x), 0u, 0u, 0u}, outs + 2 * my_idx + 1);
                const uint32_t ps = (pool_head + (uint32_t)my_slot) % (uint32_t)DEEP_POOL;
                pidx[ps] = my_idx;
                lds_f32 *po = pout + ps * 8;
                po[0] = pr.x; po[1] = pr.y; po[2] = pr.z; po[3] = pr.w; po[4] = vv.x;
                hot_st_link(my_idx, hot_ld_link(my_idx) | LK_OUT);
                my_take = false;
            }
            pool_head = (pool_head + (uint32_t)n_spec) % (uint32_t)DEEP_POOL;
            // the demanded output
            float probs[4], nn_value;
            {
                const float4 *src = reinterpret_cast<const float4 *>(res + my_base * 8);
                const float4 pr = src[0]; const float4 vv = src[1];
                probs[0] = unif(pr.x); probs[1] = unif(pr.y); probs[2] = unif(pr.z); probs[3] = unif(pr.w); nn_value = unif(vv.x);
            }
            // the output of every node that gets expanded goes into the table, under its board
            if (!yielded) tbl_put(phase == DP_ROOT ? st.board : cur.board, probs, nn_value);

            const int ca = lane & 3;                               // the child / action this lane works on
            // expand (search.rs:56-75): one child per action with prior > 0, state = clone + step; four lanes, one child each
            float pri[4] = {0.0f, 0.0f, 0.0f, 0.0f};               // priors of the children just created, in child order
            uint32_t act_mask = 0;                                 // bit a: action a got a child
            auto expand = [&](uint32_t idx, uint32_t idx_link, const PuzzleLane s, const float (&pb)[4]) -> uint32_t {
                const float mine = ca == 0 ? pb[0] : (ca == 1 ? pb[1] : (ca == 2 ? pb[2] : pb[3]));
                const bool has = mine > 0.0f;
                act_mask = (uint32_t)(__builtin_amdgcn_ballot_w64(has && lane < 4)) & 15u;
                const uint32_t cnt = (uint32_t)__builtin_popcount(act_mask);
                const uint32_t pos = (uint32_t)__builtin_popcount(act_mask & ((1u << ca) - 1u));
                PuzzleLane c = s;
                puzzle_step_legal(c, env, ca);                                // (a child exists only for a legal move: prior > 0; the other lanes' result is not used)
                if (has && lane < 4) {
                    const uint32_t ni = n_nodes + pos;
                    const uint32_t undo = (idx != 0u && (uint32_t)ca == (((idx_link >> 27) & 3u) ^ 2u)) ? LK_UNDO : 0u;   // 0 left, 1 up, 2 right, 3 down
                    hot_st(ni, ux4{0u, 0u, __float_as_uint(mine), ((uint32_t)ca << 27) | undo});
                    hot2_st(ni, 0.0f, DNONE);                                // value_sum 0 (and q = 0: search.rs:31); no children
                    brdq[ni] = make_uint4((uint32_t)c.board, (uint32_t)(c.board >> 32), idx, (uint32_t)c.depth);
                }
                if (lane == 0) hot_st_link(idx, (idx_link & ~(LK_CB | (7u << 24))) | n_nodes | (cnt << 24));
                // priors in child order (uniform)
                uint32_t k = 0;
#pragma unroll
                for (int act = 0; act < 4; ++act) {
                    if (!((act_mask >> act) & 1u)) continue;
                    if (k == 0) pri[0] = pb[act]; else if (k == 1) pri[1] = pb[act]; else if (k == 2) pri[2] = pb[act]; else pri[3] = pb[act];
                    ++k;
                }
                n_nodes += cnt;
                return cnt;
            };
            // next_sample (search.rs:94-100): a child of `node` by its priors; `cur` follows
            auto sample_child = [&](uint32_t cb, uint32_t nch) {
                // (Philox is ~110 instructions for one draw: the lanes compute 64 consecutive ones in one go)
                const uint32_t di = it * MED + expanded;
                if ((di & ~63u) != rng_base) {
                    rng_base = di & ~63u;
                    rng_buf = rng_draw(a.seed, e_global, rng_base + (uint32_t)lane, STREAM_MCTS | ((uint32_t)t << 8)).x;
                }
                const int k = sample_weighted4(pri, (int)nch, u32_to_unit(rdl(rng_buf, (int)(di & 63u))));
                int act = 0, seen = 0;
#pragma unroll
                for (int x = 0; x < 4; ++x) if ((act_mask >> x) & 1u) { if (seen == k) act = x; ++seen; }
                const uint32_t undo = (node != 0u && (uint32_t)act == (((cur_link >> 27) & 3u) ^ 2u)) ? LK_UNDO : 0u;
                node = cb + (uint32_t)k;
                puzzle_step_legal(cur, env, act);                             // (the sampled child exists: its move is legal)
                cur_link = ((uint32_t)act << 27) | undo;
                push(node);
            };
            // next (search.rs:77-91) of a node, from its children's stored statistics: the first maximum of
            // ucb = q + C * sqrt(N) / (n + 1) * prior (search.rs:29-39), strict '>' from -inf (a NaN never wins), as child index | action << 27;
            // DNONE without children or when nothing beats -inf (the reference panics there).  Per-lane arguments.
            auto best_child = [&](uint32_t cb, uint32_t nch, float sqN) -> uint32_t {
                if (nch == 0u) return DNONE;
                ux4 k[4];                                                              // (the children of a node are contiguous)
                if (cb + nch <= NL) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) k[c] = tbl[cb + ((uint32_t)c < nch ? (uint32_t)c : nch - 1u)];
                } else if (cb >= NL) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) k[c] = hotq[cb + ((uint32_t)c < nch ? (uint32_t)c : nch - 1u)];
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) k[c] = hot_ld(cb + ((uint32_t)c < nch ? (uint32_t)c : nch - 1u));
                }
                uint32_t best = DNONE; float bu = -__builtin_inff();
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float d = sqN / ((float)k[c].y + 1.0f);
                    d = a.C * d;
                    d = d * __uint_as_float(k[c].z);
                    const float u = __uint_as_float(k[c].x) + d;
                    const bool better = ((uint32_t)c < nch) & (u > bu);
                    bu = better ? u : bu;
                    best = better ? ((cb + (uint32_t)c) | (k[c].w & (3u << 27))) : best;
                }
                return best;
            };
            // backpropagate (search.rs:45-53): value_sum += v, visit_count += 1 on every node of the path -- lane l = path level l -- and,
            // from the new statistics, the q and the best child the next descents read (header: hot2)
            auto backprop = [&](uint32_t idx, float val) {
                uint32_t b = DNONE;
                if (!overflow) {
                    const bool on = lane < plen, inner = on && lane != 0;                // level 0 is the root: its record lives in registers
                    const bool in_lds = p_idx < NL;
                    uint32_t vis = root_visit, link = root_cb | (root_nc << 24); float vs = root_vs;
                    if (inner) {
                        if (in_lds) { const ux4 h = tbl[p_idx]; vis = h.y; link = h.w; vs = __uint_as_float(reinterpret_cast<lds_u32 *>(tq + p_idx)[0]); }
                        else { const ux4 h = hotq[p_idx]; vis = h.y; link = h.w; vs = __uint_as_float(reinterpret_cast<const uint32_t *>(hot2 + p_idx)[0]); }
                    }
                    const uint32_t nv = vis + 1u; const float nvs = vs + val;
                    const float nq = nvs / (float)nv;
                    ux2 w; w.x = __float_as_uint(nq); w.y = nv;
                    if (inner) {
                        if (in_lds) { *reinterpret_cast<lds_u2 *>(tbl + p_idx) = w; reinterpret_cast<lds_u32 *>(tq + p_idx)[0] = __float_as_uint(nvs); }
                        else { *reinterpret_cast<ux2 *>(hotq + p_idx) = w; reinterpret_cast<uint32_t *>(hot2 + p_idx)[0] = __float_as_uint(nvs); }
                    }
                    // level l reads the statistics level l+1 has just stored (same wave: its memory operations stay in order)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (on) b = best_child(link & LK_CB, lk_nch(link), sqrtf((float)nv));
                    if (inner) {
                        if (in_lds) reinterpret_cast<lds_u32 *>(tq + p_idx)[1] = b;
                        else reinterpret_cast<uint32_t *>(hot2 + p_idx)[1] = b;
                    }
                } else if (lane == 0) {
                    // a path deeper than the 64 levels the lanes hold: bottom-up through the parent links
                    while (idx != DNONE && idx != 0u) {
                        const ux4 h = hot_ld(idx);
                        const uint32_t nv = h.y + 1u; const float nvs = __uint_as_float(hot2_ld_vs(idx)) + val;
                        hot_st(idx, ux4{__float_as_uint(nvs / (float)nv), nv, h.z, h.w});
                        hot2_st(idx, nvs, best_child(h.w & LK_CB, lk_nch(h.w), sqrtf((float)nv)));
                        idx = brdq[idx].z;
                    }
                    b = best_child(root_cb, root_nc, sqrtf((float)(root_visit + 1u)));
                }
                root_best = rdl(b, 0);
                root_vs = root_vs + val; root_visit += 1u;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            };

            // root node of a move's tree (search.rs:120-129): visit_count 1, expanded with the root priors
            auto start_move = [&](const float (&pb)[4]) {
                ++evals;
                n_nodes = 1; cursor = 1;
                pidx[lane < DEEP_POOL ? lane : 0] = DNONE;               // outputs of the previous move's tree (every lane stores: no lane branch in the walk)
                root_vs = 0.0f; root_visit = 1u; root_cb = 1u;
                rng_base = 0xffffffffu;                                   // (the draws are keyed by the move)
                root_nc = expand(0u, 0u, st, pb);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                root_best = rdl(best_child(root_cb, root_nc, 1.0f), 0);         // sqrt(visit_count = 1)
                it = 0;
                phase = DP_LEAF;
            };
            bool resume = false;
            TW_DS(y1);
            TW_DA(c_pre, y0, y1);
            const unsigned long long tree_t0 = __builtin_readcyclecounter();
            if (yielded) {
                yielded = false;                                 // stopped between two searches: go on searching
            } else if (phase == DP_ROOT) {
                start_move(probs);
            } else {
                resume = true;                                   // the leaf `node` just got its output
            }

            for (;;) {
#ifdef TW_ABLATE
                ++c_search;
#endif
                bool need_nn = false;
                TW_DS(y2);
                if (!resume) {
                    if (!DEC && it != S && (NWK > 1 || a.tree_budget != 0xffffffffu)) { // (a lone walker keeps nobody waiting; decoupled walkers never do)
                        const unsigned long long walked = __builtin_readcyclecounter() - tree_t0;
                        bool stop = walked > (unsigned long long)a.tree_budget;
                        if (!stop && walked > (unsigned long long)a.tree_budget_min) {        // somebody waits for a forward: do not keep it waiting
                            const ux4 wa = *(volatile lds_u4 *)(res + 8 * C + 16), wb = *(volatile lds_u4 *)(res + 8 * C + 20);
                            stop = uniu(wa.x | wa.y | wa.z | wa.w | wb.x | wb.y | wb.z | wb.w) != 0u;
                        }
                        if (stop) { yielded = true; dem_idx = DNONE; break; }
                    }
                    if (it == S) {
                        // ---- move finished: visit counts -> probs (search.rs:166-188) ------------------------------
                        float mp[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                        if (root_nc > 0) {
                            const ux4 rq = hot_ld(root_cb + ((uint32_t)ca < root_nc ? (uint32_t)ca : root_nc - 1u));
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                if ((uint32_t)c >= root_nc) continue;
                                const int act = lk_act(rdl(rq.w, c));
                                const float vis = (float)rdl(rq.y, c);
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
                        int action = 0;
                        bool over;
                        if (sv_on) {
                            // solve.rs:31-58: total += reward; action = argmax | sample of the MCTS probs; step
                            total = total + puzzle_reward(st, env);
                            if (uniu(svp->deterministic)) {
                                float bv = mp[0];
#pragma unroll
                                for (int i = 1; i < 4; ++i) if (mp[i] > bv) { bv = mp[i]; action = i; }
                            } else {
                                const u32x4 w = rng_draw(a.seed, e_global, (uint32_t)t, STREAM_SOLVE);
                                action = sample_weighted4(mp, 4, u32_to_unit(w.x));
                            }
                            if (svp->actions && lane == 0) svp->actions[e_local * (uint64_t)svp->act_pad + (uint64_t)t] = (uint8_t)action;
                            puzzle_step(st, env, action);
                            ++t;
                            over = puzzle_final(st, env);
                            if (over) finish_attempt();
                        } else {
                            // az.rs:72-81: action = sample(mcts_probs); val = env.reward(); store the record
                            const u32x4 w = rng_draw(a.seed, e_global, (uint32_t)t, STREAM_AZ_ACT);
                            action = sample_weighted4(mp, 4, u32_to_unit(w.x));
                            if (lane == 0) {
                                uint32_t pk[4];
                                obs_bytes(st.board, obs_base, pk);
                                store_rec(a.out.rec + rec_base + (uint64_t)t, pk, mp, 0.0f, puzzle_reward(st, env), 0, -1);
                            }
                            over = puzzle_final(st, env);                                                // az.rs:84
                            if (over && lane == 0) a.out.ep_len[e_local] = (uint32_t)t + 1u;
                        }
                        if (over) {
                            unsigned got = 0xffffffffu;
                            if (more) {
                                if (lane == 0) got = atomicAdd(a.queue, 1u);
                                got = (unsigned)uni((int)got);
                            }
                            if ((uint64_t)got < E) take(nth((uint64_t)got));
                            else { more = false; phase = DP_DEAD; }
                            TW_DS(y9); TW_DA(c_fin, y2, y9);
                            break;
                        }
                        if (!sv_on) { puzzle_step(st, env, action); ++t; }                               // az.rs:89
                        // The next move's root holds a board this episode has most likely expanded a node with (the child just
                        // chosen, if a search went through it): the new tree then starts from the table's output at once
                        // instead of waiting for a forward.
                        {
                            float rp[4]; float rv;
                            if (tbl_get(st.board, rp, rv)) {
                                start_move(rp);
                                ++reused;
                                TW_DS(y9); TW_DA(c_fin, y2, y9);
                                continue;
                            }
                        }
                        phase = DP_ROOT;
                        TW_DS(y9); TW_DA(c_fin, y2, y9);
                        break;
                    }
                    // ---- descend to a leaf by UCB (search.rs:133-138, next :77-91, ucb :29-39): four lanes score the four
                    //      children; the state follows the chosen actions (a child's state IS step(parent state, action))
                    // ---- descend to a leaf (search.rs:133-138): follow the stored `next` choices; the state follows the actions (a
                    //      child's state IS step(parent state, action), and a child exists only for a legal move)
                    node = 0; plen = 0; overflow = false;
                    push(0u);
                    cur = st; cur_link = 0u;                   // (a root without children is evaluated again, like any childless node)
                    for (uint32_t b = root_best; b != DNONE; b = uniu(hot2_ld_best(node))) {
                        node = b & LK_CB;
                        puzzle_step_legal(cur, env, (int)((b >> 27) & 3u));
                        push(node);
#ifdef TW_ABLATE
                        ++c_lvl;
#endif
                    }
                    if (node != 0u) cur_link = uniu(hot_ld_link(node));       // the leaf's flags and action
                    value = 0.0f; expanded = 0;
                    TW_DS(y3); TW_DA(c_desc, y2, y3);
                } else {
                    // ---- the demanded leaf's output has arrived (search.rs:154-159): expand, sample a child by the priors ---
                    resume = false;
                    ++evals;
                    const uint32_t cb = n_nodes;
                    const uint32_t nch = expand(node, cur_link, cur, probs);
                    if (nch > 0) sample_child(cb, nch);
                    value = nn_value;
                    ++expanded;
                    TW_DS(y5); TW_DA(c_res, y2, y5);
                }
                {
                    TW_DS(y3);
                    // leaf phase (search.rs:143-160); after a demand: the further expansion levels of the same search (max_expand_depth > 1)
                    while (expanded < MED) {
                        value = puzzle_reward(cur, env);                                  // :146
                        if (puzzle_final(cur, env)) break;                                // :149
                        float lp[4]; float lv;
                        if (cur_link & LK_OUT) {
                            // the node was evaluated ahead of this search: take its output (LDS pool, else the arena) and go on
                            const bool hit = lane < DEEP_POOL && pidx[lane < DEEP_POOL ? lane : 0] == node;
                            const unsigned long long hm = __builtin_amdgcn_ballot_w64(hit);
                            if (hm != 0ull) {
                                const int ps = __builtin_ctzll(hm);
                                lds_f32 *po = pout + ps * 8;
                                lp[0] = unif(po[0]); lp[1] = unif(po[1]); lp[2] = unif(po[2]); lp[3] = unif(po[3]); lv = unif(po[4]);
                                if (lane == ps) pidx[ps] = DNONE;
                            } else {
                                const ux4 o2 = __builtin_nontemporal_load(outs + 2 * node), o3 = __builtin_nontemporal_load(outs + 2 * node + 1);
                                lp[0] = unif(__uint_as_float(o2.x)); lp[1] = unif(__uint_as_float(o2.y)); lp[2] = unif(__uint_as_float(o2.z));
                                lp[3] = unif(__uint_as_float(o2.w)); lv = unif(__uint_as_float(o3.x));
                            }
                            tbl_put(cur.board, lp, lv);
                        } else {
                            // a board this walker has expanded a node with before (the grandparent's, when the move took the parent's
                            // move back; a sibling subtree's; an earlier move's; an earlier episode's): the network would compute the
                            // same bits again
                            if (!tbl_get(cur.board, lp, lv)) { need_nn = true; break; }       // :154 needs the network: demand it
                            ++reused;
                        }
                        ++evals;
#ifdef TW_ABLATE
                        ++c_hits;
#endif
                        const uint32_t cb = n_nodes;
                        const uint32_t nch = expand(node, cur_link, cur, lp);             // :156
                        cur_link &= ~LK_OUT;                                              // (no children: the same node is looked at again)
                        if (nch > 0) sample_child(cb, nch);                               // :157
                        value = lv;                                                       // :158
                        ++expanded;
                    }
                    TW_DS(y4); TW_DA(c_leaf, y3, y4);
                }
                if (need_nn) { dem_idx = node; break; }
                TW_DS(y6);
                backprop(node, value);                                                    // :163
                ++it;
                TW_DS(y7); TW_DA(c_bp, y6, y7);
            }
            if constexpr (!DEC) wait_f[wave] = (phase != DP_DEAD && !yielded) ? 1 : 0;    // stopped in front of a forward it needs
        }
        TW_DS(z3);
        TW_DA(c_tree, z2, z3);
        assemble();
        if constexpr (DEC) {
            if (phase != DP_DEAD) post(); else mark_dead();
        } else {
            live = phase != DP_DEAD ? 1 : 0;
            park();
        }
        TW_DS(z4);
        TW_DA(c_asm, z3, z4);
    }
    if constexpr (DEC) { if (walker && aborted) mark_dead(); }     // (a walker that left through the watchdog must not keep the engine waves waiting)
    unpark();
    if (lane == 0) { atomicAdd(a.eval_count, evals); atomicAdd(a.eval_count + 1, spec_evals); atomicAdd(a.eval_count + 2, reused); }
#ifdef TW_ABLATE
    if (lane == 0) {
        atomicAdd(&g_deep_stamps[0], c_fwd); atomicAdd(&g_deep_stamps[1], c_tree); atomicAdd(&g_deep_stamps[2], c_bar);
        atomicAdd(&g_deep_stamps[3], c_trips); atomicAdd(&g_deep_stamps[4], c_search); atomicAdd(&g_deep_stamps[5], c_hits);
        if (wave < NWK) { atomicAdd(&g_deep_stamps[15], c_yield); }
        atomicAdd(&g_deep_extra[0], c_root); atomicAdd(&g_deep_extra[1], c_dead); atomicAdd(&g_deep_extra[2], c_nspec); atomicAdd(&g_deep_extra[3], wave < NWK ? c_trips : 0ull);
        atomicAdd(&g_deep_stamps[6], 1ull); atomicAdd(&g_deep_stamps[7], c_asm);
        atomicAdd(&g_deep_stamps[8], c_pre); atomicAdd(&g_deep_stamps[9], c_desc); atomicAdd(&g_deep_stamps[10], c_leaf);
        atomicAdd(&g_deep_stamps[11], c_bp); atomicAdd(&g_deep_stamps[12], c_fin); atomicAdd(&g_deep_stamps[13], c_res); atomicAdd(&g_deep_stamps[14], c_lvl);
    }
#endif
    if (engw) eng.end();
}

// ---- the engine of the split shape -----------------------------------------------------------------------------------------
// One workgroup = the four engine waves of Engine3T and nothing else; it serves the walkers [e * split_wpe, (e + 1) * split_wpe) of
// mcts_deep_kernel<.., SPL> through their mailboxes: every thread watches one walker's request number; up to sixteen pending requests
// go into one forward -- four columns each (the demand and three boards evaluated ahead) while at most four wait, two for up to eight,
// the demand alone beyond (a busy engine serves demands first) --, taken round-robin from behind the last one served.  Leaves when
// every one of its walkers has posted 0xffffffff.  Every wait is bounded (watchdog -> eval_count[13]).
// Holds the caller's stream until every engine workgroup is resident: with two walker workgroups per CU the walkers would otherwise be
// free to take one slot on EVERY CU before the engines are placed, and an engine workgroup (127 KB of LDS) fits beside none of them.
__global__ void split_gate_kernel(const uint32_t *resident, uint32_t engines, unsigned long long *eval_count)
{
    uint32_t spins = 0;
    while (__hip_atomic_load(resident, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < engines) {
        __builtin_amdgcn_s_sleep(32);
        if (++spins > (1u << 22)) { if (threadIdx.x == 0) atomicAdd(eval_count + 13, 1ull); break; }      // (seconds: the collect fails instead of hanging)
    }
}

template <int NT, int NC>
__global__ void __launch_bounds__(256, 1) mcts_engine_kernel(const MctsArgs a)
{
    using Eng = Engine3T<NT, NC>;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Eng eng;
    eng.begin1(a.pol, lds);
    const PuzzleConsts env = a.env;
    const int tid = threadIdx.x, lane = eng.lane, wave = eng.wave, col = eng.ep_lane();
    float *xbase = lds + Eng::lds_floats(a.pol);
    uint2 *req = reinterpret_cast<uint2 *>(xbase);                       // [16] boards of this forward
    volatile lds_u32 *sel = (volatile lds_u32 *)(xbase + 32);             // [16][2]: walker (thread number) and request number served in column group r
    volatile lds_u32 *wcnt = (volatile lds_u32 *)(xbase + 64);            // [4] pending requests per wave | [4] shape of the forward: k, columns per request
    const uint32_t w0 = blockIdx.x * a.split_wpe;
    const uint32_t nW = w0 >= a.split_walkers ? 0u : (a.split_walkers - w0 < a.split_wpe ? a.split_walkers - w0 : a.split_wpe);
    uint32_t *mb = a.mailbox + ((size_t)w0 + (size_t)tid) * 64;           // this thread's walker
    uint32_t served = 0u, rr = 0u, idle = 0u;
    bool aborted = false;
    if (tid < 16) req[tid] = make_uint2((uint32_t)env.ident, (uint32_t)(env.ident >> 32));
    if (tid == 0) atomicAdd(a.mailbox + (size_t)a.split_walkers * 64, 1u);          // this engine workgroup is resident (split_gate_kernel waits for all of them)
    eng.begin2();
    __syncthreads();
    for (;;) {
        uint32_t r = 0xffffffffu;
        if ((uint32_t)tid < nW) r = __hip_atomic_load(mb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool dead = r == 0xffffffffu;
        const bool pending = !dead && r != served;
        if (__syncthreads_and(dead ? 1 : 0)) break;                        // every walker of this engine is done
        // rank of the pending requests, counted round-robin from thread rr on (fairness: nobody waits for more than one round)
        const uint32_t pos = ((uint32_t)tid + 256u - rr) & 255u;          // position in the rotated order
        const unsigned long long m = __builtin_amdgcn_ballot_w64(pending);
        if (lane == 0) wcnt[wave] = (uint32_t)__builtin_popcountll(m);
        __syncthreads();
        const uint32_t c0 = wcnt[0], c1 = wcnt[1], c2 = wcnt[2], c3 = wcnt[3];
        const uint32_t n_pending = c0 + c1 + c2 + c3;
        if (n_pending == 0u) {
            __syncthreads();                                               // (wcnt is rewritten next round)
            __builtin_amdgcn_s_sleep(8);
            if (++idle > (1u << 21)) { aborted = true; if (tid == 0) atomicAdd(a.eval_count + 13, 1ull); break; }     // (seconds without a request)
            continue;
        }
        idle = 0u;
        // pending requests in front of this thread in the rotated order: those of earlier positions.  The rotation splits the threads in two
        // runs -- [rr, 256) then [0, rr) --, so: (pending threads t' >= rr with t' < tid, if tid >= rr) or (all pending >= rr + pending t' < tid, if tid < rr)
        const uint32_t wave_base = wave == 0 ? 0u : wave == 1 ? c0 : wave == 2 ? c0 + c1 : c0 + c1 + c2;
        const uint32_t before_me = wave_base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));      // pending threads below tid
        // pending threads below rr: computed by thread rr's own before_me -- publish it
        if ((uint32_t)tid == rr) wcnt[6] = before_me;
        __syncthreads();
        const uint32_t below_rr = wcnt[6];
        const uint32_t rank = (uint32_t)tid >= rr ? before_me - below_rr : before_me + (n_pending - below_rr);
        (void)pos;
        const uint32_t k = n_pending < 16u ? n_pending : 16u;
        const uint32_t cols = k <= 4u ? 4u : (k <= 8u ? 2u : 1u);
        const bool take = pending && rank < k;
        if (take) {
            sel[2 * rank] = (uint32_t)tid; sel[2 * rank + 1] = r;
            for (uint32_t c = 0; c < cols; ++c) {
                const unsigned long long b = __hip_atomic_load(reinterpret_cast<unsigned long long *>(mb + 2) + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                req[rank * cols + c] = make_uint2((uint32_t)b, (uint32_t)(b >> 32));
            }
        }
        if (tid == 0) { wcnt[4] = k; wcnt[5] = cols; }
        __syncthreads();
        // ---- Policy::full_predict of the 16 boards (policy.rs:102-126), as the walker kernel's own forward
        {
            const uint2 rb = req[col];
            const uint64_t board = ((uint64_t)rb.y << 32) | rb.x;
            float lsum[4] = {0.0f, 0.0f, 0.0f, 0.0f}, vsum = 0.0f;
            const int n_pass = eng.pol.n_perms > 0 ? eng.pol.n_perms : 1;
            const float np = (float)eng.pol.n_perms;
            for (int pass = 0; pass < n_pass; ++pass) {
                const int perm = eng.pol.n_perms > 0 ? pass : -1;
                int rowoff[NC];
                eng.rows_of(board, env.n_cells, perm, rowoff);
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
            PuzzleLane lf; lf.board = board; lf.depth = 0;
            const int z = blank_cell(board);
            lf.zx = z % env.width; lf.zy = z / env.width;
            float probs[4];
            masked_softmax4(lsum, puzzle_maskbits(lf, env), probs);
            // every engine wave holds every column's output: wave w writes every fourth column into its request's mailbox
            const uint32_t kk = wcnt[4], cc = wcnt[5];
            const uint32_t rq = (uint32_t)col / cc, ci = (uint32_t)col - rq * cc;
            if (eng.h == 0 && (col & 3) == wave && rq < kk) {
                uint32_t *dst = a.mailbox + ((size_t)w0 + (size_t)sel[2 * rq]) * 64 + 36 + 5 * ci;
                __hip_atomic_store(dst + 0, __float_as_uint(probs[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(dst + 1, __float_as_uint(probs[1]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(dst + 2, __float_as_uint(probs[2]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(dst + 3, __float_as_uint(probs[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(dst + 4, __float_as_uint(vsum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");              // (s_waitcnt: this wave's outputs are out)
        __syncthreads();                                                    // ... and everybody's
        if (take) {
            __hip_atomic_store(mb + 33, cols, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __hip_atomic_store(mb + 32, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // the walker goes on
            served = r;
        }
        // the next round starts behind the last request served
        if (take && rank == k - 1u) wcnt[7] = ((uint32_t)tid + 1u) & 255u;
        __syncthreads();
        rr = wcnt[7];
        if ((uint32_t)tid < 16u && (uint32_t)tid >= k * cols) req[tid] = make_uint2((uint32_t)env.ident, (uint32_t)(env.ident >> 32));   // (unused columns hold a valid board)
    }
    (void)aborted;
    eng.end();
}

// ---- launch ---------------------------------------------------------------------------------------------------------
// The deep shape serves AlphaZero self-play of up to CUs x 8 episodes (short searches) .. CUs x 256 (from 100 searches on: 65,536 on
// an MI355X; the reference's per-GPU batch is 4,096) on policies the 16- / 32-column engines support (128 or 256 hidden units); everything else
// runs the lane-per-episode kernel of tw_mcts.hip.  Measured (scripts/bench_az.py, Puzzle-15, 512/256 policy, difficulty 8;
// walker kernel vs lane-per-episode kernel, end of round 2):
//   1,024 x 100: 14.1 vs 48.9 ms    1,024 x 1,000: 115 vs 498 ms    4,096 x 100: 26.4 vs 42.3 ms    4,096 x 1,000: 174 vs 509 ms
//   8,192 x 100: 45.9 vs 61.1 ms    12,288 x 100: 64.1 vs 68.0 ms   8,192 x 32: 19.7 vs 17.8 ms
// One episode's chain of searches runs about 3x faster here (2 - 3 evaluations consumed per forward and walker instead of one),
// but at most CUs x 8 episodes are in flight: beyond that many the lane-per-episode kernel's 16+ columns of distinct episodes win.

bool mcts_deep_applies(const MctsArgs &a)
{
    const int force = launch_options().force_geom;
    if ((a.pol.hidden != 128 && a.pol.hidden != 256) || a.num_episodes == 0) return false;      // (solve mode: num_episodes = attempts)
    if (force == 8 || force == 1 || (launch_options().az_variant & 7) == 2) return false;     // diagnostic: pin the lane-per-episode shapes
    if ((launch_options().az_variant & 7) >= 3) return true;                              // diagnostic: a pinned walker shape, whatever the batch
    // (eight walkers per workgroup against lane-per-episode: 8,192 x 100 45.9 / 61.1 ms, 12,288 x 100 64.1 / 68.0, 16,384 x 100 81.4 / 78.6,
    //  8,192 x 50 25.3 / 28.4, 6,144 x 50 21.7 / 28.1, 8,192 x 32 19.7 / 17.8, 6,144 x 32 14.9 / 17.5, 4,096 x 24 9.2 / 9.5,
    //  4,096 x 16 6.9 / 6.4, 4,096 x 8 4.4 / 3.5, 2,048 x 16 5.3 / 6.4, 2,048 x 8 3.4 / 3.4)
    // round 3 (board-keyed output table, best-child links; walker with eight per workgroup / lane-per-episode, ms): 16,384 x 100 63.7 / 75.5,
    // 32,768 x 100 117 / 141, 8,192 x 400 96.5 / 235, 8,192 x 32 15.3 / 17.7, 16,384 x 32 27.3 / 25.2, 6,144 x 16 7.7 / 9.0, 8,192 x 16 9.6 / 9.1,
    // 4,096 x 8 4.1 / 3.6
    // end of round 3 (streaks not cut short, longest-looking episodes first, 56 k wait budget): 49,152 x 100 154.9 / 193.8, 65,536 x 100 203 / 233, 131,072 x 100
    // 395 / 291, 24,576 x 48 49.8 / 57.0, 32,768 x 48 65.4 / 66.8, 65,536 x 48 124 / 108, 32,768 x 64 79.4 / 88.7, 131,072 x 64 294 / 186, 12,288 x 32 20.3 / 19.8,
    // 16,384 x 32 26.1 / 25.3, 8,192 x 16 9.0 / 9.2, 12,288 x 16 12.5 / 10.3, 4,096 x 8 4.0 / 3.6
    // ... and with the shape the automatic choice takes there (eight walkers on the 16-column engine; the pinned variant 6 is the 32-column one), walker /
    // lane-per-episode, ms: 65,536 x 100 181.7 / 233, 98,304 x 100 266.7 / 255.9, 49,152 x 48 86.1 / 97.7, 65,536 x 48 112.6 / 108.2, 65,536 x 64 137.0 / 145.9,
    // 16,384 x 32 24.5 / 25.3, 24,576 x 32 34.7 / 37.7, 8,192 x 16 8.5 / 9.2, 12,288 x 16 11.9 / 10.3, 6,144 x 24 9.0 / 13.3, 4,096 x 8 3.61 / 3.56
    const uint32_t S = a.num_searches;
    return a.num_episodes <= (uint64_t)device_cus() * (S >= 64 ? 256u : (S >= 48 ? 192u : (S >= 32 ? 96u : (S >= 16 ? 32u : 8u))));
}

// Shape of a launch: walkers per workgroup and engine width.  As few walkers as keep every CU busy -- with fewer walkers each one
// owns more of the forward's columns, i.e. more of its tree is evaluated ahead of the search and fewer searches wait for a
// forward (misses of a demand: 38 % with three columns of look-ahead, 22 % with seven, 13 % with fifteen); more episodes than
// walkers go through the episode queue in rounds.  From two walkers on they share the 32-column engine (its forward costs 1.45x
// the 16-column one and carries twice the columns).  With many episodes and short searches EIGHT walkers do: four waves that
// only walk beside the four that also run the forward, two waves per SIMD, 4 columns each -- the kernel then has 256
// registers per lane instead of 512 and a third of the tree statistics in LDS.
// Measured (256 CUs, ms per collect; 16-column engine with 1 / 2 / 4 walkers | 32-column engine with 1 / 2 / 4 / 8;
// scripts/az_shape_grid.sh, profiles/r02_az_shape_grid.txt):
//   100 searches    256 episodes  8.5 11.5 15.0 |  8.9 11.3 13.5 18.2     512: 10.5 11.6 15.4 | 11.0 11.5 14.2 18.4
//                   768: 15.9 12.6 16.3 | 16.9 12.6 14.5 19.6              1,024: 18.2 14.1 16.3 | 19.1 14.1 14.8 21.1
//                   1,536: 25.2 20.7 17.2 | 26.4 21.5 15.8 21.3            2,048: 31.3 24.6 18.8 | 33.2 25.1 17.7 21.6
//                   3,072: 44.4 34.0 27.1 | 46.9 34.0 26.1 23.6            4,096: 57.7 42.3 33.7 | 60.8 42.3 31.3 26.4
//   1,000 searches  512: 96 103 129 | 97 97 115 129    1,024: 126 123 130 | 127 115 116 136    2,048: 195 153 157 | 197 148 139 140
//                   4,096: 324 241 197 | 330 225 174 187
//   4,096 x 200: 94 70 58 | 97 67 51.9 47.3     3,072 x 200: 73 54 43 | 75 52 39.4 41.0     4,096 x 400: 157 114 91 | 160 110 84.0 82.5
struct DeepShape { int walkers; bool wide; bool dec; bool split; int engines; };
static std::atomic<int> g_split_disabled{0};
void mcts_deep_disable_split() { g_split_disabled.store(1); }
// The split shape's numbers (round 4, ms per collect, `profiles/r04_az_split_shape.txt`; E x S: engines 48 / 64 / 96 / 128 / 160 with 16 walkers per CU |
// 128 engines with 2 x 12 walkers per CU | best shape inside one workgroup):
//   4,096 x 1,000: - / 77.9 / 76.6 / 73.5 / 77.7 | 77.7 | 87.8      4,096 x 100: - / 15.4 / 13.7 / 13.4 / 15.7 | 13.0 | 17.4      4,096 x 400: 35.5 at 128 | 35.8 | 51.1
//   8,192 x 100: 22.0 at 128 | 17.8 | 30.8      16,384 x 100: 36.5 / 35.3 / 37.0 at 64 / 96 / 128 | 30.6 (29.3 at 96) | 50.3      2,048 x 1,000: 70.7 at 96 | - | 75.6
// Half of the CUs run engines: a forward costs its CU 17 us whatever it carries, a saturated engine packs up to sixteen demands into one,
// and at 100 searches per move a walker spends more time waiting for outputs than walking (tree 43 k cycles per demand, wait 74 k).
static int split_walkers_per_group(uint32_t num_searches)
{
#ifdef TW_ABLATE
    if (const char *e = getenv("TW_SPLIT_WALKERS")) { const int n = atoi(e); if (n == 8 || n == 12 || n == 16) return n; }
#endif
    return num_searches >= 400 ? 16 : 12;              // 1 x 16 walker waves per CU for long searches, 2 x 12 for short ones
}
static int split_groups_per_cu(int walkers_per_group) { return walkers_per_group <= 12 ? 2 : 1; }
static DeepShape deep_shape(uint64_t num_episodes, int reserve_cus, uint32_t num_searches, bool solve = false)
{
    const int cus = device_cus();
    const int r = reserve_cus < 0 ? 0 : (reserve_cus > cus - 1 ? cus - 1 : reserve_cus);
    const uint64_t avail = (uint64_t)(cus - r);
    DeepShape sh;
    // (round 3, ms for two / four walkers: 600 x 100 11.7 / 13.1, 768 x 100 12.7 / 13.1, 900 x 100 13.2 / 13.2, 1,024 x 100 14.1 / 13.1, 768 x 1,000 93.2 / 90.6,
    //  1,024 x 1,000 100.5 / 91.3: four as soon as they fill every CU, and always for long searches)
    // (end of round 3 -- streaks not cut short, longest-looking episodes first: rounds through the queue cost little now, sharing a forward costs
    //  what it did --, ms for one / two / four walkers: x 1,000: 640 61.4 / 65.5 / -, 768 64.2 / 66.3 / 71.2, 1,024 - / 72.5 / 77.8, 1,280 - / 75.8 / 81.8,
    //  1,536 - / 75.6 / 81.7, 2,048 - / 83.2 / 83.0, 4,096 - / 127 / 98.4; x 400: 1,024 - / 32.4 / 34.9, 1,536 - / 33.1 / 36.5; x 100: 640 9.2 / 8.9 / -,
    //  768 11.7 / 9.3 / 10.7, 1,024 - / 9.8 / 10.5, 1,280 - / 11.7 / 11.1, 1,536 - / 15.1 / 11.3)
    const bool long_search = num_searches >= 400;
    if (num_episodes <= (long_search ? 3 : 2) * avail) sh.walkers = 1;
    else if (long_search ? num_episodes < 8 * avail : 2 * num_episodes <= 9 * avail) sh.walkers = 2;
    else sh.walkers = 4;
    if (num_searches < 800 && num_episodes > 10 * avail) sh.walkers = 8;     // (eight x 2 columns / four x 4 on the 16-column engine: 4,096 x 400 56.4 / 60.3 ms, x 600 78.6 / 83.3, x 1,000 128.7 / 126.6, 8,192 x 400 78.6 / 98.6)
    // earlier:     // (with the 80 k budget below: 4,096 x 200 35.1 / 37.7 ms for eight / four, 8,192 x 200 54.8 / 61.7, 4,096 x 400 64.3 / 59.4)
    // (before that budget, eight / four walkers, ms: 3,072 x 100 18.9 / 19.2, 4,096 x 100 21.8 / 23.5, 4,096 x 200 37.8 / 37.4, 4,096 x 400 66.2 / 62.1)
    // (round 3: with the table serving most outputs a forward's look-ahead columns matter less than its cost -- the 16-column forward is
    //  40 k cycles, the 32-column one 59 k: four walkers x 4 columns against four x 8, ms: 4,096 x 1,000 131.6 / 135.0, 2,048 x 1,000 94.1 / 107.3,
    //  1,024 x 1,000 82.0 / 90.0, 2,048 x 100 13.4 / 15.5, 1,024 x 100 11.7 / 13.2; two walkers, 768 x 100: 10.6 / 12.3.  (Eight walkers seemed to
    //  need the 32 columns: 4,096 x 100 21.8 against 23.5 for four x 4) -- until the 16-column engine learnt to carry walk-only waves: eight walkers x
    //  2 columns against eight x 4 on the 32-column engine: 4,096 x 100 19.1 / 19.8 ms, 6,144 x 100 24.8 / 27.9, 16,384 x 100 53.9 / 56.6, 8,192 x 32 14.4 / 14.6
    sh.wide = false;
    // diagnostic (TW_OPT_AZ_VARIANT): 3 / 4 / 5 pin two / one / four walkers per workgroup, + 16 / + 32 the 16- / 32-column engine
    const int v = launch_options().az_variant;
    if ((v & 7) == 3) sh.walkers = 2;
    if ((v & 7) == 4) sh.walkers = 1;
    if ((v & 7) == 5) sh.walkers = 4;
    if ((v & 7) == 6) { sh.walkers = 8; sh.wide = true; }
    if (v & 16) sh.wide = false;
    if (v & 32) sh.wide = true;
    if (!(v & 48) && launch_options().force_geom == 32) sh.wide = true;
    if (solve) { sh.wide = false; if (sh.walkers == 8) sh.walkers = 4; }       // solve mode: the 16-column engine, one / two / four walkers
    // The decoupled shape (four engine-only waves + the walkers, mcts_deep_kernel<.., DEC>): from two walkers per workgroup on, 16-column engine,
    // self-play.  TW_OPT_AZ_VARIANT + 128 pins it on (where it exists), + 256 pins it off.
    sh.dec = !solve && !sh.wide && sh.walkers >= 2 && sh.walkers <= 8;
    if (v & 256) sh.dec = false;
    if ((v & 128) && !solve && !sh.wide && sh.walkers >= 2) sh.dec = true;
    // The split shape (walkers and engine as two kernels): from eight episodes per CU on.  Four engine workgroups per XCD serve the walker
    // workgroups on the other CUs.  TW_OPT_AZ_VARIANT + 512 pins it on, + 1024 off.
    sh.engines = (int)avail / 2;
#ifdef TW_ABLATE
    if (const char *e = getenv("TW_SPLIT_ENGINES")) { const int n = atoi(e); if (n >= 1 && n < (int)avail) sh.engines = n; }
#endif
    sh.split = !solve && !sh.wide && (v & 7) == 0 && avail >= 16 && num_episodes >= 8 * avail && !(v & 256);
    if ((v & 1024) || g_split_disabled.load()) sh.split = false;
    if ((v & 512) && !solve && avail >= 16 && !g_split_disabled.load()) { sh.split = true; sh.wide = false; }
    if (sh.split) { sh.dec = true; sh.walkers = split_walkers_per_group(num_searches); }
    return sh;
}
static int deep_walkers_per_group(uint64_t num_episodes, int reserve_cus, uint32_t num_searches, bool solve = false) { return deep_shape(num_episodes, reserve_cus, num_searches, solve).walkers; }

uint64_t mcts_deep_walkers(uint64_t num_episodes, int reserve_cus, uint32_t num_searches, bool solve)
{
    const int cus = device_cus();
    const int r = reserve_cus < 0 ? 0 : (reserve_cus > cus - 1 ? cus - 1 : reserve_cus);
    const DeepShape sh = deep_shape(num_episodes, reserve_cus, num_searches, solve);
    const uint64_t nwk = (uint64_t)sh.walkers;
    const uint64_t blocks = (num_episodes + nwk - 1) / nwk;
    // (split: the engine workgroups have CUs of their own; one or two walker workgroups on each of the others)
    const uint64_t room = sh.split ? ((uint64_t)(cus - r) - (uint64_t)sh.engines) * (uint64_t)split_groups_per_cu(sh.walkers) : (uint64_t)(cus - r);
    return (blocks < room ? blocks : room) * nwk;
}
bool mcts_deep_split(uint64_t num_episodes, int reserve_cus, uint32_t num_searches, bool solve) { return deep_shape(num_episodes, reserve_cus, num_searches, solve).split; }

// The split shape: mcts_engine_kernel on a side stream (first: its workgroups are resident before the walkers need them; it ends when every
// walker has said it is done), mcts_deep_kernel<.., SPL> on the caller's stream, which then waits for the engine's end.  Walker workgroups
// + engine workgroups <= CUs: every one of them is resident whatever the placement (neither kind fits twice beside the other: 127 KB and
// ~150 KB of LDS), so no workgroup waits for one that cannot start.
static std::mutex g_split_mutex;
static hipStream_t g_split_stream[64] = {};
static hipEvent_t g_split_ready[64] = {}, g_split_done[64] = {};
template <int NT, int NC, int NWK>
static int launch_deep_split(const MctsArgs &a, const DeepShape &sh, hipStream_t s, uint32_t *blocks, uint32_t *threads)
{
    using G = Geom<NT, NC, 0, -16>;
    if (!a.mailbox) { set_error("mcts (deep, split): no mailboxes"); return TW_ERR_INVALID; }
    const uint64_t walkers = mcts_deep_walkers(a.num_episodes, a.reserve_cus, a.num_searches, false);
    const uint64_t nb = walkers / NWK;
    uint64_t ne = (uint64_t)sh.engines;
    if (ne > nb) ne = nb;                                   // (never more engines than walker workgroups)
    const uint64_t wpe = (walkers + ne - 1) / ne;
    if (nb == 0 || wpe > 256) { set_error("mcts (deep, split): %llu walkers for %llu engines", (unsigned long long)walkers, (unsigned long long)ne); return TW_ERR_INVALID; }
    MctsArgs b = a;
    const size_t budget = (size_t)159 * 1024 / sizeof(float) / (size_t)split_groups_per_cu(NWK);
    size_t nl = (budget - deep_extra_floats(4 * NWK, 0, NWK, true)) / ((size_t)NWK * 6);
    if (nl > a.node_cap) nl = a.node_cap;
    b.lds_nodes = (uint32_t)nl;
    b.split_walkers = (uint32_t)walkers; b.split_wpe = (uint32_t)wpe;
    deep_tree_budgets(&b.tree_budget_min, &b.tree_budget);
    const size_t lds_w = deep_extra_floats(4 * NWK, b.lds_nodes, NWK, true) * sizeof(float);
    const size_t lds_e = (G::Eng::lds_floats(a.pol) + 80) * sizeof(float);
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&mcts_deep_kernel<NT, NC, -16, NWK, false, true, true>), lds_w)) return rc;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&mcts_engine_kernel<NT, NC>), lds_e)) return rc;
    int dev = 0; TW_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) { set_error("mcts (deep, split): device %d", dev); return TW_ERR_INVALID; }
    hipStream_t es; hipEvent_t ready, done;
    {
        std::lock_guard<std::mutex> lk(g_split_mutex);
        if (!g_split_stream[dev]) {
            TW_HIP(hipStreamCreateWithFlags(&g_split_stream[dev], hipStreamNonBlocking));
            TW_HIP(hipEventCreateWithFlags(&g_split_ready[dev], hipEventDisableTiming));
            TW_HIP(hipEventCreateWithFlags(&g_split_done[dev], hipEventDisableTiming));
        }
        es = g_split_stream[dev]; ready = g_split_ready[dev]; done = g_split_done[dev];
    }
#ifdef TW_ABLATE
    { unsigned long long zeros[16] = {0}; if (getenv("TW_STAMPS")) { TW_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_deep_stamps), zeros, sizeof(zeros))); TW_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_deep_extra), zeros, 32)); } }
#endif
    // (TW_OPT_AZ_VARIANT + 2048, a test hook: both kernels on the caller's stream, ONE AFTER THE OTHER -- what a tool that serialises kernel
    //  launches does to them; both then run into their watchdogs and tw_az_collect falls back to the single-kernel shapes)
    if (launch_options().az_variant & 2048) es = s;
    TW_HIP(hipEventRecord(ready, s));                       // tables, mailboxes, start boards: written on the caller's stream
    if (es != s) TW_HIP(hipStreamWaitEvent(es, ready, 0));
    hipLaunchKernelGGL((mcts_engine_kernel<NT, NC>), dim3((unsigned)ne), dim3(256), lds_e, es, b);
    TW_HIP(hipGetLastError());
    TW_HIP(hipEventRecord(done, es));
    hipLaunchKernelGGL(split_gate_kernel, dim3(1), dim3(64), 0, s, (const uint32_t *)(b.mailbox + (size_t)walkers * 64), (uint32_t)ne, b.eval_count);
    hipLaunchKernelGGL((mcts_deep_kernel<NT, NC, -16, NWK, false, true, true>), dim3((unsigned)nb), dim3(64 * NWK), lds_w, s, b);
    TW_HIP(hipGetLastError());
    TW_HIP(hipStreamWaitEvent(s, done, 0));                 // the collect goes on when both kernels have ended
#ifdef TW_ABLATE
    if (getenv("TW_STAMPS")) {
        unsigned long long h[16];
        TW_HIP(hipStreamSynchronize(s));
        TW_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_deep_stamps), sizeof(h)));
        const double w = (double)h[6], tr = (double)h[3];
        fprintf(stderr, "split stamps: walkers %.0f (x %d per workgroup, %llu engines), lds nodes %u | per walker: tree %.0f, assembly %.0f, waiting for results %.0f cycles, trips %.1f | per trip: tree %.0f, wait %.0f | "
                        "search-loop passes per trip %.2f, stored outputs consumed per trip %.2f\n",
                w, NWK, (unsigned long long)ne, b.lds_nodes, h[1] / w, h[7] / w, h[2] / w, tr / w, h[1] / tr, h[2] / tr, (double)h[4] / tr, (double)h[5] / tr);
        fprintf(stderr, "  tree phase per trip: store ahead-outputs + read demand %.0f | descents %.0f (%.2f levels per trip) | leaf phase incl. stored-output expansions %.0f | "
                        "resume (expand demanded leaf) %.0f | backprops %.0f | move finish %.0f\n",
                h[8] / tr, h[9] / tr, h[14] / tr, h[10] / tr, h[13] / tr, h[11] / tr, h[12] / tr);
    }
#endif
    if (blocks) *blocks = (uint32_t)nb;
    if (threads) *threads = 64 * NWK;
    return TW_OK;
}


template <int NT, int NC, int NW, int NWK, bool SOLVE = false, bool DEC = false>
static int launch_deep_nwk(const MctsArgs &a, hipStream_t s, uint32_t *blocks, uint32_t *threads)
{
    using G = Geom<NT, NC, 0, NW>;
    constexpr int C = G::Eng::EPB;
    constexpr unsigned THREADS = DEC ? 64u * (DEEP_WAVES + NWK) : (NWK > DEEP_WAVES ? 64u * NWK : 64u * DEEP_WAVES);
    const uint64_t nb = mcts_deep_walkers(a.num_episodes, a.reserve_cus, a.num_searches, SOLVE) / NWK;
    // the hot quads of the first lds_nodes nodes of every tree live in LDS: as many as fit beside the engine
    MctsArgs b = a;
    const size_t eng_floats = G::Eng::lds_floats(a.pol);
    const size_t budget = (size_t)159 * 1024 / sizeof(float);
    if (eng_floats + deep_extra_floats(C, 0, NWK, DEC) > budget) { set_error("mcts (deep): the policy engine alone needs %zu bytes of LDS", eng_floats * 4); return TW_ERR_UNSUPPORTED; }
    size_t nl = (budget - eng_floats - deep_extra_floats(C, 0, NWK, DEC)) / ((size_t)NWK * 6);
    if (nl > a.node_cap) nl = a.node_cap;
    b.lds_nodes = (uint32_t)nl;
    deep_tree_budgets(&b.tree_budget_min, &b.tree_budget);
    // (round 3, with the table: minimum 16 / 32 / 48 / 80 / 120 / 200 k cycles -> 4,096 x 100 (eight walkers) 25.3 / 22.4 / 21.2 / 19.9 / 20.2 / 22.7 ms,
    //  4,096 x 1,000 (four) 125 / 126 / 124 / 124 / 132 / 144, 1,024 x 1,000 80.3 / 80.5 / 81.6 / 84.5 / 88.8 / 98.2: eight walkers share a forward
    //  of 59 k cycles and wait a little longer for it)
    // (re-measured at the end of round 3 -- streaks not cut short, longest-looking episodes first --, minimum 32 / 48 / 64 / 80 / 100 / 140 k cycles, eight
    //  walkers: 4,096 x 100 17.3 / 16.6 / 16.8 / 17.4 / 18.4 / 20.0 ms, 6,144 x 100 26.7 / 24.0 / 24.2 / 24.4 / 24.9 / 26.3, 16,384 x 100 56.1 / 53.0 / 51.6 / 52.6 /
    //  53.2 / 57.4; two and four walkers, 24 / 36 / 48 / 64 / 90 k: 4,096 x 1,000 96.6 / 97.9 / 100.4 / 100.7 / 102.1, 2,048 x 100 12.7 / 12.5 / 12.2 / 12.7 / 13.2,
    //  1,024 x 1,000 76.1 / 76.9 / 77.3 / 77.8 / 81.1, 1,024 x 100 10.3 / 10.3 / 10.5 / 10.6 / 11.1: 48 k stays, eight walkers take 56 k)
    if (NWK == 8 && launch_options().az_tree_budget_min == 0 && b.tree_budget > 56000u) b.tree_budget_min = 56000u;
    const size_t lds_bytes = (eng_floats + deep_extra_floats(C, b.lds_nodes, NWK, DEC)) * sizeof(float);
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&mcts_deep_kernel<NT, NC, NW, NWK, SOLVE, DEC>), lds_bytes)) return rc;
#ifdef TW_ABLATE
    unsigned long long zeros[16] = {0};
    if (getenv("TW_STAMPS")) { TW_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_deep_stamps), zeros, sizeof(zeros))); TW_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_deep_extra), zeros, 32)); }
#endif
    hipLaunchKernelGGL((mcts_deep_kernel<NT, NC, NW, NWK, SOLVE, DEC>), dim3((unsigned)nb), dim3(THREADS), lds_bytes, s, b);
    TW_HIP(hipGetLastError());
#ifdef TW_ABLATE
    if (getenv("TW_STAMPS")) {
        unsigned long long h[16];
        TW_HIP(hipStreamSynchronize(s));
        TW_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_deep_stamps), sizeof(h)));
        unsigned long long x[4];
        TW_HIP(hipMemcpyFromSymbol(x, HIP_SYMBOL(g_deep_extra), sizeof(x)));
        fprintf(stderr, "  walker trips %llu: dead %llu, yielded (no demand) %llu, root evaluations %llu, leaf demands %llu | columns evaluated ahead %llu, consumed %llu\n",
                x[3], x[1], h[15], x[0], x[3] - x[1] - h[15] - x[0], x[2], h[5]);
        const double w = (double)h[6], tr = (double)h[3];
        fprintf(stderr, "deep stamps: waves %.0f, lds nodes %u | per wave: fwd %.0f, tree %.0f, assembly %.0f, barrier wait %.0f cycles, trips %.1f | per trip: fwd %.0f, tree %.0f, assembly %.0f, barrier %.0f | "
                        "search-loop passes per trip %.2f, stored outputs consumed per trip %.2f\n",
                w, b.lds_nodes, h[0] / w, h[1] / w, h[7] / w, h[2] / w, tr / w, h[0] / tr, h[1] / tr, h[7] / tr, h[2] / tr, (double)h[4] / tr, (double)h[5] / tr);
        fprintf(stderr, "  tree phase per trip: store ahead-outputs + read demand %.0f | descents %.0f (%.2f levels per trip) | leaf phase incl. stored-output expansions %.0f | "
                        "resume (expand demanded leaf) %.0f | backprops %.0f | move finish %.0f\n",
                h[8] / tr, h[9] / tr, h[14] / tr, h[10] / tr, h[13] / tr, h[11] / tr, h[12] / tr);
    }
#endif
    if (blocks) *blocks = (uint32_t)nb;
    if (threads) *threads = THREADS;
    return TW_OK;
}

template <int NT, int NC, int NW>
static int launch_deep_geom(const MctsArgs &a, hipStream_t s, uint32_t *blocks, uint32_t *threads)
{
    if constexpr (NW == -17) {       // solve mode (MCTS-guided evaluate / solve): its own instantiations, one / two / four walkers
        switch (deep_walkers_per_group(a.num_episodes, a.reserve_cus, a.num_searches, true)) {
            case 1: return launch_deep_nwk<NT, NC, NW, 1, true>(a, s, blocks, threads);
            case 2: return launch_deep_nwk<NT, NC, NW, 2, true>(a, s, blocks, threads);
            default: return launch_deep_nwk<NT, NC, NW, 4, true>(a, s, blocks, threads);
        }
    }
    const DeepShape sh = deep_shape(a.num_episodes, a.reserve_cus, a.num_searches);
    if constexpr (NW == -16) {      // the decoupled shapes exist on the 16-column engine
        if (sh.split) switch (sh.walkers) {
            case 8:  return launch_deep_split<NT, NC, 8>(a, sh, s, blocks, threads);
            case 12: return launch_deep_split<NT, NC, 12>(a, sh, s, blocks, threads);
            default: return launch_deep_split<NT, NC, 16>(a, sh, s, blocks, threads);
        }
        if (sh.dec) switch (sh.walkers) {
            case 2: return launch_deep_nwk<NT, NC, NW, 2, false, true>(a, s, blocks, threads);
            case 8: return launch_deep_nwk<NT, NC, NW, 8, false, true>(a, s, blocks, threads);
            default: return launch_deep_nwk<NT, NC, NW, 4, false, true>(a, s, blocks, threads);
        }
    }
    switch (sh.walkers) {
        case 1: return launch_deep_nwk<NT, NC, NW, 1>(a, s, blocks, threads);
        case 2: return launch_deep_nwk<NT, NC, NW, 2>(a, s, blocks, threads);
        case 8: return launch_deep_nwk<NT, NC, NW, 8>(a, s, blocks, threads);
        default: return launch_deep_nwk<NT, NC, NW, 4>(a, s, blocks, threads);
    }
}

template <int NT>
static int launch_deep_nt(const MctsArgs &a, hipStream_t s, uint32_t *blocks, uint32_t *threads)
{
    const int nc = a.env.n_cells;
    const bool wide = deep_shape(a.num_episodes, a.reserve_cus, a.num_searches, a.solve.on != 0).wide;
    if (wide) {
        if (nc <= 4) return launch_deep_geom<NT, 4, -4>(a, s, blocks, threads);
        if (nc <= 9) return launch_deep_geom<NT, 9, -4>(a, s, blocks, threads);
        return launch_deep_geom<NT, 16, -4>(a, s, blocks, threads);
    }
    if (a.solve.on) {
        if (nc <= 4) return launch_deep_geom<NT, 4, -17>(a, s, blocks, threads);
        if (nc <= 9) return launch_deep_geom<NT, 9, -17>(a, s, blocks, threads);
        return launch_deep_geom<NT, 16, -17>(a, s, blocks, threads);
    }
    if (nc <= 4) return launch_deep_geom<NT, 4, -16>(a, s, blocks, threads);
    if (nc <= 9) return launch_deep_geom<NT, 9, -16>(a, s, blocks, threads);
    return launch_deep_geom<NT, 16, -16>(a, s, blocks, threads);
}

int launch_mcts_deep(const MctsArgs &a, hipStream_t s, uint32_t *blocks, uint32_t *threads)
{
    const uint64_t need = 5ull + 4ull * a.num_searches * (a.max_expand_depth ? a.max_expand_depth : 1u);
    if (a.env.n_cells < 1 || a.env.n_cells > 16 || a.pol.obs_size != a.env.n_cells * a.env.n_cells || a.pol.obs_size > 256 ||
        a.pol.n_actions != 4 || a.pol.emb % 32 != 0 || a.pol.emb < 32 || (!a.solve.on && a.out.t_pad < a.env.depth0 + 1) || a.node_cap < need ||
        !a.arena || !a.eval_count || !a.queue || (!a.init_boards && !(a.solve.on && a.solve.from_state)) || !a.tbl || a.tbl_entries == 0 ||
        (a.tbl_entries & (a.tbl_entries - 1u)) ||
        (a.solve.on && (!a.solve.success || !a.solve.total || !a.solve.n_steps || a.solve.num_searches == 0 || a.order || !a.solve_dev)) ||
        (!a.solve.on && a.solve_dev)) {
        set_error("mcts (deep): unsupported shape (n_cells=%d obs_size=%d actions=%d emb=%d hidden=%d t_pad=%d node_cap=%u need=%llu)",
                  a.env.n_cells, a.pol.obs_size, a.pol.n_actions, a.pol.emb, a.pol.hidden, a.out.t_pad, a.node_cap, (unsigned long long)need);
        return TW_ERR_UNSUPPORTED;
    }
    switch (a.pol.hidden) {
        case 128: return launch_deep_nt<4>(a, s, blocks, threads);
        case 256: return launch_deep_nt<8>(a, s, blocks, threads);
        default: set_error("mcts (deep): hidden size %d not in {128,256}", a.pol.hidden); return TW_ERR_UNSUPPORTED;
    }
}

}  // namespace tw
