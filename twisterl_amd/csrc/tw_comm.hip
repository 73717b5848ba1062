// tw_comm.hip -- the multi-GPU exchange of the collectors behind the C ABI: RCCL over xGMI, one process per GPU.
//
// The reference has no distributed backend (single process, rayon); episodes are independent
// (rust/src/collector/ppo.rs:59, :110-124), so G ranks collect disjoint episode ranges with the RNG keyed by the GLOBAL
// episode index and three exchanges remain:
//   * tw_comm_broadcast_policy   ncclBroadcast of the policy's device image (every weight image, ~1.6 MB for Puzzle-15) --
//                                the multi-GPU half of Algorithm.sync_rs_policy (src/twisterl/rl/algorithm.py:90-93)
//   * tw_gather_*                finished trajectories to the root in the reference merge order [E-1, 0, .., E-2]
//                                (rust/src/collector/collector.rs:40-46): per step one ncclAllGather of four counters per rank
//                                and grouped ncclSend / ncclRecv of every field, received AT ITS FINAL OFFSET in the root's
//                                result (nothing staged, nothing re-ordered).  Steps are chunk-major (in step s rank r holds
//                                global chunk s*G + r), so the offset of a chunk depends only on counts already gathered and
//                                the transfer of step s overlaps with the collection of step s+1; the records of episode E-1
//                                land in front of everything else, in slack the result keeps there.
// Every non-root rank has its own xGMI link to the root: the G-1 transfers of a step run concurrently.
// RCCL is resolved at run time (dlopen "librccl.so.1"): a single-GPU host of this library does not need it.
#include "tw_common.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace tw {

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
};

Rccl g_rccl;
std::mutex g_rccl_mutex;

int rccl_load()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.lib) return TW_OK;
    // TW_RCCL_LIBRARY names the library that provides the nccl* symbols (a path dlopen takes).  A host sets it to pick one RCCL
    // build out of several; tests/test_gpu_multirank.py sets it to a host-staged stand-in (tests/stub_rccl.hip) so that two and
    // three ranks can exchange through tw_comm_* / tw_gather_* on ONE GPU, which RCCL itself refuses.  A named library that does
    // not load is an error, never a silent switch to another one.
    void *h = nullptr;
    if (const char *named = getenv("TW_RCCL_LIBRARY"); named && *named) {
        h = dlopen(named, RTLD_NOW | RTLD_LOCAL);
        if (!h) { set_error("TW_RCCL_LIBRARY=%s does not load (%s)", named, dlerror()); return TW_ERR_UNSUPPORTED; }
    }
    // a process that already holds an RCCL (PyTorch-ROCm bundles one) must keep using THAT one
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { set_error("RCCL not found (dlopen librccl.so.1: %s)", dlerror()); return TW_ERR_UNSUPPORTED; }
    Rccl r; r.lib = h;
    auto sym = [&](const char *name) { return dlsym(h, name); };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(sym("ncclCommAbort"));      // (optional: old libraries lack it)
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.GetErrorString || !r.AllGather || !r.Broadcast || !r.Send || !r.Recv ||
        !r.GroupStart || !r.GroupEnd) {
        set_error("RCCL library lacks a required symbol");
        return TW_ERR_UNSUPPORTED;
    }
    g_rccl = r;
    return TW_OK;
}

int nccl_fail(ncclResult_t e, const char *what, int line)
{
    set_error("RCCL error %d (%s) at tw_comm.hip:%d: %s", (int)e, g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?", line, what);
    return TW_ERR_HIP;
}
#define TW_NCCL(call) do { ncclResult_t _e = (call); if (_e != ncclSuccess) return nccl_fail(_e, #call, __LINE__); } while (0)

}  // namespace

// bytes per record of every gathered field (tw_collected field order); 0 = not gathered
void gather_field_widths(int is_ppo, uint32_t n_cells, uint32_t obs_width, uint32_t n_actions, size_t (&w)[TW_F_COUNT])
{
    for (int f = 0; f < TW_F_COUNT; ++f) w[f] = 0;
    w[TW_F_OBS] = (size_t)n_cells * obs_width; w[TW_F_LOGITS] = (size_t)n_actions * 4; w[TW_F_PERMS] = 1;
    if (is_ppo) { w[TW_F_VALUES] = 4; w[TW_F_REWARDS] = 4; w[TW_F_ACTIONS] = 1; w[TW_F_ADVS] = 4; w[TW_F_RETS] = 4; }
    else w[TW_F_REMAINING] = 4;
}

}  // namespace tw

using namespace tw;

// ------------------------------------------------------------------------------------------------ placement (pure host code)
// Where one step's chunks go in the root's result.  No HIP, no RCCL: tw_gather_submit calls exactly this, and so does the CPU
// test that compares it with twisterl_amd.dist.plan_step (the gloo-tested twin) at world 2 / 3 / 4 / 8.
extern "C" int tw_gather_plan(tw_gather_state *st, int world, const uint64_t *counts, int32_t *tail_rank_out, uint32_t *n_pieces,
                              tw_gather_piece *pieces)
{
    if (!st || !counts || !n_pieces || !pieces || world < 1) { set_error("tw_gather_plan: bad argument"); return TW_ERR_INVALID; }
    if (st->steps == 0 || st->step >= st->steps) { set_error("tw_gather_plan: step %u of %u", st->step, st->steps); return TW_ERR_INVALID; }
    if (st->steps > 1 && (st->max_records == 0 || st->max_episode_records == 0)) {
        set_error("tw_gather_plan: more than one step needs max_records and max_episode_records (the result is allocated before the totals are known)");
        return TW_ERR_INVALID;
    }
    const bool last_step = st->step + 1 == st->steps;
    auto cnt = [&](int r, int k) { return counts[(size_t)r * TW_GATHER_COUNTS + k]; };
    uint64_t total = 0;
    for (int r = 0; r < world; ++r) {
        if (cnt(r, 1) > cnt(r, 0) || (cnt(r, 0) == 0) != (cnt(r, 2) == 0)) { set_error("tw_gather_plan: inconsistent counts of rank %d", r); return TW_ERR_INVALID; }
        total += cnt(r, 0);
    }
    // episode E-1 is the last episode of the last non-empty chunk of the last step: its records go in front of everything
    uint64_t tail = 0; int tail_rank = -1;
    if (last_step)
        for (int r = world - 1; r >= 0; --r) if (cnt(r, 0) > 0) { tail = cnt(r, 1); tail_rank = r; break; }
    if (st->step == 0) {
        // one step: everything is known (exact size, the tail goes to the very front); several: capacity + slack for the tail
        if (st->steps == 1) { st->front = tail; st->cap = total; }
        else { st->front = st->max_episode_records; st->cap = st->max_records + st->max_episode_records; }
        st->pos = 0; st->tail = 0;
    }
    if (tail > st->front) { set_error("tw_gather_plan: the last episode has %llu records, max_episode_records is %llu", (unsigned long long)tail, (unsigned long long)st->front); return TW_ERR_INVALID; }
    uint64_t p = st->pos;
    for (int r = 0; r < world; ++r) {
        const uint64_t n = cnt(r, 0), at = st->front + p;
        tw_gather_piece *pc = pieces + (size_t)r * 2;
        pc[0] = tw_gather_piece{0, 0, 0}; pc[1] = tw_gather_piece{0, 0, 0};
        if (n == 0) n_pieces[r] = 0;
        else if (r == tail_rank) {
            const uint64_t body = n - tail;
            pc[0] = tw_gather_piece{body, n, st->front - tail};
            n_pieces[r] = 1;
            if (body) { pc[1] = tw_gather_piece{0, body, at}; n_pieces[r] = 2; }
        } else { pc[0] = tw_gather_piece{0, n, at}; n_pieces[r] = 1; }
        p += n - (r == tail_rank ? tail : 0);
    }
    // every rank computes the same capacity from the same numbers: all of them fail here or none does (nobody is left waiting)
    if (st->front + p > st->cap) { set_error("tw_gather_plan: %llu records exceed max_records", (unsigned long long)p); return TW_ERR_INVALID; }
    if (tail_rank_out) *tail_rank_out = tail_rank;
    if (last_step) st->tail = tail;
    st->pos = p;
    st->step += 1;
    return TW_OK;
}

struct tw_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;          // the exchange runs beside the collectors' stream
    uint64_t *counts_dev = nullptr;        // [TW_GATHER_COUNTS] mine | [world][TW_GATHER_COUNTS] all
    // The same, PINNED on the host, + one word for a chunk's last episode length.  Copies from / to pageable memory are not
    // asynchronous: the call itself waits until the stream gets there, i.e. for ever behind the transfer of a peer that died, and
    // the bounded wait below would never be reached (found by tests/test_gpu_multirank.py: a rank killed before a count exchange).
    uint64_t *counts_host = nullptr;
    uint32_t timeout_ms = 0;               // tw_comm_set_timeout_ms: how long tw_gather_finish waits for the transfers (0 = for ever)
    bool dead = false;                     // aborted (a local failure inside a group, or a timeout): every later call fails
};

// A rank that cannot go on takes its communicator down instead of leaving half a group behind: ncclCommAbort ends this rank's
// in-flight RCCL kernels and releases the communicator; the peers' matching operations then never complete, which their own
// tw_gather_finish turns into TW_ERR_HIP after tw_comm_set_timeout_ms (and an abort of THEIR communicator).
static void comm_abort(tw_comm *c)
{
    if (!c || c->dead) return;
    c->dead = true;
    if (c->comm && g_rccl.CommAbort) (void)g_rccl.CommAbort(c->comm);
    else if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    c->comm = nullptr;
}

extern "C" int tw_comm_get_unique_id(tw_comm_id *out)
{
    if (!out) { set_error("tw_comm_get_unique_id: null output"); return TW_ERR_INVALID; }
    int rc = rccl_load(); if (rc) return rc;
    static_assert(sizeof(tw_comm_id) == sizeof(ncclUniqueId), "tw_comm_id must be an ncclUniqueId");
    ncclUniqueId id;
    TW_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(out->bytes, id.internal, sizeof(id.internal));
    return TW_OK;
}

extern "C" int tw_comm_init(int rank, int world, const tw_comm_id *id, tw_comm **out)
{
    if (!id || !out || world < 1 || rank < 0 || rank >= world) { set_error("tw_comm_init: bad argument (rank %d of %d)", rank, world); return TW_ERR_INVALID; }
    *out = nullptr;
    int rc = rccl_load(); if (rc) return rc;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); set_error("tw_comm_init: no HIP device available"); return TW_ERR_NO_DEVICE; }
    tw_comm *c = new tw_comm();
    c->rank = rank; c->world = world;
    hipError_t e = hipGetDevice(&c->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void **)&c->counts_dev, (size_t)(world + 1) * TW_GATHER_COUNTS * sizeof(uint64_t));
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->counts_host, ((size_t)(world + 1) * TW_GATHER_COUNTS + 1) * sizeof(uint64_t), hipHostMallocDefault);
    if (e != hipSuccess) { rc = hip_fail(e, "tw_comm_init", __FILE__, __LINE__); tw_comm_destroy(c); return rc; }
    ncclUniqueId nid;
    memcpy(nid.internal, id->bytes, sizeof(nid.internal));
    ncclResult_t ne = g_rccl.CommInitRank(&c->comm, world, nid, rank);
    if (ne != ncclSuccess) { rc = nccl_fail(ne, "ncclCommInitRank", __LINE__); c->comm = nullptr; tw_comm_destroy(c); return rc; }
    *out = c;
    return TW_OK;
}

extern "C" void tw_comm_destroy(tw_comm *c)
{
    if (!c) return;
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    if (c->counts_dev) (void)hipFree(c->counts_dev);
    if (c->counts_host) (void)hipHostFree(c->counts_host);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int tw_comm_set_timeout_ms(tw_comm *c, uint32_t ms)
{
    if (!c) { set_error("tw_comm_set_timeout_ms: null communicator"); return TW_ERR_INVALID; }
    c->timeout_ms = ms;
    return TW_OK;
}

extern "C" int tw_comm_rank(const tw_comm *c) { return c ? c->rank : -1; }
extern "C" int tw_comm_world(const tw_comm *c) { return c ? c->world : 0; }

extern "C" int tw_comm_broadcast_policy(tw_comm *c, tw_policy *p, int root)
{
    if (!c || !p || root < 0 || root >= c->world) { set_error("tw_comm_broadcast_policy: bad argument"); return TW_ERR_INVALID; }
    if (c->dead) { set_error("tw_comm_broadcast_policy: the communicator was aborted"); return TW_ERR_HIP; }
    void *img = nullptr; size_t bytes = 0;
    int rc = policy_device_image(p, &img, &bytes); if (rc) return rc;
    // the collectors' stream may still read the image: the broadcast is ordered behind it and waited for
    hipStream_t s = current_stream();
    TW_NCCL(g_rccl.Broadcast(img, img, bytes, ncclChar, root, c->comm, s));
    rc = policy_restore_local_tables(p, s); if (rc) return rc;
    TW_HIP(hipStreamSynchronize(s));
    return TW_OK;
}

// ---------------------------------------------------------------------------------------------------------------- gather
struct tw_gather {
    tw_comm *c = nullptr;
    int root = 0, is_ppo = 1;
    uint32_t n_cells = 0, n_actions = 0, obs_width = 0;       // n_actions / obs_width: taken from the first non-empty chunk any rank submits
    uint64_t total_episodes = 0;
    tw_gather_state st{};                  // steps, step, pos, front, cap, tail: advanced by tw_gather_plan
    size_t width[TW_F_COUNT] = {};
    // root: the result, every field allocated at capacity; pointers are fixed up in finish()
    void *arena = nullptr; size_t arena_bytes = 0;
    uint8_t *base[TW_F_COUNT] = {};
    uint32_t *ep_len = nullptr; uint64_t *ep_start = nullptr; uint64_t *scan_total = nullptr; void *scan_scratch = nullptr;
    bool allocated = false;
};

extern "C" int tw_gather_begin(tw_comm *c, int root, uint32_t steps, uint64_t max_records, uint32_t max_episode_records,
                               uint64_t total_episodes, int is_ppo, uint32_t n_cells, tw_gather **out)
{
    if (!c || !out || root < 0 || root >= c->world || steps == 0 || n_cells == 0 || n_cells > 64 || total_episodes == 0) {
        set_error("tw_gather_begin: bad argument"); return TW_ERR_INVALID;
    }
    if (c->dead) { set_error("tw_gather_begin: the communicator was aborted"); return TW_ERR_HIP; }
    if (steps > 1 && (max_records == 0 || max_episode_records == 0)) {
        set_error("tw_gather_begin: more than one step needs max_records and max_episode_records (the result is allocated before the totals are known)");
        return TW_ERR_INVALID;
    }
    tw_gather *g = new tw_gather();
    g->c = c; g->root = root; g->is_ppo = is_ppo ? 1 : 0; g->n_cells = n_cells; g->total_episodes = total_episodes;
    g->st.steps = steps; g->st.max_records = max_records; g->st.max_episode_records = max_episode_records;
    *out = g;
    return TW_OK;
}

static int gather_alloc(tw_gather *g)
{
    const uint64_t E = g->total_episodes;
    size_t cur = 0, off[TW_F_COUNT] = {};
    auto seg = [&](size_t bytes) { size_t o = cur; cur = (cur + bytes + 255) / 256 * 256; return o; };
    for (int f = 0; f < TW_F_COUNT; ++f) if (g->width[f]) off[f] = seg((size_t)g->st.cap * g->width[f] + 16);
    const size_t o_len = seg(E * 4), o_start = seg(E * 8), o_tot = seg(8), o_scan = seg(scan_scratch_bytes(E));
    TW_HIP(hipMalloc(&g->arena, cur));
    g->arena_bytes = cur;
    uint8_t *a = reinterpret_cast<uint8_t *>(g->arena);
    for (int f = 0; f < TW_F_COUNT; ++f) g->base[f] = g->width[f] ? a + off[f] : nullptr;
    g->ep_len = reinterpret_cast<uint32_t *>(a + o_len); g->ep_start = reinterpret_cast<uint64_t *>(a + o_start);
    g->scan_total = reinterpret_cast<uint64_t *>(a + o_tot); g->scan_scratch = a + o_scan;
    g->allocated = true;
    return TW_OK;
}

// Waits for everything queued on the exchange stream.  With tw_comm_set_timeout_ms the wait is bounded: a peer that failed or was
// killed never posts what this rank waits for, so after the limit the communicator is aborted and the call fails (TW_ERR_HIP)
// instead of hanging the host.
static int wait_exchange(tw_comm *c, const char *what)
{
    hipError_t e = hipSuccess;
    if (c->timeout_ms) {
        const auto t0 = std::chrono::steady_clock::now();
        while ((e = hipStreamQuery(c->stream)) == hipErrorNotReady) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(c->timeout_ms)) break;
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        if (e == hipErrorNotReady) {
            (void)hipGetLastError();
            set_error("%s did not complete within %u ms (a peer failed?); communicator aborted", what, c->timeout_ms);
            comm_abort(c);                                   // (ncclCommAbort returns once this rank's RCCL kernels have ended)
            (void)hipStreamSynchronize(c->stream);           // what was queued behind them (copies into host memory of the caller) drains
            return TW_ERR_HIP;
        }
    } else e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return hip_fail(e, what, __FILE__, __LINE__);
    return TW_OK;
}

// one ncclAllGather of TW_GATHER_COUNTS numbers per rank on the exchange stream, read back
static int exchange_counts(tw_comm *c, const uint64_t (&mine)[TW_GATHER_COUNTS], std::vector<uint64_t> &all)
{
    hipStream_t s = c->stream;
    const size_t n_all = (size_t)c->world * TW_GATHER_COUNTS;
    memcpy(c->counts_host, mine, sizeof(mine));
    TW_HIP(hipMemcpyAsync(c->counts_dev, c->counts_host, sizeof(mine), hipMemcpyHostToDevice, s));
    TW_NCCL(g_rccl.AllGather(c->counts_dev, c->counts_dev + TW_GATHER_COUNTS, TW_GATHER_COUNTS, ncclUint64, c->comm, s));
    TW_HIP(hipMemcpyAsync(c->counts_host + TW_GATHER_COUNTS, c->counts_dev + TW_GATHER_COUNTS, n_all * 8, hipMemcpyDeviceToHost, s));
    int rc = wait_exchange(c, "the count exchange");          // bounded: a peer that died never joins the all-gather
    if (rc) return rc;
    all.assign(c->counts_host + TW_GATHER_COUNTS, c->counts_host + TW_GATHER_COUNTS + n_all);
    return TW_OK;
}

extern "C" int tw_gather_submit(tw_gather *g, const tw_collected *local, uint64_t episode_offset)
{
    if (!g) { set_error("tw_gather_submit: null gather"); return TW_ERR_INVALID; }
    if (g->st.step >= g->st.steps) { set_error("tw_gather_submit: more submits than steps"); return TW_ERR_INVALID; }
    tw_comm *c = g->c;
    if (c->dead) { set_error("tw_gather_submit: the communicator was aborted"); return TW_ERR_HIP; }
    const int world = c->world, rank = c->rank;
    // A chunk that does not fit the gather is reported THROUGH the count exchange (status word), so that every rank sees it and
    // all of them return the error together: a rank that left before the collective would leave the others waiting in it.
    uint64_t status = 0;
    int is_ppo = g->is_ppo; uint32_t nc = g->n_cells, na = 0, ow = 0; uint64_t n_local = 0, e_local = 0;
    if (local) {
        int rc = collected_describe(local, &is_ppo, &nc, &na, &n_local, &e_local);
        if (rc || is_ppo != g->is_ppo || nc != g->n_cells) status = 1;          // layout differs from tw_gather_begin's
        ow = tw_collected_obs_width(local);
        if (!status && n_local == 0) { na = 0; ow = 0; }                        // an empty chunk has no say in the layout
    }
    // (records, records of the chunk's last episode, episodes, first global episode, bytes per obs id, actions, status) of every rank
    uint64_t mine[TW_GATHER_COUNTS] = {status ? 0 : n_local, 0, status ? 0 : e_local, episode_offset, ow, na, status, 0};
    hipStream_t s = c->stream;
    if (local && e_local && !status) {
        // (behind the previous step's transfer on the exchange stream: the wait is the bounded one)
        uint32_t *last_len = reinterpret_cast<uint32_t *>(c->counts_host + (size_t)(world + 1) * TW_GATHER_COUNTS);
        *last_len = 0;
        hipError_t e = hipMemcpyAsync(last_len, reinterpret_cast<const uint32_t *>(collected_field(local, TW_F_EP_LEN)) + (e_local - 1), 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) { const int wrc = wait_exchange(c, "tw_gather_submit: the previous step's transfer"); if (wrc) return wrc; }
        if (e != hipSuccess) { (void)hip_fail(e, "tw_gather_submit: reading the last episode length", __FILE__, __LINE__); mine[0] = mine[2] = 0; mine[6] = status = 2; }
        mine[1] = *last_len;
    }
    std::vector<uint64_t> all;
    int rc = exchange_counts(c, mine, all);
    if (rc) { comm_abort(c); return rc; }
    auto cnt = [&](int r, int k) { return all[(size_t)r * TW_GATHER_COUNTS + k]; };
    for (int r = 0; r < world; ++r) if (cnt(r, 6)) {
        if (r != rank || status == 1) set_error("tw_gather_submit: rank %d reports a chunk that does not fit the gather (status %llu)", r, (unsigned long long)cnt(r, 6));
        return r == rank && status == 2 ? TW_ERR_HIP : TW_ERR_INVALID;
    }
    // layout of the records: from the first non-empty chunk any rank has submitted; every later one has to agree
    for (int r = 0; r < world; ++r) if (cnt(r, 0)) {
        if (!g->obs_width) { g->obs_width = (uint32_t)cnt(r, 4); g->n_actions = (uint32_t)cnt(r, 5); }
        if (cnt(r, 4) != g->obs_width || cnt(r, 5) != g->n_actions || (g->obs_width != 1 && g->obs_width != 2) || g->n_actions == 0) {
            set_error("tw_gather_submit: rank %d holds %llu-byte obs ids and %llu actions, the gather %u and %u", r, (unsigned long long)cnt(r, 4),
                      (unsigned long long)cnt(r, 5), g->obs_width, g->n_actions);
            return TW_ERR_INVALID;
        }
    }
    if (g->obs_width) gather_field_widths(g->is_ppo, g->n_cells, g->obs_width, g->n_actions, g->width);

    // placement: the pure host function the CPU tests exercise at world 2 / 3 / 4 / 8
    std::vector<uint32_t> np((size_t)world);
    std::vector<tw_gather_piece> pc((size_t)world * 2);
    int32_t tail_rank = -1;
    const bool first = g->st.step == 0;
    rc = tw_gather_plan(&g->st, world, all.data(), &tail_rank, np.data(), pc.data());
    if (rc) return rc;                                        // the same on every rank (same numbers)
    if (first || (rank == g->root && !g->allocated && g->obs_width)) {
        // the root's allocation is the one step that can fail on one rank alone: its outcome goes round before anybody sends
        uint64_t st2[TW_GATHER_COUNTS] = {0};
        if (rank == g->root && !g->allocated && g->obs_width) { rc = gather_alloc(g); st2[6] = rc ? 3 : 0; }
        if (first) {
            std::vector<uint64_t> all2;
            const int rc2 = exchange_counts(c, st2, all2);
            if (rc2) { comm_abort(c); return rc2; }
            if (all2[(size_t)g->root * TW_GATHER_COUNTS + 6]) {
                if (rank != g->root) set_error("tw_gather_submit: the root could not allocate the result");
                return rank == g->root ? rc : TW_ERR_HIP;
            }
        } else if (rc) { comm_abort(c); return rc; }
    }

    // a failure inside the group closes it (RCCL keeps a thread-local group depth) and takes the communicator down
    auto transfers = [&]() -> int {
        if (rank == g->root) {
            for (int r = 0; r < world; ++r) {
                const uint64_t er = cnt(r, 2), eo = cnt(r, 3);
                if (er && (eo > g->total_episodes || er > g->total_episodes - eo)) { set_error("tw_gather_submit: episodes [%llu, +%llu) of rank %d lie outside the gathered range", (unsigned long long)eo, (unsigned long long)er, r); return TW_ERR_INVALID; }
                for (uint32_t i = 0; i < np[(size_t)r]; ++i) {
                    const tw_gather_piece &q = pc[(size_t)r * 2 + i];
                    for (int f = 0; f < TW_F_COUNT; ++f) if (g->width[f]) {
                        uint8_t *dst = g->base[f] + q.dst * g->width[f];
                        const size_t bytes = (size_t)(q.src_hi - q.src_lo) * g->width[f];
                        if (r == rank) TW_HIP(hipMemcpyAsync(dst, reinterpret_cast<const uint8_t *>(collected_field(local, f)) + q.src_lo * g->width[f], bytes, hipMemcpyDeviceToDevice, s));
                        else TW_NCCL(g_rccl.Recv(dst, bytes, ncclChar, r, c->comm, s));
                    }
                }
                if (er && r == rank) TW_HIP(hipMemcpyAsync(g->ep_len + eo, collected_field(local, TW_F_EP_LEN), er * 4, hipMemcpyDeviceToDevice, s));
                else if (er) TW_NCCL(g_rccl.Recv(g->ep_len + eo, er, ncclUint32, r, c->comm, s));
            }
        } else {
            for (uint32_t i = 0; i < np[(size_t)rank]; ++i) {
                const tw_gather_piece &q = pc[(size_t)rank * 2 + i];
                for (int f = 0; f < TW_F_COUNT; ++f) if (g->width[f])
                    TW_NCCL(g_rccl.Send(reinterpret_cast<const uint8_t *>(collected_field(local, f)) + q.src_lo * g->width[f], (size_t)(q.src_hi - q.src_lo) * g->width[f],
                                        ncclChar, g->root, c->comm, s));
            }
            if (cnt(rank, 2)) TW_NCCL(g_rccl.Send(collected_field(local, TW_F_EP_LEN), cnt(rank, 2), ncclUint32, g->root, c->comm, s));
        }
        return TW_OK;
    };
    TW_NCCL(g_rccl.GroupStart());
    const int trc = transfers();
    const ncclResult_t gend = g_rccl.GroupEnd();
    if (trc) { comm_abort(c); return trc; }
    if (gend != ncclSuccess) { rc = nccl_fail(gend, "ncclGroupEnd", __LINE__); comm_abort(c); return rc; }
    return TW_OK;
}

extern "C" int tw_gather_finish(tw_gather *g, tw_collected **merged)
{
    if (!g) { set_error("tw_gather_finish: null gather"); return TW_ERR_INVALID; }
    if (merged) *merged = nullptr;
    tw_comm *c = g->c;
    int rc = TW_OK;
    if (c->dead) { set_error("tw_gather_finish: the communicator was aborted"); rc = TW_ERR_HIP; }
    else if (g->st.step != g->st.steps) { set_error("tw_gather_finish: %u of %u steps submitted", g->st.step, g->st.steps); rc = TW_ERR_INVALID; }
    hipError_t e = hipSuccess;
    if (rc == TW_OK) rc = wait_exchange(c, "tw_gather_finish: the transfers");
    else if (!c->dead) (void)wait_exchange(c, "tw_gather_finish");     // (steps missing: what WAS posted still targets the arena freed below)
    if (rc == TW_OK && c->rank == g->root && merged) {
        if (!g->allocated) { set_error("tw_gather_finish: no records were submitted"); rc = TW_ERR_EMPTY; }
    }
    if (rc == TW_OK && c->rank == g->root && merged) {
        const uint64_t total = g->st.pos + g->st.tail, a0 = g->st.front - g->st.tail, E = g->total_episodes;
        // first record of every episode in the merged order (the scan of tw_finalize.hip over the gathered lengths)
        rc = launch_scan(g->ep_len, E, 1, g->ep_start, g->scan_total, g->scan_scratch, scan_scratch_bytes(E), c->stream);
        if (rc == TW_OK) { e = hipStreamSynchronize(c->stream); if (e != hipSuccess) rc = hip_fail(e, "tw_gather_finish: scan", __FILE__, __LINE__); }
        if (rc == TW_OK) {
            void *fp[TW_F_COUNT] = {}; size_t fb[TW_F_COUNT] = {};
            for (int f = 0; f < TW_F_COUNT; ++f) if (g->width[f]) { fp[f] = g->base[f] + a0 * g->width[f]; fb[f] = (size_t)total * g->width[f]; }
            fp[TW_F_EP_LEN] = g->ep_len; fb[TW_F_EP_LEN] = E * 4; fp[TW_F_EP_START] = g->ep_start; fb[TW_F_EP_START] = E * 8;
            rc = collected_adopt(g->arena, g->arena_bytes, c->device, g->is_ppo, g->n_cells, g->n_actions, total, E, fp, fb, merged);
            if (rc == TW_OK) { g->arena = nullptr; collected_adopt_obs_width(*merged, g->obs_width); }    // owned by the result now
        }
    }
    if (g->arena) (void)hipFree(g->arena);
    delete g;
    return rc;
}
