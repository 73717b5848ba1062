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

#include <cstring>
#include <mutex>
#include <vector>

namespace tw {

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
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
    // a process that already holds an RCCL (PyTorch-ROCm bundles one) must keep using THAT one
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
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
void gather_field_widths(int is_ppo, uint32_t n_cells, uint32_t n_actions, size_t (&w)[TW_F_COUNT])
{
    for (int f = 0; f < TW_F_COUNT; ++f) w[f] = 0;
    w[TW_F_OBS] = n_cells; w[TW_F_LOGITS] = (size_t)n_actions * 4; w[TW_F_PERMS] = 1;
    if (is_ppo) { w[TW_F_VALUES] = 4; w[TW_F_REWARDS] = 4; w[TW_F_ACTIONS] = 1; w[TW_F_ADVS] = 4; w[TW_F_RETS] = 4; }
    else w[TW_F_REMAINING] = 4;
}

}  // namespace tw

using namespace tw;

struct tw_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;          // the exchange runs beside the collectors' stream
    uint64_t *counts_dev = nullptr;        // [4] mine | [world][4] all
};

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
    if (e == hipSuccess) e = hipMalloc((void **)&c->counts_dev, (size_t)(world + 1) * 4 * sizeof(uint64_t));
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
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int tw_comm_rank(const tw_comm *c) { return c ? c->rank : -1; }
extern "C" int tw_comm_world(const tw_comm *c) { return c ? c->world : 0; }

extern "C" int tw_comm_broadcast_policy(tw_comm *c, tw_policy *p, int root)
{
    if (!c || !p || root < 0 || root >= c->world) { set_error("tw_comm_broadcast_policy: bad argument"); return TW_ERR_INVALID; }
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
    uint32_t steps = 1, step = 0, n_cells = 0, n_actions = 4, max_ep = 0;
    uint64_t max_records = 0, total_episodes = 0;
    uint64_t pos = 0, front = 0, tail = 0, cap = 0;
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
    if (steps > 1 && (max_records == 0 || max_episode_records == 0)) {
        set_error("tw_gather_begin: more than one step needs max_records and max_episode_records (the result is allocated before the totals are known)");
        return TW_ERR_INVALID;
    }
    tw_gather *g = new tw_gather();
    g->c = c; g->root = root; g->steps = steps; g->max_records = max_records; g->max_ep = max_episode_records;
    g->total_episodes = total_episodes; g->is_ppo = is_ppo ? 1 : 0; g->n_cells = n_cells;
    gather_field_widths(g->is_ppo, n_cells, 4, g->width);
    *out = g;
    return TW_OK;
}

static int gather_alloc(tw_gather *g, uint64_t total_first, uint64_t tail_first)
{
    // one step: everything is known (exact size, the tail goes to the very front); several: capacity + slack for the tail
    if (g->steps == 1) { g->front = tail_first; g->cap = total_first; }
    else { g->front = g->max_ep; g->cap = g->max_records + g->max_ep; }
    const uint64_t E = g->total_episodes;
    size_t cur = 0, off[TW_F_COUNT] = {};
    auto seg = [&](size_t bytes) { size_t o = cur; cur = (cur + bytes + 255) / 256 * 256; return o; };
    for (int f = 0; f < TW_F_COUNT; ++f) if (g->width[f]) off[f] = seg((size_t)g->cap * g->width[f] + 16);
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

extern "C" int tw_gather_submit(tw_gather *g, const tw_collected *local, uint64_t episode_offset)
{
    if (!g) { set_error("tw_gather_submit: null gather"); return TW_ERR_INVALID; }
    if (g->step >= g->steps) { set_error("tw_gather_submit: more submits than steps"); return TW_ERR_INVALID; }
    tw_comm *c = g->c;
    const int world = c->world, rank = c->rank;
    const bool last_step = g->step + 1 == g->steps;
    int is_ppo = g->is_ppo; uint32_t nc = g->n_cells, na = 4; uint64_t n_local = 0, e_local = 0;
    if (local) {
        int rc = collected_describe(local, &is_ppo, &nc, &na, &n_local, &e_local); if (rc) return rc;
        if (is_ppo != g->is_ppo || nc != g->n_cells) { set_error("tw_gather_submit: the chunk's layout differs from tw_gather_begin's"); return TW_ERR_INVALID; }
    }
    // (records, records of the chunk's last episode, episodes, first global episode) of every rank
    uint64_t mine[4] = {n_local, 0, e_local, episode_offset};
    hipStream_t s = c->stream;
    if (local && e_local) {
        uint32_t last_len = 0;
        TW_HIP(hipMemcpyAsync(&last_len, reinterpret_cast<const uint32_t *>(collected_field(local, TW_F_EP_LEN)) + (e_local - 1), 4, hipMemcpyDeviceToHost, s));
        TW_HIP(hipStreamSynchronize(s));
        mine[1] = last_len;
    }
    TW_HIP(hipMemcpyAsync(c->counts_dev, mine, sizeof(mine), hipMemcpyHostToDevice, s));
    TW_NCCL(g_rccl.AllGather(c->counts_dev, c->counts_dev + 4, 4, ncclUint64, c->comm, s));
    std::vector<uint64_t> all((size_t)world * 4);
    TW_HIP(hipMemcpyAsync(all.data(), c->counts_dev + 4, all.size() * 8, hipMemcpyDeviceToHost, s));
    TW_HIP(hipStreamSynchronize(s));

    // episode E-1 is the last episode of the last non-empty chunk of the last step
    uint64_t tail = 0; int tail_rank = -1;
    if (last_step)
        for (int r = world - 1; r >= 0; --r) if (all[(size_t)r * 4] > 0) { tail = all[(size_t)r * 4 + 1]; tail_rank = r; break; }
    if (rank == g->root && !g->allocated) {
        uint64_t total = 0; for (int r = 0; r < world; ++r) total += all[(size_t)r * 4];
        int rc = gather_alloc(g, total, tail); if (rc) return rc;
    }
    if (last_step) g->tail = tail;
    std::vector<uint64_t> pos_of((size_t)world);
    uint64_t p = g->pos;
    for (int r = 0; r < world; ++r) { pos_of[(size_t)r] = p; p += all[(size_t)r * 4] - (r == tail_rank ? tail : 0); }
    if (rank == g->root && g->front + p > g->cap) { set_error("tw_gather_submit: %llu records exceed max_records", (unsigned long long)p); return TW_ERR_INVALID; }

    struct Piece { uint64_t lo, hi, dst; };
    auto pieces = [&](int r, Piece (&out)[2]) -> int {
        const uint64_t n = all[(size_t)r * 4];
        if (n == 0) return 0;
        if (r == tail_rank) {
            const uint64_t body = n - tail;
            out[0] = Piece{body, n, g->front - tail};
            if (body == 0) return 1;
            out[1] = Piece{0, body, g->front + pos_of[(size_t)r]};
            return 2;
        }
        out[0] = Piece{0, n, g->front + pos_of[(size_t)r]};
        return 1;
    };

    // (a failure inside the group still closes it: RCCL keeps a thread-local group depth)
    auto transfers = [&]() -> int {
        if (rank == g->root) {
            for (int r = 0; r < world; ++r) {
                Piece pc[2]; const int np = pieces(r, pc);
                const uint64_t er = all[(size_t)r * 4 + 2], eo = all[(size_t)r * 4 + 3];
                if (r == rank) {
                    for (int i = 0; i < np; ++i)
                        for (int f = 0; f < TW_F_COUNT; ++f) if (g->width[f])
                            TW_HIP(hipMemcpyAsync(g->base[f] + pc[i].dst * g->width[f], reinterpret_cast<const uint8_t *>(collected_field(local, f)) + pc[i].lo * g->width[f],
                                                  (pc[i].hi - pc[i].lo) * g->width[f], hipMemcpyDeviceToDevice, s));
                    if (er) TW_HIP(hipMemcpyAsync(g->ep_len + eo, collected_field(local, TW_F_EP_LEN), er * 4, hipMemcpyDeviceToDevice, s));
                } else {
                    for (int i = 0; i < np; ++i)
                        for (int f = 0; f < TW_F_COUNT; ++f) if (g->width[f])
                            TW_NCCL(g_rccl.Recv(g->base[f] + pc[i].dst * g->width[f], (pc[i].hi - pc[i].lo) * g->width[f], ncclChar, r, c->comm, s));
                    if (er) TW_NCCL(g_rccl.Recv(g->ep_len + eo, er, ncclUint32, r, c->comm, s));
                }
            }
        } else {
            Piece pc[2]; const int np = pieces(rank, pc);
            for (int i = 0; i < np; ++i)
                for (int f = 0; f < TW_F_COUNT; ++f) if (g->width[f])
                    TW_NCCL(g_rccl.Send(reinterpret_cast<const uint8_t *>(collected_field(local, f)) + pc[i].lo * g->width[f], (pc[i].hi - pc[i].lo) * g->width[f],
                                        ncclChar, g->root, c->comm, s));
            if (e_local) TW_NCCL(g_rccl.Send(collected_field(local, TW_F_EP_LEN), e_local, ncclUint32, g->root, c->comm, s));
        }
        return TW_OK;
    };
    TW_NCCL(g_rccl.GroupStart());
    const int trc = transfers();
    const ncclResult_t gend = g_rccl.GroupEnd();
    if (trc) return trc;
    if (gend != ncclSuccess) return nccl_fail(gend, "ncclGroupEnd", __LINE__);
    g->pos = p;
    g->step += 1;
    return TW_OK;
}

extern "C" int tw_gather_finish(tw_gather *g, tw_collected **merged)
{
    if (!g) { set_error("tw_gather_finish: null gather"); return TW_ERR_INVALID; }
    if (merged) *merged = nullptr;
    tw_comm *c = g->c;
    int rc = TW_OK;
    if (g->step != g->steps) { set_error("tw_gather_finish: %u of %u steps submitted", g->step, g->steps); rc = TW_ERR_INVALID; }
    hipError_t e = hipStreamSynchronize(c->stream);
    if (rc == TW_OK && e != hipSuccess) rc = hip_fail(e, "tw_gather_finish", __FILE__, __LINE__);
    if (rc == TW_OK && c->rank == g->root && merged) {
        const uint64_t total = g->pos + g->tail, a0 = g->front - g->tail, E = g->total_episodes;
        // first record of every episode in the merged order (the scan of tw_finalize.hip over the gathered lengths)
        rc = launch_scan(g->ep_len, E, 1, g->ep_start, g->scan_total, g->scan_scratch, scan_scratch_bytes(E), c->stream);
        if (rc == TW_OK) { e = hipStreamSynchronize(c->stream); if (e != hipSuccess) rc = hip_fail(e, "tw_gather_finish: scan", __FILE__, __LINE__); }
        if (rc == TW_OK) {
            void *fp[TW_F_COUNT] = {}; size_t fb[TW_F_COUNT] = {};
            for (int f = 0; f < TW_F_COUNT; ++f) if (g->width[f]) { fp[f] = g->base[f] + a0 * g->width[f]; fb[f] = (size_t)total * g->width[f]; }
            fp[TW_F_EP_LEN] = g->ep_len; fb[TW_F_EP_LEN] = E * 4; fp[TW_F_EP_START] = g->ep_start; fb[TW_F_EP_START] = E * 8;
            rc = collected_adopt(g->arena, g->arena_bytes, c->device, g->is_ppo, g->n_cells, 4, total, E, fp, fb, merged);
            if (rc == TW_OK) g->arena = nullptr;         // owned by the result now
        }
    }
    if (g->arena) (void)hipFree(g->arena);
    delete g;
    return rc;
}
