// tw_api.hip -- C ABI (include/twisterl_hip.h): host-side logic of the collector library.
//
// Host mirror of the reference's compiled (Rust) layer for the hot path:
//   * tw_puzzle_*      Env for Puzzle         rust/src/envs/puzzle.rs:20-185 (single-env API
//                                             behind PyBaseEnv, python_interface/env.rs:44-160)
//   * tw_policy_*      Policy/Linear/EmbeddingBag/Sequential ctors, nn/policy.rs:29-32
//   * tw_ppo_collect   PPOCollector::collect  rust/src/collector/ppo.rs:108-126 + merge
//                                             (collector/collector.rs:40-46)
// All arithmetic of the hot path runs in the HIP kernels; nothing here computes a trajectory.
#include "tw_common.hpp"

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

namespace tw {

// ---------------------------------------------------------------------------------- errors
static thread_local std::string g_err;
static thread_local hipStream_t g_stream = nullptr;

void set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap; va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

int hip_fail(hipError_t e, const char *what, const char *file, int line)
{
    set_error("HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
    (void)hipGetLastError();
    return (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? TW_ERR_NO_DEVICE : TW_ERR_HIP;
}

hipStream_t current_stream() { return g_stream; }

// ---------------------------------------------------------------------------------- launch options / per-device caches
static std::atomic<int> g_force_geom{0}, g_no_persist{0}, g_az_variant{0}, g_az_tree_budget{0}, g_az_tree_budget_min{0}, g_az_reuse{0};
LaunchOptions launch_options() { return LaunchOptions{g_force_geom.load(), g_no_persist.load(), g_az_variant.load(), g_az_tree_budget.load(), g_az_tree_budget_min.load(), g_az_reuse.load()}; }
// counters of the last self-play launch (eval_count[0..15], see MctsArgs): tw_debug_counters
static std::mutex g_dbg_mutex;
static unsigned long long g_dbg_counters[16];

static std::mutex g_dev_mutex;

int device_cus()
{
    static std::map<int, int> cus;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
    std::lock_guard<std::mutex> lock(g_dev_mutex);
    auto it = cus.find(dev);
    if (it != cus.end()) return it->second;
    hipDeviceProp_t p;
    int n = 256;
    if (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) n = p.multiProcessorCount;
    else (void)hipGetLastError();
    cus[dev] = n;
    return n;
}

int ensure_dynamic_lds(const void *kernel, size_t bytes)
{
    static std::map<std::pair<const void *, int>, size_t> granted;
    int dev = 0; TW_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_dev_mutex);
    size_t &have = granted[std::make_pair(kernel, dev)];
    if (bytes > have) {
        TW_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        have = bytes;
    }
    return TW_OK;
}

int require_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        set_error("no HIP device available (hipGetDeviceCount: %s); this library has no CPU fallback",
                  e == hipSuccess ? "0 devices" : hipGetErrorString(e));
        return TW_ERR_NO_DEVICE;
    }
    return TW_OK;
}

// ---------------------------------------------------------------------------------- workspace
// Padded trajectory buffers are large (E * t_pad * 48 B) and have the same size every
// iteration of the trainer, so they are cached per device instead of hipMalloc'ed per call.
struct Workspace {
    void *ptr = nullptr; size_t cap = 0; int device = -1;
};
static std::mutex g_ws_mutex;
static Workspace g_ws;
static void pool_drop();     // frees every pooled result arena (below)

static int ws_reserve(size_t bytes, void **out)
{
    int dev = 0; TW_HIP(hipGetDevice(&dev));
    if (g_ws.ptr && (g_ws.device != dev || g_ws.cap < bytes)) {
        TW_HIP(hipFree(g_ws.ptr)); g_ws.ptr = nullptr; g_ws.cap = 0;
    }
    if (!g_ws.ptr) {
        hipError_t e = hipMalloc(&g_ws.ptr, bytes);
        if (e != hipSuccess) { (void)hipGetLastError(); pool_drop(); e = hipMalloc(&g_ws.ptr, bytes); }   // pooled result arenas go first
        if (e != hipSuccess) { g_ws.ptr = nullptr; return hip_fail(e, "hipMalloc(trajectory workspace)", __FILE__, __LINE__); }
        g_ws.cap = bytes; g_ws.device = dev;
    }
    *out = g_ws.ptr;
    return TW_OK;
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Result arenas (the compact trajectories a collect hands to its caller) have the same size every iteration of the
// trainer too: a freed arena goes to a small per-process pool instead of back to the driver (hipMalloc / hipFree of a
// few GB cost milliseconds each and serialise with every stream).
// hipFree waits for the device; the pool does not.  A released arena therefore carries an event recorded on the library's stream
// at release, and whoever takes it out of the pool makes its stream wait for that event: work the library itself had queued on
// the arena (a gather's copies, a pack kernel) is finished before the next collect writes into it.  Consumers on OTHER streams
// (zero-copy torch tensors on a side stream) must be finished before tw_collected_free -- as they would have to be before a
// hipFree they do not wait for.
struct PooledArena { void *ptr; size_t cap; int device; hipEvent_t released; };
static std::mutex g_pool_mutex;
static std::vector<PooledArena> g_pool;
constexpr size_t POOL_ENTRIES = 16;     // (a pipelined multi-GPU step keeps one arena per pipeline step alive until its gather is done)
constexpr size_t POOL_BYTES = (size_t)48 << 30;      // memory the pool may hold back from other allocators (torch does not see it: tw_release_cached_memory frees it)

static void pool_free(PooledArena &a) { (void)hipFree(a.ptr); if (a.released) (void)hipEventDestroy(a.released); }

static void pool_drop()
{
    std::vector<PooledArena> drop;
    { std::lock_guard<std::mutex> lock(g_pool_mutex); drop.swap(g_pool); }
    for (auto &a : drop) pool_free(a);
}

static int arena_acquire(size_t bytes, void **out, size_t *cap)
{
    int dev = 0; TW_HIP(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        size_t best = g_pool.size();
        for (size_t i = 0; i < g_pool.size(); ++i)
            if (g_pool[i].device == dev && g_pool[i].cap >= bytes && g_pool[i].cap <= 2 * bytes + (1u << 20) &&
                (best == g_pool.size() || g_pool[i].cap < g_pool[best].cap)) best = i;
        if (best != g_pool.size()) {
            const PooledArena a = g_pool[best];
            g_pool.erase(g_pool.begin() + (long)best);
            *out = a.ptr; *cap = a.cap;
            if (a.released) {
                const hipError_t e = hipStreamWaitEvent(current_stream(), a.released, 0);
                (void)hipEventDestroy(a.released);
                if (e != hipSuccess) { (void)hipFree(a.ptr); return hip_fail(e, "hipStreamWaitEvent(pooled arena)", __FILE__, __LINE__); }
            }
            return TW_OK;
        }
    }
    // a new arena gets 1/16 of headroom: the record count of a collect varies with the seed by a fraction of a percent, and an
    // arena that is a few KB short of the next result is a pool miss -- a 4 GB hipMalloc is 60 ms of a 142 ms step
    size_t want = bytes ? bytes + bytes / 16 + 256 : 256;
    hipError_t e = hipMalloc(out, want);
    if (e != hipSuccess) {      // out of memory: give the pooled arenas back and try once more, without the headroom
        (void)hipGetLastError();
        pool_drop();
        want = bytes ? bytes : 256;
        e = hipMalloc(out, want);
    }
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(compact result)", __FILE__, __LINE__);
    *cap = want;
    return TW_OK;
}

// `producer`: the stream the result was produced on.  tw_set_stream is thread-local, and a result may be freed by another thread
// (a Python finalizer, a thread that never called tw_set_stream): the fence has to cover what the PRODUCING stream still has
// queued on the arena as well as what the freeing thread's stream has (a pack kernel, a gather's copies), so the producer is
// made to wait for the freeing stream and the release event is recorded on the producer.
static void arena_release(void *ptr, size_t cap, int device, hipStream_t producer)
{
    if (!ptr) return;
    hipEvent_t ev = nullptr;
    const hipStream_t here = current_stream();
    bool ok = hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
    if (ok && here != producer) ok = hipEventRecord(ev, here) == hipSuccess && hipStreamWaitEvent(producer, ev, 0) == hipSuccess;
    if (!ok || hipEventRecord(ev, producer) != hipSuccess) {
        (void)hipGetLastError();
        if (ev) (void)hipEventDestroy(ev);
        (void)hipFree(ptr);                          // no fence to hand on: let the driver's own (synchronising) free do it
        return;
    }
    std::vector<PooledArena> victims;
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        g_pool.push_back(PooledArena{ptr, cap, device, ev});
        size_t held = 0;
        for (auto &a : g_pool) held += a.cap;
        while (g_pool.size() > POOL_ENTRIES || (held > POOL_BYTES && g_pool.size() > 1)) {          // evict the smallest
            size_t v = 0;
            for (size_t i = 1; i < g_pool.size(); ++i) if (g_pool[i].cap < g_pool[v].cap) v = i;
            held -= g_pool[v].cap;
            victims.push_back(g_pool[v]);
            g_pool.erase(g_pool.begin() + (long)v);
        }
    }
    for (auto &a : victims) pool_free(a);
}

}  // namespace tw

using namespace tw;

// ====================================================================================== misc
extern "C" int tw_abi_version(void) { return TW_ABI_VERSION; }
extern "C" const char *tw_last_error(void) { return g_err.c_str(); }

extern "C" int tw_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

extern "C" int tw_set_device(int device)
{
    int rc = require_device(); if (rc) return rc;
    TW_HIP(hipSetDevice(device));
    return TW_OK;
}

extern "C" int tw_set_launch_option(int option, int value)
{
    switch (option) {
        case TW_OPT_FORCE_GEOM:
            if (value != 0 && value != 1 && value != 8 && value != 32) { set_error("TW_OPT_FORCE_GEOM: value %d not in {0, 1, 8, 32}", value); return TW_ERR_INVALID; }
            g_force_geom.store(value); return TW_OK;
        case TW_OPT_NO_PERSIST: g_no_persist.store(value ? 1 : 0); return TW_OK;
        case TW_OPT_AZ_VARIANT:
            if (value < 0 || (value & 7) > 6 || (value & ~4087) != 0 || (value & 48) == 48 || (value & 384) == 384 || (value & 1536) == 1536) { set_error("TW_OPT_AZ_VARIANT: value %d is not {0 .. 6} (+ 16 | 32) (+ 64) (+ 128 | 256) (+ 512 | 1024) (+ 2048)", value); return TW_ERR_INVALID; }
            g_az_variant.store(value); return TW_OK;
        case TW_OPT_AZ_TREE_BUDGET:
            if (value < 0 || (value != 0 && value < 1000)) { set_error("TW_OPT_AZ_TREE_BUDGET: %d cycles (0 = automatic, else >= 1000)", value); return TW_ERR_INVALID; }
            g_az_tree_budget.store(value); return TW_OK;
        case TW_OPT_AZ_TREE_BUDGET_MIN:
            if (value < 0 || (value != 0 && value < 1000)) { set_error("TW_OPT_AZ_TREE_BUDGET_MIN: %d cycles (0 = automatic, else >= 1000)", value); return TW_ERR_INVALID; }
            g_az_tree_budget_min.store(value); return TW_OK;
        case TW_OPT_AZ_REUSE:
            if (value < 0 || value > 4) { set_error("TW_OPT_AZ_REUSE: value %d not in {0 .. 4}", value); return TW_ERR_INVALID; }
#ifndef TW_ABLATE
            // 2 reads the path level whatever the depth -- another board's output below PATH_DEPTH -- and skips the board check: it
            // returns different bytes (profiles/r03_az_reuse_probe.txt).  Kept for the record in the diagnostic build only.
            if (value == 2) { set_error("TW_OPT_AZ_REUSE: 2 is a diagnostic form (wrong below the path depth); the product build takes 0, 1, 3 or 4"); return TW_ERR_INVALID; }
#endif
            g_az_reuse.store(value); return TW_OK;
        default: set_error("tw_set_launch_option: unknown option %d", option); return TW_ERR_INVALID;
    }
}

extern "C" int tw_debug_counters(uint64_t *out, int n)
{
    if (!out || n < 0) { set_error("tw_debug_counters: null argument"); return TW_ERR_INVALID; }
    std::lock_guard<std::mutex> lock(g_dbg_mutex);
    for (int i = 0; i < n; ++i) out[i] = i < 16 ? g_dbg_counters[i] : 0;
    return TW_OK;
}

extern "C" int tw_release_cached_memory(void)
{
    {
        std::lock_guard<std::mutex> lock(g_ws_mutex);
        if (g_ws.ptr) { (void)hipFree(g_ws.ptr); g_ws.ptr = nullptr; g_ws.cap = 0; g_ws.device = -1; }
    }
    pool_drop();
    return TW_OK;
}

extern "C" int tw_set_stream(void *hip_stream) { g_stream = reinterpret_cast<hipStream_t>(hip_stream); return TW_OK; }

extern "C" int tw_get_device_info(tw_device_info *out)
{
    if (!out) { set_error("tw_get_device_info: null output"); return TW_ERR_INVALID; }
    int rc = require_device(); if (rc) return rc;
    int dev = 0; TW_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p; TW_HIP(hipGetDeviceProperties(&p, dev));
    memset(out, 0, sizeof(*out));
    snprintf(out->name, sizeof(out->name), "%s", p.name);
    snprintf(out->arch, sizeof(out->arch), "%s", p.gcnArchName);
    out->compute_units = p.multiProcessorCount;
    out->wavefront_size = p.warpSize;
    out->total_mem_bytes = p.totalGlobalMem;
    out->lds_bytes_per_block = p.sharedMemPerBlock;
    return TW_OK;
}

// ====================================================================================== Puzzle (host object)
struct tw_puzzle {
    std::vector<int64_t> state;            // puzzle.rs:21
    int64_t zx = 0, zy = 0, depth = 1;      // puzzle.rs:22-23, depth 1 from Puzzle::new (:41)
    int64_t width = 0, height = 0, difficulty = 0, depth_slope = 0, max_depth = 0;
};

extern "C" tw_puzzle *tw_puzzle_create(uint32_t width, uint32_t height, uint32_t difficulty, uint32_t depth_slope,
                                       uint32_t max_depth)
{
    if (width == 0 || height == 0) { set_error("Puzzle: width and height must be positive"); return nullptr; }
    tw_puzzle *p = new tw_puzzle();
    p->width = width; p->height = height; p->difficulty = difficulty; p->depth_slope = depth_slope; p->max_depth = max_depth;
    p->state.resize((size_t)width * height);
    for (size_t i = 0; i < p->state.size(); ++i) p->state[i] = (int64_t)i;
    return p;
}
extern "C" tw_puzzle *tw_puzzle_clone(const tw_puzzle *p) { return p ? new tw_puzzle(*p) : nullptr; }
extern "C" void tw_puzzle_destroy(tw_puzzle *p) { delete p; }

extern "C" int tw_puzzle_get_desc(const tw_puzzle *p, tw_puzzle_desc *out)
{
    if (!p || !out) { set_error("tw_puzzle_get_desc: null argument"); return TW_ERR_INVALID; }
    out->width = (uint32_t)p->width; out->height = (uint32_t)p->height; out->difficulty = (uint32_t)p->difficulty;
    out->depth_slope = (uint32_t)p->depth_slope; out->max_depth = (uint32_t)p->max_depth;
    return TW_OK;
}
extern "C" uint32_t tw_puzzle_num_actions(const tw_puzzle *) { return 4; }                      // puzzle.rs:90-92
extern "C" int tw_puzzle_obs_shape(const tw_puzzle *p, uint32_t out[2])                          // puzzle.rs:94-97
{
    out[0] = out[1] = (uint32_t)p->state.size(); return TW_OK;
}
extern "C" int tw_puzzle_set_difficulty(tw_puzzle *p, uint32_t d) { p->difficulty = d; return TW_OK; }   // :99-101
extern "C" uint32_t tw_puzzle_get_difficulty(const tw_puzzle *p) { return (uint32_t)p->difficulty; }      // :103-105
extern "C" uint32_t tw_puzzle_depth(const tw_puzzle *p) { return (uint32_t)p->depth; }

extern "C" int tw_puzzle_set_state(tw_puzzle *p, const int64_t *state, size_t n)                 // puzzle.rs:107-117
{
    if (!p || !state) { set_error("set_state: null argument"); return TW_ERR_INVALID; }
    p->state.assign(state, state + n);
    p->depth = p->max_depth;
    for (size_t i = 0; i < n; ++i)
        if (state[i] == 0) { p->zx = (int64_t)i % p->width; p->zy = (int64_t)i / p->width; break; }
    return TW_OK;
}

extern "C" int tw_puzzle_step(tw_puzzle *p, uint32_t action)                                     // puzzle.rs:135-160
{
    const int64_t zx = p->zx, zy = p->zy, w = p->width;
    int64_t nx = zx, ny = zy; bool ok = false;
    if (action == 0 && zx > 0) { nx = zx - 1; ok = true; }
    else if (action == 1 && zy > 0) { ny = zy - 1; ok = true; }
    else if (action == 2 && zx < p->width - 1) { nx = zx + 1; ok = true; }
    else if (action == 3 && zy < p->height - 1) { ny = zy + 1; ok = true; }
    if (ok) {
        const size_t zi = (size_t)(zy * w + zx), ti = (size_t)(ny * w + nx);
        if (zi >= p->state.size() || ti >= p->state.size()) { set_error("step: blank outside the board"); return TW_ERR_INVALID; }
        p->state[zi] = p->state[ti]; p->state[ti] = 0; p->zx = nx; p->zy = ny;
    }
    p->depth = p->depth > 0 ? p->depth - 1 : 0;
    return TW_OK;
}

extern "C" int tw_puzzle_reset(tw_puzzle *p, uint64_t seed, uint64_t episode)                    // puzzle.rs:119-133
{
    for (size_t i = 0; i < p->state.size(); ++i) p->state[i] = (int64_t)i;
    p->zx = 0; p->zy = 0;
    for (int64_t d = 0; d < p->difficulty; ++d) {
        const u32x4 w = rng_draw(seed, episode, (uint32_t)d, STREAM_SCRAMBLE);
        tw_puzzle_step(p, u32_below(w.x, 4u));
    }
    p->depth = p->depth_slope * p->difficulty;
    return TW_OK;
}

extern "C" int tw_puzzle_masks(const tw_puzzle *p, uint8_t out[4])                               // puzzle.rs:162-165
{
    out[0] = p->zx > 0; out[1] = p->zy > 0; out[2] = p->zx < p->width - 1; out[3] = p->zy < p->height - 1;
    return TW_OK;
}
extern "C" int tw_puzzle_solved(const tw_puzzle *p)                                              // puzzle.rs:44-50
{
    for (size_t i = 0; i < p->state.size(); ++i) if (p->state[i] != (int64_t)i) return 0;
    return 1;
}
extern "C" int tw_puzzle_is_final(const tw_puzzle *p) { return p->depth == 0 || tw_puzzle_solved(p); }   // :167-169
extern "C" float tw_puzzle_reward(const tw_puzzle *p)                                            // puzzle.rs:171-177
{
    if (tw_puzzle_solved(p)) return 1.0f;
    return p->depth == 0 ? -0.5f : -0.5f / (float)p->max_depth;
}
extern "C" int tw_puzzle_observe(const tw_puzzle *p, int64_t *out)                               // puzzle.rs:183-185
{
    const int64_t n = (int64_t)p->state.size();
    for (int64_t i = 0; i < n; ++i) out[i] = i * n + p->state[(size_t)i];
    return TW_OK;
}
extern "C" int tw_puzzle_get_state(const tw_puzzle *p, int64_t *out)
{
    memcpy(out, p->state.data(), p->state.size() * sizeof(int64_t)); return TW_OK;
}
extern "C" int tw_puzzle_set_position(tw_puzzle *p, uint32_t x, uint32_t y, int64_t val)         // puzzle.rs:71-73
{
    const size_t i = (size_t)y * p->width + x;
    if (i >= p->state.size()) { set_error("set_position: index out of bounds"); return TW_ERR_INVALID; }
    p->state[i] = val; return TW_OK;
}
extern "C" int64_t tw_puzzle_get_position(const tw_puzzle *p, uint32_t x, uint32_t y)            // puzzle.rs:75-77
{
    const size_t i = (size_t)y * p->width + x;
    return i < p->state.size() ? p->state[i] : -1;
}

// ====================================================================================== Policy
struct tw_policy {
    PolicyDev dev{};
    void *arena = nullptr;
    size_t arena_bytes = 0;
    int device = -1;
    uint32_t n16 = 0, sp16 = 0, sps = 0;      // f16 / split-f16 image geometry (tw_policy_update_device)
    std::vector<LayerDev> gen_layers;          // generic stacks: the layer table (device pointers) and the un-padded widths
    std::vector<uint32_t> gen_out;
};

namespace {
int hid_row(int r, int i) { return 32 * r + 2 * ((i & 3) + 4 * (i >> 3)) + ((i >> 2) & 1); }   // == tw::hid() in tw_rollout.hip
}

namespace {

// The shape the MFMA engines implement (the BasicPolicy of both Puzzle configs, examples/ppo_puzzle{8,15}_v1.json):
// embedding -> ONE common Linear of 32..256 units -> linear heads.  Everything else runs the generic engine.
bool is_mfma_shape(const tw_policy_desc *d)
{
    if (d->n_common != 1 || d->n_action != 1 || d->n_value != 1 || !d->common || !d->action || !d->value) return false;
    const tw_linear_desc &c = d->common[0], &a = d->action[0], &v = d->value[0];
    const uint32_t E = d->emb_size, H = c.out_features, A = a.out_features;
    return c.in_features == E && a.in_features == H && v.in_features == H && v.out_features == 1 && A == d->n_actions && A != 0 && A <= 31 &&
           E != 0 && E % 32 == 0 && H % 32 == 0 && H != 0 && H <= 256 && (H & (H - 1)) == 0 && d->obs_size != 0 && d->obs_size <= 256 &&
           !a.apply_relu && !v.apply_relu;
}

// Any Sequential stack (modules.rs:28-34): natural-layout weights + a layer table for EngineV (tw_engine_generic.hpp).
tw_policy *create_generic_policy(const tw_policy_desc *d)
{
    const uint32_t E = d->emb_size, OS = d->obs_size;
    if (E == 0 || E % 4 != 0 || E > 512 || OS == 0 || OS > 65535) {
        set_error("policy: generic stacks need an embedding size that is a multiple of 4 and <= 512 and obs_size <= 65535 (got emb=%u obs_size=%u)", E, OS);
        return nullptr;
    }
    const uint32_t n_layers = d->n_common + d->n_action + d->n_value;
    if (d->n_common > 8 || d->n_action > 8 || d->n_value > 8) { set_error("policy: at most 8 layers per stack"); return nullptr; }
    if ((d->n_common && !d->common) || (d->n_action && !d->action) || (d->n_value && !d->value)) { set_error("policy: null layer array"); return nullptr; }
    auto chain = [&](const tw_linear_desc *ls, uint32_t n, uint32_t in, const char *what, uint32_t *out) -> bool {
        for (uint32_t i = 0; i < n; ++i) {
            if (!ls[i].weights || !ls[i].bias || ls[i].in_features != in || ls[i].out_features == 0 || ls[i].out_features > 512) {
                set_error("policy: %s layer %u has %u -> %u features where %u inputs arrive (widths up to 512)", what, i, ls[i].in_features, ls[i].out_features, in);
                return false;
            }
            in = ls[i].out_features;
        }
        *out = in;
        return true;
    };
    uint32_t cw = 0, aw = 0, vw = 0;
    if (!chain(d->common, d->n_common, E, "common", &cw) || !chain(d->action, d->n_action, cw, "action", &aw) ||
        !chain(d->value, d->n_value, cw, "value", &vw)) return nullptr;
    if (aw != d->n_actions || aw < 1 || aw > 31) { set_error("policy: the action head ends in %u outputs, n_actions is %u", aw, d->n_actions); return nullptr; }
    const uint32_t A = aw;
    size_t cur = 0;
    auto seg = [&](size_t bytes) { size_t o = cur; cur = align_up(cur + bytes, 256); return o; };
    const size_t o_emb = seg((size_t)(OS + 2) * E * 4), o_tab = seg((size_t)(n_layers ? n_layers : 1) * sizeof(LayerDev)),
                 o_op = seg((size_t)(d->n_perms ? d->n_perms : 1) * OS), o_ap = seg((size_t)(d->n_perms ? d->n_perms : 1) * A),
                 o_op16 = seg(OS > 256 ? (size_t)(d->n_perms ? d->n_perms : 1) * OS * 2 : 0);
    std::vector<size_t> o_w(n_layers), o_b(n_layers);
    std::vector<const tw_linear_desc *> all;
    for (uint32_t i = 0; i < d->n_common; ++i) all.push_back(&d->common[i]);
    for (uint32_t i = 0; i < d->n_action; ++i) all.push_back(&d->action[i]);
    for (uint32_t i = 0; i < d->n_value; ++i) all.push_back(&d->value[i]);
    // every layer's outputs are padded to a multiple of four (zero weight columns, zero bias): the engine works on output quads
    auto pad4 = [](uint32_t x) { return (x + 3u) & ~3u; };
    // matrix-core image of a layer (LayerDev::wm): tiles of 16 outputs, tb tiles per block, nb blocks, kg groups of four inputs
    auto tiles_per_block = [](uint32_t out) { const uint32_t tiles = (out + 15u) / 16u; return tiles >= 4u ? 4u : tiles; };
    auto blocks_of = [&](uint32_t out) { const uint32_t tiles = (out + 15u) / 16u, tb = tiles_per_block(out); return (tiles + tb - 1u) / tb; };
    std::vector<size_t> o_wm(n_layers);
    for (uint32_t i = 0; i < n_layers; ++i) {
        const uint32_t in = all[i]->in_features, out = all[i]->out_features, tb = tiles_per_block(out), nb = blocks_of(out), kg = (in + 3u) / 4u;
        o_w[i] = seg((size_t)in * pad4(out) * 4); o_b[i] = seg((size_t)nb * tb * 16 * 4 + 16);
        o_wm[i] = seg((size_t)(kg + 8) * 4 * nb * 16 * tb * 4 + 64);          // (+ 8 k-groups of read slack: EngineV loads that far ahead)
    }
    std::vector<uint8_t> img(cur, 0);
    float *emb = reinterpret_cast<float *>(img.data() + o_emb);
    memcpy(emb, d->emb_vectors, (size_t)OS * E * 4);
    memcpy(emb + (size_t)OS * E, d->emb_bias, (size_t)E * 4);
    for (uint32_t i = 0; i < n_layers; ++i) {
        const uint32_t in = all[i]->in_features, out = all[i]->out_features, outp = pad4(out);
        float *wd = reinterpret_cast<float *>(img.data() + o_w[i]);
        for (uint32_t k = 0; k < in; ++k) memcpy(wd + (size_t)k * outp, all[i]->weights + (size_t)k * out, (size_t)out * 4);
        memcpy(img.data() + o_b[i], all[i]->bias, (size_t)out * 4);
        const uint32_t tb = tiles_per_block(out), nb = blocks_of(out), kg = (in + 3u) / 4u;
        float *wm = reinterpret_cast<float *>(img.data() + o_wm[i]);
        for (uint32_t k = 0; k < kg * 4; ++k)
            for (uint32_t b = 0; b < nb; ++b)
                for (uint32_t r = 0; r < 16; ++r)
                    for (uint32_t t = 0; t < tb; ++t) {
                        const uint32_t o = (b * tb + t) * 16 + r;
                        wm[(((size_t)k * nb + b) * 16 + r) * tb + t] = (k < in && o < out) ? all[i]->weights[(size_t)k * out + o] : -0.0f;
                    }
    }
    for (uint32_t p = 0; p < d->n_perms; ++p) {
        for (uint32_t i = 0; i < OS; ++i) {
            img[o_op + (size_t)p * OS + i] = (uint8_t)d->obs_perms[(size_t)p * OS + i];
            if (OS > 256) { const uint16_t v16 = (uint16_t)d->obs_perms[(size_t)p * OS + i]; memcpy(img.data() + o_op16 + ((size_t)p * OS + i) * 2, &v16, 2); }
        }
        for (uint32_t i = 0; i < A; ++i) img[o_ap + (size_t)p * A + i] = (uint8_t)d->act_perms[(size_t)p * A + i];
    }
    tw_policy *pol = new tw_policy();
    pol->arena_bytes = img.size();
    hipError_t e = hipGetDevice(&pol->device);
    if (e == hipSuccess) e = hipMalloc(&pol->arena, img.size());
    if (e != hipSuccess) { hip_fail(e, "policy upload", __FILE__, __LINE__); delete pol; return nullptr; }
    const uint8_t *base = reinterpret_cast<const uint8_t *>(pol->arena);
    pol->dev.obs_perms16 = OS > 256 ? reinterpret_cast<const uint16_t *>(base + o_op16) : nullptr;
    LayerDev *tab = reinterpret_cast<LayerDev *>(img.data() + o_tab);          // device pointers go into the table before the upload
    for (uint32_t i = 0; i < n_layers; ++i) {
        tab[i].w = reinterpret_cast<const float *>(base + o_w[i]); tab[i].b = reinterpret_cast<const float *>(base + o_b[i]);
        tab[i].in = (int32_t)all[i]->in_features; tab[i].out = (int32_t)pad4(all[i]->out_features); tab[i].relu = all[i]->apply_relu ? 1 : 0; tab[i].pad = 0;
        tab[i].wm = reinterpret_cast<const float *>(base + o_wm[i]);
        tab[i].kg = (int32_t)((all[i]->in_features + 3u) / 4u); tab[i].nb = (int32_t)blocks_of(all[i]->out_features); tab[i].tb = (int32_t)tiles_per_block(all[i]->out_features); tab[i].pad2 = 0;
        pol->gen_layers.push_back(tab[i]); pol->gen_out.push_back(all[i]->out_features);
    }
    e = hipMemcpy(pol->arena, img.data(), img.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { hip_fail(e, "policy upload", __FILE__, __LINE__); (void)hipFree(pol->arena); delete pol; return nullptr; }
    PolicyDev &pd = pol->dev;
    pd.obs_size = (int)OS; pd.emb = (int)E; pd.hidden = 0; pd.n_actions = (int)A; pd.n_perms = (int)d->n_perms;
    pd.emb_relu = d->emb_apply_relu ? 1 : 0; pd.common_relu = 0;
    pd.emb_rows = reinterpret_cast<const float *>(base + o_emb);
    pd.obs_perms = base + o_op; pd.act_perms = base + o_ap;
    {   // rows of EngineV's three activation buffers: replay the buffer choices of EngineV::stack() / forward() (tw_engine_generic.hpp)
        int rows[3] = {(int)E, 0, 0};
        auto need = [&](int b, uint32_t r) { if ((int)r > rows[b]) rows[b] = (int)r; };
        auto outp = [&](uint32_t i) { return (uint32_t)(pol->gen_layers[i].nb * pol->gen_layers[i].tb * 16); };
        auto inp = [&](uint32_t i) { return (uint32_t)(pol->gen_layers[i].kg * 4); };
        auto run = [&](uint32_t first, uint32_t n, int src, int keep) -> int {
            int c = src;
            for (uint32_t l = 0; l < n; ++l) {
                int dst = 0;
                while (dst == c || dst == keep) ++dst;
                need(c, inp(first + l)); need(dst, outp(first + l));
                c = dst;
            }
            return c;
        };
        const int co = run(0, d->n_common, 0, -1);
        if (d->n_value == 1 && d->n_action == 1) {
            const int vo = co == 0 ? 1 : 0, ao = 3 - co - vo;
            need(co, inp(d->n_common)); need(co, inp(d->n_common + d->n_action));
            need(ao, outp(d->n_common)); need(vo, outp(d->n_common + d->n_action));
        } else {
            (void)run(d->n_common + d->n_action, d->n_value, co, co);
            (void)run(d->n_common, d->n_action, co, co);
        }
        pd.gen_rows0 = rows[0] > 16 ? rows[0] : 16; pd.gen_rows1 = rows[1] > 16 ? rows[1] : 16; pd.gen_rows2 = rows[2] > 16 ? rows[2] : 16;
    }
    pd.generic = 1; pd.n_common = (int)d->n_common; pd.n_action = (int)d->n_action; pd.n_value = (int)d->n_value; pd.value_out = (int)vw;
    pd.layers = reinterpret_cast<const LayerDev *>(base + o_tab);
    return pol;
}

}  // namespace

extern "C" tw_policy *tw_policy_create(const tw_policy_desc *d)
{
    if (!d || !d->emb_vectors || !d->emb_bias) { set_error("tw_policy_create: null descriptor"); return nullptr; }
    if (require_device()) return nullptr;
    if (d->n_perms > 0 && (!d->obs_perms || !d->act_perms)) { set_error("policy: n_perms > 0 but perms are null"); return nullptr; }
    if (d->n_perms > 127) { set_error("policy: at most 127 twists"); return nullptr; }
    {
        const uint32_t A0 = d->n_actions;
        for (uint32_t p = 0; p < d->n_perms; ++p) {
            for (uint32_t i = 0; i < d->obs_size; ++i)
                if (d->obs_perms[(size_t)p * d->obs_size + i] < 0 || (uint32_t)d->obs_perms[(size_t)p * d->obs_size + i] >= d->obs_size) {
                    set_error("policy: obs_perms[%u][%u] out of range", p, i); return nullptr;
                }
            for (uint32_t i = 0; i < A0; ++i)
                if (d->act_perms[(size_t)p * A0 + i] < 0 || (uint32_t)d->act_perms[(size_t)p * A0 + i] >= A0) {
                    set_error("policy: act_perms[%u][%u] out of range", p, i); return nullptr;
                }
        }
    }
    // ---- the MFMA engines implement the BasicPolicy of the Puzzle configs; any other Sequential stack runs the generic engine
    if (!is_mfma_shape(d)) return create_generic_policy(d);
    const tw_linear_desc &c = d->common[0], &a = d->action[0], &v = d->value[0];
    const uint32_t E = d->emb_size, H = c.out_features, A = a.out_features;

    const uint32_t NT = H / 32, OS = d->obs_size;
    // ---- host image ------------------------------------------------------------------------
    struct Seg { size_t off, bytes; };
    size_t cur = 0;
    auto seg = [&](size_t bytes) { Seg s{cur, bytes}; cur = align_up(cur + bytes, 256); return s; };
    const uint32_t NQ = (NT + 3) / 4;     // float4 groups of row-tiles per (k, i)
    const size_t T16_SLOT = 21 * 256;     // floats per 16-column table chunk image (tw_engine.hpp R3_TSLOT)
    const Seg s_emb = seg((size_t)(OS + 2) * E * 4), s_w1p = seg((size_t)E * NQ * 128 * 4), s_b1 = seg((size_t)H * 4),
              s_t16 = seg((size_t)(E / 16) * T16_SLOT * 4),
              s_wh8 = seg((size_t)H * 8 * 4), s_bh8 = seg(8 * 4), s_w1 = seg((size_t)E * H * 4), s_wa = seg((size_t)H * A * 4),
              s_ba = seg((size_t)A * 4), s_wv = seg((size_t)H * 4), s_bv = seg(4),
              s_op = seg((size_t)(d->n_perms ? d->n_perms : 1) * OS), s_ap = seg((size_t)(d->n_perms ? d->n_perms : 1) * A);
    // ---- f16 image (Engine16, tw_engine16.hpp): available when obs ids are cell*n+tile with n <= 16 and
    //      every twist maps cells to cells (true for any symmetry of a cell-wise one-hot encoding)
    uint32_t n16 = 1; while (n16 * n16 < OS) ++n16;
    bool f16_ok = n16 * n16 == OS && n16 <= 16 && d->n_perms <= 4 && A <= 4;
    std::vector<uint8_t> srcmap((size_t)(d->n_perms + 1) * 16, 0), vmap((size_t)(d->n_perms + 1) * 256, 0xFF);
    if (f16_ok) {
        for (uint32_t c2 = 0; c2 < n16; ++c2) { srcmap[c2] = (uint8_t)c2; for (uint32_t v2 = 0; v2 < n16; ++v2) vmap[c2 * 16 + v2] = (uint8_t)v2; }
        for (uint32_t p = 0; p < d->n_perms && f16_ok; ++p) {
            std::vector<int> cm(n16, -1), seen(n16, 0);
            for (uint32_t sc = 0; sc < n16 && f16_ok; ++sc) {
                for (uint32_t v2 = 0; v2 < n16; ++v2) {
                    const uint32_t id2 = (uint32_t)d->obs_perms[(size_t)p * OS + sc * n16 + v2];
                    if (v2 == 0) cm[sc] = (int)(id2 / n16);
                    else if ((int)(id2 / n16) != cm[sc]) { f16_ok = false; break; }
                }
                if (f16_ok) { if (seen[cm[sc]]) f16_ok = false; seen[cm[sc]] = 1; }
            }
            if (!f16_ok) break;
            for (uint32_t sc = 0; sc < n16; ++sc) {
                const uint32_t tc = (uint32_t)cm[sc];
                srcmap[(size_t)(p + 1) * 16 + tc] = (uint8_t)sc;
                for (uint32_t v2 = 0; v2 < n16; ++v2)
                    vmap[((size_t)(p + 1) * 16 + tc) * 16 + v2] = (uint8_t)((uint32_t)d->obs_perms[(size_t)p * OS + sc * n16 + v2] % n16);
            }
        }
    }
    const uint32_t nc16 = !f16_ok ? 0u : (n16 <= 4 ? 4u : (n16 <= 9 ? 9u : 16u));
    const uint32_t SP16 = (nc16 + 2 * NT + 3) / 4 * 4, NKT = E / 32;     // stage image padded to whole rounds of 4 DMA pieces (Engine16::SBYTES)
    const Seg s_st16 = seg(f16_ok ? (size_t)NKT * SP16 * 1024 : 0), s_hd16 = seg(f16_ok ? (size_t)NT * 2048 : 0),
              s_eb16 = seg((size_t)NKT * 32 * 4), s_b116 = seg((size_t)NT * 32 * 4), s_bh16 = seg(8 * 4),
              s_src16 = seg(srcmap.size()), s_vm16 = seg(vmap.size());
    // split-f16 image (EngineS, tw_engine16x2.hpp)
    const uint32_t SPS = f16_ok ? ((std::max(nc16 + 2 * NT, 4 * NT) + 3) / 4 * 4) : 0;
    const bool split_ok = f16_ok && NKT >= 2;
    const Seg s_stS = seg(split_ok ? (size_t)(2 * NKT + 1) * SPS * 1024 : 0), s_t0S = seg(split_ok ? (size_t)2 * nc16 * 1024 : 0);
    std::vector<uint8_t> img(cur, 0);
    float *emb = reinterpret_cast<float *>(img.data() + s_emb.off);
    memcpy(emb, d->emb_vectors, (size_t)OS * E * 4);
    memcpy(emb + (size_t)OS * E, d->emb_bias, (size_t)E * 4);                 // bias row; row OS+1 stays zero
    float *w1p = reinterpret_cast<float *>(img.data() + s_w1p.off);
    // MFMA A-operand image [k][q][i][4]: lane i of k-step k reads row-tiles 4q..4q+3 as one float4
    for (uint32_t k = 0; k < E; ++k)
        for (uint32_t q = 0; q < NQ; ++q)
            for (uint32_t i = 0; i < 32; ++i)
                for (uint32_t cc = 0; cc < 4; ++cc) {
                    const uint32_t r = 4 * q + cc;
                    w1p[(((size_t)k * NQ + q) * 32 + i) * 4 + cc] =
                        r < NT ? c.weights[(size_t)k * H + hid_row((int)r, (int)i)] : 0.0f;
                }
    // Engine3 table image: chunk c, row r, position p of 20: p<8 -> column 16c+2p (even k), p<16 -> 16c+2(p-8)+1, else 0
    float *t16 = reinterpret_cast<float *>(img.data() + s_t16.off);
    for (uint32_t ch = 0; ch < E / 16; ++ch)
        for (uint32_t r = 0; r < OS + 2; ++r)
            for (uint32_t pp = 0; pp < 16; ++pp) {
                const uint32_t k = pp < 8 ? 2 * pp : 2 * (pp - 8) + 1;
                t16[(size_t)ch * T16_SLOT + (size_t)r * 20 + pp] = emb[(size_t)r * E + ch * 16 + k];
            }
    memcpy(img.data() + s_b1.off, c.bias, (size_t)H * 4);
    float *wh8 = reinterpret_cast<float *>(img.data() + s_wh8.off);
    float *bh8 = reinterpret_cast<float *>(img.data() + s_bh8.off);
    if (A <= 4) {
        for (uint32_t n = 0; n < H; ++n) {
            for (uint32_t i = 0; i < A; ++i) wh8[(size_t)n * 8 + i] = a.weights[(size_t)n * A + i];
            wh8[(size_t)n * 8 + 4] = v.weights[n];
        }
        for (uint32_t i = 0; i < A; ++i) bh8[i] = a.bias[i];
        bh8[4] = v.bias[0];
    }
    memcpy(img.data() + s_w1.off, c.weights, (size_t)E * H * 4);
    memcpy(img.data() + s_wa.off, a.weights, (size_t)H * A * 4);
    memcpy(img.data() + s_ba.off, a.bias, (size_t)A * 4);
    memcpy(img.data() + s_wv.off, v.weights, (size_t)H * 4);
    memcpy(img.data() + s_bv.off, v.bias, 4);
    for (uint32_t p = 0; p < d->n_perms; ++p) {
        for (uint32_t i = 0; i < OS; ++i) img[s_op.off + (size_t)p * OS + i] = (uint8_t)d->obs_perms[(size_t)p * OS + i];
        for (uint32_t i = 0; i < A; ++i) img[s_ap.off + (size_t)p * A + i] = (uint8_t)d->act_perms[(size_t)p * A + i];
    }

    if (f16_ok) {
        auto put16 = [&](size_t byte_off, float x) { const _Float16 hx = (_Float16)x; memcpy(img.data() + byte_off, &hx, 2); };
        auto rho = [](uint32_t r, uint32_t hh2) { return 8 * (r >> 2) + 4 * hh2 + (r & 3); };   // accumulator register -> tile row
        for (uint32_t kt = 0; kt < NKT; ++kt) {
            const size_t sbase = s_st16.off + (size_t)kt * SP16 * 1024;
            for (uint32_t l = 0; l < 64; ++l)
                for (uint32_t jx = 0; jx < 8; ++jx) {
                    const uint32_t hh2 = l >> 5, row = l & 31;
                    for (uint32_t c2 = 0; c2 < nc16; ++c2) {               // table chunk = cell c2, k slot = tile value
                        const uint32_t val = 8 * hh2 + jx;
                        const uint32_t kte = (kt + 1) % NKT;                // the table part of stage kt belongs to the NEXT tile (pipeline)
                        const float x = (c2 < n16 && val < n16) ? d->emb_vectors[(size_t)(c2 * n16 + val) * E + 32 * kte + row] : 0.0f;
                        put16(sbase + (size_t)c2 * 1024 + l * 16 + jx * 2, x);
                    }
                    for (uint32_t ht = 0; ht < NT; ++ht)
                        for (uint32_t m = 0; m < 2; ++m) {                 // W1 chunk: k slot (hh2, jx) = embedding row of register 8m+jx
                            const uint32_t k = 32 * kt + rho(8 * m + jx, hh2);
                            put16(sbase + (size_t)(nc16 + ht * 2 + m) * 1024 + l * 16 + jx * 2, c.weights[(size_t)k * H + 32 * ht + row]);
                        }
                }
        }
        for (uint32_t ht = 0; ht < NT; ++ht)
            for (uint32_t m = 0; m < 2; ++m)
                for (uint32_t l = 0; l < 64; ++l)
                    for (uint32_t jx = 0; jx < 8; ++jx) {
                        const uint32_t hh2 = l >> 5, row = l & 31, hid = 32 * ht + rho(8 * m + jx, hh2);
                        float x = 0.0f;
                        if (row < 8) { if ((row & 3) < A) x = a.weights[(size_t)hid * A + (row & 3)]; }
                        else if (row == 8 || row == 12) x = v.weights[hid];
                        put16(s_hd16.off + (size_t)(ht * 2 + m) * 1024 + l * 16 + jx * 2, x);
                    }
        float *eb16 = reinterpret_cast<float *>(img.data() + s_eb16.off);
        for (uint32_t kt = 0; kt < NKT; ++kt)
            for (uint32_t hh2 = 0; hh2 < 2; ++hh2)
                for (uint32_t r = 0; r < 16; ++r) eb16[(kt * 2 + hh2) * 16 + r] = d->emb_bias[32 * kt + rho(r, hh2)];
        float *b116 = reinterpret_cast<float *>(img.data() + s_b116.off);
        for (uint32_t ht = 0; ht < NT; ++ht)
            for (uint32_t hh2 = 0; hh2 < 2; ++hh2)
                for (uint32_t r = 0; r < 16; ++r) b116[(ht * 2 + hh2) * 16 + r] = c.bias[32 * ht + rho(r, hh2)];
        float *bh16 = reinterpret_cast<float *>(img.data() + s_bh16.off);
        for (uint32_t i = 0; i < A; ++i) bh16[i] = a.bias[i];
        bh16[4] = v.bias[0];
        memcpy(img.data() + s_src16.off, srcmap.data(), srcmap.size());
        memcpy(img.data() + s_vm16.off, vmap.data(), vmap.size());
        if (split_ok) {
            // x = x_hi + x_lo with both terms binary16, operands pre-scaled by 16 (exact): hi = f16(16x), lo = f16(16x - hi)
            auto put2 = [&](size_t off_hi, size_t off_lo, float x) {
                const float sx = 16.0f * x;
                const _Float16 hi = (_Float16)sx, lo = (_Float16)(sx - (float)hi);
                memcpy(img.data() + off_hi, &hi, 2); memcpy(img.data() + off_lo, &lo, 2);
            };
            for (uint32_t kt = 0; kt < NKT; ++kt) {
                const size_t bhi = s_stS.off + (size_t)(2 * kt) * SPS * 1024, blo = bhi + (size_t)SPS * 1024;
                const uint32_t kte = (kt + 1) % NKT;
                for (uint32_t l = 0; l < 64; ++l)
                    for (uint32_t jx = 0; jx < 8; ++jx) {
                        const uint32_t hh2 = l >> 5, row = l & 31, o = l * 16 + jx * 2;
                        for (uint32_t c2 = 0; c2 < nc16; ++c2) {
                            const uint32_t val = 8 * hh2 + jx;
                            const float x = (c2 < n16 && val < n16) ? d->emb_vectors[(size_t)(c2 * n16 + val) * E + 32 * kte + row] : 0.0f;
                            put2(bhi + (size_t)c2 * 1024 + o, blo + (size_t)c2 * 1024 + o, x);
                            if (kt == NKT - 1)   // tile 0 also goes to the resident copy [hi chunks | lo chunks]
                                put2(s_t0S.off + (size_t)c2 * 1024 + o, s_t0S.off + (size_t)(nc16 + c2) * 1024 + o, x);
                        }
                        for (uint32_t ht = 0; ht < NT; ++ht)
                            for (uint32_t m = 0; m < 2; ++m) {
                                const uint32_t k = 32 * kt + rho(8 * m + jx, hh2);
                                const size_t po = (size_t)(nc16 + ht * 2 + m) * 1024 + o;
                                put2(bhi + po, blo + po, c.weights[(size_t)k * H + 32 * ht + row]);
                            }
                    }
            }
            const size_t bh = s_stS.off + (size_t)(2 * NKT) * SPS * 1024;
            for (uint32_t ht = 0; ht < NT; ++ht)
                for (uint32_t m = 0; m < 2; ++m)
                    for (uint32_t l = 0; l < 64; ++l)
                        for (uint32_t jx = 0; jx < 8; ++jx) {
                            const uint32_t hh2 = l >> 5, row = l & 31, hid = 32 * ht + rho(8 * m + jx, hh2);
                            float x = 0.0f;
                            if (row < 8) { if ((row & 3) < A) x = a.weights[(size_t)hid * A + (row & 3)]; }
                            else if (row == 8 || row == 12) x = v.weights[hid];
                            const size_t po = (size_t)(ht * 2 + m) * 1024 + l * 16 + jx * 2;
                            put2(bh + po, bh + (size_t)2 * NT * 1024 + po, x);
                        }
        }
    }

    tw_policy *pol = new tw_policy();
    hipError_t e = hipGetDevice(&pol->device);
    pol->arena_bytes = img.size();
    if (e == hipSuccess) e = hipMalloc(&pol->arena, img.size());
    if (e == hipSuccess) e = hipMemcpy(pol->arena, img.data(), img.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        hip_fail(e, "policy upload", __FILE__, __LINE__);
        if (pol->arena) (void)hipFree(pol->arena);
        delete pol; return nullptr;
    }
    const uint8_t *base = reinterpret_cast<const uint8_t *>(pol->arena);
    PolicyDev &pd = pol->dev;
    pd.obs_size = (int)OS; pd.emb = (int)E; pd.hidden = (int)H; pd.n_actions = (int)A; pd.n_perms = (int)d->n_perms;
    pd.emb_relu = d->emb_apply_relu ? 1 : 0; pd.common_relu = c.apply_relu ? 1 : 0;
    pd.emb_rows = reinterpret_cast<const float *>(base + s_emb.off);
    pd.w1p = reinterpret_cast<const float *>(base + s_w1p.off);
    pd.t_img16 = reinterpret_cast<const float *>(base + s_t16.off);
    pd.b1 = reinterpret_cast<const float *>(base + s_b1.off);
    pd.wh8 = reinterpret_cast<const float *>(base + s_wh8.off);
    pd.bh8 = reinterpret_cast<const float *>(base + s_bh8.off);
    pd.w1 = reinterpret_cast<const float *>(base + s_w1.off);
    pd.wa = reinterpret_cast<const float *>(base + s_wa.off);
    pd.ba = reinterpret_cast<const float *>(base + s_ba.off);
    pd.wv = reinterpret_cast<const float *>(base + s_wv.off);
    pd.bv = reinterpret_cast<const float *>(base + s_bv.off);
    pd.obs_perms = base + s_op.off;
    pd.act_perms = base + s_ap.off;
    pd.f16_nc = (int)nc16;
    pd.stage16 = base + s_st16.off; pd.head16 = base + s_hd16.off;
    pd.ebias16 = reinterpret_cast<const float *>(base + s_eb16.off);
    pd.b1img16 = reinterpret_cast<const float *>(base + s_b116.off);
    pd.bh16 = reinterpret_cast<const float *>(base + s_bh16.off);
    pd.srcmap16 = base + s_src16.off; pd.vmap16 = base + s_vm16.off;
    pd.stageS = split_ok ? base + s_stS.off : nullptr; pd.t0S = split_ok ? base + s_t0S.off : nullptr;
    pol->n16 = n16; pol->sp16 = SP16; pol->sps = split_ok ? SPS : 0;
    return pol;
}

extern "C" int tw_policy_update_device(tw_policy *p, const float *emb_w, const float *emb_b, const float *w1, const float *b1,
                                       const float *wa, const float *ba, const float *wv, const float *bv)
{
    if (!p || !emb_w || !emb_b || !w1 || !b1 || !wa || !ba || !wv || !bv) { set_error("tw_policy_update_device: null argument"); return TW_ERR_INVALID; }
    int rc = require_device(); if (rc) return rc;
    const PolicyDev &d = p->dev;
    if (d.generic) { set_error("tw_policy_update_device: policies of another depth take tw_policy_update_device_layers"); return TW_ERR_UNSUPPORTED; }
    SyncArgs a{};
    a.emb_w = emb_w; a.emb_b = emb_b; a.w1 = w1; a.b1 = b1; a.wa = wa; a.ba = ba; a.wv = wv; a.bv = bv;
    a.OS = d.obs_size; a.E = d.emb; a.H = d.hidden; a.A = d.n_actions; a.NT = d.hidden / 32; a.NQ = (a.NT + 3) / 4;
    a.n16 = (int)p->n16; a.nc16 = d.f16_nc; a.SP16 = (int)p->sp16; a.NKT = d.emb / 32;
    auto w = [](const void *q) { return const_cast<void *>(q); };
    a.emb_rows = (float *)w(d.emb_rows); a.w1p = (float *)w(d.w1p); a.t_img16 = (float *)w(d.t_img16); a.b1_d = (float *)w(d.b1);
    a.wh8 = (float *)w(d.wh8); a.bh8 = (float *)w(d.bh8); a.w1_nat = (float *)w(d.w1); a.wa_nat = (float *)w(d.wa);
    a.ba_nat = (float *)w(d.ba); a.wv_nat = (float *)w(d.wv); a.bv_nat = (float *)w(d.bv);
    a.stage16 = (uint8_t *)w(d.stage16); a.head16 = (uint8_t *)w(d.head16); a.ebias16 = (float *)w(d.ebias16);
    a.b1img16 = (float *)w(d.b1img16); a.bh16 = (float *)w(d.bh16);
    a.stageS = (uint8_t *)w(d.stageS); a.t0S = (uint8_t *)w(d.t0S); a.SPS = (int)p->sps;
    const bool f16 = d.f16_nc != 0;
    const bool split = f16 && p->sps != 0 && d.stageS && d.t0S;
    const unsigned long long cnt[18] = {
        (unsigned long long)(a.OS + 2) * a.E, (unsigned long long)a.E * a.NQ * 128, (unsigned long long)(a.E / 16) * 21 * 256,
        (unsigned long long)a.H, (unsigned long long)a.H * 8, 8ull, (unsigned long long)a.E * a.H, (unsigned long long)a.H * a.A,
        (unsigned long long)a.A, (unsigned long long)a.H, 1ull,
        f16 ? (unsigned long long)a.NKT * a.SP16 * 512 : 0ull, f16 ? (unsigned long long)a.NT * 1024 : 0ull,
        f16 ? (unsigned long long)a.NKT * 32 : 0ull, f16 ? (unsigned long long)a.NT * 32 : 0ull, f16 ? 8ull : 0ull,
        split ? (unsigned long long)(2 * a.NKT + 1) * a.SPS * 512 : 0ull, split ? (unsigned long long)2 * a.nc16 * 512 : 0ull};
    unsigned long long run = 0;
    for (int i = 0; i < 18; ++i) { run += cnt[i]; a.seg_end[i] = run; }
    return launch_policy_sync(a, current_stream());
}

extern "C" int tw_policy_update_device_layers(tw_policy *p, const float *emb_w, const float *emb_b, const float *const *weights,
                                              const float *const *biases, uint32_t n_layers)
{
    if (!p || !emb_w || !emb_b || (n_layers && (!weights || !biases))) { set_error("tw_policy_update_device_layers: null argument"); return TW_ERR_INVALID; }
    int rc = require_device(); if (rc) return rc;
    const PolicyDev &d = p->dev;
    if (!d.generic) { set_error("tw_policy_update_device_layers: this policy has the one-common-layer shape: tw_policy_update_device"); return TW_ERR_UNSUPPORTED; }
    if (n_layers != p->gen_layers.size() || n_layers > (uint32_t)GEN_SYNC_MAX_LAYERS) { set_error("tw_policy_update_device_layers: the policy has %zu layers, %u given", p->gen_layers.size(), n_layers); return TW_ERR_INVALID; }
    GenSyncArgs a{};
    a.emb_w = emb_w; a.emb_b = emb_b; a.emb_rows = const_cast<float *>(d.emb_rows); a.OS = d.obs_size; a.E = d.emb; a.n_layers = (int)n_layers;
    unsigned long long run = (unsigned long long)(a.OS + 2) * a.E;
    a.seg_end[0] = run;
    for (uint32_t l = 0; l < n_layers; ++l) {
        if (!weights[l] || !biases[l]) { set_error("tw_policy_update_device_layers: null layer %u", l); return TW_ERR_INVALID; }
        const LayerDev &L = p->gen_layers[l];
        a.w[l] = weights[l]; a.b[l] = biases[l];
        a.w_nat[l] = const_cast<float *>(L.w); a.b_img[l] = const_cast<float *>(L.b); a.wm[l] = const_cast<float *>(L.wm);
        a.in[l] = L.in; a.out[l] = (int)p->gen_out[l]; a.outp[l] = L.out; a.kg[l] = L.kg; a.nb[l] = L.nb; a.tb[l] = L.tb;
        run += (unsigned long long)L.in * L.out; a.seg_end[1 + 3 * l] = run;
        run += (unsigned long long)L.nb * L.tb * 16; a.seg_end[2 + 3 * l] = run;
        run += (unsigned long long)L.kg * 4 * L.nb * 16 * L.tb; a.seg_end[3 + 3 * l] = run;
    }
    return launch_policy_sync_generic(a, current_stream());
}

extern "C" void tw_policy_destroy(tw_policy *p)
{
    if (!p) return;
    if (p->arena) (void)hipFree(p->arena);
    delete p;
}
extern "C" uint32_t tw_policy_num_actions(const tw_policy *p) { return p ? (uint32_t)p->dev.n_actions : 0; }
extern "C" uint32_t tw_policy_num_perms(const tw_policy *p) { return p ? (uint32_t)p->dev.n_perms : 0; }

extern "C" int tw_policy_evaluate(const tw_policy *p, int mode, uint32_t precision, const int32_t *obs, uint32_t n,
                                  uint32_t n_obs, const uint8_t *masks, const int32_t *perms, float *out_actions,
                                  float *out_values)
{
    if (!p || !obs || !masks || !out_actions || !out_values) { set_error("tw_policy_evaluate: null argument"); return TW_ERR_INVALID; }
    if (mode < TW_EVAL_FORWARD || mode > TW_EVAL_FULL_PREDICT) { set_error("tw_policy_evaluate: bad mode %d", mode); return TW_ERR_INVALID; }
    if (precision != TW_PREC_F32_EXACT) { set_error("tw_policy_evaluate: only TW_PREC_F32_EXACT is implemented"); return TW_ERR_UNSUPPORTED; }
    if (n == 0) return TW_OK;
    if (n_obs == 0 || n_obs > 64) { set_error("tw_policy_evaluate: n_obs %u out of range", n_obs); return TW_ERR_INVALID; }
    const int A = p->dev.n_actions;
    for (size_t i = 0; i < (size_t)n * n_obs; ++i)
        if (obs[i] < 0 || obs[i] >= p->dev.obs_size) { set_error("index out of bounds: obs id %d, obs_size %d", obs[i], p->dev.obs_size); return TW_ERR_INVALID; }
    if (perms)
        for (uint32_t i = 0; i < n; ++i)
            if (perms[i] >= p->dev.n_perms) { set_error("perm index %d out of range (%d twists)", perms[i], p->dev.n_perms); return TW_ERR_INVALID; }
    int rc = require_device(); if (rc) return rc;
    hipStream_t s = current_stream();
    const size_t b_obs = (size_t)n * n_obs * 4, b_m = (size_t)n * A, b_p = (size_t)n * 4, b_oa = (size_t)n * A * 4, b_ov = (size_t)n * 4;
    const size_t o_obs = 0, o_m = align_up(o_obs + b_obs, 256), o_p = align_up(o_m + b_m, 256),
                 o_oa = align_up(o_p + b_p, 256), o_ov = align_up(o_oa + b_oa, 256), tot = align_up(o_ov + b_ov, 256);
    uint8_t *buf = nullptr;
    TW_HIP(hipMalloc((void **)&buf, tot));
    auto cleanup = [&]() { (void)hipFree(buf); };
#define TW_HIP_C(call) do { hipError_t _e = (call); if (_e != hipSuccess) { cleanup(); return hip_fail(_e, #call, __FILE__, __LINE__); } } while (0)
    TW_HIP_C(hipMemcpyAsync(buf + o_obs, obs, b_obs, hipMemcpyHostToDevice, s));
    TW_HIP_C(hipMemcpyAsync(buf + o_m, masks, b_m, hipMemcpyHostToDevice, s));
    if (perms) TW_HIP_C(hipMemcpyAsync(buf + o_p, perms, b_p, hipMemcpyHostToDevice, s));
    rc = launch_policy_eval(p->dev, mode, reinterpret_cast<const int32_t *>(buf + o_obs), n, n_obs, buf + o_m,
                            perms ? reinterpret_cast<const int32_t *>(buf + o_p) : nullptr,
                            reinterpret_cast<float *>(buf + o_oa), reinterpret_cast<float *>(buf + o_ov), s);
    if (rc) { cleanup(); return rc; }
    TW_HIP_C(hipMemcpyAsync(out_actions, buf + o_oa, b_oa, hipMemcpyDeviceToHost, s));
    TW_HIP_C(hipMemcpyAsync(out_values, buf + o_ov, b_ov, hipMemcpyDeviceToHost, s));
    TW_HIP_C(hipStreamSynchronize(s));
#undef TW_HIP_C
    cleanup();
    return TW_OK;
}

// ====================================================================================== Collected
struct tw_collected {
    void *arena = nullptr;                  // one allocation holding every compact field (from the arena pool)
    size_t arena_cap = 0; int device = -1;
    hipStream_t stream = nullptr;           // the library stream of the thread that produced it (tw_set_stream is thread-local)
    uint32_t obs_width = 1;                 // bytes per obs id (2: an environment with more than 256 ids, tw_ppo_collect_env)
    void *field_ptr[TW_F_COUNT] = {};
    size_t field_bytes[TW_F_COUNT] = {};
    uint64_t n_records = 0, n_episodes = 0;
    uint32_t n_cells = 0, n_actions = 0;
    int is_ppo = 0;
    tw_collect_stats stats{};
};

extern "C" uint64_t tw_collected_num_records(const tw_collected *c) { return c ? c->n_records : 0; }
extern "C" uint64_t tw_collected_num_episodes(const tw_collected *c) { return c ? c->n_episodes : 0; }
extern "C" uint32_t tw_collected_num_cells(const tw_collected *c) { return c ? c->n_cells : 0; }
extern "C" uint32_t tw_collected_num_actions(const tw_collected *c) { return c ? c->n_actions : 0; }
extern "C" int tw_collected_is_ppo(const tw_collected *c) { return c ? c->is_ppo : 0; }
extern "C" uint32_t tw_collected_obs_width(const tw_collected *c) { return c ? c->obs_width : 0; }

extern "C" void *tw_collected_device_ptr(const tw_collected *c, int field, size_t *bytes)
{
    if (!c || field < 0 || field >= TW_F_COUNT) { if (bytes) *bytes = 0; return nullptr; }
    if (bytes) *bytes = c->field_bytes[field];
    return c->field_ptr[field];
}

extern "C" int tw_collected_copy_to_host(const tw_collected *c, int field, void *dst, size_t bytes)
{
    if (!c || !dst || field < 0 || field >= TW_F_COUNT) { set_error("copy_to_host: bad argument"); return TW_ERR_INVALID; }
    if (bytes != c->field_bytes[field]) {
        set_error("copy_to_host: field %d holds %zu bytes, caller asked for %zu", field, c->field_bytes[field], bytes);
        return TW_ERR_INVALID;
    }
    if (bytes == 0) return TW_OK;
    TW_HIP(hipMemcpy(dst, c->field_ptr[field], bytes, hipMemcpyDeviceToHost));
    return TW_OK;
}

extern "C" int tw_collected_stats(const tw_collected *c, tw_collect_stats *out)
{
    if (!c || !out) { set_error("tw_collected_stats: null argument"); return TW_ERR_INVALID; }
    *out = c->stats; return TW_OK;
}

// ---- trainer hand-off (tw_trainer.hip) ---------------------------------------------------------
extern "C" int tw_collected_adv_stats(const tw_collected *c, double *mean, double *std_unbiased)
{
    if (!c || !mean || !std_unbiased) { set_error("tw_collected_adv_stats: null argument"); return TW_ERR_INVALID; }
    if (!c->is_ppo || !c->field_ptr[TW_F_ADVS]) { set_error("tw_collected_adv_stats: no advantages in this result (AlphaZero data?)"); return TW_ERR_INVALID; }
    hipStream_t s = current_stream();
    double *acc = nullptr;
    TW_HIP(hipMalloc((void **)&acc, sum_scratch_doubles() * sizeof(double)));
    const float *adv = reinterpret_cast<const float *>(c->field_ptr[TW_F_ADVS]);
    const uint64_t n = c->n_records;
    double sum = 0.0, ss = 0.0;
    int rc = launch_sum(adv, n, 0.0, 0, acc, s);
    hipError_t e = hipSuccess;
    if (rc == TW_OK) { e = hipMemcpyAsync(&sum, acc, 8, hipMemcpyDeviceToHost, s); if (e == hipSuccess) e = hipStreamSynchronize(s); }
    const double m = n ? sum / (double)n : 0.0;
    if (rc == TW_OK && e == hipSuccess) rc = launch_sum(adv, n, m, 1, acc, s);
    if (rc == TW_OK && e == hipSuccess) { e = hipMemcpyAsync(&ss, acc, 8, hipMemcpyDeviceToHost, s); if (e == hipSuccess) e = hipStreamSynchronize(s); }
    (void)hipFree(acc);
    if (rc != TW_OK) return rc;
    if (e != hipSuccess) return hip_fail(e, "advantage statistics", __FILE__, __LINE__);
    *mean = m;
    *std_unbiased = n > 1 ? __builtin_sqrt(ss / (double)(n - 1)) : __builtin_nan("");     // torch.std of one element is nan
    return TW_OK;
}

extern "C" int tw_collected_pack_trainer(const tw_collected *c, uint32_t obs_size, int normalize_advantage, uint64_t row_begin,
                                         uint64_t row_count, float *obs_onehot, float *log_probs, int64_t *actions, int64_t *perms,
                                         float *advs)
{
    if (!c) { set_error("tw_collected_pack_trainer: null result"); return TW_ERR_INVALID; }
    if (row_begin > c->n_records || row_count > c->n_records - row_begin) {
        set_error("tw_collected_pack_trainer: rows [%llu, +%llu) outside the %llu records", (unsigned long long)row_begin,
                  (unsigned long long)row_count, (unsigned long long)c->n_records);
        return TW_ERR_INVALID;
    }
    if (obs_onehot && c->obs_width != 1) { set_error("tw_collected_pack_trainer: one-hot packing exists for one-byte obs ids (obs_size <= 256)"); return TW_ERR_UNSUPPORTED; }
    if (obs_onehot && (obs_size == 0 || obs_size > 256 || obs_size < c->n_cells)) { set_error("tw_collected_pack_trainer: obs_size %u", obs_size); return TW_ERR_INVALID; }
    if ((log_probs || actions || advs) && !c->is_ppo) { set_error("tw_collected_pack_trainer: log_probs / actions / advs exist for PPO data only"); return TW_ERR_INVALID; }
    hipStream_t s = current_stream();
    int rc = TW_OK;
    if (obs_onehot)
        rc = launch_onehot(reinterpret_cast<const uint8_t *>(c->field_ptr[TW_F_OBS]), row_begin, row_count, (int)c->n_cells, (int)obs_size, obs_onehot, s);
    if (rc) return rc;
    float mean = 0.0f, denom = 1.0f;
    if (advs && normalize_advantage) {
        double m, sd;
        rc = tw_collected_adv_stats(c, &m, &sd); if (rc) return rc;
        mean = (float)m; denom = (float)sd + 1e-8f;                                  // ppo.py:56, f32 like torch
    }
    if (log_probs || actions || perms || advs)
        rc = launch_ppo_pack(reinterpret_cast<const float *>(c->field_ptr[TW_F_LOGITS]), reinterpret_cast<const uint8_t *>(c->field_ptr[TW_F_ACTIONS]),
                             reinterpret_cast<const int8_t *>(c->field_ptr[TW_F_PERMS]), reinterpret_cast<const float *>(c->field_ptr[TW_F_ADVS]),
                             row_begin, row_count, (int)c->n_actions, mean, denom, normalize_advantage ? 1 : 0,
                             c->is_ppo ? log_probs : nullptr, c->is_ppo ? actions : nullptr, perms, c->is_ppo ? advs : nullptr, s);
    return rc;
}

namespace tw {
const PolicyDev *policy_dev(const tw_policy *p) { return &p->dev; }
void collected_adopt_obs_width(tw_collected *c, uint32_t obs_width) { c->obs_width = obs_width; }
int policy_device_image(tw_policy *p, void **image, size_t *bytes)
{
    if (!p || !p->arena) { set_error("policy: no device image"); return TW_ERR_INVALID; }
    *image = p->arena; *bytes = p->arena_bytes;
    return TW_OK;
}
// after the arena was overwritten with another process's image (tw_comm_broadcast_policy): the layer table of a generic stack
// holds device POINTERS, which are this process's own -- put them back
int policy_restore_local_tables(tw_policy *p, hipStream_t s)
{
    if (!p->dev.generic || p->gen_layers.empty()) return TW_OK;
    TW_HIP(hipMemcpyAsync(const_cast<LayerDev *>(p->dev.layers), p->gen_layers.data(), p->gen_layers.size() * sizeof(LayerDev), hipMemcpyHostToDevice, s));
    return TW_OK;
}
int collected_describe(const tw_collected *c, int *is_ppo, uint32_t *n_cells, uint32_t *n_actions, uint64_t *n_records, uint64_t *n_episodes)
{
    if (!c) { set_error("null collected data"); return TW_ERR_INVALID; }
    *is_ppo = c->is_ppo; *n_cells = c->n_cells; *n_actions = c->n_actions; *n_records = c->n_records; *n_episodes = c->n_episodes;
    return TW_OK;
}
const void *collected_field(const tw_collected *c, int field) { return c->field_ptr[field]; }
int collected_adopt(void *arena, size_t arena_bytes, int device, int is_ppo, uint32_t n_cells, uint32_t n_actions, uint64_t n_records,
                    uint64_t n_episodes, void *const (&field_ptr)[TW_F_COUNT], const size_t (&field_bytes)[TW_F_COUNT], tw_collected **out)
{
    tw_collected *c = new tw_collected();
    c->stream = current_stream();
    c->arena = arena; c->arena_cap = arena_bytes; c->device = device;
    c->is_ppo = is_ppo; c->n_cells = n_cells; c->n_actions = n_actions; c->n_records = n_records; c->n_episodes = n_episodes;
    for (int f = 0; f < TW_F_COUNT; ++f) { c->field_ptr[f] = field_ptr[f]; c->field_bytes[f] = field_bytes[f]; }
    c->stats.records = n_records; c->stats.episodes = n_episodes;
    *out = c;
    return TW_OK;
}
}  // namespace tw

extern "C" void tw_collected_free(tw_collected *c)
{
    if (!c) return;
    arena_release(c->arena, c->arena_cap, c->device, c->stream);
    delete c;
}

// ====================================================================================== PPO collect
namespace {

struct EventSet {
    hipEvent_t ev[5] = {};
    int n = 0;
    ~EventSet() { for (int i = 0; i < n; ++i) (void)hipEventDestroy(ev[i]); }
    int init() { for (; n < 5; ++n) TW_HIP(hipEventCreate(&ev[n])); return TW_OK; }
};

int make_env_consts(const tw_puzzle_desc *env, PuzzleConsts *out, uint64_t max_cells = 16)
{
    const uint64_t nc = (uint64_t)env->width * env->height;
    if (env->width == 0 || env->height == 0 || nc > max_cells) {
        set_error("Puzzle %ux%u: the HIP path packs the board as 16 nibbles (width*height <= 16)", env->width, env->height);
        return TW_ERR_UNSUPPORTED;
    }
    const uint64_t depth0 = (uint64_t)env->depth_slope * env->difficulty;
    if (depth0 > 1022 || env->max_depth == 0) {
        set_error("Puzzle: depth_slope*difficulty = %llu exceeds the supported 1022 (or max_depth == 0)", (unsigned long long)depth0);
        return TW_ERR_UNSUPPORTED;
    }
    out->width = (int)env->width; out->height = (int)env->height; out->n_cells = (int)nc;
    out->difficulty = (int)env->difficulty; out->depth0 = (int)depth0;
    out->r_step = -0.5f / (float)env->max_depth;           // puzzle.rs:175
    uint64_t id = 0;
    for (uint64_t i = 0; i < nc && i < 16; ++i) id |= i << (4 * i);          // (boards above 16 cells have their own packing: tw_rollout_big.hip)
    out->ident = id;
    return TW_OK;
}

}  // namespace

// test hook: the start boards of episodes [episode_offset, episode_offset + n) and the order the self-play walkers take them in
// (init_boards_kernel + episode_order_kernel, exactly as tw_az_collect launches them), copied to the host
extern "C" int tw_debug_episode_order(const tw_puzzle_desc *env, uint64_t seed, uint64_t episode_offset, uint64_t n, uint64_t *boards_out, uint32_t *order_out)
{
    if (!env || !boards_out || !order_out) { set_error("tw_debug_episode_order: null argument"); return TW_ERR_INVALID; }
    if (n == 0 || n > (1ull << 24)) { set_error("tw_debug_episode_order: n = %llu out of range", (unsigned long long)n); return TW_ERR_INVALID; }
    int rc = require_device(); if (rc) return rc;
    PuzzleConsts envc; rc = make_env_consts(env, &envc, 16); if (rc) return rc;
    hipStream_t s = current_stream();
    void *buf = nullptr;
    TW_HIP(hipMalloc(&buf, n * 12 + 256 + episode_order_scratch_bytes(n)));
    uint64_t *b = reinterpret_cast<uint64_t *>(buf); uint32_t *o = reinterpret_cast<uint32_t *>(b + n);
    void *scr = reinterpret_cast<uint8_t *>(buf) + align_up(n * 12, 256);
    rc = launch_init_boards(envc, seed, episode_offset, n, b, s);
    if (!rc) rc = launch_episode_order(envc, b, n, o, scr, s);
    hipError_t e = rc ? hipSuccess : hipMemcpyAsync(boards_out, b, n * 8, hipMemcpyDeviceToHost, s);
    if (!rc && e == hipSuccess) e = hipMemcpyAsync(order_out, o, n * 4, hipMemcpyDeviceToHost, s);
    if (!rc && e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(buf);
    if (rc) return rc;
    TW_HIP(e);
    return TW_OK;
}

// Boards above 16 cells (puzzle.rs:34-42 takes any width x height; the kernels pack a board as 16 nibbles): the Puzzle steps
// on the host through the any-environment collectors (tw_env_generic.hip) -- same RNG spec, same arithmetic, the policy
// evaluations of a moment in one batched launch.  f32; evaluate / solve of such boards are not implemented.
static tw_env_vtable puzzle_env_table(tw_puzzle *proto)
{
    tw_env_vtable vt{};
    vt.prototype = proto; vt.num_actions = 4; vt.n_obs = (uint32_t)proto->state.size(); vt.obs_size = vt.n_obs * vt.n_obs;
    vt.clone = [](void *e) -> void * { return tw_puzzle_clone(static_cast<tw_puzzle *>(e)); };
    vt.destroy = [](void *e) { tw_puzzle_destroy(static_cast<tw_puzzle *>(e)); };
    vt.reset = [](void *e, uint64_t seed, uint64_t episode) { (void)tw_puzzle_reset(static_cast<tw_puzzle *>(e), seed, episode); };
    vt.step = [](void *e, uint32_t a) { (void)tw_puzzle_step(static_cast<tw_puzzle *>(e), a); };
    vt.observe = [](void *e, int32_t *o) {
        const tw_puzzle *q = static_cast<const tw_puzzle *>(e);
        const int64_t n = (int64_t)q->state.size();
        for (int64_t i = 0; i < n; ++i) o[i] = (int32_t)(i * n + q->state[(size_t)i]);       // puzzle.rs:183-185
    };
    vt.masks = [](void *e, uint8_t *m) { (void)tw_puzzle_masks(static_cast<tw_puzzle *>(e), m); };
    vt.reward = [](void *e) -> float { return tw_puzzle_reward(static_cast<tw_puzzle *>(e)); };
    vt.is_final = [](void *e) -> int { return tw_puzzle_is_final(static_cast<tw_puzzle *>(e)); };
    vt.success = [](void *e) -> int { return tw_puzzle_solved(static_cast<tw_puzzle *>(e)); };   // puzzle.rs:179-181
    return vt;
}

static int big_board_checks(const tw_puzzle_desc *env, uint32_t precision, uint64_t *depth0)
{
    if (precision != TW_PREC_F32_EXACT) { set_error("boards above 16 cells run in f32 only"); return TW_ERR_UNSUPPORTED; }
    if ((uint64_t)env->width * env->height > 64) { set_error("Puzzle %ux%u: at most 64 cells", env->width, env->height); return TW_ERR_UNSUPPORTED; }
    *depth0 = (uint64_t)env->depth_slope * env->difficulty;
    if (*depth0 + 1 > 0xfffffffeull) { set_error("Puzzle: depth_slope*difficulty too large"); return TW_ERR_UNSUPPORTED; }
    return TW_OK;
}

static int collect_big_board(const tw_puzzle_desc *env, const tw_policy *policy, const tw_ppo_params *ppo, const tw_az_params *az, tw_collected **out)
{
    uint64_t depth0 = 0;
    int rc = big_board_checks(env, ppo ? ppo->precision : az->precision, &depth0); if (rc) return rc;
    tw_puzzle *proto = tw_puzzle_create(env->width, env->height, env->difficulty, env->depth_slope, env->max_depth);
    if (!proto) return TW_ERR_INVALID;
    tw_env_vtable vt = puzzle_env_table(proto);
    rc = ppo ? tw_ppo_collect_env(&vt, policy, ppo, (uint32_t)(depth0 + 1), out) : tw_az_collect_env(&vt, policy, az, (uint32_t)(depth0 + 1), out);
    tw_puzzle_destroy(proto);
    return rc;
}

extern "C" int tw_ppo_collect(const tw_puzzle_desc *env, const tw_policy *policy, const tw_ppo_params *prm,
                              tw_collected **out)
{
    if (!env || !policy || !prm || !out) { set_error("tw_ppo_collect: null argument"); return TW_ERR_INVALID; }
    *out = nullptr;
    if (prm->num_episodes == 0) {
        set_error("Something went wrong. No data in collected data chunks to merge. ");   // collector.rs:41
        return TW_ERR_EMPTY;
    }
    if (prm->precision > TW_PREC_F16X2) { set_error("tw_ppo_collect: unknown precision %u", prm->precision); return TW_ERR_INVALID; }
    int rc = require_device(); if (rc) return rc;

    // boards of 17 .. 64 cells roll out on the device too (tw_rollout_big.hip: 5-bit or one-byte cells, two-byte obs ids, the generic
    // engine); whatever that kernel does not take steps on the host (tw_env_generic.hip)
    const uint64_t cells = (uint64_t)env->width * env->height;
    const bool big = cells > 16;
    if (big && !(cells <= 64 && policy->dev.generic && prm->precision == TW_PREC_F32_EXACT && (uint64_t)env->depth_slope * env->difficulty <= 1022 &&
                 env->max_depth != 0 && !launch_options().force_geom))
        return collect_big_board(env, policy, prm, nullptr, out);

    RolloutArgs ra{};
    rc = make_env_consts(env, &ra.env, 64); if (rc) return rc;
    ra.pol = policy->dev;
    if (ra.pol.obs_size != ra.env.n_cells * ra.env.n_cells) {
        set_error("index out of bounds: policy obs_size %d != Puzzle obs ids %d", ra.pol.obs_size, ra.env.n_cells * ra.env.n_cells);
        return TW_ERR_INVALID;
    }
    if (ra.pol.n_actions != 4) { set_error("Puzzle has 4 actions, policy has %d", ra.pol.n_actions); return TW_ERR_INVALID; }
    if (ra.pol.generic && prm->precision != TW_PREC_F32_EXACT) { set_error("tw_ppo_collect: the f16 modes exist for the one-common-layer policy shape only"); return TW_ERR_UNSUPPORTED; }
    const uint64_t E = prm->num_episodes;
    const int t_pad = ra.env.depth0 + 1;
    ra.num_episodes = E; ra.episode_offset = prm->episode_offset; ra.seed = prm->seed;

    std::lock_guard<std::mutex> lock(g_ws_mutex);
    hipStream_t s = current_stream();

    // ---- workspace carve ---------------------------------------------------------------------
    const uint64_t R = E * (uint64_t)t_pad;
    size_t cur = 0;
    auto seg = [&](size_t bytes) { size_t o = cur; cur = align_up(cur + bytes, 256); return o; };
    // persistent-lane mode (more episodes than resident lanes): start boards + episode queue
    ra.reserve_cus = (int)(prm->reserve_cus > 0x7fffu ? 0x7fffu : prm->reserve_cus);
    // (generic policy stacks: one 16-episode workgroup per CU -- rollout_f32_resident_episodes counts 256 per CU)
    const uint64_t resident = ra.pol.generic ? rollout_generic_resident_episodes(ra.pol, ra.env.n_cells, ra.reserve_cus)
                            : prm->precision == TW_PREC_F32_EXACT ? f32_resident_episodes(E, (int)ra.pol.hidden, false, ra.reserve_cus)
                                                                  : rollout_f32_resident_episodes(ra.reserve_cus);
    const bool persist = E > resident && !launch_options().no_persist && !big;
    const size_t o_rec = seg(R * sizeof(PaddedRec)), o_len = seg(E * 4), o_start = seg(E * 8), o_total = seg(8),
                 o_scan = seg(scan_scratch_bytes(E)), o_init = seg(persist ? E * 16 : 0), o_queue = seg(persist ? 4 : 0),
                 o_obs16 = seg(big ? R * cells * 2 : 0), o_boards = seg(persist ? E * 8 : 0), o_ordscr = seg(persist ? episode_order_scratch_bytes(E) : 0);
    void *wsp = nullptr;
    rc = ws_reserve(cur, &wsp); if (rc) return rc;
    uint8_t *ws = reinterpret_cast<uint8_t *>(wsp);
    ra.out.rec = reinterpret_cast<PaddedRec *>(ws + o_rec);
    ra.out.ep_len = reinterpret_cast<uint32_t *>(ws + o_len); ra.out.t_pad = t_pad;
    uint64_t *ep_start_ws = reinterpret_cast<uint64_t *>(ws + o_start);
    uint64_t *total_d = reinterpret_cast<uint64_t *>(ws + o_total);

    EventSet ev; rc = ev.init(); if (rc) return rc;
    tw_collect_stats st{};
    if (persist) {
        ra.init_boards = reinterpret_cast<const uint4 *>(ws + o_init);
        ra.queue = reinterpret_cast<unsigned int *>(ws + o_queue);
        const unsigned int first = (unsigned int)resident;      // episodes handed out at launch
        TW_HIP(hipMemcpyAsync(ws + o_queue, &first, 4, hipMemcpyHostToDevice, s));
        // the lanes take the episodes longest-looking first (as the self-play walkers do): with a policy that solves the puzzle an episode
        // is about as long as its start board is far from the solved one, and the collect ends with its last long episode -- trained
        // Puzzle-8 policy, 262,144 envs: D = 32 12.2 -> 11.5 ms, D = 12 4.85 -> 3.36 ms; 1,048,576 envs 25.2 -> 22.8 ms
        // (scripts/ragged_trained.py).  Diagnostic, TW_OPT_AZ_VARIANT + 64: by index
        const bool ordered = !(launch_options().az_variant & 64);
        rc = launch_init_boards(ra.env, ra.seed, ra.episode_offset, E, ordered ? reinterpret_cast<uint64_t *>(ws + o_boards) : nullptr, s,
                                ordered ? nullptr : reinterpret_cast<uint4 *>(ws + o_init));
        if (rc) return rc;
        if (ordered) {
            rc = launch_episode_order(ra.env, reinterpret_cast<const uint64_t *>(ws + o_boards), E, nullptr, ws + o_ordscr, s, reinterpret_cast<uint4 *>(ws + o_init));
            if (rc) return rc;
        }
    }
    TW_HIP(hipEventRecord(ev.ev[0], s));
    rc = big                               ? launch_rollout_big(ra, reinterpret_cast<uint16_t *>(ws + o_obs16), s, &st.rollout_blocks, &st.rollout_threads)
         : prm->precision == TW_PREC_F16   ? launch_rollout_f16(ra, s, &st.rollout_blocks, &st.rollout_threads)
         : prm->precision == TW_PREC_F16X2 ? launch_rollout_f16x2(ra, s, &st.rollout_blocks, &st.rollout_threads)
                                           : launch_rollout_f32(ra, s, &st.rollout_blocks, &st.rollout_threads);
    if (rc) return rc;
    TW_HIP(hipEventRecord(ev.ev[1], s));
    rc = launch_scan(ra.out.ep_len, E, prm->merge_order ? 1 : 0, ep_start_ws, total_d, ws + o_scan, scan_scratch_bytes(E), s);
    if (rc) return rc;
    TW_HIP(hipEventRecord(ev.ev[2], s));
    uint64_t total = 0;
    TW_HIP(hipMemcpyAsync(&total, total_d, 8, hipMemcpyDeviceToHost, s));
    TW_HIP(hipStreamSynchronize(s));
    if (total == 0 || total > R) { set_error("collect: inconsistent record count %llu (max %llu)", (unsigned long long)total, (unsigned long long)R); return TW_ERR_HIP; }

    // ---- compact result ----------------------------------------------------------------------
    tw_collected *c = new tw_collected();
    c->stream = current_stream();
    c->n_records = total; c->n_episodes = E; c->n_cells = (uint32_t)ra.env.n_cells; c->n_actions = 4; c->is_ppo = 1;
    size_t ccur = 0;
    auto cseg = [&](int f, size_t bytes) { c->field_bytes[f] = bytes; size_t o = ccur; ccur = align_up(ccur + bytes, 256); return o; };
    if (big) c->obs_width = 2;
    const size_t c_obs = cseg(TW_F_OBS, total * c->n_cells * c->obs_width), c_lg = cseg(TW_F_LOGITS, total * 16), c_prm = cseg(TW_F_PERMS, total),
                 c_val = cseg(TW_F_VALUES, total * 4), c_rew = cseg(TW_F_REWARDS, total * 4), c_act = cseg(TW_F_ACTIONS, total),
                 c_adv = cseg(TW_F_ADVS, total * 4), c_ret = cseg(TW_F_RETS, total * 4), c_len = cseg(TW_F_EP_LEN, E * 4),
                 c_start = cseg(TW_F_EP_START, E * 8);
    rc = hipGetDevice(&c->device) == hipSuccess ? arena_acquire(ccur, &c->arena, &c->arena_cap) : TW_ERR_HIP;
    if (rc) { delete c; return rc; }
    uint8_t *ca = reinterpret_cast<uint8_t *>(c->arena);
    const size_t offs[TW_F_COUNT] = {c_obs, c_lg, c_prm, c_val, c_rew, c_act, c_adv, c_ret, 0, c_len, c_start};
    for (int f = 0; f < TW_F_COUNT; ++f) c->field_ptr[f] = c->field_bytes[f] ? ca + offs[f] : nullptr;
    CompactTraj ct{};
    ct.obs = ca + c_obs; ct.logits = reinterpret_cast<float *>(ca + c_lg); ct.perms = reinterpret_cast<int8_t *>(ca + c_prm);
    ct.values = reinterpret_cast<float *>(ca + c_val); ct.rewards = reinterpret_cast<float *>(ca + c_rew);
    ct.actions = ca + c_act; ct.advs = reinterpret_cast<float *>(ca + c_adv); ct.rets = reinterpret_cast<float *>(ca + c_ret);

#define TW_HIP_C(call) do { hipError_t _e = (call); if (_e != hipSuccess) { tw_collected_free(c); return hip_fail(_e, #call, __FILE__, __LINE__); } } while (0)
    TW_HIP_C(hipEventRecord(ev.ev[3], s));
    rc = launch_finalize_ppo(ra.out, ep_start_ws, E, big ? 0 : ra.env.n_cells, prm->gamma, prm->lambda, ct, s);      // (0 cells: the obs ids come from their own array)
    if (rc == TW_OK && big)
        rc = launch_compact_obs16(reinterpret_cast<const uint16_t *>(ws + o_obs16), ra.out.ep_len, ep_start_ws, E, t_pad, ra.env.n_cells,
                                  reinterpret_cast<uint16_t *>(ca + c_obs), s);
    if (rc) { tw_collected_free(c); return rc; }
    TW_HIP_C(hipMemcpyAsync(ca + c_len, ra.out.ep_len, E * 4, hipMemcpyDeviceToDevice, s));
    TW_HIP_C(hipMemcpyAsync(ca + c_start, ep_start_ws, E * 8, hipMemcpyDeviceToDevice, s));
    TW_HIP_C(hipEventRecord(ev.ev[4], s));
    TW_HIP_C(hipStreamSynchronize(s));
    float ms = 0;
    TW_HIP_C(hipEventElapsedTime(&ms, ev.ev[0], ev.ev[1])); st.ms_rollout = ms;
    TW_HIP_C(hipEventElapsedTime(&ms, ev.ev[1], ev.ev[2])); st.ms_scan = ms;
    TW_HIP_C(hipEventElapsedTime(&ms, ev.ev[3], ev.ev[4])); st.ms_finalize = ms;
    TW_HIP_C(hipEventElapsedTime(&ms, ev.ev[0], ev.ev[4])); st.ms_total = ms;
#undef TW_HIP_C
    st.records = total; st.episodes = E; st.padded_bytes = cur; st.forward_evals = total;
    c->stats = st;
    *out = c;
    return TW_OK;
}

static int az_collect_once(const tw_puzzle_desc *env, const tw_policy *policy, const tw_az_params *prm, tw_collected **out, bool *split_watchdog);

// The split shape of the walker kernel needs its two kernels on the chip at the same time.  Where something keeps them apart -- a tool that
// serialises kernel launches (rocprofv3 --pmc), another process holding the CUs with persistent kernels of its own -- the first such collect
// runs into its watchdog (seconds, never a hang); the library then says so once on stderr, stops using the split shape in this process and
// runs the collect again on the shapes inside one workgroup: same bytes, the HIP path throughout.
extern "C" int tw_az_collect(const tw_puzzle_desc *env, const tw_policy *policy, const tw_az_params *prm, tw_collected **out)
{
    bool split_watchdog = false;
    int rc = az_collect_once(env, policy, prm, out, &split_watchdog);
    if (rc != TW_OK && split_watchdog) {
        fprintf(stderr, "[twisterl_hip] self-play: the split shape's kernels did not run side by side (a profiler that serialises kernels? a shared GPU?): "
                        "%s -- retrying on the single-kernel shapes, which this process keeps from now on\n", tw_last_error());
        mcts_deep_disable_split();
        rc = az_collect_once(env, policy, prm, out, &split_watchdog);
    }
    return rc;
}

static int az_collect_once(const tw_puzzle_desc *env, const tw_policy *policy, const tw_az_params *prm, tw_collected **out, bool *split_watchdog)
{
    *split_watchdog = false;
    if (!env || !policy || !prm || !out) { set_error("tw_az_collect: null argument"); return TW_ERR_INVALID; }
    *out = nullptr;
    if (prm->num_episodes == 0) {
        set_error("Something went wrong. No data in collected data chunks to merge. ");   // collector.rs:41
        return TW_ERR_EMPTY;
    }
    if (prm->precision != TW_PREC_F32_EXACT) { set_error("tw_az_collect: precision %u not implemented", prm->precision); return TW_ERR_UNSUPPORTED; }
    int rc = require_device(); if (rc) return rc;
    // boards of 17 .. 64 cells: self-play on the device too (tw_mcts_big.hip: boards of 5-bit or one-byte cells, nodes without a
    // board, two-byte obs ids, the generic engine); whatever that kernel does not take steps on the host (tw_env_generic.hip)
    const uint64_t cells = (uint64_t)env->width * env->height;
    const bool big = cells > 16;
    if (big && !(cells <= 64 && policy->dev.generic && (uint64_t)env->depth_slope * env->difficulty <= 1022 && env->max_depth != 0 &&
                 !launch_options().force_geom))
        return collect_big_board(env, policy, nullptr, prm, out);

    MctsArgs ma{};
    rc = make_env_consts(env, &ma.env, 64); if (rc) return rc;
    ma.pol = policy->dev;
    if (ma.pol.obs_size != ma.env.n_cells * ma.env.n_cells) {
        set_error("index out of bounds: policy obs_size %d != Puzzle obs ids %d", ma.pol.obs_size, ma.env.n_cells * ma.env.n_cells);
        return TW_ERR_INVALID;
    }
    if (ma.pol.n_actions != 4) { set_error("Puzzle has 4 actions, policy has %d", ma.pol.n_actions); return TW_ERR_INVALID; }
    const uint64_t E = prm->num_episodes;
    const int t_pad = ma.env.depth0 + 1;
    const uint64_t cap64 = 5ull + 4ull * prm->num_mcts_searches * (prm->max_expand_depth ? prm->max_expand_depth : 1u);
    if (cap64 > 0x7fffffffull || (uint64_t)prm->num_mcts_searches * (prm->max_expand_depth ? prm->max_expand_depth : 1u) > 0xffffffffull) {
        set_error("tw_az_collect: num_mcts_searches x max_expand_depth too large"); return TW_ERR_UNSUPPORTED;
    }
    ma.num_episodes = E; ma.episode_offset = prm->episode_offset; ma.seed = prm->seed;
    ma.num_searches = prm->num_mcts_searches; ma.max_expand_depth = prm->max_expand_depth; ma.C = prm->C;
    ma.node_cap = (uint32_t)cap64;

    std::lock_guard<std::mutex> lock(g_ws_mutex);
    hipStream_t s = current_stream();

    const uint64_t R = E * (uint64_t)t_pad;
    size_t cur = 0;
    auto seg = [&](size_t bytes) { size_t o = cur; cur = align_up(cur + bytes, 256); return o; };
    // persistent lanes (more episodes than resident lanes): one tree arena per LANE, start boards + episode queue
    ma.reserve_cus = (int)(prm->reserve_cus > 0x7fffu ? 0x7fffu : prm->reserve_cus);
    // few, deep searches: the walker-per-wave shape (tw_mcts_deep.hip) -- always persistent, 64-byte nodes, one arena per walker
    const bool deep = !big && mcts_deep_applies(ma);
    const uint64_t resident = big ? E : deep ? mcts_deep_walkers(E, ma.reserve_cus, ma.num_searches) : f32_resident_episodes(E, (int)ma.pol.hidden, true, ma.reserve_cus);
    const bool persist = !big && (deep || (E > resident && !launch_options().no_persist && !ma.pol.generic));
    const uint64_t arenas = persist ? resident : E;
    const size_t o_rec = seg(R * sizeof(PaddedRec)), o_len = seg(E * 4), o_start = seg(E * 8),
                 o_total = seg(136), o_scan = seg(scan_scratch_bytes(E)),
                 o_arena = seg(big ? arenas * cap64 * mcts_big_node_bytes() : deep ? arenas * mcts_deep_arena_bytes(cap64) : arenas * cap64 * mcts_node_bytes()),
                 o_init = seg(persist ? E * 8 : 0), o_queue = seg(persist ? 4 : 0), o_order = seg(deep ? E * 4 : 0), o_ordscr = seg(deep ? episode_order_scratch_bytes(E) : 0),
                 o_obs16 = seg(big ? R * cells * 2 : 0);
    const uint32_t tbl_entries = deep ? mcts_deep_table_entries(ma.num_searches, ma.max_expand_depth) : 0;
    const size_t tbl_bytes = (size_t)arenas * tbl_entries * 32;
    const size_t o_tbl = seg(tbl_bytes);
    const bool split = deep && mcts_deep_split(E, ma.reserve_cus, ma.num_searches);      // walkers and engine as two kernels: a 256-byte mailbox per walker
    const size_t o_mb = seg(split ? (size_t)arenas * 256 + 256 : 0);          // (+ the engines' residency counter)
    size_t free_b = 0, total_b = 0;
    TW_HIP(hipMemGetInfo(&free_b, &total_b));
    {   // what can actually be had: free memory plus the cached workspace this call would replace
        int dev_now = 0; TW_HIP(hipGetDevice(&dev_now));
        const size_t avail = free_b + (g_ws.ptr && g_ws.device == dev_now ? g_ws.cap : 0);
        if (cur > g_ws.cap && cur > (size_t)(0.95 * (double)avail)) {
            set_error("tw_az_collect: %zu bytes of tree arenas + trajectories exceed the free device memory (%zu free of %zu)", cur, avail, total_b);
            return TW_ERR_UNSUPPORTED;
        }
    }
    void *wsp = nullptr;
    rc = ws_reserve(cur, &wsp); if (rc) return rc;
    uint8_t *ws = reinterpret_cast<uint8_t *>(wsp);
    ma.out.rec = reinterpret_cast<PaddedRec *>(ws + o_rec);
    ma.out.ep_len = reinterpret_cast<uint32_t *>(ws + o_len); ma.out.t_pad = t_pad;
    uint64_t *ep_start_ws = reinterpret_cast<uint64_t *>(ws + o_start);
    uint64_t *total_d = reinterpret_cast<uint64_t *>(ws + o_total);
    ma.eval_count = reinterpret_cast<unsigned long long *>(ws + o_total + 8);
    ma.arena = reinterpret_cast<MctsNode *>(ws + o_arena);

    EventSet ev; rc = ev.init(); if (rc) return rc;
    tw_collect_stats st{};
    TW_HIP(hipMemsetAsync(ws + o_total, 0, 136, s));
    ma.reuse_mode = (uint32_t)launch_options().az_reuse;
    if (deep) {          // the walkers' board-keyed output tables start empty (the policy may have changed since the last collect)
        ma.tbl = ws + o_tbl; ma.tbl_entries = tbl_entries;
        TW_HIP(hipMemsetAsync(ws + o_tbl, 0, tbl_bytes, s));
    }
    if (split) {
        ma.mailbox = reinterpret_cast<uint32_t *>(ws + o_mb);
        TW_HIP(hipMemsetAsync(ws + o_mb, 0, (size_t)arenas * 256 + 256, s));
    }
    if (persist) {
        ma.init_boards = reinterpret_cast<const uint64_t *>(ws + o_init);
        ma.queue = reinterpret_cast<unsigned int *>(ws + o_queue);
        const unsigned int first = (unsigned int)(resident < E ? resident : E);      // episodes handed out at launch
        TW_HIP(hipMemcpyAsync(ws + o_queue, &first, 4, hipMemcpyHostToDevice, s));
        rc = launch_init_boards(ma.env, ma.seed, ma.episode_offset, E, reinterpret_cast<uint64_t *>(ws + o_init), s);
        if (rc) return rc;
        // walker kernel: longest-looking episodes first (diagnostic, TW_OPT_AZ_VARIANT + 64: by index).  (The lane-per-episode kernel's
        // queue gains nothing from it: 16,384 x 100 71.5 against 75.3 ms, but 262,144 x 32 186.9 against 182.2, 32,768 x 32 47.1 against 46.0)
        if (deep && !(launch_options().az_variant & 64)) {
            rc = launch_episode_order(ma.env, ma.init_boards, E, reinterpret_cast<uint32_t *>(ws + o_order), ws + o_ordscr, s);
            if (rc) return rc;
            ma.order = reinterpret_cast<const uint32_t *>(ws + o_order);
            ma.order_across = 1;
        }
    }
    TW_HIP(hipEventRecord(ev.ev[0], s));
    rc = big  ? launch_mcts_big(ma, reinterpret_cast<uint16_t *>(ws + o_obs16), s, &st.rollout_blocks, &st.rollout_threads)
       : deep ? launch_mcts_deep(ma, s, &st.rollout_blocks, &st.rollout_threads)
              : launch_mcts_f32(ma, s, &st.rollout_blocks, &st.rollout_threads);
    if (rc) return rc;
    TW_HIP(hipEventRecord(ev.ev[1], s));
    rc = launch_scan(ma.out.ep_len, E, prm->merge_order ? 1 : 0, ep_start_ws, total_d, ws + o_scan, scan_scratch_bytes(E), s);
    if (rc) return rc;
    TW_HIP(hipEventRecord(ev.ev[2], s));
    uint64_t host_tot[17] = {0};
    TW_HIP(hipMemcpyAsync(host_tot, ws + o_total, 136, hipMemcpyDeviceToHost, s));
    TW_HIP(hipStreamSynchronize(s));
    const uint64_t total = host_tot[0];
    {
        std::lock_guard<std::mutex> dl(g_dbg_mutex);
        for (int i = 0; i < 16; ++i) g_dbg_counters[i] = host_tot[1 + i];
    }
    if (host_tot[1 + 12] != 0) {       // eval_count[12]: a stored output offered for a board it was not computed for (never, by construction)
        set_error("az collect: %llu reused network outputs failed the board check", (unsigned long long)host_tot[1 + 12]);
        return TW_ERR_HIP;
    }
    if (host_tot[1 + 13] != 0) {       // eval_count[13]: the decoupled walker shape's watchdog (a request / completion hand-shake that never completed)
        *split_watchdog = split;
        set_error("az collect: the walker kernel's watchdog fired (%llu waves gave up waiting).  The split shape needs its two kernels to run at the same time: "
                  "under a tool that serialises kernels (rocprofv3 --pmc) pin the shapes inside one workgroup, TW_OPT_AZ_VARIANT + 1024", (unsigned long long)host_tot[1 + 13]);
        return TW_ERR_HIP;
    }
    if (total == 0 || total > R) { set_error("az collect: inconsistent record count %llu (max %llu)", (unsigned long long)total, (unsigned long long)R); return TW_ERR_HIP; }

    tw_collected *c = new tw_collected();
    c->stream = current_stream();
    c->n_records = total; c->n_episodes = E; c->n_cells = (uint32_t)ma.env.n_cells; c->n_actions = 4; c->is_ppo = 0;
    size_t ccur = 0;
    auto cseg = [&](int f, size_t bytes) { c->field_bytes[f] = bytes; size_t o = ccur; ccur = align_up(ccur + bytes, 256); return o; };
    if (big) c->obs_width = 2;
    const size_t c_obs = cseg(TW_F_OBS, total * c->n_cells * c->obs_width), c_lg = cseg(TW_F_LOGITS, total * 16), c_prm = cseg(TW_F_PERMS, total),
                 c_rem = cseg(TW_F_REMAINING, total * 4), c_len = cseg(TW_F_EP_LEN, E * 4), c_start = cseg(TW_F_EP_START, E * 8);
    rc = hipGetDevice(&c->device) == hipSuccess ? arena_acquire(ccur, &c->arena, &c->arena_cap) : TW_ERR_HIP;
    if (rc) { delete c; return rc; }
    uint8_t *ca = reinterpret_cast<uint8_t *>(c->arena);
    c->field_ptr[TW_F_OBS] = ca + c_obs; c->field_ptr[TW_F_LOGITS] = ca + c_lg; c->field_ptr[TW_F_PERMS] = ca + c_prm;
    c->field_ptr[TW_F_REMAINING] = ca + c_rem; c->field_ptr[TW_F_EP_LEN] = ca + c_len; c->field_ptr[TW_F_EP_START] = ca + c_start;

#define TW_HIP_C(call) do { hipError_t _e = (call); if (_e != hipSuccess) { tw_collected_free(c); return hip_fail(_e, #call, __FILE__, __LINE__); } } while (0)
    TW_HIP_C(hipEventRecord(ev.ev[3], s));
    rc = launch_finalize_az(ma.out, ep_start_ws, E, big ? 0 : ma.env.n_cells, ca + c_obs, reinterpret_cast<float *>(ca + c_lg),      // (0 cells: the obs ids come from their own array)
                            reinterpret_cast<int8_t *>(ca + c_prm), reinterpret_cast<float *>(ca + c_rem), s);
    if (rc == TW_OK && big)
        rc = launch_compact_obs16(reinterpret_cast<const uint16_t *>(ws + o_obs16), ma.out.ep_len, ep_start_ws, E, t_pad, (int)cells,
                                  reinterpret_cast<uint16_t *>(ca + c_obs), s);
    if (rc) { tw_collected_free(c); return rc; }
    TW_HIP_C(hipMemcpyAsync(ca + c_len, ma.out.ep_len, E * 4, hipMemcpyDeviceToDevice, s));
    TW_HIP_C(hipMemcpyAsync(ca + c_start, ep_start_ws, E * 8, hipMemcpyDeviceToDevice, s));
    TW_HIP_C(hipEventRecord(ev.ev[4], s));
    TW_HIP_C(hipStreamSynchronize(s));
    float ms = 0;
    TW_HIP_C(hipEventElapsedTime(&ms, ev.ev[0], ev.ev[1])); st.ms_rollout = ms;
    TW_HIP_C(hipEventElapsedTime(&ms, ev.ev[1], ev.ev[2])); st.ms_scan = ms;
    TW_HIP_C(hipEventElapsedTime(&ms, ev.ev[3], ev.ev[4])); st.ms_finalize = ms;
    TW_HIP_C(hipEventElapsedTime(&ms, ev.ev[0], ev.ev[4])); st.ms_total = ms;
#undef TW_HIP_C
    st.records = total; st.episodes = E; st.padded_bytes = cur;
    st.forward_evals = host_tot[1] * (uint64_t)(ma.pol.n_perms > 0 ? ma.pol.n_perms : 1);
    st.speculative_evals = host_tot[2] * (uint64_t)(ma.pol.n_perms > 0 ? ma.pol.n_perms : 1);
    st.reused_evals = host_tot[3] * (uint64_t)(ma.pol.n_perms > 0 ? ma.pol.n_perms : 1);
    c->stats = st;
    *out = c;
    return TW_OK;
}

// ====================================================================================== solve / evaluate
namespace {

// runs the attempts and reduces best-of-N per episode on the host in the reference's order
int run_solve(const PuzzleConsts &envc, const tw_policy *policy, const tw_solve_params *prm, uint64_t n_episodes,
              uint64_t episode_offset, bool from_state, const tw_puzzle *start, int max_steps, bool want_actions,
              std::vector<float> &best_s, std::vector<float> &best_r, std::vector<uint8_t> &best_actions)
{
    if (prm->num_searches == 0) { best_s.assign(n_episodes, 0.0f); best_r.assign(n_episodes, -__builtin_inff()); return TW_OK; }
    if (prm->precision != TW_PREC_F32_EXACT) { set_error("solve: precision %u not implemented", prm->precision); return TW_ERR_UNSUPPORTED; }
    SolveArgs sa{};
    sa.env = envc; sa.pol = policy->dev;
    if (sa.pol.obs_size != envc.n_cells * envc.n_cells) {
        set_error("index out of bounds: policy obs_size %d != Puzzle obs ids %d", sa.pol.obs_size, envc.n_cells * envc.n_cells);
        return TW_ERR_INVALID;
    }
    if (sa.pol.n_actions != 4) { set_error("Puzzle has 4 actions, policy has %d", sa.pol.n_actions); return TW_ERR_INVALID; }
    const uint64_t N = prm->num_searches, A = n_episodes * N;
    sa.num_attempts = A; sa.episode_offset = episode_offset; sa.seed = prm->seed;
    sa.num_searches = prm->num_searches; sa.deterministic = prm->deterministic ? 1u : 0u;
    sa.from_state = from_state ? 1u : 0u;
    std::vector<uint8_t> start_cells;
    if (from_state) {
        // the reference's set_state takes any vector (puzzle.rs:107-117); the device board is n_cells nibbles holding a
        // permutation of 0..n_cells-1 whose blank is where zero_location says -- anything else is refused here
        if (start->state.size() != (size_t)envc.n_cells) { set_error("solve: state has %zu entries, the board %d cells", start->state.size(), envc.n_cells); return TW_ERR_INVALID; }
        uint64_t b = 0, seen = 0;
        for (size_t i = 0; i < start->state.size(); ++i) {
            const int64_t v = start->state[i];
            if (v < 0 || v >= envc.n_cells || ((seen >> v) & 1ull)) { set_error("solve: state is not a permutation of 0..%d (entry %zu = %lld)", envc.n_cells - 1, i, (long long)v); return TW_ERR_INVALID; }
            seen |= 1ull << v;
            if (i < 16) b |= (uint64_t)v << (4 * i);                          // (boards above 16 cells: one byte per cell, start_cells below)
            start_cells.push_back((uint8_t)v);
        }
        const int64_t zi = start->zy * start->width + start->zx;
        if (zi < 0 || zi >= envc.n_cells || start->state[(size_t)zi] != 0) { set_error("solve: zero_location (%lld, %lld) does not hold the blank", (long long)start->zx, (long long)start->zy); return TW_ERR_INVALID; }
        sa.start_board = b; sa.start_zx = (int)start->zx; sa.start_zy = (int)start->zy; sa.start_depth = (int)start->depth;
    }
    sa.t_pad = max_steps > 0 ? max_steps : 1;

    hipStream_t s = current_stream();
    size_t cur = 0;
    auto seg = [&](size_t bytes) { size_t o = cur; cur = align_up(cur + bytes, 256); return o; };
    const size_t o_s = seg(A * 4), o_r = seg(A * 4), o_n = seg(A * 4), o_a = seg(want_actions ? A * (size_t)sa.t_pad : 0),
                 o_cells = seg(from_state && envc.n_cells > 16 ? 64 : 0);
    // MCTS-guided inference (solve.rs:41-47): per-attempt node arenas for the search trees
    const bool mcts = prm->num_mcts_searches != 0;
    const uint64_t node_cap = 5ull + 4ull * prm->num_mcts_searches * (prm->max_expand_depth ? prm->max_expand_depth : 1u);
    const bool big_mcts = mcts && sa.env.n_cells > 16;                   // boards of 17 .. 64 cells: tw_mcts_big.hip (32-byte nodes without a board)
    const size_t node_bytes = big_mcts ? mcts_big_node_bytes() : mcts_node_bytes();
    if (mcts && (node_cap > 0xffffffffull || A * node_cap * node_bytes > (200ull << 30))) {
        set_error("solve: MCTS arenas of %llu attempts x %llu nodes do not fit", (unsigned long long)A, (unsigned long long)node_cap);
        return TW_ERR_UNSUPPORTED;
    }
    // few deep searches on the MFMA policy shapes: the walker kernel (tw_mcts_deep.hip) in its solve mode -- one walker per attempt, persistent
    MctsArgs ma{};
    bool deep = false; uint64_t walkers = 0; uint32_t tbl_entries = 0;
    if (mcts) {
        ma.env = envc; ma.pol = sa.pol; ma.num_episodes = A; ma.episode_offset = episode_offset; ma.seed = prm->seed;
        ma.num_searches = prm->num_mcts_searches; ma.max_expand_depth = prm->max_expand_depth; ma.C = prm->C;
        ma.node_cap = (uint32_t)node_cap;
        ma.solve.on = 1; ma.solve.deterministic = sa.deterministic; ma.solve.num_searches = sa.num_searches;
        ma.solve.from_state = sa.from_state; ma.solve.start_board = sa.start_board; ma.solve.start_zx = sa.start_zx;
        ma.solve.start_zy = sa.start_zy; ma.solve.start_depth = sa.start_depth;
        ma.solve.act_pad = sa.t_pad;
        ma.reuse_mode = (uint32_t)launch_options().az_reuse;
        deep = !big_mcts && mcts_deep_applies(ma);
        if (deep) { walkers = mcts_deep_walkers(A, 0, ma.num_searches, true); tbl_entries = mcts_deep_table_entries(ma.num_searches, ma.max_expand_depth); }
    }
    const size_t o_cnt = seg(mcts ? 136 : 0),
                 o_arena = seg(!mcts ? 0 : deep ? (size_t)walkers * mcts_deep_arena_bytes(node_cap) : (size_t)(A * node_cap) * node_bytes),
                 o_tbl = seg(deep ? (size_t)walkers * tbl_entries * 32 : 0), o_init = seg(deep && !from_state ? n_episodes * 8 : 0), o_queue = seg(deep ? 4 : 0),
                 o_sv = seg(deep ? sizeof(MctsSolve) : 0);
    uint8_t *buf = nullptr;
    TW_HIP(hipMalloc((void **)&buf, cur ? cur : 256));
    sa.success = reinterpret_cast<float *>(buf + o_s); sa.total = reinterpret_cast<float *>(buf + o_r);
    sa.n_steps = reinterpret_cast<uint32_t *>(buf + o_n); sa.actions = want_actions ? buf + o_a : nullptr;
    if (from_state && envc.n_cells > 16) {
        hipError_t ce = hipMemcpyAsync(buf + o_cells, start_cells.data(), start_cells.size(), hipMemcpyHostToDevice, s);
        if (ce == hipSuccess) ce = hipStreamSynchronize(s);                 // (start_cells is a local)
        if (ce != hipSuccess) { (void)hipFree(buf); return hip_fail(ce, "hipMemcpyAsync(start state)", __FILE__, __LINE__); }
        sa.start_cells = buf + o_cells;
    }
    int rc;
    if (mcts) {
        ma.arena = reinterpret_cast<MctsNode *>(buf + o_arena);
        ma.eval_count = reinterpret_cast<unsigned long long *>(buf + o_cnt);
        ma.solve.start_cells = sa.start_cells;
        ma.solve.success = sa.success; ma.solve.total = sa.total; ma.solve.n_steps = sa.n_steps; ma.solve.actions = sa.actions;
        hipError_t me = hipMemsetAsync(buf + o_cnt, 0, 136, s);
        if (me == hipSuccess && deep) {             // walker arenas need no clearing; their output tables start empty; the queue starts behind the first walkers
            ma.tbl = buf + o_tbl; ma.tbl_entries = tbl_entries;
            me = hipMemsetAsync(buf + o_tbl, 0, (size_t)walkers * tbl_entries * 32, s);
            ma.queue = reinterpret_cast<unsigned int *>(buf + o_queue);
            const unsigned int first = (unsigned int)(walkers < A ? walkers : A);
            if (me == hipSuccess) me = hipMemcpyAsync(buf + o_queue, &first, 4, hipMemcpyHostToDevice, s);
            if (me == hipSuccess) me = hipMemcpyAsync(buf + o_sv, &ma.solve, sizeof(MctsSolve), hipMemcpyHostToDevice, s);      // (see MctsArgs::solve_dev)
            ma.solve_dev = reinterpret_cast<const MctsSolve *>(buf + o_sv);
            if (me == hipSuccess) me = hipStreamSynchronize(s);                               // (`first` is a local)
        }
        if (me != hipSuccess) { (void)hipFree(buf); return hip_fail(me, "solve: MCTS workspace", __FILE__, __LINE__); }
        rc = TW_OK;
        if (deep && !from_state) {
            ma.init_boards = reinterpret_cast<const uint64_t *>(buf + o_init);
            rc = launch_init_boards(envc, prm->seed, episode_offset, n_episodes, reinterpret_cast<uint64_t *>(buf + o_init), s);
        }
        if (rc == TW_OK)
            rc = big_mcts ? launch_mcts_big(ma, nullptr, s, nullptr, nullptr) : deep ? launch_mcts_deep(ma, s, nullptr, nullptr) : launch_mcts_f32(ma, s, nullptr, nullptr);
    } else rc = sa.env.n_cells > 16 ? launch_solve_big(sa, s) : launch_solve_f32(sa, s);
    std::vector<float> hs(A), hr(A); std::vector<uint32_t> hn(A);
    hipError_t e = hipSuccess;
    if (rc == TW_OK) {
        e = hipMemcpyAsync(hs.data(), sa.success, A * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(hr.data(), sa.total, A * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(hn.data(), sa.n_steps, A * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    if (rc != TW_OK || e != hipSuccess) { (void)hipFree(buf); return rc != TW_OK ? rc : hip_fail(e, "solve readback", __FILE__, __LINE__); }
    best_s.assign(n_episodes, 0.0f); best_r.assign(n_episodes, -__builtin_inff());
    std::vector<uint64_t> best_att(n_episodes, (uint64_t)-1);
    for (uint64_t ep = 0; ep < n_episodes; ++ep)
        for (uint64_t a = 0; a < N; ++a) {        // solve.rs:84-98: `if next_val.0 > best.0` on (success, total) tuples
            const uint64_t i = ep * N + a;
            if (hs[i] > best_s[ep] || (hs[i] == best_s[ep] && hr[i] > best_r[ep])) { best_s[ep] = hs[i]; best_r[ep] = hr[i]; best_att[ep] = i; }
        }
    if (want_actions && n_episodes == 1 && best_att[0] != (uint64_t)-1) {
        const uint64_t i = best_att[0];
        best_actions.resize(hn[i]);
        if (hn[i]) e = hipMemcpy(best_actions.data(), sa.actions + i * (uint64_t)sa.t_pad, hn[i], hipMemcpyDeviceToHost);
    }
    (void)hipFree(buf);
    if (e != hipSuccess) return hip_fail(e, "solve actions readback", __FILE__, __LINE__);
    return TW_OK;
}

}  // namespace

extern "C" int tw_evaluate(const tw_puzzle_desc *env, const tw_policy *policy, const tw_solve_params *prm,
                           uint64_t num_episodes, uint64_t episode_offset, float *success_rate, float *mean_reward)
{
    if (!env || !policy || !prm || !success_rate || !mean_reward) { set_error("tw_evaluate: null argument"); return TW_ERR_INVALID; }
    int rc = require_device(); if (rc) return rc;
    // boards of 17 .. 64 cells: on the device (solve_big_kernel; MCTS-guided: mcts_big_kernel); whatever they do not take: the any-environment path (host env)
    const uint64_t cells = (uint64_t)env->width * env->height;
    const bool big_dev = cells > 16 && cells <= 64 && policy->dev.generic && prm->precision == TW_PREC_F32_EXACT &&
                         (uint64_t)env->depth_slope * env->difficulty <= 1022 && env->max_depth != 0 && !launch_options().force_geom;
    if (cells > 16 && !big_dev) {
        uint64_t depth0 = 0;
        rc = big_board_checks(env, prm->precision, &depth0); if (rc) return rc;
        tw_puzzle *proto = tw_puzzle_create(env->width, env->height, env->difficulty, env->depth_slope, env->max_depth);
        if (!proto) return TW_ERR_INVALID;
        tw_env_vtable vt = puzzle_env_table(proto);
        rc = tw_evaluate_env(&vt, policy, prm, num_episodes, episode_offset, (uint32_t)(depth0 + 1), success_rate, mean_reward);
        tw_puzzle_destroy(proto);
        return rc;
    }
    PuzzleConsts envc; rc = make_env_consts(env, &envc, 64); if (rc) return rc;
    if (num_episodes == 0) { *success_rate = __builtin_nanf(""); *mean_reward = __builtin_nanf(""); return TW_OK; }   // 0/0 (evaluate.rs:52)
    std::vector<float> bs, br; std::vector<uint8_t> acts;
    rc = run_solve(envc, policy, prm, num_episodes, episode_offset, false, nullptr, envc.depth0 + 1, false, bs, br, acts);
    if (rc) return rc;
    float successes = 0.0f, rewards = 0.0f;       // serial accumulation, episode order (evaluate.rs:36-52)
    for (uint64_t e = 0; e < num_episodes; ++e) { successes = successes + bs[e]; rewards = rewards + br[e]; }
    *success_rate = successes / (float)num_episodes;
    *mean_reward = rewards / (float)num_episodes;
    return TW_OK;
}

extern "C" int tw_solve(const tw_puzzle *env, const tw_policy *policy, const tw_solve_params *prm, float *success,
                        float *reward, uint8_t *actions_out, uint32_t actions_cap, uint32_t *n_actions)
{
    if (!env || !policy || !prm || !success || !reward) { set_error("tw_solve: null argument"); return TW_ERR_INVALID; }
    int rc = require_device(); if (rc) return rc;
    // boards of 17 .. 64 cells: on the device too (solve_big_kernel / mcts_big_kernel); whatever they do not take: the any-environment path (host env)
    bool big_dev = env->state.size() > 16 && env->state.size() <= 64 && policy->dev.generic && prm->precision == TW_PREC_F32_EXACT &&
                   env->depth >= 0 && env->depth <= 1022 && env->max_depth != 0 && !launch_options().force_geom &&
                   env->width * env->height == (int64_t)env->state.size();
    if (big_dev) {                                                         // (the device board is a permutation with the blank where zero_location says; set_state
        uint64_t seen = 0;                                                 //  takes any vector, puzzle.rs:107-117: anything else steps on the host as before)
        for (int64_t v : env->state) { if (v < 0 || v >= (int64_t)env->state.size() || ((seen >> v) & 1ull)) { big_dev = false; break; } seen |= 1ull << v; }
        const int64_t zi = env->zy * env->width + env->zx;
        if (big_dev && (zi < 0 || zi >= (int64_t)env->state.size() || env->state[(size_t)zi] != 0)) big_dev = false;
    }
    if (env->state.size() > 16 && !big_dev) {
        tw_puzzle_desc bd; tw_puzzle_get_desc(env, &bd);
        uint64_t depth0 = 0;
        rc = big_board_checks(&bd, prm->precision, &depth0); if (rc) return rc;
        if (env->depth > 0xfffffff0ll) { set_error("tw_solve: depth %lld too large", (long long)env->depth); return TW_ERR_UNSUPPORTED; }
        tw_puzzle *proto = tw_puzzle_clone(env);                           // (the caller's env is not modified)
        tw_env_vtable vt = puzzle_env_table(proto);
        rc = tw_solve_env(&vt, policy, prm, (uint32_t)env->depth + 1u, success, reward, actions_out, actions_cap, n_actions);
        tw_puzzle_destroy(proto);
        return rc;
    }
    tw_puzzle_desc d; tw_puzzle_get_desc(env, &d);
    PuzzleConsts envc; rc = make_env_consts(&d, &envc, 64); if (rc) return rc;
    if (env->depth > 4096) { set_error("tw_solve: depth %lld too large", (long long)env->depth); return TW_ERR_UNSUPPORTED; }
    std::vector<float> bs, br; std::vector<uint8_t> acts;
    rc = run_solve(envc, policy, prm, 1, 0, true, env, (int)env->depth + 1, true, bs, br, acts);
    if (rc) return rc;
    *success = bs[0]; *reward = br[0];
    if (n_actions) *n_actions = (uint32_t)acts.size();
    if (actions_out) {
        if (acts.size() > actions_cap) { set_error("tw_solve: %zu actions, caller's buffer holds %u", acts.size(), actions_cap); return TW_ERR_INVALID; }
        if (!acts.empty()) memcpy(actions_out, acts.data(), acts.size());
    }
    return TW_OK;
}
