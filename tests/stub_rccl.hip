// stub_rccl.hip -- TEST INFRASTRUCTURE, not a product path.
//
// A stand-in for the eleven nccl* symbols twisterl_amd/csrc/tw_comm.hip resolves with dlopen, so that the library's own
// multi-rank exchange (tw_comm_* / tw_gather_*: count all-gather, grouped send/recv at final offsets, policy broadcast, the
// abort / bounded-wait path) can run with 2 and 3 ranks as separate processes on ONE GPU.  RCCL itself refuses two ranks on one
// device, and this pool hands a test one GPU; the real RCCL runs the same calls at world 1 (tests/test_gpu_parity.py,
// tests/test_gpu_full_size.py) and on the driver's 8-GPU node.  Selected with TW_RCCL_LIBRARY=<path of the built .so>
// (tw_comm.hip: rccl_load).
//
// What it keeps of the real thing, because the library relies on it:
//   - every operation is ASYNCHRONOUS and ordered on the hipStream it was given; the caller returns at once and
//     hipStreamQuery / hipStreamSynchronize see the transfer:
//       send = device -> pinned-host copy on the stream + an event; the communicator's worker thread waits for the event and
//              publishes the bytes (a buffered send);
//       recv = a one-lane kernel on the stream that polls a flag in pinned host memory, followed by the pinned-host -> device
//              copy; the worker thread waits for the message, fills the staging buffer and raises the flag.
//     (No blocking host callbacks: a hipLaunchHostFunc body that waits can run inside the caller's own hipStreamQuery.)
//   - operations between ncclGroupStart and ncclGroupEnd are posted at ncclGroupEnd, in call order.
//   - messages between one (sender, receiver) pair match in posting order and must agree in size (a mismatch marks the
//     communicator failed: its waiting kernels are released and later calls return ncclSystemError).
//   - a peer that died is NOT noticed: the receive never completes (the library's bounded wait is what is under test);
//     ncclCommAbort releases this rank's waiting kernels.  The polling kernel gives up by itself after ~2 minutes.
// Transport: one file per message under /dev/shm, "<token>.<kind>.<src>.<dst>.<seq>", written under a temporary name and
// renamed, consumed (unlinked) by the receiver.  The token is the ncclUniqueId.
//
// Test knob (environment of the rank's process): TWSTUB_DROP_SENDS=1 -- point-to-point sends are accepted and never published
// (a rank that dies between the count exchange and its transfers).
//
// Build: hipcc -O2 --offload-arch=gfx950 -fPIC -shared stub_rccl.hip -o libstub_rccl.so -lpthread
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <time.h>
#include <unistd.h>

namespace {

constexpr int STUB_MAX_RANKS = 16;
constexpr unsigned long long WAIT_ITERS = 30ull * 1000 * 1000;        // x (s_sleep 127 + one host read) ~ 2-3 minutes, then give up

// the stream stops here until the worker thread raises *flag (1: data staged, 2: aborted / failed)
__global__ void stub_wait_flag(const uint32_t *flag)
{
    for (unsigned long long i = 0; i < WAIT_ITERS; ++i) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return;
        __builtin_amdgcn_s_sleep(127);
    }
}

struct Task {
    enum Kind { SEND, RECV, QUIT } kind = SEND;
    hipEvent_t ev = nullptr;                       // SEND: the device -> host copy
    std::vector<std::string> paths;                // one message each
    std::vector<char *> hosts;
    size_t bytes = 0;
    uint32_t *flag = nullptr;                      // RECV
};

}  // namespace

struct ncclComm {
    std::string token;
    int rank = 0, world = 1, device = 0;
    std::atomic<int> abort{0}, failed{0};
    uint64_t p2p_send_seq[STUB_MAX_RANKS] = {}, p2p_recv_seq[STUB_MAX_RANKS] = {};
    uint64_t coll_seq = 0;
    bool drop_sends = false;
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Task> queue;
    // Nothing is released before the communicator goes: hipHostFree may wait for the whole device, i.e. for a stream that stands
    // in stub_wait_flag behind a peer that died.
    std::vector<void *> host_allocs;
    std::vector<hipEvent_t> events;
};

namespace {

thread_local int g_group_depth = 0;
struct GroupOp { int kind; const void *src; void *dst; size_t bytes; int peer; ncclComm *c; hipStream_t s; };
thread_local std::vector<GroupOp> g_group;

size_t type_size(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 1;
    }
}

std::string msg_path(const ncclComm *c, const char *kind, int src, int dst, uint64_t seq)
{
    char b[256];
    snprintf(b, sizeof(b), "/dev/shm/%s.%s.%d.%d.%llu", c->token.c_str(), kind, src, dst, (unsigned long long)seq);
    return b;
}

bool publish(ncclComm *c, const std::string &path, const void *data, size_t bytes)
{
    const std::string tmp = path + ".tmp";
    int fd = open(tmp.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0600);
    if (fd < 0) { c->failed = 1; return false; }
    const char *p = (const char *)data; size_t left = bytes;
    while (left) {
        ssize_t w = write(fd, p, left);
        if (w < 0) { if (errno == EINTR) continue; close(fd); c->failed = 1; return false; }
        p += w; left -= (size_t)w;
    }
    close(fd);
    if (rename(tmp.c_str(), path.c_str()) != 0) { c->failed = 1; return false; }
    return true;
}

// waits for the message, checks its size, reads and consumes it; gives up on abort
bool consume(ncclComm *c, const std::string &path, void *data, size_t bytes)
{
    int fd;
    while ((fd = open(path.c_str(), O_RDONLY)) < 0) {
        if (c->abort || c->failed) return false;
        struct timespec ts = {0, 100000};        // 100 us
        nanosleep(&ts, nullptr);
    }
    struct stat st;
    if (fstat(fd, &st) != 0 || (size_t)st.st_size != bytes) {
        fprintf(stderr, "[stub_rccl] rank %d: message %s holds %lld bytes, the receive expects %zu\n", c->rank, path.c_str(), (long long)st.st_size, bytes);
        close(fd); unlink(path.c_str()); c->failed = 1; return false;
    }
    char *p = (char *)data; size_t left = bytes;
    while (left) {
        ssize_t r = read(fd, p, left);
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) { close(fd); c->failed = 1; return false; }
        p += r; left -= (size_t)r;
    }
    close(fd);
    unlink(path.c_str());
    return true;
}

void worker_main(ncclComm *c)
{
    (void)hipSetDevice(c->device);
    for (;;) {
        Task t;
        {
            std::unique_lock<std::mutex> lk(c->mu);
            c->cv.wait(lk, [&] { return !c->queue.empty(); });
            t = std::move(c->queue.front());
            c->queue.pop_front();
        }
        if (t.kind == Task::QUIT) return;
        if (t.kind == Task::SEND) {
            if (hipEventSynchronize(t.ev) != hipSuccess) { c->failed = 1; continue; }
            for (size_t i = 0; i < t.paths.size(); ++i)
                if (!c->abort && !c->failed) publish(c, t.paths[i], t.hosts[i], t.bytes);
        } else {
            bool ok = true;
            for (size_t i = 0; i < t.paths.size() && ok; ++i) ok = consume(c, t.paths[i], t.hosts[i], t.bytes);
            __atomic_store_n(t.flag, ok ? 1u : 2u, __ATOMIC_RELEASE);        // the stream goes on (after a failure: with garbage, the call sites report it)
        }
    }
}

void post(ncclComm *c, Task &&t)
{
    { std::lock_guard<std::mutex> lk(c->mu); c->queue.push_back(std::move(t)); }
    c->cv.notify_one();
}

char *host_alloc(ncclComm *c, size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 4, hipHostMallocDefault) != hipSuccess) return nullptr;
    c->host_allocs.push_back(p);
    return (char *)p;
}

uint32_t *flag_alloc(ncclComm *c)
{
    uint32_t *f = (uint32_t *)host_alloc(c, 64);
    if (f) __atomic_store_n(f, 0u, __ATOMIC_RELEASE);
    return f;
}

#define STUB_HIP(call) do { if ((call) != hipSuccess) { (void)hipGetLastError(); return ncclUnhandledCudaError; } } while (0)

ncclResult_t do_send(ncclComm *c, const void *buf, size_t bytes, int peer, hipStream_t s)
{
    if (c->abort || c->failed) return ncclSystemError;
    if (peer < 0 || peer >= c->world || peer == c->rank) return ncclInvalidArgument;
    const uint64_t seq = c->p2p_send_seq[peer]++;
    if (c->drop_sends) return ncclSuccess;
    char *host = host_alloc(c, bytes);
    if (!host) return ncclUnhandledCudaError;
    Task t; t.kind = Task::SEND; t.bytes = bytes;
    t.paths.push_back(msg_path(c, "p2p", c->rank, peer, seq)); t.hosts.push_back(host);
    if (bytes) STUB_HIP(hipMemcpyAsync(host, buf, bytes, hipMemcpyDeviceToHost, s));
    STUB_HIP(hipEventCreateWithFlags(&t.ev, hipEventDisableTiming));
    c->events.push_back(t.ev);
    STUB_HIP(hipEventRecord(t.ev, s));
    post(c, std::move(t));
    return ncclSuccess;
}

ncclResult_t do_recv(ncclComm *c, void *buf, size_t bytes, int peer, hipStream_t s)
{
    if (c->abort || c->failed) return ncclSystemError;
    if (peer < 0 || peer >= c->world || peer == c->rank) return ncclInvalidArgument;
    const uint64_t seq = c->p2p_recv_seq[peer]++;
    char *host = host_alloc(c, bytes);
    uint32_t *flag = flag_alloc(c);
    if (!host || !flag) return ncclUnhandledCudaError;
    Task t; t.kind = Task::RECV; t.bytes = bytes; t.flag = flag;
    t.paths.push_back(msg_path(c, "p2p", peer, c->rank, seq)); t.hosts.push_back(host);
    hipLaunchKernelGGL(stub_wait_flag, dim3(1), dim3(1), 0, s, flag);
    STUB_HIP(hipGetLastError());
    if (bytes) STUB_HIP(hipMemcpyAsync(buf, host, bytes, hipMemcpyHostToDevice, s));
    post(c, std::move(t));
    return ncclSuccess;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ the nccl* symbols
extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    struct timeval tv; gettimeofday(&tv, nullptr);
    unsigned r = 0;
    int fd = open("/dev/urandom", O_RDONLY);
    if (fd >= 0) { if (read(fd, &r, sizeof(r)) != (ssize_t)sizeof(r)) r = (unsigned)tv.tv_usec; close(fd); }
    snprintf(id->internal, 48, "twstub_%d_%lld_%08x", (int)getpid(), (long long)tv.tv_sec, r);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || nranks > STUB_MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    if (strncmp(id.internal, "twstub_", 7) != 0) return ncclInvalidArgument;
    id.internal[47] = 0;
    ncclComm *c = new ncclComm();
    c->token = id.internal;
    c->rank = rank; c->world = nranks;
    if (hipGetDevice(&c->device) != hipSuccess) { delete c; return ncclUnhandledCudaError; }
    const char *d = getenv("TWSTUB_DROP_SENDS");
    c->drop_sends = d && *d && strcmp(d, "0") != 0;
    // collective: every rank says hello to every other one (bounded wait: a missing rank fails the init)
    char one = 1;
    for (int r = 0; r < nranks; ++r) if (r != rank && !publish(c, msg_path(c, "init", rank, r, 0), &one, 1)) { delete c; return ncclSystemError; }
    for (int r = 0; r < nranks; ++r) if (r != rank) {
        const std::string path = msg_path(c, "init", r, rank, 0);
        int fd, tries = 0;
        while ((fd = open(path.c_str(), O_RDONLY)) < 0) {
            if (++tries > 600000) { delete c; return ncclSystemError; }        // 60 s
            struct timespec ts = {0, 100000};
            nanosleep(&ts, nullptr);
        }
        close(fd); unlink(path.c_str());
    }
    c->worker = std::thread(worker_main, c);
    *comm = c;
    return ncclSuccess;
}

static void comm_end(ncclComm *c, bool aborting)
{
    if (aborting) c->abort = 1;                    // the worker gives up its wait and raises every flag still queued: the kernels leave
    { Task q; q.kind = Task::QUIT; post(c, std::move(q)); }
    if (c->worker.joinable()) c->worker.join();
    for (hipEvent_t e : c->events) { (void)hipEventSynchronize(e); (void)hipEventDestroy(e); }
    (void)hipDeviceSynchronize();                  // (every waiting kernel of this communicator has its flag by now)
    for (void *p : c->host_allocs) (void)hipHostFree(p);
    (void)hipGetLastError();
    delete c;
}

ncclResult_t ncclCommAbort(ncclComm_t c) { if (!c) return ncclInvalidArgument; comm_end(c, true); return ncclSuccess; }
ncclResult_t ncclCommDestroy(ncclComm_t c) { if (!c) return ncclInvalidArgument; comm_end(c, false); return ncclSuccess; }

const char *ncclGetErrorString(ncclResult_t e)
{
    switch (e) {
    case ncclSuccess: return "no error (stub transport)";
    case ncclUnhandledCudaError: return "unhandled HIP error (stub transport)";
    case ncclSystemError: return "unhandled system error (stub transport)";
    case ncclInvalidArgument: return "invalid argument (stub transport)";
    default: return "error (stub transport)";
    }
}

static ncclResult_t enqueue(int kind, const void *src, void *dst, size_t bytes, int peer, ncclComm_t c, hipStream_t s)
{
    if (g_group_depth == 0) return kind == 0 ? do_send(c, src, bytes, peer, s) : do_recv(c, dst, bytes, peer, s);
    g_group.push_back(GroupOp{kind, src, dst, bytes, peer, c, s});
    return ncclSuccess;
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s)
{
    if (!c) return ncclInvalidArgument;
    return enqueue(0, buf, nullptr, count * type_size(t), peer, c, s);
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s)
{
    if (!c) return ncclInvalidArgument;
    return enqueue(1, nullptr, buf, count * type_size(t), peer, c, s);
}

ncclResult_t ncclGroupStart(void) { g_group_depth++; return ncclSuccess; }

ncclResult_t ncclGroupEnd(void)
{
    if (g_group_depth <= 0) return ncclInvalidUsage;
    if (--g_group_depth > 0) return ncclSuccess;
    ncclResult_t rc = ncclSuccess;
    for (size_t i = 0; i < g_group.size() && rc == ncclSuccess; ++i) {
        const GroupOp &o = g_group[i];
        rc = o.kind == 0 ? do_send(o.c, o.src, o.bytes, o.peer, o.s) : do_recv(o.c, o.dst, o.bytes, o.peer, o.s);
    }
    g_group.clear();
    return rc;
}

ncclResult_t ncclAllGather(const void *sendbuf, void *recvbuf, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t s)
{
    if (!c) return ncclInvalidArgument;
    if (c->abort || c->failed) return ncclSystemError;
    const size_t bytes = count * type_size(t);
    const uint64_t seq = c->coll_seq++;
    const int W = c->world;
    char *host = host_alloc(c, (size_t)W * bytes);
    uint32_t *flag = flag_alloc(c);
    if (!host || !flag) return ncclUnhandledCudaError;
    Task so, ro;
    so.kind = Task::SEND; ro.kind = Task::RECV; so.bytes = ro.bytes = bytes; ro.flag = flag;
    for (int r = 0; r < W; ++r) if (r != c->rank) {
        so.paths.push_back(msg_path(c, "coll", c->rank, r, seq)); so.hosts.push_back(host + (size_t)c->rank * bytes);
        ro.paths.push_back(msg_path(c, "coll", r, c->rank, seq)); ro.hosts.push_back(host + (size_t)r * bytes);
    }
    STUB_HIP(hipMemcpyAsync(host + (size_t)c->rank * bytes, sendbuf, bytes, hipMemcpyDeviceToHost, s));
    STUB_HIP(hipEventCreateWithFlags(&so.ev, hipEventDisableTiming));
    c->events.push_back(so.ev);
    STUB_HIP(hipEventRecord(so.ev, s));
    hipLaunchKernelGGL(stub_wait_flag, dim3(1), dim3(1), 0, s, flag);
    STUB_HIP(hipGetLastError());
    STUB_HIP(hipMemcpyAsync(recvbuf, host, (size_t)W * bytes, hipMemcpyHostToDevice, s));
    post(c, std::move(so));
    post(c, std::move(ro));
    return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void *sendbuf, void *recvbuf, size_t count, ncclDataType_t t, int root, ncclComm_t c, hipStream_t s)
{
    if (!c || root < 0 || root >= c->world) return ncclInvalidArgument;
    if (c->abort || c->failed) return ncclSystemError;
    const size_t bytes = count * type_size(t);
    const uint64_t seq = c->coll_seq++;
    char *host = host_alloc(c, bytes);
    if (!host) return ncclUnhandledCudaError;
    Task o; o.bytes = bytes;
    if (c->rank == root) {
        o.kind = Task::SEND;
        for (int r = 0; r < c->world; ++r) if (r != root) { o.paths.push_back(msg_path(c, "coll", root, r, seq)); o.hosts.push_back(host); }
        STUB_HIP(hipMemcpyAsync(host, sendbuf, bytes, hipMemcpyDeviceToHost, s));
        STUB_HIP(hipEventCreateWithFlags(&o.ev, hipEventDisableTiming));
        c->events.push_back(o.ev);
        STUB_HIP(hipEventRecord(o.ev, s));
        if (recvbuf != sendbuf) STUB_HIP(hipMemcpyAsync(recvbuf, sendbuf, bytes, hipMemcpyDeviceToDevice, s));
    } else {
        o.kind = Task::RECV;
        o.flag = flag_alloc(c);
        if (!o.flag) return ncclUnhandledCudaError;
        o.paths.push_back(msg_path(c, "coll", root, c->rank, seq)); o.hosts.push_back(host);
        hipLaunchKernelGGL(stub_wait_flag, dim3(1), dim3(1), 0, s, o.flag);
        STUB_HIP(hipGetLastError());
        STUB_HIP(hipMemcpyAsync(recvbuf, host, bytes, hipMemcpyHostToDevice, s));
    }
    post(c, std::move(o));
    return ncclSuccess;
}

}  // extern "C"
