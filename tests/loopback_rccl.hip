// loopback_rccl.hip -- TEST INFRASTRUCTURE, not product.  An in-process stand-in for the nine librccl entry points the library's native
// transport binds (csrc/sph_mi355x.hip: RcclApi), so that the code path a multi-GPU node runs -- ncclSend / ncclRecv / ncclAllReduce enqueued by
// the library itself on its main, halo and reduction streams, no host waits, edge / interior split, speculative divergence correction -- can be
// driven on ONE GPU: the "ranks" are handles stepped by threads of one process (tests/loopback_worker.py), the transfers are device-to-device
// copies ordered by HIP events exactly as RCCL orders its kernels:
//   * a send is complete on the sender's stream when the receiver's copy is (the send buffer may be reused after it),
//   * a receive is ordered on the receiver's stream behind everything the sender had enqueued before its send,
//   * an all-reduce is ordered on every rank's stream behind every rank's contribution, and sums in rank order on every rank alike.
// The host side differs from RCCL in one respect: a rank's ncclGroupEnd / ncclAllReduce returns only after its peers have POSTED the matching
// calls (RCCL returns at once and lets the kernels meet on the device).  The library issues the same sequence on every rank, so that is a
// rendezvous, not a deadlock; every wait is bounded (kWaitSeconds) and ends in ncclSystemError.
// Record / replay (tools/loopback_rehearsal.sh): after loopback_record_begin(k, bytes) everything rank k RECEIVES -- the bytes of every ncclRecv, the result
// of every ncclAllReduce -- is also appended to a device-side log; a communicator opened on the id loopback_replay_id() hands out then plays rank k
// ALONE against that log (sends vanish, receives and reductions are copies out of the log): the same computation as in the full run, bit for bit,
// with nothing else on the GPU -- the device time of ONE rank of a sharded run, its link time set to zero, and a clean rocprofv3 profile of it.
// Loaded through the development override SPH_RCCL_LIB (needs SPH_DEV=1); never shipped, never linked.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tests/loopback_rccl.hip -o tests/_build/libloopback_rccl.so
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

constexpr int kWaitSeconds = 120;
constexpr size_t kStageBytes = 16u << 20;      // per rank and parity: the largest all-reduce the library issues is 4 x (rigid samples) doubles

struct SendDesc {
    const void *src = nullptr;
    size_t bytes = 0;
    hipEvent_t ready = nullptr, done = nullptr;
    bool consumed = false;
};

struct World {
    int n = 0, joined = 0, left = 0;
    std::mutex m;
    std::condition_variable cv;
    std::vector<std::vector<std::deque<std::shared_ptr<SendDesc>>>> box;     // box[src][dst]
    std::vector<long long> posted;                                            // all-reduces posted per rank
    void *stage[2] = {nullptr, nullptr};                                      // n slots of kStageBytes per parity
    std::vector<hipEvent_t> ready[2], done[2];
    std::atomic<long long> p2p_bytes{0}, p2p_msgs{0}, allreduces{0};
};

struct Comm {
    std::shared_ptr<World> w;
    int rank = 0;
    long long ar_seq = 0;
    bool replay = false;
    size_t cursor = 0;          // replay: next log entry
};

// what one rank received, in the order it asked for it
struct Recorder {
    int rank = -1;
    char *log = nullptr;
    size_t cap = 0, used = 0;
    std::vector<std::pair<size_t, size_t>> entries;      // (offset, bytes)
    std::mutex m;
    bool overflow = false;
};
Recorder g_rec;

// rank g_rec.rank only: append `bytes` at `src` (complete on `stream` at this point) to the log
ncclResult_t record(Comm *c, const void *src, size_t bytes, hipStream_t stream)
{
    if (c->rank != g_rec.rank || !g_rec.log || c->replay) return ncclSuccess;
    size_t off;
    {
        std::lock_guard<std::mutex> lk(g_rec.m);
        const size_t padded = (bytes + 255) & ~(size_t)255;
        if (g_rec.used + padded > g_rec.cap) { g_rec.overflow = true; return ncclSuccess; }
        off = g_rec.used; g_rec.used += padded;
        g_rec.entries.emplace_back(off, bytes);
    }
    if (bytes && hipMemcpyAsync(g_rec.log + off, src, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}
ncclResult_t replay_into(Comm *c, void *dst, size_t bytes, hipStream_t stream, const char *what)
{
    if (c->cursor >= g_rec.entries.size() || g_rec.entries[c->cursor].second != bytes) {
        fprintf(stderr, "loopback_rccl replay: %s of %zu bytes at entry %zu, the log holds %zu entries and %zu bytes there\n", what, bytes, c->cursor, g_rec.entries.size(),
                c->cursor < g_rec.entries.size() ? g_rec.entries[c->cursor].second : (size_t)0);
        return ncclInvalidArgument;
    }
    const size_t off = g_rec.entries[c->cursor++].first;
    if (bytes && hipMemcpyAsync(dst, g_rec.log + off, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

std::mutex g_reg_m;
std::map<std::string, std::shared_ptr<World>> g_worlds;
std::atomic<unsigned> g_next_id{1};

struct PendingOp { bool send; void *buf; size_t bytes; int peer; Comm *comm; hipStream_t stream; };
thread_local int t_depth = 0;
thread_local std::vector<PendingOp> t_ops;

size_t type_size(ncclDataType_t t)
{
    switch (t) {
    case ncclChar: case ncclUint8: return 1;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

template <class Pred>
bool wait_for(World &w, std::unique_lock<std::mutex> &lk, Pred p)
{
    return w.cv.wait_for(lk, std::chrono::seconds(kWaitSeconds), p);
}

#define LB_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "loopback_rccl: %s: %s\n", #x, hipGetErrorString(e_)); return ncclUnhandledCudaError; } } while (0)

void delay_on(hipStream_t stream);
constexpr int kBatchMax = 24;
struct CopyBatch { int n; char *dst[kBatchMax]; const char *src[kBatchMax]; size_t bytes[kBatchMax]; };
__global__ void k_copy_batch(CopyBatch b)
{
    const int j = blockIdx.y;
    char *d = b.dst[j]; const char *s = b.src[j]; const size_t n = b.bytes[j];
    const size_t stride = (size_t)gridDim.x * blockDim.x * 16, start = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(s)) & 15) == 0) {
        for (size_t o = start; o + 16 <= n; o += stride) *reinterpret_cast<uint4 *>(d + o) = *reinterpret_cast<const uint4 *>(s + o);
        if (blockIdx.x == 0 && threadIdx.x < (n & 15)) d[(n & ~(size_t)15) + threadIdx.x] = s[(n & ~(size_t)15) + threadIdx.x];
    } else {
        for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += (size_t)gridDim.x * blockDim.x) d[o] = s[o];
    }
}
ncclResult_t run_group(std::vector<PendingOp> &ops)
{
    if (!ops.empty() && ops[0].comm->replay) {
        // one launch for the whole group, as RCCL makes one kernel of a group of transfers (a copy per receive would charge the replay a launch
        // per message that a real group does not pay)
        CopyBatch b;
        b.n = 0;
        hipStream_t st = nullptr;
        for (PendingOp &op : ops) {
            if (op.send) continue;
            Comm *c = op.comm;
            if (c->cursor >= g_rec.entries.size() || g_rec.entries[c->cursor].second != op.bytes) {
                fprintf(stderr, "loopback_rccl replay: receive of %zu bytes at entry %zu does not match the log\n", op.bytes, c->cursor);
                return ncclInvalidArgument;
            }
            if (!st) { st = op.stream; delay_on(st); }
            if (op.stream != st || b.n == kBatchMax) {          // (never in the library's groups; fall back to plain copies)
                ncclResult_t r = replay_into(c, op.buf, op.bytes, op.stream, "receive");
                if (r != ncclSuccess) return r;
                continue;
            }
            b.dst[b.n] = static_cast<char *>(op.buf); b.src[b.n] = g_rec.log + g_rec.entries[c->cursor++].first; b.bytes[b.n] = op.bytes; b.n += 1;
        }
        if (b.n) hipLaunchKernelGGL(k_copy_batch, dim3(64, b.n), dim3(256), 0, st, b);
        return ncclSuccess;
    }
    std::vector<std::shared_ptr<SendDesc>> sends;
    // every send is posted before anything waits
    for (PendingOp &op : ops) {
        if (!op.send) continue;
        World &w = *op.comm->w;
        auto d = std::make_shared<SendDesc>();
        d->src = op.buf; d->bytes = op.bytes;
        LB_HIP(hipEventCreateWithFlags(&d->ready, hipEventDisableTiming));
        LB_HIP(hipEventRecord(d->ready, op.stream));
        {
            std::lock_guard<std::mutex> lk(w.m);
            w.box[op.comm->rank][op.peer].push_back(d);
        }
        w.cv.notify_all();
        w.p2p_bytes += (long long)op.bytes; w.p2p_msgs += 1;
        sends.push_back(d);
    }
    for (PendingOp &op : ops) {
        if (op.send) continue;
        World &w = *op.comm->w;
        std::shared_ptr<SendDesc> d;
        {
            std::unique_lock<std::mutex> lk(w.m);
            auto &q = w.box[op.peer][op.comm->rank];
            if (!wait_for(w, lk, [&] { return !q.empty(); })) { fprintf(stderr, "loopback_rccl: rank %d waited %d s for a send of rank %d\n", op.comm->rank, kWaitSeconds, op.peer); return ncclSystemError; }
            d = q.front(); q.pop_front();
        }
        if (d->bytes != op.bytes) { fprintf(stderr, "loopback_rccl: rank %d receives %zu bytes from rank %d, which sends %zu\n", op.comm->rank, op.bytes, op.peer, d->bytes); return ncclInvalidArgument; }
        LB_HIP(hipStreamWaitEvent(op.stream, d->ready, 0));
        LB_HIP(hipMemcpyAsync(op.buf, d->src, op.bytes, hipMemcpyDeviceToDevice, op.stream));
        { ncclResult_t r = record(op.comm, op.buf, op.bytes, op.stream); if (r != ncclSuccess) return r; }
        hipEvent_t done;
        LB_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
        LB_HIP(hipEventRecord(done, op.stream));
        {
            std::lock_guard<std::mutex> lk(w.m);
            d->done = done; d->consumed = true;
        }
        w.cv.notify_all();
    }
    size_t k = 0;
    for (PendingOp &op : ops) {
        if (!op.send) continue;
        World &w = *op.comm->w;
        std::shared_ptr<SendDesc> d = sends[k++];
        {
            std::unique_lock<std::mutex> lk(w.m);
            if (!wait_for(w, lk, [&] { return d->consumed; })) { fprintf(stderr, "loopback_rccl: rank %d waited %d s for rank %d to receive\n", op.comm->rank, kWaitSeconds, op.peer); return ncclSystemError; }
        }
        LB_HIP(hipStreamWaitEvent(op.stream, d->done, 0));      // the send buffer is free once the receiver has copied it
        (void)hipEventDestroy(d->ready);
        (void)hipEventDestroy(d->done);
    }
    return ncclSuccess;
}

__global__ void loopback_marker_kernel(int *p) { if (p) *p = 1; }

// LOOPBACK_LATENCY_US=<n>: every group of point-to-point transfers and every all-reduce is preceded, on its stream, by a kernel that does nothing
// for n microseconds -- a stand-in for what a real link and a real collective take before the first byte arrives (one workgroup of one wave
// waiting on the 100 MHz wall clock: bounded by construction).  With it the replay of one rank prices the two slab protocols against a link that
// is not free (tools/loopback_replay.sh).
__global__ void loopback_delay_kernel(long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
int g_latency_us = -1;
void delay_on(hipStream_t stream)
{
    if (g_latency_us < 0) { const char *e = getenv("LOOPBACK_LATENCY_US"); g_latency_us = e ? atoi(e) : 0; if (g_latency_us > 1000) g_latency_us = 1000; }
    if (g_latency_us > 0) hipLaunchKernelGGL(loopback_delay_kernel, dim3(1), dim3(64), 0, stream, (long long)g_latency_us * 100);
}

template <class T>
__global__ void k_reduce(T *__restrict__ out, const char *__restrict__ stage, size_t slot_bytes, int ranks, size_t n, int is_max)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    T acc = reinterpret_cast<const T *>(stage)[i];
    for (int r = 1; r < ranks; ++r) {                    // rank order on every rank: all copies of the result are the same bits
        const T v = reinterpret_cast<const T *>(stage + (size_t)r * slot_bytes)[i];
        acc = is_max ? (v > acc ? v : acc) : acc + v;
    }
    out[i] = acc;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id->internal, 0, NCCL_UNIQUE_ID_BYTES);
    snprintf(id->internal, NCCL_UNIQUE_ID_BYTES, "loopback-%u", g_next_id.fetch_add(1));
    return ncclSuccess;
}

// ---- record / replay controls (called by tests/loopback_worker.py through ctypes) ----
int loopback_record_begin(int rank, size_t log_bytes)
{
    std::lock_guard<std::mutex> lk(g_rec.m);
    if (g_rec.log) (void)hipFree(g_rec.log);
    g_rec.log = nullptr; g_rec.rank = rank; g_rec.used = 0; g_rec.cap = 0; g_rec.entries.clear(); g_rec.overflow = false;
    if (hipMalloc((void **)&g_rec.log, log_bytes) != hipSuccess) { g_rec.rank = -1; return 1; }
    g_rec.cap = log_bytes;
    return 0;
}
// entries recorded, bytes used (negative: the log overflowed and is useless)
long long loopback_record_size(long long *bytes)
{
    std::lock_guard<std::mutex> lk(g_rec.m);
    if (bytes) *bytes = (long long)g_rec.used;
    return g_rec.overflow ? -1 : (long long)g_rec.entries.size();
}
// the log to / from a file: the replay may then run in a process of its own (a profiler around ONE thread and ONE handle)
int loopback_log_save(const char *path)
{
    std::lock_guard<std::mutex> lk(g_rec.m);
    if (!g_rec.log || g_rec.overflow) return 1;
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    FILE *f = fopen(path, "wb");
    if (!f) return 3;
    const long long hdr[3] = {(long long)g_rec.rank, (long long)g_rec.entries.size(), (long long)g_rec.used};
    fwrite(hdr, sizeof(hdr), 1, f);
    for (auto &e : g_rec.entries) { const long long v[2] = {(long long)e.first, (long long)e.second}; fwrite(v, sizeof(v), 1, f); }
    std::vector<char> chunk(64u << 20);
    for (size_t off = 0; off < g_rec.used; off += chunk.size()) {
        const size_t n = std::min(chunk.size(), g_rec.used - off);
        if (hipMemcpy(chunk.data(), g_rec.log + off, n, hipMemcpyDeviceToHost) != hipSuccess) { fclose(f); return 4; }
        if (fwrite(chunk.data(), 1, n, f) != n) { fclose(f); return 5; }
    }
    fclose(f);
    return 0;
}
int loopback_log_load(const char *path)
{
    std::lock_guard<std::mutex> lk(g_rec.m);
    FILE *f = fopen(path, "rb");
    if (!f) return 3;
    long long hdr[3];
    if (fread(hdr, sizeof(hdr), 1, f) != 1) { fclose(f); return 6; }
    if (g_rec.log) (void)hipFree(g_rec.log);
    g_rec.log = nullptr; g_rec.rank = (int)hdr[0]; g_rec.used = (size_t)hdr[2]; g_rec.cap = g_rec.used; g_rec.overflow = false;
    g_rec.entries.resize((size_t)hdr[1]);
    for (auto &e : g_rec.entries) { long long v[2]; if (fread(v, sizeof(v), 1, f) != 1) { fclose(f); return 6; } e.first = (size_t)v[0]; e.second = (size_t)v[1]; }
    if (hipMalloc((void **)&g_rec.log, std::max<size_t>(g_rec.used, 256)) != hipSuccess) { fclose(f); return 1; }
    std::vector<char> chunk(64u << 20);
    for (size_t off = 0; off < g_rec.used; off += chunk.size()) {
        const size_t n = std::min(chunk.size(), g_rec.used - off);
        if (fread(chunk.data(), 1, n, f) != n) { fclose(f); return 6; }
        if (hipMemcpy(g_rec.log + off, chunk.data(), n, hipMemcpyHostToDevice) != hipSuccess) { fclose(f); return 4; }
    }
    fclose(f);
    return 0;
}
void loopback_replay_id(ncclUniqueId *id)
{
    memset(id->internal, 0, NCCL_UNIQUE_ID_BYTES);
    snprintf(id->internal, NCCL_UNIQUE_ID_BYTES, "replay-%u", g_next_id.fetch_add(1));
}
// a named kernel on `stream`: brackets a window in a rocprofv3 kernel trace
void loopback_marker(void *stream) { hipLaunchKernelGGL(loopback_marker_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (int *)nullptr); }

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    if (!strncmp(id.internal, "replay-", 7)) {          // rank `rank` alone, against the log of what it received in the recorded run
        if (rank != g_rec.rank || !g_rec.log || g_rec.overflow) { fprintf(stderr, "loopback_rccl replay: no usable log for rank %d\n", rank); return ncclInvalidArgument; }
        Comm *c = new Comm();
        c->w = std::make_shared<World>();
        c->w->n = nranks; c->rank = rank; c->replay = true;
        *comm = reinterpret_cast<ncclComm_t>(c);
        return ncclSuccess;
    }
    std::shared_ptr<World> w;
    {
        std::lock_guard<std::mutex> lk(g_reg_m);
        std::string key(id.internal, NCCL_UNIQUE_ID_BYTES);
        auto it = g_worlds.find(key);
        if (it == g_worlds.end()) {
            w = std::make_shared<World>();
            w->n = nranks;
            w->box.assign(nranks, std::vector<std::deque<std::shared_ptr<SendDesc>>>(nranks));
            w->posted.assign(nranks, 0);
            for (int p = 0; p < 2; ++p) {
                LB_HIP(hipMalloc(&w->stage[p], kStageBytes * (size_t)nranks));
                w->ready[p].resize(nranks); w->done[p].resize(nranks);
                for (int r = 0; r < nranks; ++r) {
                    LB_HIP(hipEventCreateWithFlags(&w->ready[p][r], hipEventDisableTiming));
                    LB_HIP(hipEventCreateWithFlags(&w->done[p][r], hipEventDisableTiming));
                }
            }
            g_worlds[key] = w;
        } else {
            w = it->second;
            if (w->n != nranks) return ncclInvalidArgument;
        }
    }
    {
        std::unique_lock<std::mutex> lk(w->m);
        w->joined += 1;
        w->cv.notify_all();
        if (!wait_for(*w, lk, [&] { return w->joined >= w->n; })) { fprintf(stderr, "loopback_rccl: rank %d of %d waited %d s for the others to join\n", rank, nranks, kWaitSeconds); return ncclSystemError; }
    }
    Comm *c = new Comm();
    c->w = w; c->rank = rank;
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c) return ncclSuccess;
    if (getenv("LOOPBACK_RCCL_STATS") && c->rank == 0)
        fprintf(stderr, "loopback_rccl: %lld p2p messages, %lld bytes, %lld all-reduces (all ranks)\n", (long long)c->w->p2p_msgs, (long long)c->w->p2p_bytes, (long long)c->w->allreduces);
    delete c;          // the world (events, staging) lives until the process ends: other ranks may still be running
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "loopback: HIP call failed";
    case ncclSystemError: return "loopback: a peer did not show up";
    case ncclInvalidArgument: return "loopback: invalid argument (message sizes of a pair must match)";
    default: return "loopback: error";
    }
}

ncclResult_t ncclGroupStart() { t_depth += 1; return ncclSuccess; }

ncclResult_t ncclGroupEnd()
{
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    std::vector<PendingOp> ops;
    ops.swap(t_ops);
    return run_group(ops);
}

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    const size_t ts = type_size(datatype);
    if (!c || !ts || peer < 0 || peer >= c->w->n || peer == c->rank) return ncclInvalidArgument;
    t_ops.push_back(PendingOp{true, const_cast<void *>(sendbuff), count * ts, peer, c, stream});
    if (t_depth > 0) return ncclSuccess;
    std::vector<PendingOp> ops; ops.swap(t_ops);
    return run_group(ops);
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    const size_t ts = type_size(datatype);
    if (!c || !ts || peer < 0 || peer >= c->w->n || peer == c->rank) return ncclInvalidArgument;
    t_ops.push_back(PendingOp{false, recvbuff, count * ts, peer, c, stream});
    if (t_depth > 0) return ncclSuccess;
    std::vector<PendingOp> ops; ops.swap(t_ops);
    return run_group(ops);
}

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c || (datatype != ncclFloat64 && datatype != ncclInt32) || (op != ncclSum && op != ncclMax)) return ncclInvalidArgument;
    World &w = *c->w;
    const size_t ts = type_size(datatype), bytes = count * ts;
    if (c->replay) { delay_on(stream); return replay_into(c, recvbuff, bytes, stream, "all-reduce"); }
    if (bytes > kStageBytes) return ncclInvalidArgument;
    const long long s = c->ar_seq++;
    const int p = (int)(s & 1), me = c->rank;
    w.allreduces += 1;
    if (w.n == 1) {
        if (sendbuff != recvbuff) LB_HIP(hipMemcpyAsync(recvbuff, sendbuff, bytes, hipMemcpyDeviceToDevice, stream));
        return ncclSuccess;
    }
    // the slots of this parity were read by all-reduce s - 2: every rank recorded done[p] for it before it posted s - 1, and we are past s - 1's rendezvous
    if (s >= 2) for (int r = 0; r < w.n; ++r) LB_HIP(hipStreamWaitEvent(stream, w.done[p][r], 0));
    char *slots = static_cast<char *>(w.stage[p]);
    LB_HIP(hipMemcpyAsync(slots + (size_t)me * kStageBytes, sendbuff, bytes, hipMemcpyDeviceToDevice, stream));
    LB_HIP(hipEventRecord(w.ready[p][me], stream));
    {
        std::unique_lock<std::mutex> lk(w.m);
        w.posted[me] = s + 1;
        w.cv.notify_all();
        if (!wait_for(w, lk, [&] { for (int r = 0; r < w.n; ++r) if (w.posted[r] < s + 1) return false; return true; })) {
            fprintf(stderr, "loopback_rccl: rank %d waited %d s in all-reduce %lld\n", me, kWaitSeconds, s);
            return ncclSystemError;
        }
    }
    for (int r = 0; r < w.n; ++r) if (r != me) LB_HIP(hipStreamWaitEvent(stream, w.ready[p][r], 0));
    const unsigned blocks = (unsigned)((count + 255) / 256);
    if (datatype == ncclFloat64) hipLaunchKernelGGL(k_reduce<double>, dim3(blocks), dim3(256), 0, stream, static_cast<double *>(recvbuff), slots, kStageBytes, w.n, count, op == ncclMax ? 1 : 0);
    else hipLaunchKernelGGL(k_reduce<int>, dim3(blocks), dim3(256), 0, stream, static_cast<int *>(recvbuff), slots, kStageBytes, w.n, count, op == ncclMax ? 1 : 0);
    LB_HIP(hipGetLastError());
    LB_HIP(hipEventRecord(w.done[p][me], stream));
    return record(c, recvbuff, bytes, stream);
}

}  // extern "C"
