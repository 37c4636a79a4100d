// sph_host_transport.h -- a SECTION of csrc/sph_mi355x.hip's one translation unit (included there once, inside its anonymous namespace, in file order):
// the slab transport: callbacks or native RCCL (dlopen'ed), particle exchange, halo refreshes, re-balancing.  Not a stand-alone header: it uses SphHandle and the helpers defined above its include.

// ---------------------------------------------------------------------------------------------
// multi-GPU slab transport.  The library packs/unpacks on the device; the caller's
// callbacks move the bytes (RCCL send/recv over xGMI in production, gloo in the tests).
// ---------------------------------------------------------------------------------------------
int comm_fail(SphHandle *h, const char *what, int rc) { return fail(h, SPH_E_STATE, "comm callback %s failed (%d)", what, rc); }
inline bool slab_stream_ordered(const SphHandle *h) { return h->slab && (h->native || (!h->comm.on_host && h->comm.stream_ordered)); }
// sharded DFSPH with the device-side loop control of the single-GPU path (needs the transport's in-place all-reduce of reduce_buf)
inline bool slab_async(const SphHandle *h) { return h->slab && (h->native || h->comm.allreduce_stream) && h->red_dev; }

int native_allreduce_stream(SphHandle *h, int n, int op, hipStream_t stream = nullptr);

// all-reduce red_dev[0..n) over the slabs, ordered on `stream` (default: the handle's stream; a stream-ordered CALLBACK transport always uses the handle's)
int slab_allreduce_stream(SphHandle *h, int n, int op, hipStream_t stream = nullptr)
{
    if (!stream) stream = h->stream;
    h->comm_stat[4] += 1;
    if (h->native) return native_allreduce_stream(h, n, op, stream);
    const SphComm &cm = h->comm;
    if (cm.on_host) {                                  // host transport: stage through the caller's host buffer
        HIP_TRY(h, hipMemcpyAsync(cm.reduce_buf, h->red_dev, sizeof(double) * n, hipMemcpyDeviceToHost, stream));
        HIP_TRY(h, hipStreamSynchronize(stream));
    }
    // synchronous discipline on device buffers: the transport works on its own stream, so the pair must be complete before it reads
    // (it returns only when the reduced values are in place)
    if (!cm.on_host && !cm.stream_ordered) HIP_TRY(h, hipStreamSynchronize(stream));
    int rc = cm.allreduce_stream(cm.user, n, op);
    if (rc) return comm_fail(h, "allreduce_stream", rc);
    if (cm.on_host) HIP_TRY(h, hipMemcpyAsync(h->red_dev, cm.reduce_buf, sizeof(double) * n, hipMemcpyHostToDevice, stream));
    return SPH_OK;
}

// ---------------------------------------------------------------------------------------------
// native RCCL transport: ncclSend / ncclRecv to the left and right slab neighbour (one direct xGMI link per pair) and
// ncclAllReduce of the residual pair, issued by the library on its own stream -- no Python, no host waits.  librccl is
// dlopen'ed so that the library itself has no link-time dependency on it.
// ---------------------------------------------------------------------------------------------
struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string why;
    bool ok = false;
};

RcclApi &rccl()
{
    static RcclApi api = [] {
        RcclApi a;
        // development override (SPH_DEV=1): another library with librccl's entry points -- tests/loopback_rccl.hip drives this transport
        // with several handles of ONE process on one GPU.  sph_rccl_attach records it in the handle's overrides.
        if (const char *dev = dev_env(nullptr, "SPH_RCCL_LIB")) {
            a.lib = dlopen(dev, RTLD_NOW | RTLD_LOCAL);
            if (!a.lib) { a.why = std::string("SPH_RCCL_LIB: ") + dlerror(); return a; }
        }
        for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            if (a.lib) break;
            a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!a.lib) { a.why = "librccl.so not found"; return a; }
#define SPH_RCCL_SYM(field, sym) a.field = (decltype(a.field))dlsym(a.lib, sym); if (!a.field) { a.why = std::string("missing symbol ") + sym; return a; }
        SPH_RCCL_SYM(GetUniqueId, "ncclGetUniqueId") SPH_RCCL_SYM(CommInitRank, "ncclCommInitRank") SPH_RCCL_SYM(CommDestroy, "ncclCommDestroy")
        SPH_RCCL_SYM(GroupStart, "ncclGroupStart") SPH_RCCL_SYM(GroupEnd, "ncclGroupEnd") SPH_RCCL_SYM(Send, "ncclSend") SPH_RCCL_SYM(Recv, "ncclRecv")
        SPH_RCCL_SYM(AllReduce, "ncclAllReduce") SPH_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef SPH_RCCL_SYM
        a.ok = true;
        return a;
    }();
    return api;
}

#define NCCL_TRY(h, expr)                                                                                        \
    do {                                                                                                         \
        ncclResult_t r_ = (expr);                                                                                \
        if (r_ != ncclSuccess) return fail(h, SPH_E_HIP, "%s failed: %s", #expr, rccl().GetErrorString(r_));     \
    } while (0)

// exchange_buffers of the native transport: one group of up to four point-to-point transfers, ordered on the handle's stream
// gather_doubles > 0: the same group also carries this slab's gath_dev slot (that many doubles) to EVERY other slab and theirs back -- the residual's
// (sum, count, flags) travel with the halo, one start-up latency per solver iteration instead of the halo's plus an all-reduce's
int native_exchange(SphHandle *h, size_t sl, size_t sr, size_t rl, size_t rr, hipStream_t stream = nullptr, int gather_doubles = 0)
{
    RcclApi &n = rccl();
    if (!stream) stream = h->stream;
    const int left = h->slab_rank > 0 ? h->slab_rank - 1 : -1, right = h->slab_rank < h->nslab - 1 ? h->slab_rank + 1 : -1;
    if (!gather_doubles && !((left >= 0 && (sl || rl)) || (right >= 0 && (sr || rr)))) return SPH_OK;
    NCCL_TRY(h, n.GroupStart());
    for (int p = 0; gather_doubles && p < h->nslab; ++p) {
        if (p == h->slab_rank) continue;
        NCCL_TRY(h, n.Send(h->gath_dev + 4 * h->slab_rank, (size_t)gather_doubles, ncclDouble, p, h->nccl, stream));
        NCCL_TRY(h, n.Recv(h->gath_dev + 4 * p, (size_t)gather_doubles, ncclDouble, p, h->nccl, stream));
    }
    if (left >= 0) {
        if (sl) NCCL_TRY(h, n.Send(h->dsend[0], sl, ncclChar, left, h->nccl, stream));
        if (rl) NCCL_TRY(h, n.Recv(h->drecv[0], rl, ncclChar, left, h->nccl, stream));
    }
    if (right >= 0) {
        if (sr) NCCL_TRY(h, n.Send(h->dsend[1], sr, ncclChar, right, h->nccl, stream));
        if (rr) NCCL_TRY(h, n.Recv(h->drecv[1], rr, ncclChar, right, h->nccl, stream));
    }
    NCCL_TRY(h, n.GroupEnd());
    return SPH_OK;
}

// exchange_counts of the native transport: n ints each way with each neighbour, then the host reads what it received
constexpr int kCountInts = 8;
int native_exchange_counts_n(SphHandle *h, int n, const int32_t *sl, const int32_t *sr, int32_t *rl, int32_t *rr)
{
    RcclApi &api = rccl();
    const int left = h->slab_rank > 0 ? h->slab_rank - 1 : -1, right = h->slab_rank < h->nslab - 1 ? h->slab_rank + 1 : -1;
    memset(h->cnt_host, 0, sizeof(int) * 4 * kCountInts);
    for (int k = 0; k < n; ++k) { h->cnt_host[k] = sl[k]; h->cnt_host[kCountInts + k] = sr[k]; }
    HIP_TRY(h, hipMemcpyAsync(h->cnt_dev, h->cnt_host, sizeof(int) * 4 * kCountInts, hipMemcpyHostToDevice, h->stream));
    if (left >= 0 || right >= 0) {
        NCCL_TRY(h, api.GroupStart());
        if (left >= 0) {
            NCCL_TRY(h, api.Send(h->cnt_dev + 0, (size_t)n, ncclInt32, left, h->nccl, h->stream));
            NCCL_TRY(h, api.Recv(h->cnt_dev + 2 * kCountInts, (size_t)n, ncclInt32, left, h->nccl, h->stream));
        }
        if (right >= 0) {
            NCCL_TRY(h, api.Send(h->cnt_dev + kCountInts, (size_t)n, ncclInt32, right, h->nccl, h->stream));
            NCCL_TRY(h, api.Recv(h->cnt_dev + 3 * kCountInts, (size_t)n, ncclInt32, right, h->nccl, h->stream));
        }
        NCCL_TRY(h, api.GroupEnd());
    }
    HIP_TRY(h, hipMemcpyAsync(h->cnt_host, h->cnt_dev, sizeof(int) * 4 * kCountInts, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int k = 0; k < n; ++k) { rl[k] = h->cnt_host[2 * kCountInts + k]; rr[k] = h->cnt_host[3 * kCountInts + k]; }
    return SPH_OK;
}
int native_exchange_counts(SphHandle *h, int32_t sl, int32_t sr, int32_t *rl, int32_t *rr) { return native_exchange_counts_n(h, 1, &sl, &sr, rl, rr); }
// the same with the n ints per side already in cnt_dev[0..n) / cnt_dev[kCountInts..] (k_classify_scan): no upload; the classification's
// counters come back in the same read-back (counters_host)
int native_exchange_counts_dev(SphHandle *h, int n, int32_t *rl, int32_t *rr)
{
    RcclApi &api = rccl();
    const int left = h->slab_rank > 0 ? h->slab_rank - 1 : -1, right = h->slab_rank < h->nslab - 1 ? h->slab_rank + 1 : -1;
    h->comm_stat[3] += 1;
    static_assert(kCountInts == 8, "k_classify_scan zeroes wire[16..32), the receive half");
    if (left >= 0 || right >= 0) {
        NCCL_TRY(h, api.GroupStart());
        if (left >= 0) {
            NCCL_TRY(h, api.Send(h->cnt_dev + 0, (size_t)n, ncclInt32, left, h->nccl, h->stream));
            NCCL_TRY(h, api.Recv(h->cnt_dev + 2 * kCountInts, (size_t)n, ncclInt32, left, h->nccl, h->stream));
        }
        if (right >= 0) {
            NCCL_TRY(h, api.Send(h->cnt_dev + kCountInts, (size_t)n, ncclInt32, right, h->nccl, h->stream));
            NCCL_TRY(h, api.Recv(h->cnt_dev + 3 * kCountInts, (size_t)n, ncclInt32, right, h->nccl, h->stream));
        }
        NCCL_TRY(h, api.GroupEnd());
    }
    HIP_TRY(h, hipMemcpyAsync(h->cnt_host, h->cnt_dev, sizeof(int) * 4 * kCountInts, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->counters_host, h->counters, sizeof(int) * kSlabCounters, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int k = 0; k < n; ++k) { rl[k] = h->cnt_host[2 * kCountInts + k]; rr[k] = h->cnt_host[3 * kCountInts + k]; }
    return SPH_OK;
}

int native_allreduce_stream(SphHandle *h, int n, int op, hipStream_t stream)
{
    NCCL_TRY(h, rccl().AllReduce(h->red_dev, h->red_dev, (size_t)n, ncclDouble, op == 0 ? ncclSum : ncclMax, h->nccl, stream ? stream : h->stream));
    return SPH_OK;
}

// neighbour counts / host-side all-reduce through whichever transport the handle has
// n ints to each neighbour, n from each (absent neighbour: zeros): one host round trip where the transport can (native RCCL, a SphComm with
// exchange_counts_n), n of them through a plain exchange_counts
// doubles the transport's reduce buffer must hold on this handle (a rigid body's by-id sums: 4 per sample)
inline size_t slab_reduce_need(const SphHandle *h) { return h->rigid ? 4 * (size_t)h->Nr + 8 : 4; }

int slab_exchange_counts_n(SphHandle *h, int n, const int32_t *sl, const int32_t *sr, int32_t *rl, int32_t *rr)
{
    if (n > kCountInts) return fail(h, SPH_E_INVALID, "count exchange of %d ints", n);
    for (int k = 0; k < n; ++k) rl[k] = rr[k] = 0;
    if (h->native) { h->comm_stat[3] += 1; return native_exchange_counts_n(h, n, sl, sr, rl, rr); }
    if (h->comm.exchange_counts_n) {
        h->comm_stat[3] += 1;
        int rc = h->comm.exchange_counts_n(h->comm.user, n, sl, sr, rl, rr);
        return rc ? comm_fail(h, "exchange_counts_n", rc) : SPH_OK;
    }
    for (int k = 0; k < n; ++k) {
        h->comm_stat[3] += 1;
        int rc = h->comm.exchange_counts(h->comm.user, sl[k], sr[k], &rl[k], &rr[k]);
        if (rc) return comm_fail(h, "exchange_counts", rc);
    }
    return SPH_OK;
}

int slab_allreduce_host(SphHandle *h, double *v, int n, int op)
{
    h->comm_stat[5] += 1;
    if (!h->native) {
        int rc = h->comm.allreduce(h->comm.user, v, n, op);
        return rc ? comm_fail(h, "allreduce", rc) : SPH_OK;
    }
    if (n > h->red_cap) return fail(h, SPH_E_INVALID, "all-reduce of %d doubles exceeds the reduce buffer (%d)", n, h->red_cap);
    memcpy(h->red_host, v, sizeof(double) * n);
    HIP_TRY(h, hipMemcpyAsync(h->red_dev, h->red_host, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
    int rc = native_allreduce_stream(h, n, op);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->red_host, h->red_dev, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    memcpy(v, h->red_host, sizeof(double) * n);
    return SPH_OK;
}

// `stream`: where the packed data was produced and the unpack will run (the handle's stream, or the halo stream of an overlapped refresh).
// A stream-ordered CALLBACK transport enqueues on the handle's own stream whatever we say, so overlapped refreshes are only taken with the
// native transport or a synchronous one (slab_can_overlap).
int slab_xfer(SphHandle *h, size_t sl, size_t sr, size_t rl, size_t rr, hipStream_t stream = nullptr, int gather_doubles = 0)
{
    if (!stream) stream = h->stream;
    const SphComm &cm = h->comm;
    if (sl > cm.capacity || sr > cm.capacity || rl > cm.capacity || rr > cm.capacity)
        return fail(h, SPH_E_OVERFLOW, "halo message of %zu bytes exceeds the comm buffer capacity %zu", std::max(std::max(sl, sr), std::max(rl, rr)), cm.capacity);
    if (cm.on_host) {
        if (sl) HIP_TRY(h, hipMemcpyAsync(cm.send_left, h->dsend[0], sl, hipMemcpyDeviceToHost, stream));
        if (sr) HIP_TRY(h, hipMemcpyAsync(cm.send_right, h->dsend[1], sr, hipMemcpyDeviceToHost, stream));
    }
    if (!slab_stream_ordered(h)) HIP_TRY(h, hipStreamSynchronize(stream));     // packed data complete before the transport reads it
    h->comm_stat[0] += 1; h->comm_stat[1] += (long long)(sl + sr); h->comm_stat[2] += (long long)(rl + rr);       // (counted even when this rank's share of the exchange is empty)
    int rc;
    if (h->native) {
        if ((rc = native_exchange(h, sl, sr, rl, rr, stream, gather_doubles))) return rc;
    } else {
        rc = cm.exchange_buffers(cm.user, sl, sr, rl, rr);      // stream-ordered transports enqueue behind the pack kernels instead
        if (rc) return comm_fail(h, "exchange_buffers", rc);
    }
    if (cm.on_host) {
        if (rl) HIP_TRY(h, hipMemcpyAsync(h->drecv[0], cm.recv_left, rl, hipMemcpyHostToDevice, stream));
        if (rr) HIP_TRY(h, hipMemcpyAsync(h->drecv[1], cm.recv_right, rr, hipMemcpyHostToDevice, stream));
    }
    return SPH_OK;
}
inline bool slab_can_overlap(const SphHandle *h) { return h->overlap && h->overlap_on && (h->native || !slab_stream_ordered(h)); }

int read_counters(SphHandle *h)
{
    HIP_TRY(h, hipMemcpyAsync(h->counters_host, h->counters, sizeof(int) * kSlabCounters, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SPH_OK;
}

// Every `slab_rebalance_every` steps: global per-column particle histogram (one all-reduce of gx counts), new
// equal-count cuts on every rank alike.  Only the cuts change here; the migration that follows moves the
// particles of the shifted columns to the neighbour that now owns them.  Results do not depend on the cuts
// (every sum runs in (cell, id) order), so re-balancing is invisible in the output.
int slab_rebalance(SphHandle *h)
{
    if (!h->comm_set) return fail(h, SPH_E_STATE, "slab handle needs sph_set_comm before stepping");
    Consts &c = h->c;
    hipStream_t s = h->stream;
    HIP_TRY(h, hipMemsetAsync(h->col_hist, 0, sizeof(int) * (size_t)c.gx, s));
    {
        ProfScope ps(h, K_SLAB);
        hipLaunchKernelGGL(k_column_histogram, grid_for(c.n), dim3(kBlock), 0, s, c, h->P[h->pcur], h->id[h->icur], h->col_hist);
    }
    HIP_TRY(h, hipMemcpyAsync(h->col_hist_host, h->col_hist, sizeof(int) * (size_t)c.gx, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    std::vector<double> v((size_t)c.gx);
    for (int x = 0; x < c.gx; ++x) v[x] = (double)h->col_hist_host[x];
    int rc = slab_allreduce_host(h, v.data(), c.gx, 0);
    if (rc) return rc;
    std::vector<long long> hist((size_t)c.gx);
    for (int x = 0; x < c.gx; ++x) hist[x] = (long long)v[x];
    std::vector<int> cut;
    replan_slab_cuts(hist, c.gx, h->nslab, h->cuts, cut, h->geom.layers);
    h->cuts_moved = cut != h->cuts;
    if (h->cuts_moved) {
        h->cuts = cut;
        set_slab_geometry(h);
        ++h->n_recuts;
        if ((rc = slab_local_grid(h))) return rc;            // the slab's cell slots follow its columns
    }
    return SPH_OK;
}

// Start of a step on a slab handle: particles that left [x_lo, x_hi) move to their new owner, last step's ghosts go, and the `layers`
// columns next to each cut are copied to the neighbour as this step's ghosts.  Old ghosts and leavers are only MARKED dead; the counting
// sort drops them.
//   ordinary step   ONE message per neighbour carries migrants and ghost copies together (k_classify_slab, all three modes), after ONE
//                   count exchange of five ints per side: records, ghost copies per column, and -- because a leaver that lands in one of
//                   my ghost columns simply stays here as a ghost, the new owner does not send it back -- how many I kept per column, which
//                   is how many of the receiver's arrivals belong to the columns it copies to me.  Two host round trips per step (the
//                   counters read-back and the count exchange) where the two-round form takes four.
//   re-cut step     two rounds (migrate, then ghost copies over what arrived): moved cuts can carry whole columns across a slab, so what
//                   arrives from one side may belong to the columns copied to the other.
// Afterwards edge_n[k][l] = particles of column l of ordered edge list k (0 ghost-left, 1 send-left, 2 send-right, 3 ghost-right), known on
// both sides of a cut alike without looking at the sorted arrays.
int slab_exchange_particles(SphHandle *h)
{
    if (!h->comm_set) return fail(h, SPH_E_STATE, "slab handle needs sph_set_comm before stepping");
    Consts &c = h->c;
    hipStream_t s = h->stream;
    const dim3 b(kBlock);
    float *warm = carries_scalar(h) ? h->warm[h->wcur] : nullptr;   // dfsph warm_start_k / iisph p_past travel with the particle
    const int cap_rec = (int)std::min<size_t>(h->comm.capacity / 32, 0x7fffffff);
    int rc;
    int n_res = c.n;                                   // resident slots, dead ones included
    int ndead = 0;
    int own_ghost[2][2] = {{0, 0}, {0, 0}}, own_kept[2][2] = {{0, 0}, {0, 0}};      // [side][column]: ghost copies I sent, leavers I kept as ghosts
    int got_ghost[2][2] = {{0, 0}, {0, 0}}, got_kept[2][2] = {{0, 0}, {0, 0}};      // ... and what the neighbour on that side reported
    auto round = [&](int mode) -> int {
        {
            ProfScope ps(h, K_SLAB);
            const int nblk = (int)grid_for(n_res).x;
            hipLaunchKernelGGL(k_classify_count, dim3(nblk), b, 0, s, c, h->geom, mode, h->P[h->pcur], h->id[h->icur], h->dead, nblk, h->class_cnt);
            hipLaunchKernelGGL(k_classify_scan, dim3(kSlabCounted), dim3(kScanBlock), 0, s, nblk, h->class_cnt, h->counters, h->native ? h->cnt_dev : (int *)nullptr);
            hipLaunchKernelGGL(k_classify_write, dim3(nblk), b, 0, s, c, h->geom, mode, h->P[h->pcur], h->V[h->vcur], warm, h->id[h->icur], h->dead,
                               (float4 *)h->dsend[0], (float4 *)h->dsend[1], cap_rec, nblk, h->class_cnt, h->ds);
        }
        int r;
        const int *ct = h->counters_host;
        int32_t sl[5], sr[5], rl[5], rr[5];
        if (h->native) {
            // the counts go from device to device (k_classify_scan left them in wire order) and come back to the host together with what the
            // neighbours sent: ONE host round trip per exchange round
            if ((r = native_exchange_counts_dev(h, 5, rl, rr))) return r;
        } else if ((r = read_counters(h))) return r;
        if (ct[0] > cap_rec || ct[1] > cap_rec) return fail(h, SPH_E_OVERFLOW, "%d/%d particle records exceed the comm buffer (%d records)", ct[0], ct[1], cap_rec);
        if (mode & kSlabMigrate) ndead = ct[2];
        { const int32_t a[5] = {ct[0], ct[3], ct[4], ct[5], ct[6]}, b2[5] = {ct[1], ct[7], ct[8], ct[9], ct[10]}; for (int q = 0; q < 5; ++q) { sl[q] = a[q]; sr[q] = b2[q]; } }
        if (!h->native && (r = slab_exchange_counts_n(h, 5, sl, sr, rl, rr))) return r;
        for (int l = 0; l < 2; ++l) {
            own_ghost[0][l] += sl[1 + l]; own_kept[0][l] += sl[3 + l]; own_ghost[1][l] += sr[1 + l]; own_kept[1][l] += sr[3 + l];
            got_ghost[0][l] += rl[1 + l]; got_kept[0][l] += rl[3 + l]; got_ghost[1][l] += rr[1 + l]; got_kept[1][l] += rr[3 + l];
        }
        if ((long long)n_res + rl[0] + rr[0] > h->ncap) return fail(h, SPH_E_OVERFLOW, "slab capacity %d exceeded by the particle exchange", h->ncap);
        if ((r = slab_xfer(h, 32 * (size_t)sl[0], 32 * (size_t)sr[0], 32 * (size_t)rl[0], 32 * (size_t)rr[0]))) return r;
        {
            ProfScope ps(h, K_SLAB);
            if (rl[0]) hipLaunchKernelGGL(k_append_records, grid_for(rl[0]), b, 0, s, (const float4 *)h->drecv[0], rl[0], n_res, h->P[h->pcur], h->V[h->vcur], warm, h->id[h->icur], h->dead);
            if (rr[0]) hipLaunchKernelGGL(k_append_records, grid_for(rr[0]), b, 0, s, (const float4 *)h->drecv[1], rr[0], n_res + rl[0], h->P[h->pcur], h->V[h->vcur], warm, h->id[h->icur], h->dead);
        }
        // owned particles: migrants out, migrants in (a record is a migrant unless it is a ghost copy)
        h->n_owned += -(sl[0] - sl[1] - sl[2]) - (sr[0] - sr[1] - sr[2]) + (rl[0] - rl[1] - rl[2]) + (rr[0] - rr[1] - rr[2]);
        n_res += rl[0] + rr[0];
        c.n = n_res;
        return SPH_OK;
    };
    if (h->cuts_moved) {
        if ((rc = round(kSlabMigrate))) return rc;
        if ((rc = round(kSlabGhosts))) return rc;
        h->cuts_moved = false;
    } else {
        if ((rc = round(kSlabMigrate | kSlabGhosts | kSlabKeep))) return rc;
    }
    HIP_TRY(h, hipGetLastError());
    h->n_dead = ndead;                        // the sort runs over everything resident, dead slots included
    for (int l = 0; l < 2; ++l) {
        h->edge_n[0][l] = got_ghost[0][l] + own_kept[0][l];      // ghost-left column l: the left neighbour's copies + my leavers that stayed as ghosts
        h->edge_n[1][l] = own_ghost[0][l] + got_kept[0][l];      // send-left column l: my copies + arrivals the left neighbour kept as ghosts
        h->edge_n[2][l] = own_ghost[1][l] + got_kept[1][l];
        h->edge_n[3][l] = got_ghost[1][l] + own_kept[1][l];
    }
    h->n_ghost = h->edge_n[0][0] + h->edge_n[0][1] + h->edge_n[3][0] + h->edge_n[3][1];
    return SPH_OK;
}

// refresh one field of the ghosts after the sweep that produced it (mode: see k_pack_field).  cols: how many of the ghost columns per side
// (1 = the column next to the cut only; the lists hold it first).
int slab_exchange_field(SphHandle *h, int mode, float4 *P, float4 *V, float *rho, int cols = 2, int gather_doubles = 0)
{
    hipStream_t s = h->stream;
    const dim3 b(kBlock);
    const size_t fl = mode == 0 ? 1 : (mode == 1 ? 3 : 2);      // modes 2 and 3: two floats
    auto cnt = [&](int k) { return h->edge_n[k][0] + (cols >= 2 && h->geom.layers >= 2 ? h->edge_n[k][1] : 0); };
    const int nsl = cnt(1), nsr = cnt(2), nrl = cnt(0), nrr = cnt(3);
    float *S = h->c.kr_split ? h->krho : nullptr;               // where the per-sweep scalar k / rho lives (else P.w)
    {
        ProfScope ps(h, K_SLAB);
        if (nsl + nsr)
            hipLaunchKernelGGL(k_pack_field, grid_for(nsl + nsr), b, 0, s, h->edge_list[1], nsl, (float *)h->dsend[0], h->edge_list[2], nsr,
                               (float *)h->dsend[1], mode, P, V, S);
    }
    int rc = slab_xfer(h, 4 * fl * nsl, 4 * fl * nsr, 4 * fl * nrl, 4 * fl * nrr, nullptr, gather_doubles);
    if (rc) return rc;
    {
        ProfScope ps(h, K_SLAB);
        if (nrl + nrr)
            hipLaunchKernelGGL(k_unpack_field, grid_for(nrl + nrr), b, 0, s, h->edge_list[0], nrl, (const float *)h->drecv[0], h->edge_list[3], nrr,
                               (const float *)h->drecv[1], mode, P, V, rho, S);
    }
    HIP_TRY(h, hipGetLastError());
    return SPH_OK;
}

// The one halo refresh of a dfsph solver iteration on a two-column slab handle: the residual sweep's value (rho_derivative / rho_adv) for
// the inner ghost column, the owner's k / rho for the outer one -- 4 bytes per ghost (k_pack_resid / k_unpack_resid).  With `overlap` the
// caller has run the EDGE tiles of the sweep only: the pack waits for them (ev_edge) on the halo's own stream, and whoever reads the ghosts
// next waits for ev_halo -- the interior tiles of the sweep run under the transfer.
int slab_exchange_resid(SphHandle *h, bool dens, float *val, bool overlap, bool wait_edge = true)
{
    hipStream_t s = overlap ? h->xstream : h->stream;
    const dim3 b(kBlock);
    const int nsl = h->edge_n[1][0] + h->edge_n[1][1], nsr = h->edge_n[2][0] + h->edge_n[2][1];
    const int nrl = h->edge_n[0][0] + h->edge_n[0][1], nrr = h->edge_n[3][0] + h->edge_n[3][1];
    float *S = h->c.kr_split ? h->krho : nullptr;
    float4 *P = h->P[1 - h->pcur];
    if (overlap && wait_edge) HIP_TRY(h, hipStreamWaitEvent(s, h->ev_edge, 0));
    {
        ProfScope ps(h, K_SLAB, s);
        if (nsl + nsr)
            hipLaunchKernelGGL(k_pack_resid, grid_for(nsl + nsr), b, 0, s, h->edge_list[1], nsl, h->edge_n[1][0], (float *)h->dsend[0], h->edge_list[2], nsr,
                               h->edge_n[2][0], (float *)h->dsend[1], val, P, S);
    }
    int rc = slab_xfer(h, 4 * (size_t)nsl, 4 * (size_t)nsr, 4 * (size_t)nrl, 4 * (size_t)nrr, s);
    if (rc) return rc;
    {
        ProfScope ps(h, K_SLAB, s);
        if (nrl + nrr)
            hipLaunchKernelGGL(k_unpack_resid, grid_for(nrl + nrr), b, 0, s, h->c, h->edge_list[0], nrl, h->edge_n[0][0], (const float *)h->drecv[0], h->edge_list[3], nrr,
                               h->edge_n[3][0], (const float *)h->drecv[1], dens ? 1 : 0, h->aux, h->rho, h->ds, val, P, S, dens ? h->flow_d6 : kNoFlow);
    }
    HIP_TRY(h, hipGetLastError());
    if (overlap) HIP_TRY(h, hipEventRecord(h->ev_halo, s));
    return SPH_OK;
}
