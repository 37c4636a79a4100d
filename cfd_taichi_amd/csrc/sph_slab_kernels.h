// sph_slab_kernels.h -- device side of the multi-GPU x-slab decomposition:
// particle migration, ghost-layer packing, ordered edge lists and per-sweep ghost field refresh.
//
// Particle record on the wire: 32 bytes = (x, y, z, vx) (vy, vz, warm_start_k, id bits).
// Ghost particles carry id' = ~id (negative) in the id array; the sort orders by the true id, so the
// order inside a cell -- and with it every neighbour sum -- is the same on every decomposition.
#pragma once
#include "sph_device.h"
#include "sph_kernels.h"

namespace sph {

__device__ __forceinline__ int id_key(int id) { return id < 0 ? ~id : id; }

struct SlabGeom {
    int x_lo, x_hi;        // owned cell columns [x_lo, x_hi)
    int has_left, has_right;
    int layers;            // ghost cell columns per side (1, or 2: see the two-column protocol in sph_mi355x.hip)
    int far_left, far_right;   // the neighbours' far cuts: columns [far_left, x_lo) belong to the left neighbour, [x_hi, far_right) to the right one
};

__device__ __forceinline__ void write_record(float4 *__restrict__ buf, int slot, float4 p, float4 v, float warm, int id)
{
    buf[2 * (size_t)slot] = make_float4(p.x, p.y, p.z, v.x);
    buf[2 * (size_t)slot + 1] = make_float4(v.y, v.z, warm, __int_as_float(id));
}

// Re-balancing: owned particles per cell column.  Sorted order makes a block's particles share a few columns, so
// the block counts in LDS first and flushes only the touched columns.
__global__ __launch_bounds__(kBlock) void k_column_histogram(Consts c, const float4 *__restrict__ P, const int *__restrict__ id,
                                                              int *__restrict__ hist)
{
    constexpr int kLocal = 2048;
    __shared__ int local[kLocal];
    const bool use_lds = c.gx <= kLocal;
    if (use_lds) {
        for (int x = threadIdx.x; x < c.gx; x += kBlock) local[x] = 0;
        __syncthreads();
    }
    int s = blockIdx.x * kBlock + threadIdx.x;
    if (s < c.n && id[s] >= 0) {
        int cx = (int)floorf(P[s].x / c.hcell);
        cx = cx < 0 ? 0 : (cx >= c.gx ? c.gx - 1 : cx);
        atomicAdd(use_lds ? &local[cx] : &hist[cx], 1);
    }
    if (use_lds) {
        __syncthreads();
        for (int x = threadIdx.x; x < c.gx; x += kBlock)
            if (local[x]) atomicAdd(&hist[x], local[x]);
    }
}

// One atomicAdd per wave instead of one per particle on the same counter (same-address atomics serialise at ~12 ns each:
// 100 k edge particles would cost over a millisecond): the lanes that want a slot are counted with a ballot, the first of them
// reserves the block of slots, every lane takes base + its rank among the wanting lanes.
__device__ __forceinline__ int wave_alloc(int *__restrict__ counter, bool want)
{
    const unsigned long long m = __ballot(want);
    if (m == 0) return -1;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(m));
    base = __shfl(base, leader, 64);
    return want ? base + __popcll(m & ((1ull << lane) - 1ull)) : -1;
}

// lanes that want to be counted, one atomicAdd per wave
__device__ __forceinline__ void wave_count(int *__restrict__ counter, bool want)
{
    const unsigned long long m = __ballot(want);
    if (m != 0 && (int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicAdd(counter, __popcll(m));
}

// The particle exchange at the start of a step.  Device counters (kSlabCounters ints):
//   [0] records for the left neighbour, [1] for the right one, [2] slots that die here,
//   [3 + 4 side + l]      ghost copies of my send column l + 1 on that side (side 0 = left, 1 = right; l = 0 is the column next to the cut),
//   [3 + 4 side + 2 + l]  leavers to that side that I KEEP as ghosts of my ghost column l + 1 there.
// mode bits: kSlabMigrate -- last step's ghosts die, owned particles that left [x_lo, x_hi) are packed for the neighbour (id >= 0 on the wire);
//            kSlabGhosts  -- owned particles of the `layers` columns next to a cut are copied to that neighbour (~id on the wire);
//            kSlabKeep    -- a leaver that lands in one of my ghost columns stays resident as a ghost (id := ~id) instead of dying: the
//                            new owner would send it straight back.
// All three together are ONE message per neighbour (the merged exchange of ordinary steps: a particle moves a fraction of a cell per step,
// and slabs are at least layers + 1 columns wide, so nothing a neighbour sends me can land in the columns I copy to the OTHER neighbour).
// A step that moved the cuts runs two rounds instead -- kSlabMigrate alone, then kSlabGhosts over what arrived -- because a re-cut can
// move whole columns across a slab.  A leaver of a merged step that lands beyond the neighbour's own slab breaks the assumption: it
// raises overflow bit 2 and the step fails loudly on every rank (check_overflow_all).
constexpr int kSlabCounters = 16;
constexpr int kSlabCounted = 11;          // counters a classification fills
enum { kSlabMigrate = 1, kSlabGhosts = 2, kSlabKeep = 4 };
// What becomes of one resident slot.  which[k] = 1 if the slot counts towards counter k.
struct SlabClass {
    float4 p;
    int pid;
    bool go_left, go_right, dies, to_left, to_right;
    int gl, gr, kl, kr;       // ghost copy for the left / right neighbour, kept as my ghost on the left / right: column 1 or 2 (0: no)
    bool beyond;              // merged exchange only: a leaver that lands outside what the neighbour can take (overflow bit 2)
};
__device__ __forceinline__ SlabClass classify_slot(const Consts &c, const SlabGeom &g, int mode, int s, const float4 *__restrict__ P, const int *__restrict__ id,
                                                   const int *__restrict__ dead)
{
    SlabClass k;
    const bool in = s < c.n;
    const bool migrate = (mode & kSlabMigrate) != 0, ghosts = (mode & kSlabGhosts) != 0, keep = (mode & kSlabKeep) != 0;
    k.pid = in ? id[s] : 0;
    const bool skip = in && !migrate && dead[s] != 0;        // second round: slots the first round killed
    k.p = make_float4(0.f, 0.f, 0.f, 0.f);
    k.go_left = k.go_right = k.dies = k.beyond = false;
    k.gl = k.gr = k.kl = k.kr = 0;
    if (in && !skip) {
        if (k.pid < 0) {
            k.dies = migrate;                                // last step's ghost
        } else {
            k.p = P[s];
            // "is an edge particle" must be decided by the cell the sort bins the particle into (the ordered edge lists enumerate CELL LISTS):
            // the reference's 1-D index, ParticleSystem.py:486-494; a particle outside the grid sits in no cell of a slab handle (strict_cells)
            int cx, cy, cz;
            const int cid = cell_id_of(c, k.p.x, k.p.y, k.p.z, cx, cy, cz);
            const int col = cid < c.C ? cid % c.gx : -1;
            if (migrate) {
                k.go_left = g.has_left && cx < g.x_lo;
                k.go_right = !k.go_left && g.has_right && cx >= g.x_hi;
            }
            if (k.go_left || k.go_right) {
                if (keep && col >= 0) {
                    if (k.go_left && col >= g.x_lo - g.layers && col < g.x_lo) k.kl = g.x_lo - col;
                    if (k.go_right && col >= g.x_hi && col < g.x_hi + g.layers) k.kr = col - g.x_hi + 1;
                    // merged exchange only: a leaver must land inside the neighbour's slab, clear of the columns it copies to ITS other neighbour
                    k.beyond = (k.go_left && cx < g.far_left + g.layers && g.far_left > 0) || (k.go_right && cx >= g.far_right - g.layers && g.far_right < c.gx);
                }
                k.dies = k.kl == 0 && k.kr == 0;
            } else if (ghosts && col >= 0) {
                if (g.has_left && col >= g.x_lo && col < g.x_lo + g.layers) k.gl = col - g.x_lo + 1;
                if (g.has_right && col < g.x_hi && col >= g.x_hi - g.layers) k.gr = g.x_hi - col;
            }
        }
    }
    k.to_left = k.go_left || k.gl != 0;
    k.to_right = k.go_right || k.gr != 0;
    return k;
}
__device__ __forceinline__ bool slab_class_counts(const SlabClass &k, int q)
{
    switch (q) {
    case 0: return k.to_left;
    case 1: return k.to_right;
    case 2: return k.dies;
    case 3: return k.gl == 1;
    case 4: return k.gl == 2;
    case 5: return k.kl == 1;
    case 6: return k.kl == 2;
    case 7: return k.gr == 1;
    case 8: return k.gr == 2;
    case 9: return k.kr == 1;
    default: return k.kr == 2;
    }
}
// The classification runs in three launches WITHOUT same-address atomics.  (One atomicAdd per wave and counter on eleven shared words was
// 607 us per step on a rank of config 4 at 8 slabs -- ~50 k adds that the memory side serialises at ~12 ns each, DESIGN.md section 6 -- where
// the passes below take ~25 us together; the record order in the message is now the slot order, i.e. deterministic.)
//   k_classify_count   per workgroup: how many of its slots count towards each of the eleven counters     -> blk_cnt[q * nblk + blk]
//   k_classify_scan    one workgroup per counter: exclusive scan over the workgroups (offsets of counters 0 and 1 = the records' places
//                      in the two messages) and the totals                                                -> blk_cnt in place, counters[q]
//   k_classify_write   the same classification again; records written at offset of the workgroup + rank inside it; dead[] / id[] updated
__global__ __launch_bounds__(kBlock) void k_classify_count(Consts c, SlabGeom g, int mode, const float4 *__restrict__ P, const int *__restrict__ id,
                                                           const int *__restrict__ dead, int nblk, int *__restrict__ blk_cnt)
{
    __shared__ int s_cnt[kBlock / 64][kSlabCounted];
    const int s = blockIdx.x * kBlock + threadIdx.x;
    const SlabClass k = classify_slot(c, g, mode, s, P, id, dead);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < kSlabCounted; ++q) {
        const int n = __popcll(__ballot(slab_class_counts(k, q)));
        if (lane == 0) s_cnt[w][q] = n;
    }
    __syncthreads();
    if (threadIdx.x < kSlabCounted) {
        int t = 0;
        for (int ww = 0; ww < kBlock / 64; ++ww) t += s_cnt[ww][threadIdx.x];
        blk_cnt[(size_t)threadIdx.x * nblk + blockIdx.x] = t;
    }
}
constexpr int kScanBlock = 1024;
// wire[] (optional; the native transport's device-side count exchange): what goes to the left neighbour at wire[0..5), to the right one at
// wire[8..13) -- (records, ghost copies of column 1 / 2, leavers kept as ghosts of column 1 / 2), the five ints slab_exchange_particles trades
__global__ __launch_bounds__(kScanBlock) void k_classify_scan(int nblk, int *__restrict__ blk_cnt, int *__restrict__ counters, int *__restrict__ wire)
{
    __shared__ int s_w[kScanBlock / 64];
    int *a = blk_cnt + (size_t)blockIdx.x * nblk;
    const int per = (nblk + kScanBlock - 1) / kScanBlock;
    const int lo = min((int)threadIdx.x * per, nblk), hi = min(lo + per, nblk);
    int t = 0;
    for (int i = lo; i < hi; ++i) t += a[i];
    const int inc = wave_inclusive_scan(t);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    int before = inc - t, total = 0;
    for (int q = 0; q < kScanBlock / 64; ++q) { if (q < w) before += s_w[q]; total += s_w[q]; }
    if (blockIdx.x < 2)                               // the two allocating counters: exclusive offsets in place
        for (int i = lo; i < hi; ++i) { const int v = a[i]; a[i] = before; before += v; }
    if (threadIdx.x == 0) {
        counters[blockIdx.x] = total;
        const int q = (int)blockIdx.x;
        const int slot = q == 0 ? 0 : q == 1 ? 8 : q == 2 ? -1 : q <= 6 ? q - 2 : 8 + q - 6;       // 3..6 -> 1..4, 7..10 -> 9..12
        if (wire && slot >= 0) wire[slot] = total;
    }
    if (wire && blockIdx.x == 0 && threadIdx.x < 16) wire[16 + threadIdx.x] = 0;        // what an absent neighbour "sent" (the receive half of the count exchange: wire[16..32))
}
__global__ __launch_bounds__(kBlock) void k_classify_write(Consts c, SlabGeom g, int mode, const float4 *__restrict__ P, const float4 *__restrict__ V,
                                                           const float *__restrict__ warm, int *__restrict__ id, int *__restrict__ dead,
                                                           float4 *__restrict__ send_left, float4 *__restrict__ send_right, int cap_records,
                                                           int nblk, const int *__restrict__ blk_off, DevScalars *__restrict__ ds)
{
    __shared__ int s_cnt[kBlock / 64][2];
    const int s = blockIdx.x * kBlock + threadIdx.x;
    const bool in = s < c.n;
    const SlabClass k = classify_slot(c, g, mode, s, P, id, dead);
    if (k.beyond) atomicOr(&ds->overflow, 4);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long ml = __ballot(k.to_left), mr = __ballot(k.to_right);
    if (lane == 0) { s_cnt[w][0] = __popcll(ml); s_cnt[w][1] = __popcll(mr); }
    __syncthreads();
    int sl = blk_off[blockIdx.x] + __popcll(ml & ((1ull << lane) - 1ull)), sr = blk_off[(size_t)nblk + blockIdx.x] + __popcll(mr & ((1ull << lane) - 1ull));
    for (int q = 0; q < w; ++q) { sl += s_cnt[q][0]; sr += s_cnt[q][1]; }
    if (k.to_left && sl < cap_records) write_record(send_left, sl, k.p, V[s], warm ? warm[s] : 0.f, k.go_left ? k.pid : ~k.pid);
    if (k.to_right && sr < cap_records) write_record(send_right, sr, k.p, V[s], warm ? warm[s] : 0.f, k.go_right ? k.pid : ~k.pid);
    if (in && (mode & kSlabMigrate)) {
        dead[s] = k.dies ? 1 : 0;
        if (k.kl != 0 || k.kr != 0) id[s] = ~k.pid;
    }
}

// received records are appended behind the resident particles; the sign of the id on the wire says owned (migrant) or ghost
__global__ __launch_bounds__(kBlock) void k_append_records(const float4 *__restrict__ buf, int count, int base,
                                                           float4 *__restrict__ P, float4 *__restrict__ V, float *__restrict__ warm,
                                                           int *__restrict__ id, int *__restrict__ dead)
{
    int r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= count) return;
    float4 a = buf[2 * (size_t)r], b = buf[2 * (size_t)r + 1];
    P[base + r] = make_float4(a.x, a.y, a.z, 0.f);
    V[base + r] = make_float4(a.w, b.x, b.y, 0.f);
    if (warm) warm[base + r] = b.z;
    id[base + r] = __float_as_int(b.w);
    dead[base + r] = 0;
}

// Ordered list of the sorted slots that live in cell column `layer_cx`: cells ascending (y, then z), slots
// ascending inside a cell.  The sender's list for its edge column and the receiver's list for the matching ghost
// column enumerate the same particles in the same order, because both sides sort by (cell, true id).
// All (up to eight) edge columns of a step in one launch each: one workgroup per column scans the column's cells -- counts first (coalesced
// over the cells), then each thread the offsets of its run of consecutive cells.  (One single-workgroup launch per column, 256 cells per
// trip, was 56 us x 8 per step on config 4's 153 x 113 cell cut.)
struct LayerJobs { int n; int col[8]; int *off[8]; int *list[8]; };
// (round 6: up to kLayerDeep * kScanBlock cells per column -- config 4 has 153 x 113 = 17 289 -- every thread keeps the counts of cells t, t + 1024, ...
// in registers, all their loads in flight at once, and the offsets come from one block-wide scan per round of 1024 cells: loads and stores are
// coalesced.  The first form gave every thread 17 CONSECUTIVE cells and read its counts back from memory: 33 us; 21 us now, of which the launch
// itself is 5.  Larger columns keep that form.)
constexpr int kLayerDeep = 18;
__global__ __launch_bounds__(kScanBlock) void k_layer_offsets(Consts c, const int *__restrict__ cell_start, LayerJobs jobs, int generic)
{
    __shared__ int s_w[kScanBlock / 64];
    __shared__ int s_tot[kLayerDeep][kScanBlock / 64];
    const int layer_cx = jobs.col[blockIdx.x];
    int *__restrict__ off = jobs.off[blockIdx.x];
    const int ncol = c.gy * c.gz;
    const bool valid = layer_cx >= 0 && layer_cx < c.gx;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (ncol <= kLayerDeep * kScanBlock && !generic) {         // (generic: SPH_LAYER_GENERIC=1, so that tests reach the form larger columns take)
        int slot[kLayerDeep], v[kLayerDeep], inc[kLayerDeep];
#pragma unroll
        for (int j = 0; j < kLayerDeep; ++j) {
            const int k = (int)threadIdx.x + j * kScanBlock;
            slot[j] = -1;
            if (valid && k < ncol) {
                const int y = k / c.gz, z = k - y * c.gz;
                slot[j] = cell_slot_xyz(c, layer_cx, y, z, layer_cx + y * c.sy + z * c.sz);
            }
        }
#pragma unroll
        for (int j = 0; j < kLayerDeep; ++j) v[j] = slot[j] < 0 ? 0 : cell_start[slot[j] + 1] - cell_start[slot[j]];
#pragma unroll
        for (int j = 0; j < kLayerDeep; ++j) {
            inc[j] = wave_inclusive_scan(v[j]);
            if (lane == 63) s_tot[j][w] = inc[j];
        }
        __syncthreads();
        int carry = 0;                                     // cells of the rounds before j, then of the waves before w in round j
#pragma unroll
        for (int j = 0; j < kLayerDeep; ++j) {
            int before = carry, total = 0;
            for (int q = 0; q < kScanBlock / 64; ++q) { const int t = s_tot[j][q]; if (q < w) before += t; total += t; }
            const int k = (int)threadIdx.x + j * kScanBlock;
            if (k < ncol) off[k] = before + inc[j] - v[j];
            carry += total;
        }
        if (threadIdx.x == 0) off[ncol] = carry;
        return;
    }
    for (int k = threadIdx.x; k < ncol; k += kScanBlock) {
        int v = 0;
        if (valid) {
            const int y = k / c.gz, z = k - y * c.gz;
            const int slot = cell_slot_xyz(c, layer_cx, y, z, layer_cx + y * c.sy + z * c.sz);
            v = slot < 0 ? 0 : cell_start[slot + 1] - cell_start[slot];
        }
        off[k] = v;
    }
    __syncthreads();
    const int per = (ncol + kScanBlock - 1) / kScanBlock;
    const int lo = min((int)threadIdx.x * per, ncol), hi = min(lo + per, ncol);
    int t = 0;
    for (int i = lo; i < hi; ++i) t += off[i];
    const int inc = wave_inclusive_scan(t);
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    int before = inc - t, total = 0;
    for (int q = 0; q < kScanBlock / 64; ++q) { if (q < w) before += s_w[q]; total += s_w[q]; }
    for (int i = lo; i < hi; ++i) { const int v = off[i]; off[i] = before; before += v; }
    if (threadIdx.x == 0) off[ncol] = total;
}

// (a side's two columns share one list: the second column's entries start at list + count of the first)
__global__ __launch_bounds__(kBlock) void k_layer_list(Consts c, const int *__restrict__ cell_start, LayerJobs jobs)
{
    const int layer_cx = jobs.col[blockIdx.y];
    const int *__restrict__ off = jobs.off[blockIdx.y];
    int *__restrict__ list = jobs.list[blockIdx.y];
    int k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= c.gy * c.gz || layer_cx < 0 || layer_cx >= c.gx) return;
    int y = k / c.gz, z = k - y * c.gz;
    const int slot = cell_slot_xyz(c, layer_cx, y, z, layer_cx + y * c.sy + z * c.sz);
    if (slot < 0) return;
    int a = cell_start[slot], b = cell_start[slot + 1];
    int o = off[k];
    for (int s = a; s < b; ++s) list[o + (s - a)] = s;
}

// ghost field refresh: mode 0 = the per-sweep scalar k / rho (1 float), 1 = V.xyz (3 floats), 2 = (scalar, V.w) (2 floats; rho[] := V.w, dfsph),
// 3 = (P.w, V.w) with rho[] := P.w (pcisph / iisph: P.w carries rho).  The scalar is S[s] on kr_split handles (S = krho), else P[s].w.
// One launch serves both sides: threads [0, count_a) work on side a (left), threads [count_a, count_a + count_b) on side b (right).
__global__ __launch_bounds__(kBlock) void k_pack_field(const int *__restrict__ list_a, int count_a, float *__restrict__ out_a,
                                                       const int *__restrict__ list_b, int count_b, float *__restrict__ out_b, int mode,
                                                       const float4 *__restrict__ P, const float4 *__restrict__ V, const float *__restrict__ S)
{
    int r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= count_a + count_b) return;
    const bool b = r >= count_a;
    if (b) r -= count_a;
    const int s = (b ? list_b : list_a)[r];
    float *out = b ? out_b : out_a;
    if (mode == 0) out[r] = S ? S[s] : P[s].w;
    else if (mode == 1) { float4 v = V[s]; out[3 * (size_t)r] = v.x; out[3 * (size_t)r + 1] = v.y; out[3 * (size_t)r + 2] = v.z; }
    else { out[2 * (size_t)r] = S ? S[s] : P[s].w; out[2 * (size_t)r + 1] = V[s].w; }
}

__global__ __launch_bounds__(kBlock) void k_unpack_field(const int *__restrict__ list_a, int count_a, const float *__restrict__ in_a,
                                                         const int *__restrict__ list_b, int count_b, const float *__restrict__ in_b, int mode,
                                                         float4 *__restrict__ P, float4 *__restrict__ V, float *__restrict__ rho, float *__restrict__ S)
{
    int r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= count_a + count_b) return;
    const bool b = r >= count_a;
    if (b) r -= count_a;
    const int s = (b ? list_b : list_a)[r];
    const float *in = b ? in_b : in_a;
    if (mode == 0) { if (S) S[s] = in[r]; else P[s].w = in[r]; }
    else if (mode == 1) { V[s].x = in[3 * (size_t)r]; V[s].y = in[3 * (size_t)r + 1]; V[s].z = in[3 * (size_t)r + 2]; }
    else {
        const float aa = in[2 * (size_t)r];
        const float bb = in[2 * (size_t)r + 1];
        if (S) S[s] = aa; else P[s].w = aa;
        V[s].w = bb;
        if (rho) rho[s] = mode == 3 ? aa : bb;    // dfsph: V.w carries rho; the correction sweeps rewrite it from rho[]
    }
}

// The residual refresh of the two-column protocol (dfsph): one float per ghost.  The first n1 entries of a side's list are its INNER column,
// whose particles run the correction sweeps themselves and need the residual value (rho_derivative / rho_adv: `val`) -- their k / rho is
// re-derived on arrival with the very expression of k_residual, from the alpha and rho the ghost computed itself in D1; the entries behind them
// are the OUTER column, read only as neighbours: they get the owner's k / rho.
struct ResidLists { const int *list_a; int count_a, n1_a; float *buf_a; const int *list_b; int count_b, n1_b; float *buf_b; };
__device__ __forceinline__ void pack_resid_entry(const ResidLists &L, int r, const float *__restrict__ val, const float4 *__restrict__ P, const float *__restrict__ S)
{
    if (r >= L.count_a + L.count_b) return;
    const bool b = r >= L.count_a;
    if (b) r -= L.count_a;
    const int s = (b ? L.list_b : L.list_a)[r];
    (b ? L.buf_b : L.buf_a)[r] = r < (b ? L.n1_b : L.n1_a) ? val[s] : (S ? S[s] : P[s].w);
}
// df (the density loop on a DensFlow handle): the residual sweep pushed "run the correction" to the tiles that stage a k / rho != 0 of its OWN
// particles; a ghost's k / rho arrives here, so this kernel pushes for it -- through the row of the ghost's tile, which k_build_nl fills for
// particles without lists too
__device__ __forceinline__ void unpack_resid_entry(const Consts &c, const ResidLists &L, int r, int dens, const float *__restrict__ alpha, const float *__restrict__ rho,
                                                   const DevScalars *__restrict__ ds, float *__restrict__ val, float4 *__restrict__ P, float *__restrict__ S,
                                                   const DensFlow &df = kNoFlow)
{
    if (r >= L.count_a + L.count_b) return;
    const bool b = r >= L.count_a;
    if (b) r -= L.count_a;
    const int s = (b ? L.list_b : L.list_a)[r];
    const float x = (b ? L.buf_b : L.buf_a)[r];
    float kr = x;
    if (r < (b ? L.n1_b : L.n1_a)) {
        val[s] = x;
        if (dens) kr = ((x - c.rho0) * alpha[s] / ds->dt2) / rho[s];           // dfsph_solver.py:199,203 as k_residual writes it
        else kr = (x * alpha[s] / ds->dt) / rho[s];                            // :363,367
    }
    if (S) S[s] = kr; else P[s].w = kr;
    if (dens && df.nbr && kr != 0.f) {                       // (a NaN counts; ~1 % of the ghosts)
        const int *row = df.nbr + (size_t)(s / kBlock) * kNbrStride;
        const int hdr = row[0];
        if (hdr < 0 || (hdr & kNbrOdd) != 0) df.bcast[df.bc_out] = df.stamp_out;
        for (int q = 1; hdr >= 0 && q <= (hdr & (kNbrStride - 1)); ++q) df.need_out[row[q]] = df.stamp_out;
    }
}
__global__ __launch_bounds__(kBlock) void k_pack_resid(const int *__restrict__ list_a, int count_a, int n1_a, float *__restrict__ out_a,
                                                       const int *__restrict__ list_b, int count_b, int n1_b, float *__restrict__ out_b,
                                                       const float *__restrict__ val, const float4 *__restrict__ P, const float *__restrict__ S)
{
    pack_resid_entry(ResidLists{list_a, count_a, n1_a, out_a, list_b, count_b, n1_b, out_b}, blockIdx.x * kBlock + threadIdx.x, val, P, S);
}
// The in-order protocol (slab_overlap = 1 / sph_slab_set_overlap(h, 0)) runs a solver iteration's small launches back to back on one stream: pack,
// transfer, unpack, reduce, all-reduce, decide -- ~5 us each at 1.2 M particles per rank, a sixth of the iteration.  Two of them ride along: the
// slab's (sum, count) is reduced by the LAST workgroup of the pack launch (it only needs the residual sweep, like the pack), and the loop decision
// is taken by the last workgroup of the unpack launch (it only needs the all-reduce, which is enqueued in front of it).  Same trees, same bits.
__global__ __launch_bounds__(kFinBlock) void k_pack_resid_reduce(ResidLists L, const float *__restrict__ val, const float4 *__restrict__ P, const float *__restrict__ S,
                                                                 const double *__restrict__ psum, const int *__restrict__ pcnt, int nblocks, DevScalars *__restrict__ ds,
                                                                 int mode, double *__restrict__ red, int group, int nparts)
{
    if (blockIdx.x == gridDim.x - 1) { finalize_mean_block(psum, pcnt, nblocks, ds, mode, FINP_REDUCE, red, group, nparts, -1); return; }
    pack_resid_entry(L, blockIdx.x * kFinBlock + threadIdx.x, val, P, S);
}
__global__ __launch_bounds__(kFinBlock) void k_unpack_resid_decide(Consts c, ResidLists L, int dens, const float *__restrict__ alpha, const float *__restrict__ rho,
                                                                   float *__restrict__ val, float4 *__restrict__ P, float *__restrict__ S,
                                                                   const double *__restrict__ psum, const int *__restrict__ pcnt, int nblocks, DevScalars *__restrict__ ds,
                                                                   int mode, double *__restrict__ red, int group, int nparts, int gather_n, DensFlow df = kNoFlow)
{
    if (blockIdx.x == gridDim.x - 1) { finalize_mean_block(psum, pcnt, nblocks, ds, mode, FINP_DECIDE, red, group, nparts, -1, gather_n); return; }
    unpack_resid_entry(c, L, blockIdx.x * kFinBlock + threadIdx.x, dens, alpha, rho, ds, val, P, S, df);
}
__global__ __launch_bounds__(kBlock) void k_unpack_resid(Consts c, const int *__restrict__ list_a, int count_a, int n1_a, const float *__restrict__ in_a,
                                                         const int *__restrict__ list_b, int count_b, int n1_b, const float *__restrict__ in_b,
                                                         int dens, const float *__restrict__ alpha, const float *__restrict__ rho, const DevScalars *__restrict__ ds,
                                                         float *__restrict__ val, float4 *__restrict__ P, float *__restrict__ S, DensFlow df = kNoFlow)
{
    unpack_resid_entry(c, ResidLists{list_a, count_a, n1_a, const_cast<float *>(in_a), list_b, count_b, n1_b, const_cast<float *>(in_b)}, blockIdx.x * kBlock + threadIdx.x,
                       dens, alpha, rho, ds, val, P, S, df);
}

// Edge / interior split of the residual sweeps (slab handles, dfsph): a tile is an EDGE tile if one of its particles lies in a column that
// takes part in the halo of this step -- a ghost column or a column copied to a neighbour.  tile_order lists the edge tiles first (ascending),
// then the interior ones (ascending); tile_order[ntiles] = number of edge tiles.  The edge launch of a sweep works on [0, n_edge), its results
// are packed and sent while the interior launch works on the rest.
__global__ __launch_bounds__(kBlock) void k_tile_flags(Consts c, SlabGeom g, const float4 *__restrict__ P, int *__restrict__ flag)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    int e = 0;
    if (i < c.n) {
        const int cx = (int)floorf(P[i].x / c.hcell);
        e = (g.has_left && cx < g.x_lo + g.layers) || (g.has_right && cx >= g.x_hi - g.layers);
    }
    e = __syncthreads_or(e);
    if (threadIdx.x == 0) flag[blockIdx.x] = e ? 1 : 0;
}
// order[] = the tiles of class 2 first, then class 1 (any other nonzero flag), then the unflagged ones, each in index order; order[ntiles] = number of
// flagged tiles, order[ntiles + 1] = number of class-2 tiles.  (Slab edge tiles and the rigid shell flag with 1; the density loop's residual sweep
// with 2 = worked, 1 = left after the per-particle check: TilePhase.hot.)  ONE pass of one 1024-thread workgroup: every thread takes a run of
// consecutive tiles, the runs' counts are scanned once (wave scans + 16 wave sums), then every thread places its tiles -- one barrier (the chunked
// two-pass form this replaces took 17 us at 3907 tiles; this one 5, and it is launched every step for the density loop's working tiles).
__global__ __launch_bounds__(1024) void k_tile_order(const int *__restrict__ flag, int ntiles, int *__restrict__ order)
{
    __shared__ int wsum[2][16];
    const int per = (ntiles + 1023) / 1024, first = (int)threadIdx.x * per, last = min(first + per, ntiles);
    int mine2 = 0, mine1 = 0;
    for (int t = first; t < last; ++t) { const int f = flag[t]; mine2 += f == 2 ? 1 : 0; mine1 += (f != 0 && f != 2) ? 1 : 0; }
    const int inc2 = wave_inclusive_scan(mine2), inc1 = wave_inclusive_scan(mine1);
    if ((threadIdx.x & 63) == 63) { wsum[0][threadIdx.x >> 6] = inc2; wsum[1][threadIdx.x >> 6] = inc1; }
    __syncthreads();
    int before2 = inc2 - mine2, before1 = inc1 - mine1, total2 = 0, total1 = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < (int)(threadIdx.x >> 6)) { before2 += wsum[0][w]; before1 += wsum[1][w]; }
        total2 += wsum[0][w]; total1 += wsum[1][w];
    }
    for (int t = first; t < last; ++t) {
        const int f = flag[t];
        const int cls = f == 2 ? 2 : (f != 0 ? 1 : 0);
        order[cls == 2 ? before2 : cls == 1 ? total2 + before1 : total2 + total1 + (t - before2 - before1)] = t;
        before2 += cls == 2 ? 1 : 0; before1 += cls == 1 ? 1 : 0;
    }
    if (threadIdx.x == 0) { order[ntiles] = total2 + total1; order[ntiles + 1] = total2; }
}

// Rigid body on slab handles.  The reference's quirks read FLUID arrays with a rigid particle's local index (get_neighbour_count measures to
// fluid_particles.pos[particle_j.index], ParticleSystem.py:440-442; viscosity reads rho[particle_j.index], solver_base.py:198-199): positions and densities
// of the fluid particles with original id < Nr, wherever they are.  Every rank contributes the ones it OWNS to a zeroed array of doubles, the arrays are
// summed over the slabs (one owner per id: x + 0 + ... = x exactly) and spread into pos_orig / rho_orig.
__global__ __launch_bounds__(kBlock) void k_collect_by_id(int n, const int *__restrict__ id, const float4 *__restrict__ A4, const float *__restrict__ A1, int nr,
                                                          double *__restrict__ out)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int o = id[i];
    if (o < 0 || o >= nr) return;                            // ghosts (~id) and particles whose id no rigid particle can name
    if (A4) { const float4 a = A4[i]; out[4 * (size_t)o] = a.x; out[4 * (size_t)o + 1] = a.y; out[4 * (size_t)o + 2] = a.z; out[4 * (size_t)o + 3] = a.w; }
    else out[o] = (double)A1[i];
}
__global__ __launch_bounds__(kBlock) void k_spread_by_id(int nr, const double *__restrict__ in, float4 *__restrict__ D4, float *__restrict__ D1)
{
    const int o = blockIdx.x * kBlock + threadIdx.x;
    if (o >= nr) return;
    if (D4) D4[o] = make_float4((float)in[4 * (size_t)o], (float)in[4 * (size_t)o + 1], (float)in[4 * (size_t)o + 2], (float)in[4 * (size_t)o + 3]);
    else D1[o] = (float)in[o];
}
__global__ __launch_bounds__(kBlock) void k_floats_to_doubles(int n, const float *__restrict__ in, double *__restrict__ out)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] = (double)in[i];
}
__global__ __launch_bounds__(kBlock) void k_doubles_to_floats(int n, const double *__restrict__ in, float *__restrict__ out)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}

__global__ __launch_bounds__(kBlock) void k_unsort_ids(int n, const int *__restrict__ id, int *__restrict__ out)
{
    int s = blockIdx.x * kBlock + threadIdx.x;
    if (s < n) out[s] = id[s];
}
__global__ __launch_bounds__(kBlock) void k_copy_vec_local(int n, const float4 *__restrict__ src, float *__restrict__ dst)
{
    int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n) return;
    float4 v = src[s];
    dst[3 * (size_t)s] = v.x; dst[3 * (size_t)s + 1] = v.y; dst[3 * (size_t)s + 2] = v.z;
}

}  // namespace sph
