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
        int cx = (int)floorf(P[s].x / c.h);
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

// (1) previous ghosts die; owned particles that left the slab are packed for the neighbour and die here.
__global__ __launch_bounds__(kBlock) void k_classify_migrate(Consts c, SlabGeom g, const float4 *__restrict__ P, const float4 *__restrict__ V,
                                                             const float *__restrict__ warm, const int *__restrict__ id,
                                                             int *__restrict__ dead, float4 *__restrict__ send_left,
                                                             float4 *__restrict__ send_right, int cap_records, int *__restrict__ counters)
{
    const int s = blockIdx.x * kBlock + threadIdx.x;
    const bool in = s < c.n;
    const int pid = in ? id[s] : 0;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    bool go_left = false, go_right = false;
    if (in && pid >= 0) {
        p = P[s];
        const int cx = (int)floorf(p.x / c.h);
        go_left = g.has_left && cx < g.x_lo;
        go_right = !go_left && g.has_right && cx >= g.x_hi;
    }
    const bool dies = in && (pid < 0 || go_left || go_right);       // last step's ghosts, and owned particles that left the slab
    const int sl = wave_alloc(&counters[0], go_left);
    const int sr = wave_alloc(&counters[1], go_right);
    (void)wave_alloc(&counters[2], dies);
    if (go_left && sl < cap_records) write_record(send_left, sl, p, V[s], warm ? warm[s] : 0.f, pid);
    if (go_right && sr < cap_records) write_record(send_right, sr, p, V[s], warm ? warm[s] : 0.f, pid);
    if (in) dead[s] = dies ? 1 : 0;
}

// (2)/(4) received records are appended behind the resident particles
__global__ __launch_bounds__(kBlock) void k_append_records(const float4 *__restrict__ buf, int count, int base, int as_ghost,
                                                           float4 *__restrict__ P, float4 *__restrict__ V, float *__restrict__ warm,
                                                           int *__restrict__ id, int *__restrict__ dead)
{
    int r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= count) return;
    float4 a = buf[2 * (size_t)r], b = buf[2 * (size_t)r + 1];
    int pid = __float_as_int(b.w);
    P[base + r] = make_float4(a.x, a.y, a.z, 0.f);
    V[base + r] = make_float4(a.w, b.x, b.y, 0.f);
    if (warm) warm[base + r] = b.z;
    id[base + r] = as_ghost ? ~pid : pid;
    dead[base + r] = 0;
}

// (3) owned particles in the two edge cell columns are copied to the neighbours as ghosts
__global__ __launch_bounds__(kBlock) void k_classify_ghost(Consts c, SlabGeom g, const float4 *__restrict__ P, const float4 *__restrict__ V,
                                                           const float *__restrict__ warm, const int *__restrict__ id,
                                                           const int *__restrict__ dead, float4 *__restrict__ send_left,
                                                           float4 *__restrict__ send_right, int cap_records, int *__restrict__ counters)
{
    const int s = blockIdx.x * kBlock + threadIdx.x;
    bool to_left = false, to_right = false;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    int pid = 0;
    if (s < c.n && !dead[s]) {
        pid = id[s];
        if (pid >= 0) {
            p = P[s];
            // The ordered edge lists (k_layer_list) enumerate the CELL LISTS of the edge columns, so "is an edge particle" must be
            // decided by the cell the sort bins the particle into -- the reference's 1-D index (ParticleSystem.py:486-494, guarded only
            // by 0 <= id <= C at :393): a particle that slipped through a wall in y or z still has a valid 1-D index (it wraps into a
            // neighbouring row of the same column) and is a member of that cell; one whose index is out of range sits in no cell.
            int cx, cy, cz;
            const int cid = cell_id_of(c, p.x, p.y, p.z, cx, cy, cz);
            const int col = cid < c.C ? cid % c.gx : -1;
            to_left = g.has_left && col == g.x_lo;
            to_right = g.has_right && col == g.x_hi - 1;
        }
    }
    const int sl = wave_alloc(&counters[0], to_left);
    const int sr = wave_alloc(&counters[1], to_right);
    if (to_left && sl < cap_records) write_record(send_left, sl, p, V[s], warm ? warm[s] : 0.f, pid);
    if (to_right && sr < cap_records) write_record(send_right, sr, p, V[s], warm ? warm[s] : 0.f, pid);
}

// Ordered list of the sorted slots that live in cell column `layer_cx`: cells ascending (y, then z), slots
// ascending inside a cell.  The sender's list for its edge column and the receiver's list for the matching ghost
// column enumerate the same particles in the same order, because both sides sort by (cell, true id).
__global__ __launch_bounds__(kBlock) void k_layer_offsets(Consts c, const int *__restrict__ cell_start, int layer_cx, int *__restrict__ off)
{
    // single block: exclusive scan of the per-cell counts of the column, chunks of 256 with a carry
    __shared__ int wsum[kBlock / 64];
    __shared__ int carry_s;
    const int ncol = c.gy * c.gz;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < ncol; base += kBlock) {
        int k = base + threadIdx.x;
        int v = 0;
        if (k < ncol && layer_cx >= 0 && layer_cx < c.gx) {
            int y = k / c.gz, z = k - y * c.gz;
            const int slot = cell_slot_xyz(c, layer_cx, y, z, layer_cx + y * c.sy + z * c.sz);
            v = cell_start[slot + 1] - cell_start[slot];
        }
        int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        int inc = wave_inclusive_scan(v);
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        int woff = 0;
        for (int q = 0; q < w; ++q) woff += wsum[q];
        int carry = carry_s;
        if (k < ncol) off[k] = carry + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == kBlock - 1) carry_s = carry + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) off[ncol] = carry_s;
}

__global__ __launch_bounds__(kBlock) void k_layer_list(Consts c, const int *__restrict__ cell_start, int layer_cx, const int *__restrict__ off,
                                                       int *__restrict__ list)
{
    int k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= c.gy * c.gz || layer_cx < 0 || layer_cx >= c.gx) return;
    int y = k / c.gz, z = k - y * c.gz;
    const int slot = cell_slot_xyz(c, layer_cx, y, z, layer_cx + y * c.sy + z * c.sz);
    int a = cell_start[slot], b = cell_start[slot + 1];
    int o = off[k];
    for (int s = a; s < b; ++s) list[o + (s - a)] = s;
}

// ghost field refresh: mode 0 = P.w (1 float), 1 = V.xyz (3 floats), 2 = (P.w, V.w) (2 floats; rho[] := V.w, dfsph),
// 3 = (P.w, V.w) with rho[] := P.w (pcisph / iisph: P.w carries rho).  One launch serves both sides:
// threads [0, count_a) work on side a (left), threads [count_a, count_a + count_b) on side b (right).
__global__ __launch_bounds__(kBlock) void k_pack_field(const int *__restrict__ list_a, int count_a, float *__restrict__ out_a,
                                                       const int *__restrict__ list_b, int count_b, float *__restrict__ out_b, int mode,
                                                       const float4 *__restrict__ P, const float4 *__restrict__ V)
{
    int r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= count_a + count_b) return;
    const bool b = r >= count_a;
    if (b) r -= count_a;
    const int s = (b ? list_b : list_a)[r];
    float *out = b ? out_b : out_a;
    if (mode == 0) out[r] = P[s].w;
    else if (mode == 1) { float4 v = V[s]; out[3 * (size_t)r] = v.x; out[3 * (size_t)r + 1] = v.y; out[3 * (size_t)r + 2] = v.z; }
    else { out[2 * (size_t)r] = P[s].w; out[2 * (size_t)r + 1] = V[s].w; }
}

__global__ __launch_bounds__(kBlock) void k_unpack_field(const int *__restrict__ list_a, int count_a, const float *__restrict__ in_a,
                                                         const int *__restrict__ list_b, int count_b, const float *__restrict__ in_b, int mode,
                                                         float4 *__restrict__ P, float4 *__restrict__ V, float *__restrict__ rho)
{
    int r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= count_a + count_b) return;
    const bool b = r >= count_a;
    if (b) r -= count_a;
    const int s = (b ? list_b : list_a)[r];
    const float *in = b ? in_b : in_a;
    if (mode == 0) P[s].w = in[r];
    else if (mode == 1) { V[s].x = in[3 * (size_t)r]; V[s].y = in[3 * (size_t)r + 1]; V[s].z = in[3 * (size_t)r + 2]; }
    else {
        const float aa = in[2 * (size_t)r];
        const float bb = in[2 * (size_t)r + 1];
        P[s].w = aa;
        V[s].w = bb;
        if (rho) rho[s] = mode == 3 ? aa : bb;    // dfsph: V.w carries rho; the correction sweeps rewrite it from rho[]
    }
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
