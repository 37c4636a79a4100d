// sph_kernels.h -- gfx950 kernels of the SPH step: counting-sort cell list, neighbour-list build,
// and the list-driven WCSPH / DFSPH sweeps.  Included once by sph_mi355x.hip.
//
// Data layout in HBM (all arrays cell-sorted, x fastest / z / y like the reference's cell stride,
// ParticleSystem.py:102; ties inside a cell broken by original particle id so that every
// neighbour sum runs in the single-thread Taichi order):
//   float4 P[n]   = (x, y, z, s)   s = per-sweep scalar of that particle (rho, k/rho, ...)
//   float4 V[n]   = (vx, vy, vz, a)
//   uint32 nl[k*stride + i]        k-th fluid neighbour of particle i (row-major by k: coalesced per k)
//   uint32 nlb[k*stride + i]       k-th wall neighbour of particle i
//   int    cnt[i] = kf | kb << 16
// One thread per particle; a wave's 64 lanes are 64 consecutive sorted particles (about 8 cells),
// so gathers of P[j]/V[j] hit the same few cache lines across lanes.
#pragma once
#include <type_traits>
#include "sph_device.h"

namespace sph {

// ---- XCD-aware block mapping -------------------------------------------------------------------
// Hardware deals workgroups round-robin over the 8 XCDs (blocks b and b+8 share an XCD and its
// private 4 MiB L2).  Remap so that each XCD sweeps one CONTIGUOUS eighth of the sorted particle
// array: its L2 then only has to hold the few cell layers around its own sweep front instead of
// a slice of everything.  Bijective for any grid size (speed only, never correctness).
__device__ __forceinline__ int xcd_block(int orig, int nwg)
{
    int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// The sweeps of the DFSPH solver loops (k_correct, k_residual and their relaxed forms).  Tried instead of the contiguous eighths: chunks of C
// consecutive tiles dealt round-robin over the XCDs (the XCDs then finish together whatever the scene looks like).  At 1 M particles chunks of
// 16-64 tiles were 1.5-2 % faster (residual 64.0 -> 62.5 us, correction 51.3 -> 50.2 us; 128: nothing; profiles/r03/tune_xcd_chunk*.log) but
// moved 19 % more HBM bytes per launch (139 -> 166 MB: the halos of tiles on either side of a chunk boundary are fetched by two L2s) -- not kept.
// tools/removal_build.py xcd_chunk32 rebuilds the variant.
__device__ __forceinline__ int xcd_sweep_block(int orig, int nwg)
{
    return xcd_block(orig, nwg);
}

// Edge / interior split of a sweep (slab handles; the order comes from k_tile_order in sph_slab_kernels.h): phase 0 = every tile in one launch,
// 1 = the edge tiles (tile_order[0 .. n_edge)), 2 = the interior ones; both split launches have the full grid, surplus workgroups leave at once.
struct TilePhase { const int *order; int ntiles, phase; int shift = 0; const int *sparse = nullptr; int *hot = nullptr; };
// sparse / hot (one GPU, the density loop's change-propagated launches, in which most workgroups find their tile unchanged and leave): the launches
// of a step's second density residual note per tile whether it had work (hot[tile]); the host turns that into a permutation with the working tiles
// first (k_tile_order) and the rest of the loop's launches take their tiles through it (sparse[workgroup]) -- the few hundred workgroups with work
// start at once instead of wherever their index falls among thousands that leave.  Which workgroup serves which tile cannot change a bit.
// shift = 1: workgroup 0 of the launch is not a tile's -- it takes the loop decision of the sweep BEFORE this one (fin_ride_block) -- and workgroup
// b serves the tile that workgroup b - 1 of a grid one smaller would (phase 0, one GPU)
// Slab handles that hide the residual's all-reduce behind the next divergence correction (sph_host_dfsph.h: step_dfsph_device_loops): the correction of
// evaluation e runs before decision e is known and leaves the velocities and warm_start_k it overwrote in SpecSave; if decision e closed the loop
// (DevScalars.stop_at == e), the residual launch that follows -- its gate is closed -- puts them back, every workgroup its own 256 particles.
struct SpecSave { float4 *v; float *w; };
struct SpecUndo { float4 *v_dst; const float4 *v_src; float *w_dst; const float *w_src; int eval; };

// ---- the density loop's sparse launches: the PRODUCER says who must run (round 6; one GPU) ---------------------------------------------------
// With change propagation (stage_sources_flagged below) 70-89 % of the workgroups of a D6 / D7 launch find their tile unchanged -- but each of
// them read its staging plan, and D7's even gathered 4 bytes per staged particle, to learn that: ~3.3 us per idle workgroup, a third of the launch.
// The sweep that WRITES the operand knows which of its tiles changed anything, and k_build_nl knows which tiles stage a tile's particles (cell
// adjacency is symmetric: the tiles whose particles U stages are the tiles that stage U's): nbr[tile * kNbrStride] = count | flags, then the tiles.
//   D7 (k_correct<DENS>)  a tile that changed a velocity stores this launch's stamp into need6[t] of every tile t that stages its particles;
//   D6 (k_residual<DENS>) a tile whose own k / rho holds a nonzero (nz[tile], kept across the launches in which it idles) stores its stamp into need7[t].
// The consumer reads ONE word: need[tile] == the stamp of the launch before it, or it leaves.  A stamp is a per-handle launch counter, so a stale
// word never matches and nothing is ever cleared.  A tile that passes goes on to the exact per-particle check as before; the flags only say "maybe".
// Where the symmetry does not hold the flags say "always": a tile with a particle whose coordinates lie outside the grid (binned into a wrapped
// cell or nowhere: it walks cells it is not stored in) runs in every launch and, as a producer, raises the broadcast word instead (kNbrOdd); so
// does a tile whose cell set or tile list did not fit (header -1).
constexpr int kStageMaxCells = 640;        // cell runs a workgroup's staging plan may hold (see the plan in k_build_nl)
constexpr int kNbrStride = 64;            // ints per tile in DensFlow.nbr: header + up to 63 tiles
constexpr int kNbrOdd = 1 << 30;
struct DensFlow {
    const int *nbr;                       // nullptr: off (slab handles, SPH_DENS_PUSH=0): the consumers read their staging plans as before
    const int *need_in; int *need_out;    // stamps per tile: what this launch consumes / produces
    int *nz;                              // per tile: its own k / rho holds a nonzero
    int *bcast;                           // [0]: "every tile must run D6", [1]: "... D7" (stamps)
    int stamp_in, stamp_out, bc_in, bc_out;
    int *worked;                          // per tile, this kind of sweep: did the tile do real work in the loop's last iteration?  (a hint, see below)
};
constexpr DensFlow kNoFlow{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, nullptr};
// A sparse launch is as long as its slowest WORKING workgroup lives (tools/dens_timeline.py, profiles/r06/: the ~400 working tiles all start within
// 0.6 us, live 17 us on average and 26 us at worst, the ~3500 others are gone after 15 us), and 6 us of such a life was the chain of dependent
// trips in front of the first pair: gate, tile, need word, stage_cnt, cell runs, the checked scalars, the operands.  Two things shorten it:
//   * a tile that worked in the last iteration is almost certainly working in this one (the working set of a step barely moves): it skips the
//     per-particle check and stages its operands in one batch (computing a tile that turns out unchanged gives the bits that stand);
//   * the workgroups at the head of the working-tiles-first order request stage_cnt and their first cell run together with the need word.
struct StagePre { int have; int sw; uint2 rn0; };
constexpr StagePre kNoPre{0, 0, {0u, 0u}};
// The head of a density-loop workgroup under DensFlow: everything its verdict needs, in ONE batch of independent loads -- a load behind a
// branch (a short-circuit `||` is one) is a round trip of its own, ~0.8 us in a launch that is as long as its slowest workgroup lives.
//   need    must this tile run?
//   direct  it worked in the loop's last iteration (DensFlow.worked): no per-particle check, the operands in one batch
//   pre     the workgroups at the head of the working-tiles-first order (TilePhase.sparse: worked, then checked, then idle; n2 = sparse[ntiles + 1] =
//           how many worked when the order was taken, fetched by the caller together with its tile) get their plan's head -- stage_cnt and the
//           first cell run of every thread -- with the same batch
struct FlowHead { bool need, direct; StagePre pre; };
__device__ __forceinline__ FlowHead flow_head(const TilePhase &tp, const DensFlow &df, int tile, int n2, const uint2 *__restrict__ stage_runs,
                                              const int *__restrict__ stage_cnt)
{
    FlowHead h;
    h.pre = kNoPre;
    if ((int)blockIdx.x - (tp.shift ? 1 : 0) < n2) {          // (a run past the plan's end: never looked at)
        h.pre.rn0 = stage_runs[(size_t)tile * kStageMaxCells + threadIdx.x];
        h.pre.have = 1;
    }
    const int hdr = df.nbr[(size_t)tile * kNbrStride], nd = df.need_in[tile], bc = df.bcast[df.bc_in], wk = df.worked[tile], sw = stage_cnt[tile];
    h.pre.sw = sw;
    h.need = (((hdr < 0) ? 1 : 0) | ((hdr & kNbrOdd) != 0 ? 1 : 0) | (nd == df.stamp_in ? 1 : 0) | (bc == df.stamp_in ? 1 : 0)) != 0;
    h.direct = wk != 0;
    return h;
}
// this tile's output changed (or stands and is not zero): every tile that stages its particles must run the next sweep.  Called by whole waves;
// `mine` = nbr[tile * kNbrStride + lane], requested at the head of the kernel
__device__ __forceinline__ void flow_push(const DensFlow &df, int mine)
{
    const int lane = threadIdx.x & 63;
    const int hdr = __shfl(mine, 0, 64);
    if (hdr < 0 || (hdr & kNbrOdd) != 0) { if (lane == 0) df.bcast[df.bc_out] = df.stamp_out; }
    if (hdr >= 0 && lane >= 1 && lane <= (hdr & (kNbrStride - 1))) df.need_out[mine] = df.stamp_out;
}
__device__ __forceinline__ int sweep_tile(const TilePhase &tp, bool spread);
__device__ __forceinline__ void spec_undo(const Consts &c, const SpecUndo &un, const DevScalars *__restrict__ ds, const TilePhase &tp)
{
    if (!un.v_dst || tp.phase == 2 || ds->stop_at != un.eval) return;          // (a split sweep over a tile order: its first launch, a full grid, undoes)
    const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
    if (i < c.n) { un.v_dst[i] = un.v_src[i]; un.w_dst[i] = un.w_src[i]; }
}
__device__ __forceinline__ int sweep_tile(const TilePhase &tp, bool spread)
{
    if (tp.shift) {
        const int b = (int)blockIdx.x - 1, g = (int)gridDim.x - 1;
        if (b < 0) return -1;
        return spread ? (tp.sparse ? tp.sparse[b] : b) : xcd_sweep_block(b, g);
    }
    if (tp.phase == 0) return spread ? (tp.sparse ? tp.sparse[blockIdx.x] : (int)blockIdx.x) : xcd_sweep_block(blockIdx.x, gridDim.x);
    const int ne = tp.order[tp.ntiles];
    if (tp.phase == 1) return (int)blockIdx.x < ne ? tp.order[blockIdx.x] : -1;
    const int ni = tp.ntiles - ne;
    if ((int)blockIdx.x >= ni) return -1;
    return tp.order[ne + (spread ? (int)blockIdx.x : xcd_sweep_block(blockIdx.x, ni))];
}

// Neighbour lists are stored per 64-particle wave tile, four rows interleaved per lane:
//   entry (i, k) lives at ((i/64)*kmax + (k & ~3))*64 + (i%64)*4 + (k & 3)
// so a lane fetches neighbours k..k+3 with ONE 16-byte load, a wave's load is 1 KiB contiguous, and
// the whole tile (kmax * 256 B) is one contiguous chunk that the wave streams front to back.
__device__ __forceinline__ size_t nl_index(int i, int k, int kmax)
{
    return ((size_t)(i >> 6) * kmax + (k & ~3)) * 64 + (size_t)(i & 63) * 4 + (k & 3);
}

// ======================================================================================
// counting sort by cell                      ParticleSystem.py:368-397 (reset_grid + update_grid)
// ======================================================================================
__device__ __forceinline__ int cell_id_of(const Consts &c, float x, float y, float z, int &cx, int &cy, int &cz)
{
    // get_particle_grid_index_3d / _1d                       ParticleSystem.py:486-494
    cx = (int)floorf(x / c.hcell);
    cy = (int)floorf(y / c.hcell);
    cz = (int)floorf(z / c.hcell);
    int id = cx + cy * c.sy + cz * c.sz;
    if (id < 0 || id >= c.C) id = c.C;   // "lost" bucket (reference prints and skips, :393-395)
    // The reference guards only the 1-D index: a particle that slipped through a wall keeps a valid index and is binned into a
    // WRAPPED cell, a box length away from where it is -- a candidate that never passes the distance test there.  On a slab handle
    // that wrapped cell may be a column this rank shares with a neighbour (ordered edge / ghost lists), so such a particle is binned
    // nowhere instead: same results (nobody could see it), consistent lists.
    if (c.strict_cells && (cx < 0 || cx >= c.gx || cy < 0 || cy >= c.gy || cz < 0 || cz >= c.gz)) id = c.C;
    return id;
}

// Where a cell's particles are STORED is a free choice: every consumer reaches a cell through cell_start[slot of the cell] and
// walks the 27 cells in the reference's (dx, dy, dz) sequence with ascending particle id inside a cell, so the order of every sum
// is the reference's whatever the storage order.  The reference's own order (x fastest) makes the ~8 cells of a wave a run along
// x whose neighbourhood is 3 x 3 x 10 = 90 cells.  Stored along the Morton curve of (x, y, z) they are a 2x2x2 block with a
// 4x4x4 = 64 cell neighbourhood, the 32 cells of a workgroup a 4x4x2 block (144 cells instead of 306) and an XCD's eighth of a
// launch a compact brick -- that is what the L1/L2 hit rates of the neighbour sweeps see (dfsph 1M: 167 -> 203 Mparticle-steps/s).
// Two levels keep it arithmetic: cubic tiles of 2^tbits cells per axis, bits interleaved inside a tile, and a small table with
// the rank of every tile along the Morton curve of the tile coordinates.
// one coordinate's share of a slot: its tile term (scaled by the tile stride of the axis) and its interleaved low bits
struct SlotPart { int tile, code; };
__device__ __forceinline__ SlotPart slot_part(const Consts &c, int v, int axis, int tile_stride)
{
    if (c.tbits == 2)                                       // the default tile edge, without the loop
        return {(v >> 2) * tile_stride, ((v & 1) | (v & 2) << 2) << axis};
    int code = 0;
    for (int k = 0; k < c.tbits; ++k) code |= ((v >> k) & 1) << (3 * k + axis);
    return {(v >> c.tbits) * tile_stride, code};
}
__device__ __forceinline__ int slot_of_parts(const Consts &c, SlotPart x, SlotPart y, SlotPart z)
{
    return (c.tile_rank[x.tile + y.tile + z.tile] << (3 * c.tbits)) | (x.code | y.code | z.code);
}
__device__ __forceinline__ int cell_slot_xyz(const Consts &c, int x, int y, int z, int id)
{
    if (c.order != CELL_ORDER_TILED) return id;
    return slot_of_parts(c, slot_part(c, x, 0, 1), slot_part(c, y, 1, c.tnxz), slot_part(c, z, 2, c.tnx));
}

// from a 1-D index as cell_id_of returns it (a wrapped index of a particle outside the box is a valid cell; C = binned nowhere)
__device__ __forceinline__ int cell_slot(const Consts &c, int id)
{
    if (id >= c.C) return c.S;
    if (c.order != CELL_ORDER_TILED) return id;
    const int q = id / c.gx;                                // id = x + z*gx + y*gx*gz     ParticleSystem.py:102
    const int slot = cell_slot_xyz(c, id - q * c.gx, q / c.gz, q % c.gz, id);
    return slot < 0 ? c.S : slot;                           // (a column this slab does not hold: binned nowhere)
}

// `dead` (multi-GPU only): slots whose particle left the slab or was last step's ghost take no part in
// the sort; the sorted arrays simply end after the live particles.
// The arrays arrive in last step's cell order, so consecutive lanes mostly share a cell: a run of equal cells inside a wave
// takes ONE atomic (its head lane adds the run length, every lane gets head's base + its offset in the run).  `rank` is only
// a slot allocator -- k_order_gather establishes the canonical order inside a cell -- so any assignment is fine.
__global__ __launch_bounds__(kBlock) void k_hash_count(Consts c, const float4 *__restrict__ P, const int *__restrict__ dead,
                                                       int *__restrict__ cell_of, int *__restrict__ rank, int *__restrict__ cell_count,
                                                       DevScalars *__restrict__ ds, const int *__restrict__ gate = nullptr)
{
    if (gate && *gate == 0) return;       // Verlet handles: the lists of an earlier step still hold (DevScalars.rebuild)
    const int s = blockIdx.x * kBlock + threadIdx.x;
    // the per-build maxima of the list lengths start at zero (instead of a memset launch before the list build)
    if (ds && blockIdx.x == 0 && threadIdx.x < 2 * kNoteShards) ds->nbr_shard[threadIdx.x] = 0;
    const int lane = threadIdx.x & 63;
    int id = -1;                                            // -1: no particle in this lane (past the end, or a dead slot)
    if (s < c.n && !(dead && dead[s])) {
        float4 p = P[s];
        int cx, cy, cz;
        id = cell_slot(c, cell_id_of(c, p.x, p.y, p.z, cx, cy, cz));
    }
    if (s < c.n) cell_of[s] = id;
    const int prev = __shfl_up(id, 1, 64);
    const bool head = lane == 0 || prev != id;
    const unsigned long long heads = __ballot(head);
    const unsigned long long upto = heads & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));   // heads at lanes <= lane
    const int start = 63 - __clzll(upto);                   // head lane of this lane's run (upto != 0: lane 0 is always a head)
    const unsigned long long after = lane == 63 ? 0ull : (heads >> (lane + 1));
    const int end = after ? lane + 1 + (__ffsll((long long)after) - 1) : 64;                  // one past the run's last lane
    int base = 0;
    if (head && id >= 0) base = atomicAdd(&cell_count[id], end - lane);
    base = __shfl(base, start, 64);
    if (id >= 0) rank[s] = base + (lane - start);
}

// exclusive scan, three launches: per-tile scan, scan of tile sums, add-back
constexpr int kScanTile = kBlock * 4;   // four entries per thread
// `in` (the cell histogram) is zeroed as it is read: the next step's k_hash_count finds it clean without a memset launch
__global__ __launch_bounds__(kBlock) void k_scan_tiles(int *__restrict__ in, int *__restrict__ out,
                                                       int *__restrict__ tile_sums, int n, const int *__restrict__ gate = nullptr)
{
    if (gate && *gate == 0) return;
    __shared__ int wsum[kBlock / 64];
    int base = blockIdx.x * kScanTile + threadIdx.x * 4;
    int v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        if (base + k < n) in[base + k] = 0;
    }
    int tsum = v[0] + v[1] + v[2] + v[3];
    // inclusive wave scan of tsum
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = wave_inclusive_scan(tsum);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int woff = 0;
    for (int k = 0; k < w; ++k) woff += wsum[k];
    int excl = woff + inc - tsum;
    int run = excl;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == kBlock - 1) tile_sums[blockIdx.x] = woff + inc;
}

__global__ __launch_bounds__(kBlock) void k_scan_sums(int *__restrict__ tile_sums, int ntiles, const int *__restrict__ gate = nullptr)
{
    if (gate && *gate == 0) return;
    // single block; serial over chunks of 256 with a carry (ntiles is small: cells / 1024)
    __shared__ int wsum[kBlock / 64];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < ntiles; base += kBlock) {
        int i = base + threadIdx.x;
        int v = (i < ntiles) ? tile_sums[i] : 0;
        int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        int inc = wave_inclusive_scan(v);
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < w; ++k) woff += wsum[k];
        int carry = carry_s;
        if (i < ntiles) tile_sums[i] = carry + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == kBlock - 1) carry_s = carry + woff + inc;
        __syncthreads();
    }
}

// fold != 0: tile_sums still holds the RAW per-tile totals (k_scan_sums was not launched: grids of up to kScanFoldTiles tiles) and every
// workgroup adds up the totals of the tiles in front of its own -- a few hundred L2-resident ints against a dependent launch (~3 us)
constexpr int kScanFoldTiles = 1024;
__global__ __launch_bounds__(kBlock) void k_scan_add(int *__restrict__ out, const int *__restrict__ tile_sums, int n, const int *__restrict__ gate = nullptr, int fold = 0)
{
    if (gate && *gate == 0) return;
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (!fold) {
        if (i < n) out[i] += tile_sums[i / kScanTile];
        return;
    }
    __shared__ int s_w[kBlock / 64];
    const int tile = (int)(blockIdx.x * kBlock) / kScanTile;          // (kScanTile is a multiple of kBlock: a workgroup lies in one tile)
    int v = 0;
    for (int t = threadIdx.x; t < tile; t += kBlock) v += tile_sums[t];
    const int ws = wave_sum(v);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = ws;
    __syncthreads();
    int before = 0;
    for (int k = 0; k < kBlock / 64; ++k) before += s_w[k];
    if (i < n) out[i] += before;
}

__global__ __launch_bounds__(kBlock) void k_scatter(Consts c, const int *__restrict__ cell_of, const int *__restrict__ rank,
                                                    const int *__restrict__ cell_start, int *__restrict__ slot_src, const int *__restrict__ gate = nullptr)
{
    if (gate && *gate == 0) return;
    int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= c.n) return;
    int cell = cell_of[s];
    if (cell < 0) return;                       // dead slot (multi-GPU)
    slot_src[cell_start[cell] + rank[s]] = s;
}

// Canonical order inside a cell = ascending original id (the single-thread append order of
// update_grid_fluid_particles, ParticleSystem.py:389-397), then gather the state into it.
__global__ __launch_bounds__(kBlock) void k_order_gather(Consts c, const int *__restrict__ cell_of, const int *__restrict__ cell_start,
                                                         const int *__restrict__ slot_src, const float4 *__restrict__ Pin,
                                                         const float4 *__restrict__ Vin, const float *__restrict__ warm_in,
                                                         const int *__restrict__ id_in, float4 *__restrict__ Pout,
                                                         float4 *__restrict__ Vout, float *__restrict__ warm_out,
                                                         int *__restrict__ id_out, float4 *__restrict__ pos_orig,
                                                         const int *__restrict__ gate = nullptr, float4 *__restrict__ x0 = nullptr,
                                                         int *__restrict__ dead = nullptr)
{
    int d = blockIdx.x * kBlock + threadIdx.x;
    if (d >= c.n) return;
    if (dead) dead[d] = 0;              // slab handles: every sorted slot is alive (was a memset of its own, two fill launches per step)
    if (gate && *gate == 0) {           // Verlet handles between two builds: the order stands, the arrays only change roles (the host flips its buffer indices every step)
        Pout[d] = Pin[d]; Vout[d] = Vin[d]; id_out[d] = id_in[d];
        if (warm_in) warm_out[d] = warm_in[d];
        return;
    }
    int src = slot_src[d];
    int cell = cell_of[src];
    int a = cell_start[cell], b = cell_start[cell + 1];
    int raw = id_in[src];
    int key = raw < 0 ? ~raw : raw;          // ghosts carry ~id; order by the true id
    int r = 0;
    if (cell < c.S) {
        for (int e = a; e < b; ++e) {
            int o = id_in[slot_src[e]];
            r += ((o < 0 ? ~o : o) < key) ? 1 : 0;
        }
    } else {
        r = d - a;                              // "outside the grid" bucket: nobody walks it, keep arrival order
    }
    int dst = a + r;
    const float4 pp = Pin[src];
    Pout[dst] = pp;
    Vout[dst] = Vin[src];
    if (warm_in) warm_out[dst] = warm_in[src];
    id_out[dst] = raw;
    if (pos_orig) pos_orig[key] = pp;
    if (x0) x0[dst] = pp;               // Verlet handles: where this particle was when the lists were built
}

// ---- rigid body (config 5) ------------------------------------------------------------------------
// Rigid sample particles are a third species: every step they are cell-sorted like the fluid, and a fluid particle's
// neighbour list holds them in the reference's order -- per cell: fluid entries, then rigid entries (update_grid appends
// fluid first, ParticleSystem.py:383-386) -- tagged with bit 31.  vel / acc / omega / alpha are uniform over the body
// (rigid_solver.py fills them, :41,96-97,128), so only (x, y, z, V_r) is per particle.
constexpr uint32_t kRigidTag = 0x80000000u;

struct RigidView {
    const float4 *RP;         // cell-sorted rigid particles (x, y, z, V_r)
    const int *rid;           // sorted slot -> rigid particle index
    const int *rcell_start;   // rigid cell list
    const float4 *pos_orig;   // fluid positions by ORIGINAL particle id   (get_neighbour_count quirk, ParticleSystem.py:440-442)
    const float *rho_orig;    // fluid densities by ORIGINAL particle id   (viscosity quirk, solver_base.py:198-199)
    float c[3], vel[3], acc[3], omega[3], alpha[3];
    int n_fluid;
};

// predicted velocity of a rigid particle as the fluid sees it      dfsph_solver.py:292-293 / :168-169
__device__ __forceinline__ F3 rigid_velocity(const RigidView &rv, float4 pj, float dt, bool with_alpha)
{
    float wx = with_alpha ? rv.omega[0] + rv.alpha[0] * dt : rv.omega[0];
    float wy = with_alpha ? rv.omega[1] + rv.alpha[1] * dt : rv.omega[1];
    float wz = with_alpha ? rv.omega[2] + rv.alpha[2] * dt : rv.omega[2];
    float rx = pj.x - rv.c[0], ry = pj.y - rv.c[1], rz = pj.z - rv.c[2];
    float cx = wy * rz - wz * ry, cy = wz * rx - wx * rz, cz = wx * ry - wy * rx;
    F3 o;
    o.x = (rv.vel[0] + rv.acc[0] * dt) + cx;
    o.y = (rv.vel[1] + rv.acc[1] * dt) + cy;
    o.z = (rv.vel[2] + rv.acc[2] * dt) + cz;
    return o;
}

// The index stream of a list is the one part of a sweep that always comes from HBM (every row is read once per sweep), at
// ~800 cycles per request; the gathers it feeds mostly hit L2.  Rows are requested SPH_NL_AHEAD groups before they are used
// (measured at 1M particles: 2 or 3 groups ahead change nothing, 4 cost registers; non-temporal row loads are 30% slower).
#ifndef SPH_NL_AHEAD
#define SPH_NL_AHEAD 1          // index groups requested ahead of the one being processed
#endif
__device__ __forceinline__ uint4 nl_load(const uint32_t *p) { return *reinterpret_cast<const uint4 *>(p); }
struct NlAhead {
    uint4 q[SPH_NL_AHEAD];
    const uint32_t *base;
    __device__ __forceinline__ explicit NlAhead(const uint32_t *b) : base(b)
    {
#pragma unroll
        for (int d = 0; d < SPH_NL_AHEAD; ++d) q[d] = nl_load(base + (size_t)d * 256);
    }
    __device__ __forceinline__ uint4 front() const { return q[0]; }
    // drop the front group, request group (kk/4 + SPH_NL_AHEAD)
    __device__ __forceinline__ void advance(int kk)
    {
#pragma unroll
        for (int d = 0; d + 1 < SPH_NL_AHEAD; ++d) q[d] = q[d + 1];
        q[SPH_NL_AHEAD - 1] = nl_load(base + (size_t)((kk >> 2) + SPH_NL_AHEAD) * 256);
    }
};

// fluid-list walkers that understand tagged rigid entries; body(pj, vj, j): j & kRigidTag marks a rigid neighbour,
// then pj = (x, y, z, V_r) and vj is undefined
// The walk of a list, software-pipelined: the operands of group g+1 are requested before the bodies of group g run, the index
// row of group g+2 before that.  Small scenes are bound by the latency of one wave's dependent gathers (one wave per SIMD or
// less): +12 % on DFSPH at 30 k particles; large scenes are unaffected.  `fetch(j, slot)` loads one neighbour's operands,
// `use(slot, j)` is the pair body; bodies run in list order.  The speculative fetch past the last group reads stale but valid
// indices (see for_nbrs_p).
template <class T, class Fetch, class Use>
__device__ __forceinline__ void walk_list(const uint32_t *__restrict__ base, int cnt, Fetch fetch, Use use)
{
    if (cnt <= 0) return;
    NlAhead ahead(base);
    uint4 jj = ahead.front();
    T cur[4], nxt[4];
    fetch(jj.x, cur[0]); fetch(jj.y, cur[1]); fetch(jj.z, cur[2]); fetch(jj.w, cur[3]);
    ahead.advance(0);
    for (int kk = 0; kk < cnt; kk += 4) {
        const uint4 jn = ahead.front();
        fetch(jn.x, nxt[0]); fetch(jn.y, nxt[1]); fetch(jn.z, nxt[2]); fetch(jn.w, nxt[3]);
        ahead.advance(kk + 4);
        use(cur[0], jj.x);
        if (kk + 1 < cnt) use(cur[1], jj.y);
        if (kk + 2 < cnt) use(cur[2], jj.z);
        if (kk + 3 < cnt) use(cur[3], jj.w);
#pragma unroll
        for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
        jj = jn;
    }
}
struct Operand1 { float4 a; };
struct Operand2 { float4 a, b; };

template <bool RIGID, bool WITHV, class Body>
__device__ __forceinline__ void for_fluid_nbrs(const uint32_t *__restrict__ base, int cnt, const float4 *__restrict__ A,
                                               const float4 *__restrict__ B, const RigidView &rv, Body body)
{
    if (WITHV)
        walk_list<Operand2>(base, cnt, [&](uint32_t j, Operand2 &o) {
            const bool rg = RIGID && (j & kRigidTag);
            const uint32_t idx = RIGID ? (j & ~kRigidTag) : j;
            o.a = rg ? rv.RP[idx] : A[idx];
            o.b = B[rg ? 0u : idx];
        }, [&](const Operand2 &o, uint32_t j) { body(o.a, o.b, j); });
    else
        walk_list<Operand1>(base, cnt, [&](uint32_t j, Operand1 &o) {
            const bool rg = RIGID && (j & kRigidTag);
            const uint32_t idx = RIGID ? (j & ~kRigidTag) : j;
            o.a = rg ? rv.RP[idx] : A[idx];
        }, [&](const Operand1 &o, uint32_t j) { body(o.a, make_float4(0.f, 0.f, 0.f, 0.f), j); });
}

// ---- four lanes per particle (small scenes) ---------------------------------------------------------------------------------
// A sweep's sums are sequential per particle (the reference adds a particle's neighbours one after the other and f32 addition does
// not reassociate), so a scene of less than a wave per SIMD is as slow as ONE lane walking ~30 pairs of ~60 instructions, at one
// instruction per 4+ cycles, while most of the chip idles.  The expensive part of a pair is its TERM, not the addition: in the quad
// walks the four lanes of a quad serve one particle, lane q evaluates entry q of every group of four into zeroed accumulators
// (0 + t = t exactly; the accumulators of a sweep start at +0 or 0.001 and a sum of terms is never -0), and all four lanes then add
// the four terms in list order through DPP quad broadcasts: the same additions in the same order, a quarter of the pair bodies per
// lane.  Per group of four: one body + 4 adds per accumulator instead of four bodies.  A pair body must add to each accumulator at
// most once.  Sweep kernels select it with MODE == SWEEP_QUAD (unstaged handles below kQuadBelow particles; four times the lanes).
enum { SWEEP_PLAIN = 0, SWEEP_STAGED = 1, SWEEP_QUAD = 2 };
template <int U> __device__ __forceinline__ float quad_bcast(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), U * 0x55, 0xf, 0xf, false));
}
__device__ __forceinline__ uint32_t quad_pick(const uint4 g, int q) { return q == 0 ? g.x : q == 1 ? g.y : q == 2 ? g.z : g.w; }
template <int N, class T, class Fetch, class Use>
__device__ __forceinline__ void walk_list_quad(const uint32_t *__restrict__ base, int cnt, int q, float (&acc)[N], Fetch fetch, Use use)
{
    if (cnt <= 0) return;
    uint32_t jq = quad_pick(nl_load(base), q);
    T cur, nxt;
    fetch(jq, cur);
    for (int kk = 0; kk < cnt; kk += 4) {
        const uint32_t jn = quad_pick(nl_load(base + (size_t)((kk >> 2) + 1) * 256), q);      // one group past the end: stale but valid
        fetch(jn, nxt);
        float save[N];
#pragma unroll
        for (int n = 0; n < N; ++n) { save[n] = acc[n]; acc[n] = 0.0f; }
        if (kk + q < cnt) use(cur, jq);
#pragma unroll
        for (int n = 0; n < N; ++n) {
            const float t = acc[n];
            float a = save[n];
            a += quad_bcast<0>(t); a += quad_bcast<1>(t); a += quad_bcast<2>(t); a += quad_bcast<3>(t);
            acc[n] = a;
        }
        cur = nxt; jq = jn;
    }
}
template <bool RIGID, bool WITHV, int N, class Body>
__device__ __forceinline__ void for_fluid_nbrs_quad(const uint32_t *__restrict__ base, int cnt, int q, float (&acc)[N], const float4 *__restrict__ A,
                                                    const float4 *__restrict__ B, const RigidView &rv, Body body)
{
    if (WITHV)
        walk_list_quad<N, Operand2>(base, cnt, q, acc, [&](uint32_t j, Operand2 &o) {
            const bool rg = RIGID && (j & kRigidTag);
            const uint32_t idx = RIGID ? (j & ~kRigidTag) : j;
            o.a = rg ? rv.RP[idx] : A[idx];
            o.b = B[rg ? 0u : idx];
        }, [&](const Operand2 &o, uint32_t j) { body(o.a, o.b, j); });
    else
        walk_list_quad<N, Operand1>(base, cnt, q, acc, [&](uint32_t j, Operand1 &o) {
            const bool rg = RIGID && (j & kRigidTag);
            const uint32_t idx = RIGID ? (j & ~kRigidTag) : j;
            o.a = rg ? rv.RP[idx] : A[idx];
        }, [&](const Operand1 &o, uint32_t j) { body(o.a, make_float4(0.f, 0.f, 0.f, 0.f), j); });
}

// ======================================================================================
// neighbour-list build: the 27-cell walk of for_all_neighbor / for_all_boundary_neighbor
// (ParticleSystem.py:447-469, 337-366), done once per step because positions are frozen
// between the grid rebuild and the integrator.
// ======================================================================================
// Per-build maxima of the list lengths (health counters + the overflow flag).  Every wave used to issue two atomicMax on the
// same two words: ~31 k same-address atomics at 1 M particles, serialised at ~12 ns each = 370 of the kernel's 470 us (measured by
// removing everything else).  A relaxed device-scope load first: the stored maximum only grows during the kernel, so a wave whose
// value does not exceed what it reads has nothing to add; only the few waves that raise the maximum touch it atomically.
__device__ __forceinline__ void note_list_lengths(const Consts &c, int kf, int kb, DevScalars *__restrict__ ds)
{
    const int mf = wave_max(kf), mb = wave_max(kb);
    if ((threadIdx.x & 63) == 0) {
        int *nf = &ds->nbr_shard[blockIdx.x & (kNoteShards - 1)], *nw = &ds->wall_shard[blockIdx.x & (kNoteShards - 1)];
        if (mf > __hip_atomic_load(nf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(nf, mf);
        if (mb > __hip_atomic_load(nw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(nw, mb);
        if (mf > c.kmax || mb > c.kbmax) atomicOr(&ds->overflow, 1);
    }
}

// accept mask of four consecutive candidates: bit u set <=> !(|x_i - x_u|^2 > r2_cut), the distance formed as (dx*dx + dy*dy) + dz*dz.
// (x, y) travel as one register pair straight from the 16-byte load, so the subtraction and the squares are one packed instruction each
// and nothing has to be shuffled (left to itself the compiler pairs ACROSS candidates and spends 14 moves per batch on it).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned near_mask4(f32x2 pi_xy, float pi_z, const float4 *__restrict__ pb, float r2_cut)
{
    unsigned m = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const float4 pc = pb[u];
        const f32x2 c_xy = {pc.x, pc.y};
        const f32x2 d = pi_xy - c_xy;
        const f32x2 sq = d * d;
        const float dz = pi_z - pc.z;
        const float r2 = (sq.x + sq.y) + dz * dz;
        m |= (!(r2 > r2_cut) ? 1u : 0u) << u;
    }
    return m;
}

// The same test for the list build, as a REJECT mask built without compares: squared distances are non-negative floats, whose order is
// the order of their bit patterns, so the sign of (bits(r2_cut) - bits(r2)) says r2 > r2_cut; v_alignbit shifts each sign into the mask
// (one subtract and one funnel shift per candidate where a compare + select pair through VCC costs ~10 cycles, tools/valu_issue.hip).
// bit u set <=> |x_i - x_u|^2 > r2_cut.  (A NaN distance -- a particle with a NaN coordinate, whose cell is unspecified anyway -- counts as far.)
__device__ __forceinline__ unsigned far_mask4(f32x2 pi_xy, float pi_z, const float4 *__restrict__ pb, unsigned cut_bits)
{
    unsigned rej = 0;
#pragma unroll
    for (int u = 3; u >= 0; --u) {
        const float4 pc = pb[u];
        const f32x2 c_xy = {pc.x, pc.y};
        const f32x2 d = pi_xy - c_xy;
        const f32x2 sq = d * d;
        const float dz = pi_z - pc.z;
        const float r2 = (sq.x + sq.y) + dz * dz;
        rej = __builtin_amdgcn_alignbit(rej, cut_bits - __float_as_uint(r2), 31);
    }
    return rej;
}

// List append through an LDS staging row per lane (transposed: slot s of thread t at [s * kBlock + t], conflict-free): no
// register shuffling on (k & 3), one 16-byte store per completed group.
// Two entry formats.  32-bit: four entries per 16-byte group.  16-bit (`half`: fluid lists of a staged workgroup on an nl16 handle,
// whose entries are indices local to the workgroup's staged set, < stage_cap <= 2560): EIGHT entries per group, entry 2q in the low
// and 2q + 1 in the high half of word q -- the index stream, the one part of a sweep that always comes from HBM, is halved.
struct NlWriter {
    uint32_t *stage;        // this thread's column of the block's staging area
    uint32_t *base;
    int k, kcap;
    bool half;
    uint32_t pend;
    bool zero_next;
    __device__ __forceinline__ void push(uint32_t j)
    {
        if (half) {
            const int s = k & 7;
            if (s & 1) {
                const uint32_t w = pend | (j << 16);
                stage[(s >> 1) * kBlock] = w;
                if (s == 7 && k < kcap)
                    *reinterpret_cast<uint4 *>(base + (size_t)(k >> 3) * 256) = make_uint4(stage[0], stage[kBlock], stage[2 * kBlock], w);
            } else {
                pend = j;
            }
        } else {
            const int s = k & 3;
            stage[s * kBlock] = j;
            if (s == 3 && k < kcap)
                *reinterpret_cast<uint4 *>(base + (size_t)(k >> 2) * 256) = make_uint4(stage[0], stage[kBlock], stage[2 * kBlock], j);
        }
        ++k;
    }
    // wall lists: few particles have any, so their entries go straight to memory (4-byte stores into the 16-byte groups; the slots
    // of a last, partial group keep stale but valid indices) and the staging rows of a second writer are not needed: 4 KiB of LDS
    // less per workgroup, which is what lets eight workgroups of the staged build share a CU
    __device__ __forceinline__ void push_direct(uint32_t j)
    {
        if (k < kcap) base[(size_t)(k >> 2) * 256 + (k & 3)] = j;
        ++k;
    }
    // `self`: the particle's own local index.  The slots of a 16-bit list's last group past the count are filled with it: a walk that
    // does not mask its tail (the relaxed sweeps) then meets the particle itself there -- x_ij = 0, v_ij = 0, a term that is exactly 0
    __device__ __forceinline__ void flush(uint32_t self = 0u)
    {
        if (half) {
            const int s = k & 7;
            if (s != 0 && k < kcap) {
                uint32_t w[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t have = 2 * q + 1 < s ? stage[q * kBlock] : (2 * q < s ? (pend | (self << 16)) : (self | (self << 16)));
                    w[q] = have;
                }
                *reinterpret_cast<uint4 *>(base + (size_t)(k >> 3) * 256) = make_uint4(w[0], w[1], w[2], w[3]);
            }
        } else {
            if ((k & 3) != 0 && k < kcap)   // tail slots: stale but valid indices
                *reinterpret_cast<uint4 *>(base + (size_t)(k >> 2) * 256) = make_uint4(stage[0], stage[kBlock], stage[2 * kBlock], stage[3 * kBlock]);
            // The 32-bit walks request one group past a list's end and gather through it speculatively ("stale but valid indices").
            // On an nl16 handle a stale group may hold PACKED pairs from a step in which this workgroup was staged -- as a 32-bit
            // index far outside the arrays (this faulted in the mixed-capacity test).  Such a workgroup zeroes the one group its
            // walks can read ahead (the tile pitch keeps a spare group beyond kmax entries).
            if (zero_next && k <= kcap) *reinterpret_cast<uint4 *>(base + (size_t)((k + 3) >> 2) * 256) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
};

// ---- LDS staging of a workgroup's neighbourhood (large scenes on the Morton curve) ----------------------------------------
// The sweeps are bound by L1 tag look-ups: a 64-lane gather of P[j] touches ~44 cache lines (TCP_TOTAL_CACHE_ACCESSES / wave
// read), one look-up per cycle and CU -- 54 M look-ups = 88 us of a 104 us residual sweep at 1 M particles.  On the Morton curve
// the particles a 256-particle workgroup can see (all cells around its own cells) are only ~1300 (max ~1800): the list build
// also writes, per workgroup, the ordered set of those particles (as cell runs, stage_runs) and the lists in indices LOCAL to that
// set; a sweep then copies its operand array through that set into LDS once (coalesced: ~150 line look-ups per workgroup instead
// of ~8000) and gathers from LDS.  A workgroup whose neighbourhood does not fit (sparse regions) keeps global indices: stage_cnt < 0.
constexpr int kRunCap = 13;          // runs of equal cell per wave whose 27 cells k_build_nl looks up cooperatively (mean 9, p99 13-16);
                                     // 13: the staged build's LDS stays under 20 KiB, eight workgroups per CU
constexpr int kStageHash = 1024;       // open-addressing set of the cell slots a workgroup needs
// stage_cnt[blk] = staged particles | runs << 16 | kStageLists16 (-1: not staged).  kStageLists16: the fluid lists of this workgroup hold 16-bit
// indices local to its staged set.  Without a rigid body that is every staged workgroup of an nl16 handle; with one, the workgroups whose
// neighbourhood cells hold no rigid sample (tagged rigid entries need 32 bits) -- all but a thin shell around the body.
constexpr int kStageLists16 = 1 << 30;
constexpr int kTileSet = 128;          // open-addressing set of the tiles that hold a workgroup's staged particles (DensFlow.nbr)
__device__ __forceinline__ bool stage_lists16(const int *__restrict__ stage_cnt, int blk) { return (stage_cnt[blk] & (kStageLists16 | (int)0x80000000)) == kStageLists16; }
// Consts.stage_cap = staged particles per workgroup (16 B each for one-operand sweeps, 24 B for the residuals); 1664 keeps four
// workgroups of a residual sweep resident per CU (4 x 39 KiB of the 160 KiB LDS), ~2 % of the workgroups at 1 M particles exceed it.
__device__ __forceinline__ int stage_hash(int slot) { return (int)(((unsigned)slot * 2654435761u) >> 22); }
// every cell a particle walks was inserted by the head of its run (see the plan); the probe is bounded all the same, a miss would
// otherwise hang the GPU: it raises the overflow flag (bit 1) instead and the step reports SPH_E_OVERFLOW
__device__ __forceinline__ int stage_lookup(const int *key, const int *base, int slot, DevScalars *__restrict__ ds)
{
    int hq = stage_hash(slot);
    for (int probe = 0; probe < kStageHash; ++probe) {
        if (key[hq] == slot) return base[hq];
        hq = (hq + 1) & (kStageHash - 1);
    }
    atomicOr(&ds->overflow, 2);
    return 0;
}

// open-addressing insert into the kTileSet slots of a workgroup's tile set (k_build_nl; -1 = empty).  A full table drops the tile: the count taken
// afterwards then reaches kTileSet and the row's header says "unknown"
__device__ __forceinline__ void tile_set_insert(int *s_tset, int t)
{
    static_assert(kTileSet == 128, "7 hash bits");
    int hq = (int)(((unsigned)t * 2654435761u) >> 25);
    for (int probe = 0; probe < kTileSet; ++probe) {
        const int was = atomicCAS(&s_tset[hq], -1, t);
        if (was == -1 || was == t) break;
        hq = (hq + 1) & (kTileSet - 1);
    }
}
// who owns neighbour lists: every owned particle (id >= 0), and on two-column slab handles the ghosts of the inner ghost column (Consts.gw_*)
__device__ __forceinline__ bool list_walker(const Consts &c, int id, int cx) { return id >= 0 || (c.ghost_walk && (cx == c.gw_left || cx == c.gw_right)); }

// (Candidates are requested four at a time: twelve per batch gain where there is less than a wave per SIMD -- 30 k particles: 54 -> 48 us --
// and lose to the tests of candidates past the cell's end where issue is the limit -- 250 k: 68 -> 75 us; small scenes use k_build_nl_split.)
template <bool RIGID, bool STAGED>
__global__ __launch_bounds__(kBlock) void k_build_nl(Consts c, const float4 *__restrict__ P, const int *__restrict__ cell_start,
                                                     const float4 *__restrict__ WP, const int *__restrict__ wcell_start,
                                                     const int *__restrict__ id, uint32_t *__restrict__ nl,
                                                     uint32_t *__restrict__ nlb, int *__restrict__ cnt, DevScalars *__restrict__ ds,
                                                     RigidView rv, int *__restrict__ ncount, uint2 *__restrict__ stage_runs,
                                                     int *__restrict__ stage_cnt, const int *__restrict__ gate = nullptr, int *__restrict__ tile_nbr = nullptr)
{
    if (gate && *gate == 0) return;       // Verlet handles: the lists still hold
    __shared__ uint32_t s_stage[4 * kBlock];
    __shared__ int s_key[STAGED ? kStageHash : 1], s_base[STAGED ? kStageHash : 1], s_wsum[kBlock / 64], s_wsum_ne[kBlock / 64], s_ncell, s_ok, s_odd;
    __shared__ uint4 s_cell[kBlock / 64][kRunCap * 9];
    __shared__ int s_cslot[RIGID ? kBlock / 64 : 1][RIGID ? kRunCap * 9 : 1], s_runc[kBlock / 64][kRunCap][3];
    const int blk = xcd_block(blockIdx.x, gridDim.x);
    int i = blk * kBlock + threadIdx.x;
    int kf = 0, kb = 0;
    if (i == 0) ds->lost = cell_start[c.S + 1] - cell_start[c.S];   // size of the "outside the grid" bucket
#pragma unroll
    for (int q = 0; q < 4; ++q) s_stage[q * kBlock + threadIdx.x] = 0;
    if (STAGED) {
        // (1) the set of cell slots around the cells of this workgroup's own particles
        for (int q = threadIdx.x; q < kStageHash; q += kBlock) s_key[q] = -1;
        // (DensFlow.nbr: the set of TILES that hold those cells' particles, kTileSet slots in the last wave's cell table -- that table is first
        // written by its own wave in (3), after that wave has read the set back below)
        int *s_tset = reinterpret_cast<int *>(&s_cell[kBlock / 64 - 1][0]);
        static_assert(sizeof(s_cell[0]) >= kTileSet * sizeof(int), "the tile set lives in one wave's cell table");
        if (tile_nbr && threadIdx.x < kTileSet) s_tset[threadIdx.x] = -1;
        if (threadIdx.x == 0) { s_ncell = 0; s_ok = 1; s_odd = 0; }
        __syncthreads();
        int hcx = 0, hcy = 0, hcz = 0;
        if (i < c.n) cell_id_of(c, P[i].x, P[i].y, P[i].z, hcx, hcy, hcz);
        // a particle whose coordinates lie outside the grid is stored in a wrapped cell (or nowhere) and walks cells it is not stored in: kNbrOdd
        if (tile_nbr && i < c.n && (hcx < 0 || hcx >= c.gx || hcy < 0 || hcy >= c.gy || hcz < 0 || hcz >= c.gz)) s_odd = 1;
        // The first particle of every run of equal cell COORDINATES contributes the run's 27 cells (particles that left the box share a wrapped or
        // "outside" cell index with particles whose coordinates, and hence neighbour cells, differ).  Round 6 (tools/bnl_timeline.py): those ~32 heads
        // used to walk their 27 cells themselves, one dependent trip to the tile-rank table and one LDS atomic chain per cell, while the other 224
        // lanes waited -- 22 of a workgroup's 75 us.  Now the heads only leave their coordinates in LDS and ALL lanes share the (head, cell) pairs.
        // kind 1 (DensFlow on slab handles): a run of particles WITHOUT lists -- ghosts of the outer column -- which its neighbours' tiles stage all the
        // same: the kernel that unpacks its k / rho pushes on its behalf through THIS tile's row, so the tiles around its cell go into the tile set.
        int *s_head = reinterpret_cast<int *>(s_stage);           // (3 ints per head; the writer's staging rows are zeroed after this phase)
        static_assert(4 * kBlock >= 3 * kBlock, "one head per thread at most");
        {
            const bool in = i < c.n;
            const bool walks = in && list_walker(c, id[i], hcx);
            const int kind = walks ? 0 : 1;
            bool head = in && (walks || tile_nbr != nullptr);
            if (head && threadIdx.x != 0) {
                int px, py, pz;
                const float4 pp = P[i - 1];
                cell_id_of(c, pp.x, pp.y, pp.z, px, py, pz);
                const bool pwalks = list_walker(c, id[i - 1], px);
                head = pwalks != walks || px != hcx || py != hcy || pz != hcz;
            }
            const unsigned long long hb = __ballot(head);
            const int ln = threadIdx.x & 63, wv0 = threadIdx.x >> 6;
            if (ln == 0) s_wsum[wv0] = __popcll(hb);
            __syncthreads();
            int hbase = 0, nheads = 0;
            for (int k = 0; k < kBlock / 64; ++k) { if (k < wv0) hbase += s_wsum[k]; nheads += s_wsum[k]; }
            if (head) {
                const int e = hbase + __popcll(hb & ((1ull << ln) - 1ull));
                s_head[3 * e] = hcx; s_head[3 * e + 1] = hcy; s_head[3 * e + 2] = (hcz << 1) | kind;
            }
            __syncthreads();
            for (int t = threadIdx.x; t < nheads * 27; t += kBlock) {
                const int e = t / 27, o = t - 27 * e;
                const int o9 = o / 9, o3 = (o - 9 * o9) / 3;
                const int x = s_head[3 * e] + o9 - 1, y = s_head[3 * e + 1] + o3 - 1, zk = s_head[3 * e + 2], z = (zk >> 1) + (o - 9 * o9 - 3 * o3) - 1;
                if (x >= c.gx || y >= c.gy || z >= c.gz || x < 0 || y < 0 || z < 0) continue;
                const int slot = cell_slot_xyz(c, x, y, z, x + y * c.sy + z * c.sz);
                if (slot < 0) continue;
                if (zk & 1) {
                    const int a = cell_start[slot], b = cell_start[slot + 1];
                    for (int tt = a / kBlock; b > a && tt <= (b - 1) / kBlock; ++tt) tile_set_insert(s_tset, tt);
                    continue;
                }
                int hq = stage_hash(slot);
                for (int probe = 0; probe < kStageHash; ++probe) {
                    const int was = atomicCAS(&s_key[hq], -1, slot);
                    if (was == slot) break;
                    if (was == -1) { if (atomicAdd(&s_ncell, 1) >= kStageMaxCells) s_ok = 0; break; }
                    if (s_ok == 0) break;
                    hq = (hq + 1) & (kStageHash - 1);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) s_stage[q * kBlock + threadIdx.x] = 0;       // (the head list is done with)
        __syncthreads();
        // (2) local base of every cell of the set (table order), the ordered source list, the verdict
        int own[kStageHash / kBlock], cfirst[kStageHash / kBlock], run = 0, rig = 0;
        const bool ok = s_ok != 0;
#pragma unroll
        for (int q = 0; q < kStageHash / kBlock; ++q) {
            const int key = ok ? s_key[threadIdx.x * (kStageHash / kBlock) + q] : -1;
            cfirst[q] = key >= 0 ? cell_start[key] : 0;
            own[q] = key >= 0 ? cell_start[key + 1] - cfirst[q] : 0;
            run += own[q];
            if (RIGID && key >= 0) rig |= rv.rcell_start[key + 1] - rv.rcell_start[key];      // rigid samples in a cell this workgroup walks
        }
        const bool near_body = RIGID && __syncthreads_or(rig) != 0;
        int nonempty = 0;
#pragma unroll
        for (int q = 0; q < kStageHash / kBlock; ++q) nonempty += own[q] > 0 ? 1 : 0;
        const int inc = wave_inclusive_scan(run), inc_ne = wave_inclusive_scan(nonempty);     // particles / non-empty cells before this thread's entries
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        if (lane == 63) { s_wsum[w] = inc; s_wsum_ne[w] = inc_ne; }
        __syncthreads();
        int before = inc - run, run_idx = inc_ne - nonempty, total = 0, nruns = 0;
        for (int k = 0; k < kBlock / 64; ++k) {
            if (k < w) { before += s_wsum[k]; run_idx += s_wsum_ne[k]; }
            total += s_wsum[k]; nruns += s_wsum_ne[k];
        }
        const bool staged = ok && total <= c.stage_cap;          // (ok: at most kStageMaxCells cells, so nruns fits the run table)
        // The ordered source list of the set, as RUNS: a cell's particles are contiguous in the sorted arrays, so (first index, local
        // base | count << 16) per non-empty cell describes it -- ~1.2 KB per workgroup where the flat list was 5.2 KB, re-read by every sweep.
#pragma unroll
        for (int q = 0; q < kStageHash / kBlock; ++q) {
            const int e = threadIdx.x * (kStageHash / kBlock) + q;
            s_base[e] = before;
            if (staged && own[q] > 0) {
                const uint32_t first = (uint32_t)cfirst[q];
                stage_runs[(size_t)blk * kStageMaxCells + run_idx] = make_uint2(first, (uint32_t)before | ((uint32_t)own[q] << 16));
                ++run_idx;
            }
            before += own[q];
            if (tile_nbr && own[q] > 0)           // the tiles this cell's particles lie in (staged or not: the cell set is complete whenever `ok`)
                for (int t = cfirst[q] / kBlock; t <= (cfirst[q] + own[q] - 1) / kBlock; ++t) tile_set_insert(s_tset, t);
        }
        if (threadIdx.x == 0) {
            const bool l16 = staged && c.nl16 != 0 && !near_body;
            stage_cnt[blk] = staged ? (total | (nruns << 16) | (l16 ? kStageLists16 : 0)) : -1;
            s_ok = staged ? (l16 ? 3 : 1) : 0;
        }
        __syncthreads();
        if (tile_nbr && (int)(threadIdx.x >> 6) == kBlock / 64 - 1) {       // the set, compacted into this tile's row (the wave whose table held it)
            const int ln = threadIdx.x & 63;
            const int ta = s_tset[ln], tb = s_tset[ln + 64];
            const int na = (ta >= 0 ? 1 : 0) + (tb >= 0 ? 1 : 0);
            const int inc = wave_inclusive_scan(na);
            const int ntile = __shfl(inc, 63, 64);
            int *row = tile_nbr + (size_t)blk * kNbrStride;
            const bool fits = ok && ntile <= c.nbr_cap && ntile < kTileSet;             // (a full table may have dropped a tile)
            int pos = inc - na;
            if (fits && ta >= 0) row[1 + pos++] = ta;
            if (fits && tb >= 0) row[1 + pos] = tb;
            if (ln == 0) row[0] = fits ? (ntile | (s_odd ? kNbrOdd : 0)) : -1;
        }
    }
    constexpr int CHUNK = 4;
    const bool staged = STAGED && s_ok != 0;
    // (3) the walk.  What a particle needs to know about each of its 27 cells -- where the cell's fluid and wall particles start, how
    // many there are, the cell's base in the staged set -- used to be worked out by every lane for itself (the slot on the curve
    // through the tile-rank table, four dependent loads of cell bounds, a hash probe: ~100 instructions and two round trips to memory
    // per cell, a third of the kernel).  But the lanes of a wave are ~9 runs of particles of the SAME cell: per dx-plane the wave
    // now fills a table of (run, cell) entries with all lanes working (9 runs x 9 cells = 81 entries on 64 lanes) and every lane
    // reads its run's entry from LDS.  A wave with more runs than the table holds (sparse spray) works the entries out per lane.
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float4 pi = P[i < c.n ? i : 0];
    const f32x2 pi_xy = {pi.x, pi.y};
    const unsigned cut_bits = __float_as_uint(c.r2_cut);
    int cx, cy, cz;
    cell_id_of(c, pi.x, pi.y, pi.z, cx, cy, cz);
    const int my_raw_id = i < c.n ? id[i] : 0;
    const bool walker = i < c.n && list_walker(c, my_raw_id, cx);
    const int ghost_bit = my_raw_id < 0 ? (int)0x80000000 : 0;      // ghost (multi-GPU): owns no sums; without a list it takes part as a neighbour only
    if (i < c.n && !walker) cnt[i] = (int)0x80000000;
    const int pcx = __shfl_up(cx, 1, 64), pcy = __shfl_up(cy, 1, 64), pcz = __shfl_up(cz, 1, 64);
    const bool pwalker = __shfl_up(walker ? 1 : 0, 1, 64) != 0;
    const bool rhead = walker && (lane == 0 || !pwalker || pcx != cx || pcy != cy || pcz != cz);
    const unsigned long long rheads = __ballot(rhead);
    const int nruns = __popcll(rheads);
    const int run = __popcll(rheads & ((2ull << lane) - 1ull)) - 1;          // this lane's run (walkers only)
    const bool table = nruns <= kRunCap;                                       // wave-uniform
    if (table && rhead) { s_runc[wv][run][0] = cx; s_runc[wv][run][1] = cy; s_runc[wv][run][2] = cz; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool lists16 = STAGED && (s_ok & 2) != 0;           // (see kStageLists16)
    NlWriter wf{&s_stage[threadIdx.x], nl + nl_index(i < c.n ? i : 0, 0, c.kpitch), 0, c.kmax, lists16, 0u, !lists16 && c.nl16 != 0};
    NlWriter ww{nullptr, nlb + nl_index(i < c.n ? i : 0, 0, c.kbpitch), 0, c.kbmax, false, 0u, false};
    int nq = 0;                       // get_neighbour_count with its rigid-entry quirk (RIGID only)
    uint32_t self_local = 0u;         // this particle's own index in the staged set (16-bit lists pad their last group with it)
    const int my_id = RIGID && walker ? id[i] : 0;
    // one cell of a 27-neighbourhood: (first fluid particle, fluid count | staged base << 16, first wall particle, wall count[, slot])
    auto cell_entry = [&](int ccx, int ccy, int ccz, int dx, int o9, uint4 &e, int &eslot) {
        const int t3 = (o9 * 11) >> 5;                                          // o9 / 3 for 0 <= o9 < 9
        const int x = ccx + dx, y = ccy + t3 - 1, z = ccz + (o9 - 3 * t3) - 1;
        e = make_uint4(0u, 0u, 0u, 0u); eslot = -1;
        if (x >= c.gx || y >= c.gy || z >= c.gz || x < 0 || y < 0 || z < 0) return;   // :453-456
        const int cid = x + y * c.sy + z * c.sz;
        const int slot = cell_slot_xyz(c, x, y, z, cid);
        int a = 0, nf = 0, lbase = 0;
        if (slot >= 0) {                                                        // (< 0: a column this slab does not hold -- no fluid there for us)
            a = cell_start[slot]; nf = min(cell_start[slot + 1] - a, 0xffff);
            lbase = staged ? stage_lookup(s_key, s_base, slot, ds) : 0;
        }
        int wa = 0, nw = 0;
        if (c.boundary_handle) { wa = wcell_start[cid]; nw = wcell_start[cid + 1] - wa; }
        e = make_uint4((uint32_t)a, (uint32_t)nf | ((uint32_t)lbase << 16), (uint32_t)wa, (uint32_t)nw);
        eslot = slot;
    };
    for (int dx = -1; dx <= 1; ++dx) {
        if (table) {
            const int nent = nruns * 9;
            for (int t = lane; t < nent; t += 64) {
                const int r = (t * 57) >> 9, o9 = t - 9 * r;                    // t / 9 for 0 <= t < 144
                uint4 e; int eslot;
                cell_entry(s_runc[wv][r][0], s_runc[wv][r][1], s_runc[wv][r][2], dx, o9, e, eslot);
                s_cell[wv][t] = e;
                if (RIGID) s_cslot[RIGID ? wv : 0][RIGID ? t : 0] = eslot;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (walker) {
            for (int o9 = 0; o9 < 9; ++o9) {
                uint4 e; int eslot = -1;
                if (table) { e = s_cell[wv][run * 9 + o9]; if (RIGID) eslot = s_cslot[RIGID ? wv : 0][RIGID ? run * 9 + o9 : 0]; }
                else cell_entry(cx, cy, cz, dx, o9, e, eslot);
                const int a = (int)e.x, b = a + (int)(e.y & 0xffffu);
                const int lbase = staged ? (int)(e.y >> 16) - a : 0;            // local index = lbase + j
                // The accept bits of a cell's candidates are collected first (four candidates per trip, branch-free; up to 32 per chunk),
                // then the (few) accepted ones are appended in order: the append loop runs as often as the busiest lane has bits, and
                // per cell that is ~half of what it was per trip.  (One 32-bit byte offset per trip, the four loads differ by
                // immediates; reading up to three slots past the cell is harmless: the arrays carry 64 spare elements and the chunk's
                // validity mask drops them.)
                for (int j0 = a; j0 < b; j0 += CHUNK) {
                    const int nb = b - j0 < CHUNK ? b - j0 : CHUNK;
                    const float4 *pb = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(P) + (unsigned)j0 * 16u);
                    unsigned far = 0;
#pragma unroll
                    for (int t = 0; t < CHUNK; t += 4) far |= far_mask4(pi_xy, pi.z, pb + t, cut_bits) << t;   // :466 (norm > h)
                    unsigned m = ~far & (0xffffffffu >> (32 - nb));                              // candidates of this cell only
                    const unsigned self = (unsigned)(i - j0);                                    // :461 (j != i)
                    if (self < (unsigned)CHUNK) m &= ~(1u << self);
                    if (self < (unsigned)nb) self_local = (uint32_t)(lbase + i);                // (a chunk may reach past its cell: nb, not CHUNK)
                    if (RIGID) nq += __popc(m);
                    while (m) {
                        const int u = __ffs(m) - 1;
                        m &= m - 1;
                        wf.push((uint32_t)(lbase + j0 + u));       // staged workgroups keep LOCAL indices
                    }
                }
                if (RIGID && eslot >= 0) {
                    // rigid entries of the cell come after its fluid entries (update_grid, :383-386)
                    const int ra = rv.rcell_start[eslot], rb = rv.rcell_start[eslot + 1];
                    for (int j = ra; j < rb; ++j) {
                        const float4 pj = rv.RP[j];
                        float ddx = pi.x - pj.x, ddy = pi.y - pj.y, ddz = pi.z - pj.z;
                        float r2 = (ddx * ddx + ddy * ddy) + ddz * ddz;
                        if (!(r2 > c.r2_cut)) wf.push((uint32_t)j | kRigidTag);
                        // get_neighbour_count (:436-444): skips when particle_j.index == i (the rigid particle's LOCAL index) and
                        // measures the distance to fluid_particles.pos[particle_j.index]
                        const int jl = rv.rid[j];
                        if (jl != my_id && jl < rv.n_fluid) {
                            const float4 pq = rv.pos_orig[jl];
                            float ex = pi.x - pq.x, ey = pi.y - pq.y, ez = pi.z - pq.z;
                            float e2 = (ex * ex + ey * ey) + ez * ez;
                            if (!(e2 > c.r2_cut)) ++nq;
                        }
                    }
                }
                const int wa = (int)e.z, wb = wa + (int)e.w;
                for (int j0 = wa; j0 < wb; j0 += 32) {
                    const int nb = wb - j0 < 32 ? wb - j0 : 32;
                    unsigned far = 0;
                    for (int t = 0; t < nb; t += 4) {
                        const float4 *pb = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(WP) + (unsigned)(j0 + t) * 16u);
                        far |= far_mask4(pi_xy, pi.z, pb, cut_bits) << t;                        // :364
                    }
                    unsigned m = ~far & (0xffffffffu >> (32 - nb));
                    while (m) {
                        const int u = __ffs(m) - 1;
                        m &= m - 1;
                        ww.push_direct((uint32_t)(j0 + u));
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (walker) {
        wf.flush(self_local);
        kf = wf.k; kb = ww.k;
        int kfc = kf < c.kmax ? kf : c.kmax, kbc = kb < c.kbmax ? kb : c.kbmax;
        cnt[i] = kfc | (kbc << 16) | ghost_bit;
        if (RIGID) ncount[i] = nq;
    }
    note_list_lengths(c, kf, kb, ds);
}

// ---- the list build for small scenes: one wave per dx-plane -------------------------------------------------------------
// Below about a wave per SIMD (config 1: 29 k particles = 455 waves on 1024 SIMDs) k_build_nl is as long as ONE wave's walk of its
// 27 cells, and the chip idles.  Here a workgroup of three waves builds the lists of 64 particles: wave p walks the nine cells of the
// plane dx = p - 1 for the same 64 particles, so the walk is a third as long and three times as many waves are in flight.  A
// particle's list is the concatenation of its planes' entries (the reference's dx-outermost order), and a plane's position in the
// list is only known once the planes before it have been counted -- two passes:
//   1. every wave tests its plane's candidates, keeps the accept masks (LDS, one word per cell and lane) and counts;
//   2. after a barrier every wave knows where its entries start and writes them (4-byte stores into the 16-byte groups of the tiled
//      layout; at this size the store count does not matter).  Only the first 32 candidates of a cell have a kept mask; a fuller
//      cell, rigid entries and wall candidates are tested again in pass 2.
// Same lists, counts and health counters as k_build_nl<RIGID, false, *>; unstaged handles only (the launch picks it by size).
struct CellEntry { int a, nf, wa, nw, slot; };
__device__ __forceinline__ CellEntry split_cell_entry(const Consts &c, const int *__restrict__ cell_start, const int *__restrict__ wcell_start,
                                                       int ccx, int ccy, int ccz, int o27)
{
    const int t9 = (o27 * 57) >> 9, o9 = o27 - 9 * t9, t3 = (o9 * 11) >> 5;    // o27 / 9 and o9 / 3 for 0 <= o27 < 27
    const int x = ccx + t9 - 1, y = ccy + t3 - 1, z = ccz + (o9 - 3 * t3) - 1;
    CellEntry e = {0, 0, 0, 0, -1};
    if (x >= c.gx || y >= c.gy || z >= c.gz || x < 0 || y < 0 || z < 0) return e;     // :453-456
    const int cid = x + y * c.sy + z * c.sz;
    e.slot = cell_slot_xyz(c, x, y, z, cid);
    if (e.slot >= 0) {
        e.a = cell_start[e.slot];
        e.nf = cell_start[e.slot + 1] - e.a;
    }
    if (c.boundary_handle) { e.wa = wcell_start[cid]; e.nw = wcell_start[cid + 1] - e.wa; }
    return e;
}
// accept bits of up to 32 candidates A[j0 .. j0 + nb): twelve candidates per batch of loads
__device__ __forceinline__ unsigned split_accept32(const float4 *__restrict__ A, int j0, int nb, f32x2 pi_xy, float pi_z, unsigned cut_bits)
{
    unsigned far = 0;
    for (int t = 0; t < nb; t += 12) {
        const float4 *pb = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(A) + (unsigned)(j0 + t) * 16u);
        unsigned f = far_mask4(pi_xy, pi_z, pb, cut_bits) | far_mask4(pi_xy, pi_z, pb + 4, cut_bits) << 4 | far_mask4(pi_xy, pi_z, pb + 8, cut_bits) << 8;
        far |= f << t;                                                          // (t = 24: the bits past 32 fall off, nb <= 32 masks them anyway)
    }
    return ~far & (0xffffffffu >> (32 - nb));
}
template <bool RIGID, int NW>
__global__ __launch_bounds__(NW * 64) void k_build_nl_split(Consts c, const float4 *__restrict__ P, const int *__restrict__ cell_start,
                                                                      const float4 *__restrict__ WP, const int *__restrict__ wcell_start,
                                                                      const int *__restrict__ id, uint32_t *__restrict__ nl,
                                                                      uint32_t *__restrict__ nlb, int *__restrict__ cnt, DevScalars *__restrict__ ds,
                                                                      RigidView rv, int *__restrict__ ncount, const int *__restrict__ gate = nullptr)
{
    if (gate && *gate == 0) return;
    constexpr int CPW = 27 / NW;                                                  // cells per wave: 9 (a dx-plane) or 3 (a (dx, dy) column)
    __shared__ uint4 s_cell[NW][kRunCap * CPW];
    __shared__ int s_cslot[RIGID ? NW : 1][RIGID ? kRunCap * CPW : 1], s_runc[NW][kRunCap][3];
    __shared__ uint32_t s_mask[NW][CPW][64];
    __shared__ int s_cf[NW][64], s_cw[NW][64], s_nq[RIGID ? NW : 1][64];
    const int lane = threadIdx.x & 63, plane = threadIdx.x >> 6, o_first = plane * CPW;
    const int blk = xcd_block(blockIdx.x, gridDim.x);
    const int i = blk * 64 + lane;
    if (i == 0 && plane == 0) ds->lost = cell_start[c.S + 1] - cell_start[c.S];   // size of the "outside the grid" bucket
    const float4 pi = P[i < c.n ? i : 0];
    const f32x2 pi_xy = {pi.x, pi.y};
    const unsigned cut_bits = __float_as_uint(c.r2_cut);
    int cx, cy, cz;
    cell_id_of(c, pi.x, pi.y, pi.z, cx, cy, cz);
    const int my_raw_id = i < c.n ? id[i] : 0;
    const bool walker = i < c.n && list_walker(c, my_raw_id, cx);
    const int ghost_bit = my_raw_id < 0 ? (int)0x80000000 : 0;                    // ghost (multi-GPU), see k_build_nl
    if (i < c.n && !walker && plane == 0) cnt[i] = (int)0x80000000;
    const int pcx = __shfl_up(cx, 1, 64), pcy = __shfl_up(cy, 1, 64), pcz = __shfl_up(cz, 1, 64);
    const bool pwalker = __shfl_up(walker ? 1 : 0, 1, 64) != 0;
    const bool rhead = walker && (lane == 0 || !pwalker || pcx != cx || pcy != cy || pcz != cz);
    const unsigned long long rheads = __ballot(rhead);
    const int nruns = __popcll(rheads);
    const int run = __popcll(rheads & ((2ull << lane) - 1ull)) - 1;
    const bool table = nruns <= kRunCap;                                          // wave-uniform
    if (table && rhead) { s_runc[plane][run][0] = cx; s_runc[plane][run][1] = cy; s_runc[plane][run][2] = cz; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (table) {
        for (int t = lane; t < nruns * CPW; t += 64) {
            const int r = CPW == 9 ? (t * 57) >> 9 : (t * 171) >> 9, oc = t - CPW * r;     // t / 9 (t < 144), t / 3 (t < 48)
            const CellEntry e = split_cell_entry(c, cell_start, wcell_start, s_runc[plane][r][0], s_runc[plane][r][1], s_runc[plane][r][2], o_first + oc);
            s_cell[plane][t] = make_uint4((uint32_t)e.a, (uint32_t)e.nf, (uint32_t)e.wa, (uint32_t)e.nw);
            if (RIGID) s_cslot[RIGID ? plane : 0][RIGID ? t : 0] = e.slot;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    auto entry = [&](int oc) {
        if (!table) return split_cell_entry(c, cell_start, wcell_start, cx, cy, cz, o_first + oc);
        const uint4 w = s_cell[plane][run * CPW + oc];
        CellEntry e = {(int)w.x, (int)w.y, (int)w.z, (int)w.w, RIGID ? s_cslot[RIGID ? plane : 0][RIGID ? run * CPW + oc : 0] : 0};
        return e;
    };
    const int my_id = RIGID && walker ? id[i] : 0;
    // the walk of this plane's nine cells; WRITE = false counts (and keeps the first accept mask of every cell), WRITE = true stores
    int kf = 0, kb = 0, nq = 0;
    auto walk = [&](auto write_tag, int kf0, int kb0) {
        constexpr bool WRITE = decltype(write_tag)::value;
        uint32_t *lf = nl + nl_index(i < c.n ? i : 0, 0, c.kpitch), *lw = nlb + nl_index(i < c.n ? i : 0, 0, c.kbpitch);
        int k = kf0, kw = kb0;
        auto put_f = [&](uint32_t j) { if (WRITE && k < c.kmax) lf[(size_t)(k >> 2) * 256 + (k & 3)] = j; ++k; };
        auto put_w = [&](uint32_t j) { if (WRITE && kw < c.kbmax) lw[(size_t)(kw >> 2) * 256 + (kw & 3)] = j; ++kw; };
        for (int o9 = 0; o9 < CPW; ++o9) {
            const CellEntry e = entry(o9);
            for (int j0 = e.a; j0 < e.a + e.nf; j0 += 32) {
                const int nb = e.a + e.nf - j0 < 32 ? e.a + e.nf - j0 : 32;
                unsigned m;
                if (WRITE && j0 == e.a) {
                    m = s_mask[plane][o9][lane];
                } else {
                    m = split_accept32(P, j0, nb, pi_xy, pi.z, cut_bits);                    // :466 (norm > h)
                    const unsigned self = (unsigned)(i - j0);                                // :461 (j != i)
                    if (self < 32u) m &= ~(1u << self);
                    if (!WRITE && j0 == e.a) s_mask[plane][o9][lane] = m;
                }
                if (WRITE) {
                    while (m) { const int u = __ffs(m) - 1; m &= m - 1; put_f((uint32_t)(j0 + u)); }
                } else {
                    k += __popc(m);
                    if (RIGID) nq += __popc(m);
                }
            }
            if (RIGID && e.slot >= 0) {
                // rigid entries of the cell come after its fluid entries (update_grid, :383-386)
                const int ra = rv.rcell_start[e.slot], rb = rv.rcell_start[e.slot + 1];
                for (int j = ra; j < rb; ++j) {
                    const float4 pj = rv.RP[j];
                    const float ddx = pi.x - pj.x, ddy = pi.y - pj.y, ddz = pi.z - pj.z;
                    const float r2 = (ddx * ddx + ddy * ddy) + ddz * ddz;
                    if (!(r2 > c.r2_cut)) put_f((uint32_t)j | kRigidTag);
                    if (!WRITE) {
                        // get_neighbour_count (:436-444): skips when particle_j.index == i (the rigid particle's LOCAL index) and
                        // measures the distance to fluid_particles.pos[particle_j.index]
                        const int jl = rv.rid[j];
                        if (jl != my_id && jl < rv.n_fluid) {
                            const float4 pq = rv.pos_orig[jl];
                            const float ex = pi.x - pq.x, ey = pi.y - pq.y, ez = pi.z - pq.z;
                            const float e2 = (ex * ex + ey * ey) + ez * ez;
                            if (!(e2 > c.r2_cut)) ++nq;
                        }
                    }
                }
            }
            for (int j0 = e.wa; j0 < e.wa + e.nw; j0 += 32) {
                const int nb = e.wa + e.nw - j0 < 32 ? e.wa + e.nw - j0 : 32;
                unsigned m = split_accept32(WP, j0, nb, pi_xy, pi.z, cut_bits);              // :364
                if (WRITE) { while (m) { const int u = __ffs(m) - 1; m &= m - 1; put_w((uint32_t)(j0 + u)); } }
                else kw += __popc(m);
            }
        }
        kf = k; kb = kw;
    };
    if (walker) walk(std::false_type{}, 0, 0);
    s_cf[plane][lane] = kf; s_cw[plane][lane] = kb;
    if (RIGID) s_nq[RIGID ? plane : 0][lane] = nq;
    __syncthreads();
    int kf0 = 0, kb0 = 0, kft = 0, kbt = 0, nqt = 0;
#pragma unroll
    for (int p = 0; p < NW; ++p) {
        if (p < plane) { kf0 += s_cf[p][lane]; kb0 += s_cw[p][lane]; }
        kft += s_cf[p][lane]; kbt += s_cw[p][lane];
        if (RIGID) nqt += s_nq[RIGID ? p : 0][lane];
    }
    if (walker) walk(std::true_type{}, kf0, kb0);
    if (plane == 0) {
        if (walker) {
            cnt[i] = (kft < c.kmax ? kft : c.kmax) | ((kbt < c.kbmax ? kbt : c.kbmax) << 16) | ghost_bit;
            if (RIGID) ncount[i] = nqt;
        }
        note_list_lengths(c, walker ? kft : 0, walker ? kbt : 0, ds);
    }
}

// Relaxed arithmetic with a coupled rigid body (round 4): the tolerance-grade sweeps cover the workgroups whose lists are 16-bit (no rigid sample in any
// cell of their neighbourhood: all but a thin shell around the body, kStageLists16), the exact RIGID sweeps the rest -- two launches per sweep over the
// two halves of one tile order (k_tile_order in sph_slab_kernels.h: flagged tiles first).  flag[t] = 1: tile t keeps the exact kernels.
__global__ __launch_bounds__(kBlock) void k_tile_flags_exact(const int *__restrict__ stage_cnt, int ntiles, int *__restrict__ flag)
{
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t < ntiles) flag[t] = stage_lists16(stage_cnt, t) ? 0 : 1;
}

// ======================================================================================
// helpers for the list-driven sweeps
// ======================================================================================
// QUAD sweeps: 64 particles per workgroup, lane q = threadIdx.x & 3 of the quad serving particle blk * 64 + threadIdx.x / 4; every lane of a quad
// carries the same accumulators, lane 0 writes the results (`owner`)
#define SPH_SWEEP_PROLOGUE_M(QUAD) SPH_SWEEP_PROLOGUE_B(QUAD, xcd_block(blockIdx.x, gridDim.x))
// ... with the workgroup -> tile mapping chosen by the caller
#define SPH_SWEEP_PROLOGUE_B(QUAD, BLK) SPH_SWEEP_PROLOGUE_G(QUAD, BLK, false)
// GW ("ghosts walk"): on two-column slab handles the ghosts of the inner column have lists of their own (k_build_nl: list_walker); the sweeps
// that must run on them -- D1 and the corrections D2 / D4 / D7 -- pass true, every other sweep sees a ghost's list as empty
#define SPH_SWEEP_PROLOGUE_G(QUAD, BLK, GW)                  \
    const int blk = (BLK);                                   \
    const int q = (QUAD) ? (int)(threadIdx.x & 3) : 0;       \
    (void)q;                                                 \
    int i = (QUAD) ? blk * (kBlock / 4) + (int)(threadIdx.x >> 2) : blk * kBlock + (int)threadIdx.x; \
    const bool live = i < c.n;                               \
    const bool owner = live && q == 0;                       \
    (void)owner;                                             \
    const int ii = live ? i : 0;                             \
    const int cw = live ? cnt[ii] : 0;                       \
    const bool ghost = cw < 0;                               \
    (void)ghost;                                             \
    const int kf = (!(GW) && ghost) ? 0 : (cw & 0xffff), kb = (!(GW) && ghost) ? 0 : ((cw >> 16) & 0x7fff); \
    const float4 pi = P[ii];                                 \
    const uint32_t *nlp = nl + nl_index(ii, 0, c.kpitch);     \
    const uint32_t *nlbp = nlb ? nlb + nl_index(ii, 0, c.kbpitch) : nullptr;
#define SPH_SWEEP_PROLOGUE                                   \
    const int blk = xcd_block(blockIdx.x, gridDim.x);        \
    int i = blk * kBlock + threadIdx.x;                      \
    const bool live = i < c.n;                               \
    const int ii = live ? i : 0;                             \
    const int cw = live ? cnt[ii] : 0;                       \
    const bool ghost = cw < 0;                               \
    (void)ghost;                                             \
    const int kf = ghost ? 0 : (cw & 0xffff), kb = ghost ? 0 : ((cw >> 16) & 0x7fff); \
    const float4 pi = P[ii];                                 \
    const uint32_t *nlp = nl + nl_index(ii, 0, c.kpitch);     \
    const uint32_t *nlbp = nlb ? nlb + nl_index(ii, 0, c.kbpitch) : nullptr;

// Walk of a neighbour list in groups of four: one 16-byte index load, four independent float4
// gathers in flight, the next group's indices requested before the four bodies run.  Bodies run in
// list order, so every accumulator sees its terms in the canonical order.  Rows past a particle's
// count hold stale but valid indices (the buffer is zero-initialised, only ever holds indices < n,
// and one spare tile pads the end), so the speculative loads are always in bounds.
template <class Body>
__device__ __forceinline__ void for_nbrs_p(const uint32_t *__restrict__ base, int cnt, const float4 *__restrict__ A, Body body)
{
    walk_list<Operand1>(base, cnt, [&](uint32_t j, Operand1 &o) { o.a = A[j]; }, [&](const Operand1 &o, uint32_t) { body(o.a); });
}
template <int N, class Body>
__device__ __forceinline__ void for_nbrs_p_quad(const uint32_t *__restrict__ base, int cnt, int q, float (&acc)[N], const float4 *__restrict__ A, Body body)
{
    walk_list_quad<N, Operand1>(base, cnt, q, acc, [&](uint32_t j, Operand1 &o) { o.a = A[j]; }, [&](const Operand1 &o, uint32_t) { body(o.a); });
}

// ---- the wall terms of the DFSPH solver loops from a per-step cache ------------------------------------------------------------
// Walls are static and the positions are frozen between the list build and the integrator, so grad W_ib of a (particle, wall
// particle) pair is the same f32 triple in every sweep of a step: D1 (k_density<DFSPH>) evaluates it anyway and leaves
// (grad W_ib, V_b) in wall_gc, laid out like the wall list it belongs to -- entry k of particle i at gc_index(i, k): a wave's
// row is 1 KiB contiguous -- and D2-D7 read it back instead of gathering the wall particle and re-deriving the gradient (~65
// instructions per pair in the exact arithmetic).  Same bits: the same function of the same inputs, evaluated once.
__device__ __forceinline__ size_t gc_index(int i, int k, int pitch) { return ((size_t)(i >> 6) * pitch + k) * 64 + (size_t)(i & 63); }
template <int DEPTH = 4, class Body>
__device__ __forceinline__ void for_wall_cache(const float4 *__restrict__ base, int cnt, Body body)
{
    if (cnt <= 0) return;
    // rows in groups of DEPTH, the next group requested before the bodies of this one run (each row is its own trip to HBM); the group past the
    // end reads stale but mapped rows (the rest of the pitch, the next tile's first rows, the spare tile behind the last one)
    float4 cur[DEPTH], nxt[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) cur[u] = base[(size_t)u * 64];
    for (int k = 0; k < cnt; k += DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) nxt[u] = base[(size_t)(k + DEPTH + u) * 64];
        body(cur[0]);
#pragma unroll
        for (int u = 1; u < DEPTH; ++u)
            if (k + u < cnt) body(cur[u]);
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) cur[u] = nxt[u];
    }
}

// block partial of (sum over lanes with flag, count) in a fixed order -> deterministic
__device__ __forceinline__ void block_partial_mean(int blk, double v, int flag, double *__restrict__ psum, int *__restrict__ pcnt)
{
    __shared__ double s_sum[kBlock / 64];
    __shared__ int s_cnt[kBlock / 64];
    double ws = wave_sum(flag ? v : 0.0);
    int wc = wave_sum(flag);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { s_sum[w] = ws; s_cnt[w] = wc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0; int n = 0;
        for (int k = 0; k < kBlock / 64; ++k) { t += s_sum[k]; n += s_cnt[k]; }
        psum[blk] = t; pcnt[blk] = n;
    }
}
// QUAD sweeps: a workgroup holds 64 particles, one per quad -- exactly one WAVE of a one-lane-per-particle workgroup.  Its partial is
// that wave's butterfly: the owners' values go through LDS into particle order and wave 0 reduces them with the same tree.
// psum / pcnt then hold one entry per 64 particles; k_finalize_mean (group = 4) first adds four consecutive entries in order, which is
// the serial sum over the four waves of the 256-particle block above: same bits.
__device__ __forceinline__ void block_partial_mean_quad(int blk, double v, int flag, bool owner_lane, double *__restrict__ psum, int *__restrict__ pcnt)
{
    __shared__ double s_v[kBlock / 4];
    __shared__ int s_f[kBlock / 4];
    if ((threadIdx.x & 3) == 0) { s_v[threadIdx.x >> 2] = (owner_lane && flag) ? v : 0.0; s_f[threadIdx.x >> 2] = (owner_lane && flag) ? 1 : 0; }
    __syncthreads();
    if (threadIdx.x < 64) {
        const double ws = wave_sum(s_v[threadIdx.x]);
        const int wc = wave_sum(s_f[threadIdx.x]);
        if (threadIdx.x == 0) { psum[blk] = ws; pcnt[blk] = wc; }
    }
}

__device__ __forceinline__ void block_partial_max(int blk, float v, float *__restrict__ pmax)
{
    __shared__ float s_max[kBlock / 64];
    float wm = wave_max(v);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s_max[w] = wm;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = s_max[0];
        for (int k = 1; k < kBlock / 64; ++k) t = fmaxf(t, s_max[k]);
        pmax[blk] = t;
    }
}

// final, fixed-order reduction of the block partials; one block.  The host forms
// mean = cnt > 0 ? sum / cnt : default   (dfsph_solver.py:148-149, 278-279) after all-reducing (sum, cnt) when sharded.
enum { FIN_PLAIN = 0, FIN_DIV_FIRST = 1, FIN_DIV_LOOP = 2, FIN_DENS = 3 };

// phase FINP_ALL: reduce and decide in one launch (single GPU).  Sharded runs split it around the all-reduce over the slabs:
// FINP_REDUCE writes this slab's (sum, count) to red[0..1]; FINP_DECIDE takes the decision from the reduced pair in red.
enum { FINP_ALL = 0, FINP_REDUCE = 1, FINP_DECIDE = 2 };

// One workgroup of kFinBlock threads: every thread adds the partials t, t + kFinBlock, ... in ascending order (one batch of loads at
// 1 M particles: 3907 partials), a wave butterfly, then thread 0 adds the 16 wave sums in order -- one barrier.  (With 256 threads,
// two load rounds and an eight-level LDS tree this kernel took 5.2 us, 29 times per step.)  The order is fixed, hence deterministic.
constexpr int kFinBlock = 1024;
// group = 4: psum / pcnt hold one entry per 64 particles (QUAD sweeps, block_partial_mean_quad); `nblocks` still counts blocks of 256 particles and
// `nparts` the entries: a block's partial is the in-order sum of its (up to) four entries.
__device__ __forceinline__ void fin_partial(const double *__restrict__ psum, const int *__restrict__ pcnt, int e, int nblocks, int group, int nparts, double &v, int &m)
{
    v = 0.0; m = 0;
    if (e >= nblocks) return;
    if (group == 1) { v = psum[e]; m = pcnt[e]; return; }
    for (int u = 0; u < group; ++u) {
        const int k = e * group + u;
        if (k < nparts) { v += psum[k]; m += pcnt[k]; }          // 0.0 + w0 = w0: the serial sum over the block's waves
    }
}
// what "virtual thread" vt of a kFinBlock-thread workgroup adds up: the partials vt, vt + kFinBlock, ... in ascending order
__device__ __forceinline__ void fin_thread_sum(const double *__restrict__ psum, const int *__restrict__ pcnt, int vt, int nblocks, int group, int nparts, double &t, int &n)
{
    t = 0.0; n = 0;
    int k = vt;
    for (; k + 3 * kFinBlock < nblocks; k += 4 * kFinBlock) {
        double v[4]; int m[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) fin_partial(psum, pcnt, k + u * kFinBlock, nblocks, group, nparts, v[u], m[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) { t += v[u]; n += m[u]; }
    }
    {   // the rest, still as one batch of (predicated) loads
        double v[4]; int m[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) fin_partial(psum, pcnt, k + u * kFinBlock, nblocks, group, nparts, v[u], m[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) { t += v[u]; n += m[u]; }       // + 0.0 leaves a non-negative-zero sum unchanged
    }
}
// the reduction itself, for a workgroup of kFinBlock threads: (sum, count) over the block partials end up in s_sum[0], s_cnt[0] (thread 0's view)
__device__ __forceinline__ void fin_reduce(const double *__restrict__ psum, const int *__restrict__ pcnt, int nblocks, int group, int nparts,
                                           double *__restrict__ s_sum, long long *__restrict__ s_cnt)
{
    double t; int n;
    fin_thread_sum(psum, pcnt, threadIdx.x, nblocks, group, nparts, t, n);
    const double ws = wave_sum(t);
    const int wn = wave_sum(n);
    if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = ws; s_cnt[threadIdx.x >> 6] = wn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tt = 0.0; long long nn = 0;
        for (int w = 0; w < kFinBlock / 64; ++w) { tt += s_sum[w]; nn += s_cnt[w]; }
        s_sum[0] = tt; s_cnt[0] = nn;
    }
}
// the reference's host logic, evaluated where the data is (same f64 compares as the Python host code); one thread
__device__ __forceinline__ void fin_decide(DevScalars *__restrict__ ds, int mode, double sum, long long cnt)
{
    ds->sum = sum; ds->cnt = cnt;
    if (mode == FIN_PLAIN) return;
    if (mode == FIN_DIV_FIRST || mode == FIN_DIV_LOOP) {
        const float err = cnt > 0 ? (float)(sum / (double)cnt) : 0.0f;   // dfsph_solver.py:278-279
        int it = ds->div_it;
        int active;
        if (mode == FIN_DIV_FIRST) {                                     // :398-399
            ds->div_first = err; ds->div_err = err; ds->div_evals = 1;
            active = 1;
        } else {                                                         // :406-414
            const float past = ds->div_err;
            ds->div_past = past; ds->div_err = err; ds->div_evals += 1;
            if (fabs((double)err - (double)past) < 1e-5) active = 0;    // break before iter_cnt += 1
            else { it += 1; active = 1; }
        }
        if (active) active = ((it < ds->p_min_div || (double)err > ds->p_div_thr) && it < ds->p_max_div) ? 1 : 0;   // :400
        ds->div_it = it;
        ds->div_active = active;
    } else {
        const float avg = cnt > 0 ? (float)(sum / (double)cnt) : 1000.0f;  // :148-149
        ds->dens_avg = avg;
        ds->dens_d7_active = 1;                                          // iter_all_vel_adv of this iteration runs (:229)
        const int it = ds->dens_it + 1;                                  // :231
        ds->dens_it = it;
        int active = (it < ds->p_min_dens || (double)avg - 1000.0 > ds->p_dens_thr) ? 1 : 0;      // :225
        if (active && it >= ds->dens_cap) { active = 0; ds->dens_capped = 1; }
        ds->dens_active = active;
    }
}
// hist >= 0: this is evaluation number `hist` of its loop; the decision also goes to gate_hist[hist & 1] (see DevScalars)
// (the work of ONE workgroup of kFinBlock threads: k_finalize_mean, or the last workgroup of k_pack_resid_reduce / k_unpack_resid_decide)
// gather_n > 0 (FINP_DECIDE only): red holds gather_n slabs' (sum, count, flags) triples, four doubles apart -- gathered with the halo's own
// transfers instead of all-reduced -- and the decision sums them here, in slab order, on every slab alike
__device__ __forceinline__ void finalize_mean_block(const double *__restrict__ psum, const int *__restrict__ pcnt, int nblocks,
                                                    DevScalars *__restrict__ ds, int mode, int phase, double *__restrict__ red,
                                                    int group, int nparts, int hist, int gather_n = 0)
{
    if (mode == FIN_DIV_LOOP && ds->div_active == 0) { if (hist >= 0 && threadIdx.x == 0 && phase != FINP_REDUCE) ds->gate_hist[hist & 1] = 0; return; }
    if (mode == FIN_DENS && ds->dens_active == 0) {
        if (threadIdx.x == 0 && phase != FINP_REDUCE) { ds->dens_d7_active = 0; if (hist >= 0) ds->gate_hist[hist & 1] = 0; }
        return;
    }
    __shared__ double s_sum[kFinBlock / 64];
    __shared__ long long s_cnt[kFinBlock / 64];
    if (phase != FINP_DECIDE) fin_reduce(psum, pcnt, nblocks, group, nparts, s_sum, s_cnt);
    if (threadIdx.x != 0) return;
    // (the density loop's reduction carries a third word: this slab's overflow flags -- every slab must see a list overflow on ANY slab at the
    // same point of the step, and this way that costs no host round trip of its own)
    if (phase == FINP_REDUCE) { red[0] = s_sum[0]; red[1] = (double)s_cnt[0]; if (mode == FIN_DENS) red[2] = (double)ds->overflow; return; }
    if (phase == FINP_DECIDE) {
        double rs = red[0], rc = red[1], rf = red[2];
        for (int r = 1; r < gather_n; ++r) { rs += red[4 * r]; rc += red[4 * r + 1]; rf += red[4 * r + 2]; }
        s_sum[0] = rs; s_cnt[0] = (long long)rc;
        if (mode == FIN_DENS) ds->overflow_any = rf != 0.0 ? 1 : 0;
    }
    const int was = (mode == FIN_DENS) ? ds->dens_active : ds->div_active;
    fin_decide(ds, mode, s_sum[0], s_cnt[0]);
    if (hist >= 0) {
        const int now = (mode == FIN_DENS) ? ds->dens_active : ds->div_active;
        ds->gate_hist[hist & 1] = now;
        if (mode != FIN_DENS && was != 0 && now == 0) ds->stop_at = hist;
    }
}
__global__ __launch_bounds__(kFinBlock) void k_finalize_mean(const double *__restrict__ psum, const int *__restrict__ pcnt, int nblocks,
                                                             DevScalars *__restrict__ ds, int mode, int phase, double *__restrict__ red,
                                                             int group = 1, int nparts = 0, int hist = -1)
{
    finalize_mean_block(psum, pcnt, nblocks, ds, mode, phase, red, group, nparts, hist);
}

// ---- the loop decision riding in the NEXT sweep (one GPU) -----------------------------------------------------------------
// Between a residual sweep and the correction sweep behind it sat a single-workgroup launch, k_finalize_mean: 31 per step at dfsph_1m, 4.4 us
// each on the critical path (tools/fin_probe.py).  The correction sweep does not NEED that decision to start -- the slab path's overlapped
// protocol already runs it ahead (step_dfsph_device_loops): in the density loop D7 of iteration d runs iff iteration d runs, which the decision
// of evaluation d - 1 says (gate_hist); in the divergence loop D4 of iteration e runs ahead of decision e, keeps what it overwrites (SpecSave)
// and is undone by the gated residual launch behind it if decision e closed the loop (SpecUndo).  So the decision can be taken INSIDE the
// correction launch: one extra workgroup (workgroup 0, dispatched first) plays k_finalize_mean while the other ~3900 sweep their tiles.  It
// reads the partials of the residual launch before (complete: a kernel boundary lies in between) and writes only words of DevScalars that no
// tile of this launch reads (they read gate_hist[(e - 1) & 1], dt, dt2).  Same reduction tree as k_finalize_mean: its kBlock threads play four
// of the kFinBlock "virtual threads" each, so the f64 sum has the same bits.
// the loop state of a step's two solver loops, before the first of them starts (k_ctrl_begin; one GPU: workgroup 0 of the warm-start launch, FIN_BEGIN)
__device__ __forceinline__ void ctrl_begin_body(DevScalars *__restrict__ ds, int dens_cap)
{
    ds->div_active = 1; ds->div_it = 0; ds->div_evals = 0; ds->overflow_any = 0;
    ds->dens_active = 1; ds->dens_d7_active = 0; ds->dens_it = 0; ds->dens_cap = dens_cap; ds->dens_capped = 0;
    ds->div_err = 0.f; ds->div_past = 0.f; ds->div_first = 0.f; ds->dens_avg = 0.f;
    ds->gate_hist[0] = 1; ds->gate_hist[1] = 1; ds->stop_at = -1;
}
constexpr int FIN_BEGIN = 4;          // FinRide.mode: no decision to take -- reset the loop state (hist = the density loop's cap)
struct FinRide { const double *psum; const int *pcnt; DevScalars *ds; int nblocks, mode, group, nparts, hist; };      // mode < 0: nobody rides
constexpr FinRide kNoRide{nullptr, nullptr, nullptr, 0, -1, 1, 0, -1};
__device__ __forceinline__ void fin_ride_block(const FinRide &fr)
{
    DevScalars *ds = fr.ds;
    if (fr.mode == FIN_BEGIN) { if (threadIdx.x == 0) ctrl_begin_body(ds, fr.hist); return; }
    // (an iteration the loop does not run: what finalize_mean_block does for it)
    if (fr.mode == FIN_DIV_LOOP && ds->div_active == 0) { if (threadIdx.x == 0) ds->gate_hist[fr.hist & 1] = 0; return; }
    if (fr.mode == FIN_DENS && ds->dens_active == 0) { if (threadIdx.x == 0) { ds->dens_d7_active = 0; ds->gate_hist[fr.hist & 1] = 0; } return; }
    __shared__ double s_rsum[kFinBlock / 64];
    __shared__ long long s_rcnt[kFinBlock / 64];
#pragma unroll
    for (int j = 0; j < kFinBlock / kBlock; ++j) {
        double t; int n;
        fin_thread_sum(fr.psum, fr.pcnt, (int)threadIdx.x + j * kBlock, fr.nblocks, fr.group, fr.nparts, t, n);
        const double ws = wave_sum(t);
        const int wn = wave_sum(n);
        if ((threadIdx.x & 63) == 0) { s_rsum[(threadIdx.x >> 6) + j * (kBlock / 64)] = ws; s_rcnt[(threadIdx.x >> 6) + j * (kBlock / 64)] = wn; }
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    double tt = 0.0; long long nn = 0;
    for (int w = 0; w < kFinBlock / 64; ++w) { tt += s_rsum[w]; nn += s_rcnt[w]; }
    const int was = (fr.mode == FIN_DENS) ? ds->dens_active : ds->div_active;
    fin_decide(ds, fr.mode, tt, nn);
    const int now = (fr.mode == FIN_DENS) ? ds->dens_active : ds->div_active;
    ds->gate_hist[fr.hist & 1] = now;
    if (fr.mode != FIN_DENS && was != 0 && now == 0) ds->stop_at = fr.hist;
}


// The loop-control block handed to the host WITHOUT a copy command and an interrupt: the workgroup writes it to pinned host memory (mapped into the
// device's address space), fences at system scope and stores a sequence number behind it; the host thread spins on that number (read_scalars_fast
// in sph_host_scene.h).  A dfsph step needs the density loop's verdict before it can enqueue the integrator; through hipMemcpyAsync +
// hipStreamSynchronize that round trip left the GPU idle for 36 us per step (profiles/r05: the gap between the copy and k_dfsph_integrate).
struct DevScalarsPub { DevScalars ds; unsigned long long seq; };
__global__ __launch_bounds__(kBlock) void k_publish_scalars(const DevScalars *__restrict__ ds, DevScalarsPub *__restrict__ out, unsigned long long seq)
{
    const uint32_t *src = reinterpret_cast<const uint32_t *>(ds);
    uint32_t *dst = reinterpret_cast<uint32_t *>(&out->ds);
    for (int w = threadIdx.x; w < (int)(sizeof(DevScalars) / 4); w += kBlock) dst[w] = src[w];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&out->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_ctrl_begin(DevScalars *__restrict__ ds, int dens_cap)
{
    ctrl_begin_body(ds, dens_cap);
}

// the CFL time step from ds->vmax (global maximum)              dfsph_solver.py:104-119
__device__ __forceinline__ void apply_dt_body(const Consts &c, DevScalars *__restrict__ ds, const double *__restrict__ red, int gather_n)
{
    float max_vel = red ? (float)red[0] : ds->vmax;        // red: the maximum over all slabs (gather_n > 0: every slab's own, four doubles apart)
    for (int r = 1; r < gather_n; ++r) max_vel = fmaxf(max_vel, (float)red[4 * r]);
    if (red) ds->vmax = max_vel;
    float max_rigid_vel = ds->rigid_vmax;                         // :104-110 (0 without a rigid body)
    max_vel += max_rigid_vel;
    float max_delta_time = c.dt_cfl_num / max_vel * 0.2f;         // :112
    if (c.adaptive_dt) {                                          // :113
        float dt;
        if (max_delta_time > c.max_dt) dt = c.max_dt;             // :114-117
        else dt = rmax(max_delta_time, c.min_dt);
        ds->dt = dt;
        ds->dt2 = dt * dt;                                        // :118
        ds->ps_dt = dt;                                           // :119
    }
    ds->gate_hist[0] = 1; ds->gate_hist[1] = 1;                   // (between the two solver loops: the density loop's decisions start afresh)
}
__global__ void k_apply_dt(Consts c, DevScalars *__restrict__ ds, const double *__restrict__ red, int gather_n = 0)
{
    apply_dt_body(c, ds, red, gather_n);
}

// max |v*| over the block partials                              dfsph_solver.py:100-103
// apply != 0 (one GPU: nothing travels between the maximum and the rule): the same thread goes on to the CFL rule, one launch instead of two
// fr.mode >= 0 (one GPU): first the loop decision of the divergence loop's LAST evaluation, which has no correction launch to ride in (fin_ride_block)
__global__ __launch_bounds__(kBlock) void k_finalize_max(const float *__restrict__ pmax, int nblocks, DevScalars *__restrict__ ds,
                                                         double *__restrict__ red, Consts c, int apply, FinRide fr)
{
    if (fr.mode >= 0) { fin_ride_block(fr); __syncthreads(); }
    __shared__ float s_max[kBlock];
    float t = -INFINITY;
    for (int k = threadIdx.x; k < nblocks; k += kBlock) t = fmaxf(t, pmax[k]);
    s_max[threadIdx.x] = t;
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) s_max[threadIdx.x] = fmaxf(s_max[threadIdx.x], s_max[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        ds->vmax = s_max[0];
        if (red) red[0] = (double)s_max[0];   // red: this slab's maximum, all-reduced next
        if (apply) apply_dt_body(c, ds, nullptr, 0);
    }
}

// ======================================================================================
// W1 / D1: density (+ alpha)          solver_base.py:41-72, dfsph_solver.py:32-89, wcsph_solver.py:66-68
//   WCSPH: writes Pout = (pos, rho), Vout = (vel, p/rho^2), rho[], pressure[]
//   DFSPH: writes Pout = (pos, (warm_k/dt)/rho) for the warm start, Vout = (vel, rho), rho[], alpha[]
// ======================================================================================
// the workgroup's operand array through stage_src into LDS (see the staging plan in k_build_nl); returns false when this
// workgroup keeps global indices.  Uniform per workgroup; every thread of the workgroup must call it.
// SCALED: the positions are staged multiplied by 2^32 (exact; see norm3_scaled in sph_device.h), .w unchanged
// The staged set of a workgroup is described by cell runs (k_build_nl): stage_cnt[blk] = particles | runs << 16 (-1: not staged),
// stage_runs[blk][r] = (first sorted index, local base | count << 16).  stage_expand turns them into the flat list s_idx[e] = sorted
// index of staged element e, in LDS.  In the kernels that do not keep that list, s_idx ALIASES the start of the operand array it
// helps to fill (4 B per element inside a 16-B-per-element array): every thread first reads all of its indices into registers,
// the workgroup synchronises, and only then do the gathered operands overwrite the list.
// The copy itself is two dependent steps per element (the index, then A[index]: ~700 cycles out of L2, more from HBM): all of a
// thread's operands are requested in one batch, then the LDS stores (kStageBatch elements per thread and trip, two trips cover the
// largest capacity; a clamped index keeps the loads branch-free so that the compiler leaves them in one batch).
constexpr int kStageBatch = 7;          // 7 x 256 = 1792 >= the default capacity of 1664: one trip
constexpr int kStageTrips = 2;          // 2 x 1792 >= the largest capacity (2560)
__device__ __forceinline__ int stage_expand(const uint2 *__restrict__ stage_runs, const int *__restrict__ stage_cnt, int blk, uint32_t *__restrict__ s_idx,
                                            const StagePre &pre = kNoPre)
{
    const int w = pre.have ? pre.sw : stage_cnt[blk];       // (pre: requested by the caller ahead of time, see StagePre)
    if (w < 0) return -1;                                   // uniform per workgroup
    const int nst = w & 0xffff, nruns = (w >> 16) & 0x3fff;
    const uint2 *runs = stage_runs + (size_t)blk * kStageMaxCells;
    for (int r = threadIdx.x; r < nruns; r += kBlock) {
        const uint2 rn = (pre.have && r == (int)threadIdx.x) ? pre.rn0 : runs[r];
        const int base = (int)(rn.y & 0xffffu), n = (int)(rn.y >> 16);
        for (int k = 0; k < n; ++k) s_idx[base + k] = rn.x + (uint32_t)k;
    }
    __syncthreads();
    return nst;
}
// ---- change propagation between the sweeps of the constant-density loop (round 3) -----------------------------------------------
// rho* = max(rho + dt * sum, rho0) clamps at the rest density, and the reference's density sum has no self term, so in a collapsing
// column ~99 % of the particles sit AT rho0 with stiffness k = 0: the loop (dfsph_solver.py:221-233) iterates for the floor layer and
// a few compressed pockets (tools/zero_tiles.py: 0.5-1 % of the particles at dfsph_1m; 2-5 % lie within h of one, 5-10 % within 2h).
//   D7 (k_correct<DENS>) of a tile whose staged set (own particles + halo) holds no k / rho != 0 adds only +-0 terms: the tile checks the
//      staged k / rho FIRST (a 4-byte gather) and, if all are 0, leaves v* as it is.  Per 64-particle wave it reports whether any lane
//      applied a nonzero correction: wave_dirty[i / 64].
//   D6 (k_residual<DENS>) of a tile whose staged set lies in waves with wave_dirty == 0 would recompute, from unchanged v*, exactly what
//      it wrote in the iteration before -- rho*, k / rho and its block partial are still in memory: it returns at once.  The waves that
//      own a workgroup's staged set are read off its cell runs (a run is a cell: contiguous sorted indices, at most two waves).
// The first D6 of a step computes everywhere (`force_all`).  Bit-identical to computing everything (SPH_TILE_SKIP=0;
// tests/test_cell_order_gpu.py::test_density_loop_change_propagation_is_invisible), up to the sign of a zero velocity component
// (v - (-0) = +0 where the skipped sweep keeps -0).  Staged handles without rigid entries, slabs included: a wave that holds a ghost
// counts as changed in every iteration (its owner may have moved its v*), so the tiles along a cut always recompute.  Rigid entries fit in:
// the body's term of D7 is proportional to the particle's own k, and D6 sees the body at rest within a solver loop.
__device__ __forceinline__ bool stage_sources_flagged(const uint2 *__restrict__ stage_runs, int sw, int blk, const int *__restrict__ wave_flags)
{
    const int nruns = (sw >> 16) & 0x3fff;
    const uint2 *runs = stage_runs + (size_t)blk * kStageMaxCells;
    int f = 0;
    if (threadIdx.x < kBlock / 64) f = wave_flags[blk * (kBlock / 64) + threadIdx.x];      // the tile's own waves
    for (int r = threadIdx.x; r < nruns; r += kBlock) {
        const uint2 rn = runs[r];
        const uint32_t first = rn.x, n = rn.y >> 16;
        for (uint32_t w = first >> 6; w <= (first + n - 1u) >> 6; ++w) f |= wave_flags[w];      // (a run of > 65 particles spans three waves or more)
    }
    return __syncthreads_or(f) != 0;
}

struct StageIdx { uint32_t j[kStageTrips][kStageBatch]; };
// this thread's indices of all trips, then the barrier after which s_idx may be overwritten
__device__ __forceinline__ StageIdx stage_take(const uint32_t *__restrict__ s_idx, int nst)
{
    StageIdx x;
#pragma unroll
    for (int t = 0; t < kStageTrips; ++t) {
        if (t > 0 && t * kStageBatch * kBlock >= nst) {     // uniform: the default capacity needs one trip
#pragma unroll
            for (int u = 0; u < kStageBatch; ++u) x.j[t][u] = 0u;
            continue;
        }
#pragma unroll
        for (int u = 0; u < kStageBatch; ++u) x.j[t][u] = s_idx[min((int)threadIdx.x + (t * kStageBatch + u) * kBlock, nst - 1)];
    }
    __syncthreads();
    return x;
}
template <bool SCALED = false>
__device__ __forceinline__ bool stage_operand(const Consts &c, float4 *__restrict__ s_A, const float4 *__restrict__ A,
                                              const uint2 *__restrict__ stage_runs, const int *__restrict__ stage_cnt, int blk, const StagePre &pre = kNoPre)
{
    const int nst = stage_expand(stage_runs, stage_cnt, blk, reinterpret_cast<uint32_t *>(s_A), pre);
    if (nst < 0) return false;
    if (nst == 0) return true;                              // a workgroup of ghosts only (slab handles): nothing to stage, uniform
    const StageIdx x = stage_take(reinterpret_cast<const uint32_t *>(s_A), nst);
#pragma unroll
    for (int t = 0; t < kStageTrips; ++t) {
        const int base = threadIdx.x + t * kStageBatch * kBlock;
        if (t * kStageBatch * kBlock >= nst) break;
        float4 a[kStageBatch];
#pragma unroll
        for (int u = 0; u < kStageBatch; ++u) a[u] = A[x.j[t][u]];
#pragma unroll
        for (int u = 0; u < kStageBatch; ++u)
            if (base + u * kBlock < nst)
                s_A[base + u * kBlock] = SCALED ? make_float4(a[u].x * 0x1p32f, a[u].y * 0x1p32f, a[u].z * 0x1p32f, a[u].w) : a[u];
    }
    __syncthreads();
    return true;
}

// stage_operand<SCALED> that also reports whether any staged element has .w != 0: 1 = staged, 2 = staged and every .w is 0 (the caller's
// pair loop would add only +-0: see stage_sources_flagged), 0 = not staged.  For handles whose k / rho travels inside the (pos, k / rho)
// float4 (slab handles, SPH_KR_SPLIT=0): the verdict comes with the copy, so a zero tile saves its pair loop, not its gathers.
template <bool SCALED>
__device__ __forceinline__ int stage_operand_w_checked(const Consts &c, float4 *__restrict__ s_A, const float4 *__restrict__ A,
                                                       const uint2 *__restrict__ stage_runs, const int *__restrict__ stage_cnt, int blk, const StagePre &pre = kNoPre)
{
    const int nst = stage_expand(stage_runs, stage_cnt, blk, reinterpret_cast<uint32_t *>(s_A), pre);
    if (nst < 0) return 0;
    if (nst == 0) return 2;                                 // a workgroup of ghosts only: nothing staged, nothing to add
    const StageIdx x = stage_take(reinterpret_cast<const uint32_t *>(s_A), nst);
    int any = 0;
#pragma unroll
    for (int t = 0; t < kStageTrips; ++t) {
        const int base = threadIdx.x + t * kStageBatch * kBlock;
        if (t * kStageBatch * kBlock >= nst) break;
        float4 a[kStageBatch];
#pragma unroll
        for (int u = 0; u < kStageBatch; ++u) a[u] = A[x.j[t][u]];
#pragma unroll
        for (int u = 0; u < kStageBatch; ++u)
            if (base + u * kBlock < nst) {
                any |= a[u].w != 0.f;
                s_A[base + u * kBlock] = SCALED ? make_float4(a[u].x * 0x1p32f, a[u].y * 0x1p32f, a[u].z * 0x1p32f, a[u].w) : a[u];
            }
    }
    return __syncthreads_or(any) ? 1 : 2;
}

// kr_split handles: positions from the step's position array and the per-sweep scalar k / rho from its own 4-byte array (the sweeps
// then write 4 B per particle for their neighbours instead of a fresh (pos, k / rho) float4: 12 MB less written per launch at 1 M)
__device__ __forceinline__ bool stage_operand_ps_scaled(const Consts &c, float4 *__restrict__ s_A, const float4 *__restrict__ A, const float *__restrict__ S,
                                                        const uint2 *__restrict__ stage_runs, const int *__restrict__ stage_cnt, int blk, const StagePre &pre = kNoPre)
{
    const int nst = stage_expand(stage_runs, stage_cnt, blk, reinterpret_cast<uint32_t *>(s_A), pre);
    if (nst < 0) return false;
    if (nst == 0) return true;
    const StageIdx x = stage_take(reinterpret_cast<const uint32_t *>(s_A), nst);
#pragma unroll
    for (int t = 0; t < kStageTrips; ++t) {
        const int base = threadIdx.x + t * kStageBatch * kBlock;
        if (t * kStageBatch * kBlock >= nst) break;
        float4 a[kStageBatch]; float sc[kStageBatch];
#pragma unroll
        for (int u = 0; u < kStageBatch; ++u) { a[u] = A[x.j[t][u]]; sc[u] = S[x.j[t][u]]; }
#pragma unroll
        for (int u = 0; u < kStageBatch; ++u)
            if (base + u * kBlock < nst) s_A[base + u * kBlock] = make_float4(a[u].x * 0x1p32f, a[u].y * 0x1p32f, a[u].z * 0x1p32f, sc[u]);
    }
    __syncthreads();
    return true;
}

// The same with the scalars requested first and examined: returns 0 = not staged, 1 = staged, 2 = staged set holds no scalar != 0 (nothing
// was copied; the correction sweep of the density loop has nothing to do, see stage_sources_flagged).  One more round trip than the
// plain form for the workgroups that do have work (the positions are requested after the verdict).  (Asking the per-wave flags of the
// residual sweep instead of the staged scalars themselves -- no expansion, no gather -- was measured: fewer tiles return, 30.5 -> 35.4 us.)
template <bool SCALED>
__device__ __forceinline__ int stage_operand_ps_checked(const Consts &c, float4 *__restrict__ s_A, const float4 *__restrict__ A, const float *__restrict__ S,
                                                        const uint2 *__restrict__ stage_runs, const int *__restrict__ stage_cnt, int blk, const StagePre &pre = kNoPre)
{
    const int nst = stage_expand(stage_runs, stage_cnt, blk, reinterpret_cast<uint32_t *>(s_A), pre);
    if (nst < 0) return 0;
    if (nst == 0) return 1;
    const StageIdx x = stage_take(reinterpret_cast<const uint32_t *>(s_A), nst);
    float sc[kStageTrips][kStageBatch];
    int any = 0;
#pragma unroll
    for (int t = 0; t < kStageTrips; ++t) {
        if (t * kStageBatch * kBlock >= nst) break;
#pragma unroll
        for (int u = 0; u < kStageBatch; ++u) sc[t][u] = S[x.j[t][u]];                      // (clamped indices: duplicates of valid slots)
#pragma unroll
        for (int u = 0; u < kStageBatch; ++u) any |= sc[t][u] != 0.f;
    }
    if (!__syncthreads_or(any)) return 2;
#pragma unroll
    for (int t = 0; t < kStageTrips; ++t) {
        const int base = threadIdx.x + t * kStageBatch * kBlock;
        if (t * kStageBatch * kBlock >= nst) break;
        float4 a[kStageBatch];
#pragma unroll
        for (int u = 0; u < kStageBatch; ++u) a[u] = A[x.j[t][u]];
#pragma unroll
        for (int u = 0; u < kStageBatch; ++u)
            if (base + u * kBlock < nst)
                s_A[base + u * kBlock] = SCALED ? make_float4(a[u].x * 0x1p32f, a[u].y * 0x1p32f, a[u].z * 0x1p32f, sc[t][u]) : make_float4(a[u].x, a[u].y, a[u].z, sc[t][u]);
    }
    __syncthreads();
    return 1;
}

// two-operand variant: A staged in LDS, the global index of every staged element next to it (B is gathered from HBM/L2 through it)
__device__ __forceinline__ bool stage_operand_src(const Consts &c, float4 *__restrict__ s_A, uint32_t *__restrict__ s_src,
                                                  const float4 *__restrict__ A, const uint2 *__restrict__ stage_runs,
                                                  const int *__restrict__ stage_cnt, int blk)
{
    const int nst = stage_expand(stage_runs, stage_cnt, blk, s_src);      // the list stays: B is gathered through it
    if (nst < 0) return false;
    for (int base = threadIdx.x; base < nst; base += kStageBatch * kBlock) {
        float4 a[kStageBatch];
#pragma unroll
        for (int u = 0; u < kStageBatch; ++u) a[u] = A[s_src[min(base + u * kBlock, nst - 1)]];
#pragma unroll
        for (int u = 0; u < kStageBatch; ++u)
            if (base + u * kBlock < nst) s_A[base + u * kBlock] = a[u];
    }
    __syncthreads();
    return true;
}
// staged walkers: fluid entries are LOCAL indices into the staged arrays, tagged rigid entries stay global (rv.RP)
template <bool RIGID, bool SCALED = false, class Body>
__device__ __forceinline__ void for_staged_nbrs(const uint32_t *__restrict__ base, int cnt, const float4 *__restrict__ s_A, const RigidView &rv,
                                                Body body)
{
    NlAhead ahead(base);
    for (int kk = 0; kk < cnt; kk += 4) {
        const uint4 jj = ahead.front();
        const uint32_t j[4] = {jj.x, jj.y, jj.z, jj.w};
        float4 a[4];
        // a group without a rigid entry anywhere in the wave (nearly all of them: the body touches a thin layer of the fluid) takes the
        // plain path; the branch is wave-uniform
        if (RIGID && __any(((j[0] | j[1] | j[2] | j[3]) & kRigidTag) != 0)) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool rg = (j[u] & kRigidTag) != 0;
                if (rg) {
                    const float4 q = rv.RP[j[u] & ~kRigidTag];
                    a[u] = SCALED ? make_float4(q.x * 0x1p32f, q.y * 0x1p32f, q.z * 0x1p32f, q.w) : q;
                } else {
                    a[u] = s_A[j[u] & ~kRigidTag];
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = s_A[j[u]];
        }
        ahead.advance(kk);
        const float4 none = make_float4(0.f, 0.f, 0.f, 0.f);
        body(a[0], none, j[0]);
        if (kk + 1 < cnt) body(a[1], none, j[1]);
        if (kk + 2 < cnt) body(a[2], none, j[2]);
        if (kk + 3 < cnt) body(a[3], none, j[3]);
    }
}
template <bool RIGID, class Body>
__device__ __forceinline__ void for_staged_nbrs_pv(const uint32_t *__restrict__ base, int cnt, const float4 *__restrict__ s_A,
                                                   const uint32_t *__restrict__ s_src, const float4 *__restrict__ B, const RigidView &rv, Body body)
{
    NlAhead ahead(base);
    for (int kk = 0; kk < cnt; kk += 4) {
        const uint4 jj = ahead.front();
        const uint32_t j[4] = {jj.x, jj.y, jj.z, jj.w};
        float4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool rg = RIGID && (j[u] & kRigidTag);
            const uint32_t idx = RIGID ? (j[u] & ~kRigidTag) : j[u];
            b[u] = B[rg ? 0u : s_src[idx]];
            a[u] = rg ? rv.RP[idx] : s_A[idx];
        }
        ahead.advance(kk);
        body(a[0], b[0], j[0]);
        if (kk + 1 < cnt) body(a[1], b[1], j[1]);
        if (kk + 2 < cnt) body(a[2], b[2], j[2]);
        if (kk + 3 < cnt) body(a[3], b[3], j[3]);
    }
}

// both operands of the residual sweeps staged: (x, y, z, vx) and (vy, vz) -- 24 B per staged particle
template <bool SCALED = false>
__device__ __forceinline__ bool stage_operand_pv(const Consts &c, float4 *__restrict__ s_A, float2 *__restrict__ s_B,
                                                 const float4 *__restrict__ A, const float4 *__restrict__ B,
                                                 const uint2 *__restrict__ stage_runs, const int *__restrict__ stage_cnt, int blk, const StagePre &pre = kNoPre,
                                                 const unsigned char *__restrict__ changed = nullptr, int *any_changed = nullptr)
{
    // changed (optional): the per-particle bytes of the density loop's check, fetched with the operands; *any_changed = is one of the staged set's set?
    const int nst = stage_expand(stage_runs, stage_cnt, blk, reinterpret_cast<uint32_t *>(s_A), pre);
    if (nst < 0) return false;
    if (nst == 0) return true;                              // a workgroup of ghosts only (slab handles): nothing to stage, uniform
    const StageIdx x = stage_take(reinterpret_cast<const uint32_t *>(s_A), nst);
    int any = 0;
#pragma unroll
    for (int t = 0; t < kStageTrips; ++t) {
        const int base = threadIdx.x + t * kStageBatch * kBlock;
        if (t * kStageBatch * kBlock >= nst) break;
        float4 a[kStageBatch], b[kStageBatch];
        unsigned char f[kStageBatch];
#pragma unroll
        for (int u = 0; u < kStageBatch; ++u) { a[u] = A[x.j[t][u]]; b[u] = B[x.j[t][u]]; f[u] = changed ? changed[x.j[t][u]] : (unsigned char)0; }
#pragma unroll
        for (int u = 0; u < kStageBatch; ++u) any |= f[u];                                   // (clamped indices: duplicates of valid slots)
#pragma unroll
        for (int u = 0; u < kStageBatch; ++u)
            if (base + u * kBlock < nst) {
                const int e = base + u * kBlock;
                s_A[e] = SCALED ? make_float4(a[u].x * 0x1p32f, a[u].y * 0x1p32f, a[u].z * 0x1p32f, b[u].x) : make_float4(a[u].x, a[u].y, a[u].z, b[u].x);
                s_B[e] = make_float2(b[u].y, b[u].z);
            }
    }
    if (changed) *any_changed = __syncthreads_or(any);
    else __syncthreads();
    return true;
}
// The same with a look at a per-particle byte first: returns 0 = not staged, 1 = staged, 2 = no staged particle has its byte set (nothing was
// copied).  The residual sweep of the density loop asks it with the bytes the last correction sweep wrote ("this particle's v* changed"):
// the second, exact level of the change propagation (stage_sources_flagged is the first: per-wave flags, no expansion).
template <bool SCALED>
__device__ __forceinline__ int stage_operand_pv_checked(const Consts &c, float4 *__restrict__ s_A, float2 *__restrict__ s_B,
                                                        const float4 *__restrict__ A, const float4 *__restrict__ B, const unsigned char *__restrict__ changed,
                                                        const uint2 *__restrict__ stage_runs, const int *__restrict__ stage_cnt, int blk, const StagePre &pre = kNoPre)
{
    const int nst = stage_expand(stage_runs, stage_cnt, blk, reinterpret_cast<uint32_t *>(s_A), pre);
    if (nst < 0) return 0;
    if (nst == 0) return 1;
    const StageIdx x = stage_take(reinterpret_cast<const uint32_t *>(s_A), nst);
    int any = 0;
#pragma unroll
    for (int t = 0; t < kStageTrips; ++t) {
        if (t * kStageBatch * kBlock >= nst) break;
        unsigned char f[kStageBatch];
#pragma unroll
        for (int u = 0; u < kStageBatch; ++u) f[u] = changed[x.j[t][u]];                     // (clamped indices: duplicates of valid slots)
#pragma unroll
        for (int u = 0; u < kStageBatch; ++u) any |= f[u];
    }
    if (!__syncthreads_or(any)) return 2;
#pragma unroll
    for (int t = 0; t < kStageTrips; ++t) {
        const int base = threadIdx.x + t * kStageBatch * kBlock;
        if (t * kStageBatch * kBlock >= nst) break;
        float4 a[kStageBatch], b[kStageBatch];
#pragma unroll
        for (int u = 0; u < kStageBatch; ++u) { a[u] = A[x.j[t][u]]; b[u] = B[x.j[t][u]]; }
#pragma unroll
        for (int u = 0; u < kStageBatch; ++u)
            if (base + u * kBlock < nst) {
                const int e = base + u * kBlock;
                s_A[e] = SCALED ? make_float4(a[u].x * 0x1p32f, a[u].y * 0x1p32f, a[u].z * 0x1p32f, b[u].x) : make_float4(a[u].x, a[u].y, a[u].z, b[u].x);
                s_B[e] = make_float2(b[u].y, b[u].z);
            }
    }
    __syncthreads();
    return 1;
}
template <bool RIGID, bool SCALED = false, class Body>
__device__ __forceinline__ void for_staged_nbrs_pv2(const uint32_t *__restrict__ base, int cnt, const float4 *__restrict__ s_A,
                                                    const float2 *__restrict__ s_B, const RigidView &rv, Body body)
{
    NlAhead ahead(base);
    for (int kk = 0; kk < cnt; kk += 4) {
        const uint4 jj = ahead.front();
        const uint32_t j[4] = {jj.x, jj.y, jj.z, jj.w};
        float4 a[4], b[4];
        if (RIGID && __any(((j[0] | j[1] | j[2] | j[3]) & kRigidTag) != 0)) {      // wave-uniform, rare (see for_staged_nbrs)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool rg = (j[u] & kRigidTag) != 0;
                const uint32_t idx = j[u] & ~kRigidTag;
                if (rg) {
                    const float4 q = rv.RP[idx];                     // (x, y, z, V_r); the velocity operand is undefined for rigid entries
                    a[u] = SCALED ? make_float4(q.x * 0x1p32f, q.y * 0x1p32f, q.z * 0x1p32f, q.w) : q;
                    b[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                } else {
                    const float4 pa = s_A[idx]; const float2 pb = s_B[idx];
                    a[u] = make_float4(pa.x, pa.y, pa.z, 0.f);
                    b[u] = make_float4(pa.w, pb.x, pb.y, 0.f);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 pa = s_A[j[u]]; const float2 pb = s_B[j[u]];
                a[u] = make_float4(pa.x, pa.y, pa.z, 0.f);
                b[u] = make_float4(pa.w, pb.x, pb.y, 0.f);
            }
        }
        ahead.advance(kk);
        body(a[0], b[0], j[0]);
        if (kk + 1 < cnt) body(a[1], b[1], j[1]);
        if (kk + 2 < cnt) body(a[2], b[2], j[2]);
        if (kk + 3 < cnt) body(a[3], b[3], j[3]);
    }
}

// ---- walks of a 16-bit list (staged workgroup of an nl16 handle; no rigid entries): one 16-byte load = eight neighbours, taken
// four at a time so that the registers in flight stay those of the 32-bit walk.  The next group is requested before the bodies
// run, and only by lanes whose list goes on (the 32-bit walks read one stale group past the end: 16 B per particle and sweep).
struct Nl16Group {
    uint32_t w[4];
    __device__ __forceinline__ uint32_t lo(int q) const { return w[q] & 0xffffu; }
    __device__ __forceinline__ uint32_t hi(int q) const { return w[q] >> 16; }
};
template <class Fetch4>
__device__ __forceinline__ void walk_list16(const uint32_t *__restrict__ base, int cnt, Fetch4 fetch4)
{
    if (cnt <= 0) return;
    uint4 jn = nl_load(base);
    for (int kk = 0; kk < cnt; kk += 8) {
        const Nl16Group g = {{jn.x, jn.y, jn.z, jn.w}};
        if (kk + 8 < cnt) jn = nl_load(base + (size_t)((kk >> 3) + 1) * 256);
        fetch4(g.lo(0), g.hi(0), g.lo(1), g.hi(1), kk);
        if (kk + 4 < cnt) fetch4(g.lo(2), g.hi(2), g.lo(3), g.hi(3), kk + 4);
    }
}
template <class Body>
__device__ __forceinline__ void for_staged16_nbrs(const uint32_t *__restrict__ base, int cnt, const float4 *__restrict__ s_A, Body body)
{
    const float4 none = make_float4(0.f, 0.f, 0.f, 0.f);
    walk_list16(base, cnt, [&](uint32_t j0, uint32_t j1, uint32_t j2, uint32_t j3, int k0) {
        const float4 a0 = s_A[j0], a1 = s_A[j1], a2 = s_A[j2], a3 = s_A[j3];
        body(a0, none, 0u);
        if (k0 + 1 < cnt) body(a1, none, 0u);
        if (k0 + 2 < cnt) body(a2, none, 0u);
        if (k0 + 3 < cnt) body(a3, none, 0u);
    });
}
// A staged in LDS, B gathered from memory through the staged source index (k_dfsph_ext and the pressure solvers' three-operand sweeps)
template <class Body>
__device__ __forceinline__ void for_staged16_nbrs_pv(const uint32_t *__restrict__ base, int cnt, const float4 *__restrict__ s_A,
                                                     const uint32_t *__restrict__ s_src, const float4 *__restrict__ B, Body body)
{
    walk_list16(base, cnt, [&](uint32_t j0, uint32_t j1, uint32_t j2, uint32_t j3, int k0) {
        const float4 b0 = B[s_src[j0]], b1 = B[s_src[j1]], b2 = B[s_src[j2]], b3 = B[s_src[j3]];
        const float4 a0 = s_A[j0], a1 = s_A[j1], a2 = s_A[j2], a3 = s_A[j3];
        body(a0, b0, 0u);
        if (k0 + 1 < cnt) body(a1, b1, 0u);
        if (k0 + 2 < cnt) body(a2, b2, 0u);
        if (k0 + 3 < cnt) body(a3, b3, 0u);
    });
}
// both operands staged: (x, y, z, vx) and (vy, vz)
template <class Body>
__device__ __forceinline__ void for_staged16_nbrs_pv2(const uint32_t *__restrict__ base, int cnt, const float4 *__restrict__ s_A,
                                                      const float2 *__restrict__ s_B, Body body)
{
    walk_list16(base, cnt, [&](uint32_t j0, uint32_t j1, uint32_t j2, uint32_t j3, int k0) {
        const float4 a0 = s_A[j0], a1 = s_A[j1], a2 = s_A[j2], a3 = s_A[j3];
        const float2 b0 = s_B[j0], b1 = s_B[j1], b2 = s_B[j2], b3 = s_B[j3];
        body(make_float4(a0.x, a0.y, a0.z, 0.f), make_float4(a0.w, b0.x, b0.y, 0.f), 0u);
        if (k0 + 1 < cnt) body(make_float4(a1.x, a1.y, a1.z, 0.f), make_float4(a1.w, b1.x, b1.y, 0.f), 0u);
        if (k0 + 2 < cnt) body(make_float4(a2.x, a2.y, a2.z, 0.f), make_float4(a2.w, b2.x, b2.y, 0.f), 0u);
        if (k0 + 3 < cnt) body(make_float4(a3.x, a3.y, a3.z, 0.f), make_float4(a3.w, b3.x, b3.y, 0.f), 0u);
    });
}

template <bool DFSPH, bool RIGID, int MODE, bool RX = false>
__global__ __launch_bounds__(kBlock) void k_density(Consts c, const float4 *__restrict__ P, const float4 *V,
                                                    const float4 *__restrict__ WP, const uint32_t *__restrict__ nl,
                                                    const uint32_t *__restrict__ nlb, const int *__restrict__ cnt,
                                                    const float *__restrict__ warm, const DevScalars *__restrict__ ds,
                                                    float *__restrict__ rho_out, float *__restrict__ aux_out,
                                                    float4 *__restrict__ Pout, float4 *Vout, RigidView rv,
                                                    const int *__restrict__ id, float *__restrict__ rho_orig,
                                                    const uint2 *__restrict__ stage_src, const int *__restrict__ stage_cnt, float *__restrict__ krho,
                                                    float4 *__restrict__ wall_gc, TilePhase tp = TilePhase{nullptr, 0, 0})
{
    constexpr bool STAGED = MODE == SWEEP_STAGED, QUAD = MODE == SWEEP_QUAD;
    using K = KF<RX>;                                        // kernel functions of the sweep's arithmetic (sph_device.h); RX: unstaged handles under SPH_ARITH_RELAXED
    extern __shared__ float4 s_operand[];
    const int tile = tp.phase == 0 ? xcd_block(blockIdx.x, gridDim.x) : sweep_tile(tp, false);
    if (tile < 0) return;
    SPH_SWEEP_PROLOGUE_G(QUAD, tile, true)
    const bool staged = STAGED && stage_operand(c, s_operand, P, stage_src, stage_cnt, blk);
    float fa[5] = {0.001f, 0.f, 0.f, 0.f, 0.f};              // rho starts at 0.001, solver_base.py:44
    float &rho = fa[0], &sx = fa[1], &sy = fa[2], &sz = fa[3], &sq = fa[4];
    auto pair = [&](const float4 pj, const float4, const uint32_t j) {
        float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        float r = K::norm3(dx, dy, dz);
        const bool rg = RIGID && (j & kRigidTag);
        if (rg) rho += pj.w * K::w_in(c, r) * c.rho0;        // solver_base.py:65  (V_j * W * rho_0)
        else rho += c.m * K::w_in(c, r);                     // solver_base.py:62
        if (DFSPH) {
            F3 g = K::grad_in(c, dx, dy, dz, r);
            const float cm = rg ? pj.w * c.rho0 : c.m;       // dfsph_solver.py:62,75 / :58,70
            float rx = cm * g.x, ry = cm * g.y, rz = cm * g.z;
            sx += rx; sy += ry; sz += rz;
            sq += (rx * rx + ry * ry) + rz * rz;             // :71
        }
    };
    if (QUAD) for_fluid_nbrs_quad<RIGID, false>(nlp, kf, q, fa, P, nullptr, rv, pair);
    else if (staged && (RIGID ? stage_lists16(stage_cnt, blk) : c.nl16 != 0)) for_staged16_nbrs(nlp, kf, s_operand, pair);     // (a rigid build: no rigid cell near this tile)
    else if (staged) for_staged_nbrs<RIGID>(nlp, kf, s_operand, rv, pair);
    else for_fluid_nbrs<RIGID, false>(nlp, kf, P, nullptr, rv, pair);
    float wa[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    float &rho_b = wa[0], &bx = wa[1], &by = wa[2], &bz = wa[3], &bsq = wa[4];
    float4 *gcw = (DFSPH && !QUAD && wall_gc) ? wall_gc + gc_index(ii, 0, c.kbpitch) : nullptr;   // bodies run in list order: entry k goes to row k
    auto wall = [&](const float4 pj) {                       // pj = (x, y, z, V_b)
        float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        float r = K::norm3(dx, dy, dz);
        rho_b += pj.w * K::w_in(c, r);                       // solver_base.py:70-71
        if (DFSPH) {
            F3 g = K::grad_in(c, dx, dy, dz, r);
            float cc = pj.w * c.rho0;                        // dfsph_solver.py:82,88
            float rx = cc * g.x, ry = cc * g.y, rz = cc * g.z;
            bx += rx; by += ry; bz += rz;
            bsq += (rx * rx + ry * ry) + rz * rz;
            if (!QUAD && gcw) { *gcw = make_float4(g.x, g.y, g.z, pj.w); gcw += 64; }     // the solver loops' wall terms (for_wall_cache)
        }
    };
    if (QUAD) for_nbrs_p_quad(nlbp, kb, q, wa, WP, wall);
    else for_nbrs_p(nlbp, kb, WP, wall);
    float rho_i = c.boundary_handle ? rho + rho_b * c.rho0 : rho;   // solver_base.py:49,51
    if (!owner) return;
    rho_out[i] = rho_i;
    if (RIGID) { const int raw = id[i]; rho_orig[raw < 0 ? ~raw : raw] = rho_i; }      // (by the true id: an inner ghost of a slab handle carries ~id)
    const float4 vi = V[i];
    if (DFSPH) {
        float den;
        if (c.boundary_handle)
            den = ((((sx * sx + sy * sy) + sz * sz) + sq) + bsq) + ((bx * bx + by * by) + bz * bz);   // :45
        else
            den = ((sx * sx + sy * sy) + sz * sz) + sq;                                               // :47
        float alpha = fabsf(den) < 1e-6f ? 0.0f : rho_i / den;                                        // :48-51
        aux_out[i] = alpha;
        float dt = ds->dt;
        float k_i = warm[i] / dt;                            // dfsph_solver.py:333
        if (c.kr_split) krho[i] = k_i / rho_i;
        else Pout[i] = make_float4(pi.x, pi.y, pi.z, k_i / rho_i);
        Vout[i] = make_float4(vi.x, vi.y, vi.z, rho_i);      // velocity buffers carry rho in .w (read by D5)
    } else {
        float p = tait_pressure(rho_i);                      // wcsph_solver.py:86-90
        aux_out[i] = p;
        Pout[i] = make_float4(pi.x, pi.y, pi.z, rho_i);
        Vout[i] = make_float4(vi.x, vi.y, vi.z, p / (rho_i * rho_i));   // :109,116
    }
}

// viscosity against a rigid neighbour (solver_base.py:190-201), shared by the ext-force sweeps: the body velocity is uniform,
// and rho[particle_j.index] reads the FLUID density at the rigid particle's local index (quirk, :198-199)
__device__ __forceinline__ void rigid_viscosity(const Consts &c, const RigidView &rv, const float4 vi, float rho_i, const float4 pj, uint32_t j,
                                                float dx, float dy, float dz, float r, float &wx, float &wy, float &wz)
{
    float vx = vi.x - rv.vel[0], vy = vi.y - rv.vel[1], vz = vi.z - rv.vel[2];
    float shear = dot3(vx, vy, vz, dx, dy, dz);
    const int jl = rv.rid[j & ~kRigidTag];
    if (shear < 0.f && jl < rv.n_fluid) {
        F3 g = grad_w_in(c, dx, dy, dz, r);
        float q2 = r * r;
        float nu = c.visc_num / (rho_i + rv.rho_orig[jl]);
        float pi_ = -nu * shear / (q2 + c.visc_eps_h2);
        float sv = -1000.0f * pj.w * pi_;                               // :201
        wx += sv * g.x; wy += sv * g.y; wz += sv * g.z;
    }
}

// ======================================================================================
// W2: WCSPH pressure gradient + wall pressure + viscosity + tension + kinematic phase
//     wcsph_solver.py:70-129, solver_base.py:170-217, wcsph_solver.py:40-63
//   reads P = (pos, rho), V = (vel, p/rho^2); writes the next state Pn = (pos', .), Vn = (vel', .), acc
// ======================================================================================
template <bool RIGID, bool QUAD>
__global__ __launch_bounds__(kBlock) void k_wcsph_force(Consts c, float dt, const float4 *__restrict__ P, const float4 *__restrict__ V,
                                                        const float4 *__restrict__ WP, const uint32_t *__restrict__ nl,
                                                        const uint32_t *__restrict__ nlb, const int *__restrict__ cnt,
                                                        const float *__restrict__ pressure, float4 *__restrict__ Pn,
                                                        float4 *__restrict__ Vn, float4 *__restrict__ acc_out, RigidView rv)
{
    SPH_SWEEP_PROLOGUE_M(QUAD)
    const float4 vi = V[ii];
    const float rho_i = pi.w;
    const float a_i = vi.w;                                  // p_i / rho_i_2
    float fa[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float &gx = fa[0], &gy = fa[1], &gz = fa[2];             // pressure gradient
    float &wx = fa[3], &wy = fa[4], &wz = fa[5];             // viscosity
    float &tx = fa[6], &ty = fa[7], &tz = fa[8];             // tension
    const float p_own = (RIGID || c.boundary_handle) ? (live ? pressure[ii] : 0.f) : 0.f;
    auto pair = [&](const float4 pj, const float4 vj, const uint32_t j) {
        float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        float r = norm3(dx, dy, dz);
        F3 g = grad_w_in(c, dx, dy, dz, r);
        if (RIGID && (j & kRigidTag)) {
            const float sr = -pj.w * p_own / (rho_i * rho_i);               // wcsph_solver.py:125
            gx += sr * g.x * c.rho0; gy += sr * g.y * c.rho0; gz += sr * g.z * c.rho0;
            rigid_viscosity(c, rv, vi, rho_i, pj, j, dx, dy, dz, r, wx, wy, wz);
            return;
        }
        float s = c.m * (a_i + vj.w);                        // wcsph_solver.py:116
        gx -= s * g.x; gy -= s * g.y; gz -= s * g.z;
        float vx = vi.x - vj.x, vy = vi.y - vj.y, vz = vi.z - vj.z;
        float shear = dot3(vx, vy, vz, dx, dy, dz);          // solver_base.py:183
        if (shear < 0.f) {
            float q2 = r * r;
            float nu = c.visc_num / (rho_i + pj.w);          // :187
            float pi_ = -nu * shear / (q2 + c.visc_eps_h2);  // :188
            float sv = c.neg_m * pi_;                        // :189
            wx += sv * g.x; wy += sv * g.y; wz += sv * g.z;
        }
        float st = c.tens_c * cubic_w_in(c, r);                 // :216
        tx += st * dx; ty += st * dy; tz += st * dz;
    };
    if (QUAD) for_fluid_nbrs_quad<RIGID, true>(nlp, kf, q, fa, P, V, rv, pair);
    else for_fluid_nbrs<RIGID, true>(nlp, kf, P, V, rv, pair);
    float wa[3] = {0.f, 0.f, 0.f};
    float &bx = wa[0], &by = wa[1], &bz = wa[2];
    if (c.boundary_handle) {
        const float p_i = p_own;
        const float rho_i_2 = rho_i * rho_i;
        auto wall = [&](const float4 pj) {
            float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
            float r = norm3(dx, dy, dz);
            F3 g = grad_w_in(c, dx, dy, dz, r);
            float s = pj.w * p_i / rho_i_2;                  // wcsph_solver.py:99
            bx -= s * g.x; by -= s * g.y; bz -= s * g.z;
        };
        if (QUAD) for_nbrs_p_quad(nlbp, kb, q, wa, WP, wall);
        else for_nbrs_p(nlbp, kb, WP, wall);
    }
    if (!owner) return;
    float pg[3] = {gx, gy, gz};
    float vis[3] = {wx * c.m, wy * c.m, wz * c.m};           // solver_base.py:175
    float ten[3] = {tx * c.m, ty * c.m, tz * c.m};           // solver_base.py:209
    float bac[3] = {bx * c.rho0, by * c.rho0, bz * c.rho0};  // wcsph_solver.py:83
    float acc[3] = {c.gravity * 0.0f, c.gravity * -1.0f, c.gravity * 0.0f};   // solver_base.py:131-133
    float pos[3] = {pi.x, pi.y, pi.z};
    float vel[3] = {vi.x, vi.y, vi.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (c.boundary_handle) acc[a] += ((pg[a] + vis[a]) + ten[a]) + bac[a];   // wcsph_solver.py:44-45
        else acc[a] += (pg[a] + vis[a]) + ten[a];                                // :47
        vel[a] += acc[a] * dt;                                                   // :50
        vel[a] *= 0.9998f;                                                       // :51
        pos[a] += vel[a] * dt;                                                   // :52
    }
    if (!c.boundary_handle) {                                                    // :54-63
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (pos[a] <= c.clamp_lo[a]) { pos[a] = c.clamp_lo[a]; vel[a] *= -0.5f; }
            if (pos[a] >= c.clamp_hi[a]) { pos[a] = c.clamp_hi[a]; vel[a] *= -0.5f; }
        }
    }
    Pn[i] = make_float4(pos[0], pos[1], pos[2], 0.f);
    Vn[i] = make_float4(vel[0], vel[1], vel[2], 0.f);
    acc_out[i] = make_float4(acc[0], acc[1], acc[2], 0.f);
}

// ======================================================================================
// D2 / D4 / D7: "pressure-like" velocity corrections.  All three are
//   v_i -= dt * ( sum_F m (k_i/rho_i + k_j/rho_j) gradW + rho0 * sum_B (V_b k_i / rho_i) gradW )
// with k from warm_start_k (D2, dfsph_solver.py:314-355), rho_derivative*alpha (D4, :302-312,
// 357-391, with the 1e-5 gate and sum_up_stiff :381-384) or (rho_adv-rho0)*alpha (D7, :178-219).
// P.w of every particle holds k/rho, written by the sweep before.  Vout.w carries rho.
// ======================================================================================
enum { CORR_WARM = 0, CORR_DIV = 1, CORR_DENS = 2 };

template <int MODE, bool RIGID, int SWEEP, bool RX = false>
__global__ __launch_bounds__(kBlock) void k_correct(Consts c, const float4 *__restrict__ P, const float4 *__restrict__ WP,
                                                    const uint32_t *__restrict__ nl, const uint32_t *__restrict__ nlb,
                                                    const int *__restrict__ cnt, const float *__restrict__ rho,
                                                    const float *__restrict__ alpha, const float *__restrict__ src,   // drho (DIV) / rho_adv (DENS)
                                                    float *__restrict__ warm, const DevScalars *ds,       // (no __restrict__: workgroup 0 writes the same object through fr.ds --
                                                    // the tile workgroups may read gate_hist[], dt, dt2 and the p_* parameters only, which fin_ride_block never writes)
                                                    const float4 *Vin, float4 *Vout, RigidView rv, int gate,
                                                    const uint2 *__restrict__ stage_src, const int *__restrict__ stage_cnt, const float *__restrict__ krho,
                                                    int *__restrict__ wave_dirty, unsigned char *__restrict__ changed8, const float4 *__restrict__ wall_gc,
                                                    TilePhase tp, SpecSave sv = SpecSave{nullptr, nullptr}, FinRide fr = kNoRide, DensFlow df = kNoFlow)
{
    constexpr bool STAGED = SWEEP == SWEEP_STAGED, QUAD = SWEEP == SWEEP_QUAD;
    using K = KF<RX>;                                        // kernel functions of the sweep's arithmetic (sph_device.h); RX: unstaged handles under SPH_ARITH_RELAXED
    extern __shared__ float4 s_operand[];
    if (fr.mode >= 0 && blockIdx.x == 0) { fin_ride_block(fr); return; }          // the loop decision of the residual sweep before (whatever the gate says)
    // With change propagation most tiles of a launch return at once and the ones that work are neighbours in space (the floor layer):
    // under the XCD-contiguous mapping they would all land on one or two XCDs.  Those launches deal the tiles round-robin instead.
    // (the tile is looked up BEFORE the gate is tested: two independent scalar loads, one round trip)
    const int tile = sweep_tile(tp, MODE == CORR_DENS && wave_dirty != nullptr);
    const bool flow = MODE == CORR_DENS && STAGED && wave_dirty != nullptr && df.nbr != nullptr;
    const int n2 = (flow && tp.sparse) ? tp.sparse[tp.ntiles + 1] : 0;           // (see flow_head)
    if (gate_closed(ds, gate)) return;     // Vin may alias Vout: each thread reads and writes only its own element
    if (tile < 0) return;
    // the residual sweep before this launch said which tiles can see a k / rho != 0 (DensFlow): every other tile leaves after ONE word
    StagePre pre = kNoPre;
    bool direct = false;                                     // this tile worked in the last iteration: no per-particle check, the operands in one batch
    if (flow) {
        const FlowHead fh = flow_head(tp, df, tile, n2, stage_src, stage_cnt);
        pre = fh.pre; direct = fh.direct;
        if (!fh.need) {                                      // v* stays, as verdict 2 below would find out from the staged scalars themselves
            const int i0 = tile * kBlock + (int)threadIdx.x;
            if ((threadIdx.x & 63) == 0) wave_dirty[tile * (kBlock / 64) + (threadIdx.x >> 6)] = 0;
            if (i0 < c.n) changed8[i0] = 0;
            if (direct && threadIdx.x == 0) df.worked[tile] = 0;
            return;
        }
    }
    const int my_nbr = (flow && threadIdx.x < 64) ? df.nbr[(size_t)tile * kNbrStride + threadIdx.x] : 0;       // requested now, used by the push at the end
    SPH_SWEEP_PROLOGUE_G(QUAD, tile, true)
    // kr_split: P is the step's position array and k / rho of the neighbours comes from krho[]; else P = (pos, k / rho)
    const bool split = STAGED && c.kr_split;
    // change propagation in the density loop (stage_sources_flagged); with a body in the lists too: its term is V_r rho0 k_i / rho_i grad W, zero with k_i
    const bool track = MODE == CORR_DENS && STAGED && wave_dirty != nullptr;
    // Two-column slab handles store the ghost columns in tiles of their own (slab_cell_order): a tile in which nobody has a list -- ghosts of the
    // outer column -- has nothing to correct and nothing to stage (in place: Vin == Vout there, and nobody reads an outer ghost's velocity)
    if (c.ghost_walk && Vin == Vout && !__syncthreads_or(live && (cw & 0x7fffffff) != 0)) {
        if (track) {
            if ((threadIdx.x & 63) == 0) wave_dirty[blk * (kBlock / 64) + (threadIdx.x >> 6)] = 0;
            if (live) changed8[i] = 0;
        }
        return;
    }
    bool staged;
    if (track && !direct) {
        const int verdict = split ? stage_operand_ps_checked<true>(c, s_operand, P, krho, stage_src, stage_cnt, blk, pre)
                                  : stage_operand_w_checked<true>(c, s_operand, P, stage_src, stage_cnt, blk, pre);
        if (verdict == 2) {                                                        // every k / rho this tile can see is 0: v* stays
            // (one-column slab handles: a ghost's v* is refreshed from its owner after this sweep and may change behind this rank's back,
            // whatever this rank can see; on two-column handles the inner ghosts are corrected HERE, from the same inputs as on their owner)
            const bool foreign = live && ghost && !c.ghost_walk;
            const unsigned long long anyg = __ballot(foreign);
            if ((threadIdx.x & 63) == 0) wave_dirty[blk * (kBlock / 64) + (threadIdx.x >> 6)] = anyg != 0ull ? 1 : 0;
            if (live) changed8[i] = foreign ? 1 : 0;
            return;
        }
        staged = verdict == 1;
    } else {
        staged = STAGED && (split ? stage_operand_ps_scaled(c, s_operand, P, krho, stage_src, stage_cnt, blk, pre)
                                  : stage_operand<true>(c, s_operand, P, stage_src, stage_cnt, blk, pre));     // positions * 2^32
    }
    const float dt = ds->dt;
    const float rho_i = rho[ii];
    float k_i;
    if (MODE == CORR_WARM) k_i = warm[ii] / dt;                                   // :333
    else if (MODE == CORR_DIV) k_i = src[ii] * alpha[ii] / dt;                    // :363
    else k_i = (src[ii] - c.rho0) * alpha[ii] / ds->dt2;                          // :199
    const float kr_i = k_i / rho_i;
    float fa[3] = {0.f, 0.f, 0.f};
    float &ax = fa[0], &ay = fa[1], &az = fa[2];
    auto pair = [&](const float4 pj, const float4, const uint32_t j) {
        float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        float r = K::norm3(dx, dy, dz);
        F3 g = K::grad_in(c, dx, dy, dz, r);
        if (RIGID && (j & kRigidTag)) {
            float s = pj.w * c.rho0 * k_i / rho_i;                                // :345 / :377 / :211  (no 1e-5 gate)
            ax += s * g.x; ay += s * g.y; az += s * g.z;
        } else {
            float ks = kr_i + pj.w;
            if (MODE != CORR_DIV || ks > 1e-5f) {                                 // :367
                float s = c.m * ks;                                               // :337 / :369 / :203
                ax += s * g.x; ay += s * g.y; az += s * g.z;
            }
        }
    };
    // the same pair with every position carrying 2^32 (staged operands): two multiplications less per pair, same bits
    const float sx_i = pi.x * 0x1p32f, sy_i = pi.y * 0x1p32f, sz_i = pi.z * 0x1p32f;
    auto pair_scaled = [&](const float4 pj, const float4, const uint32_t j) {
        float dx = sx_i - pj.x, dy = sy_i - pj.y, dz = sz_i - pj.z;
        float r = norm3_scaled(dx, dy, dz);
        F3 g = grad_w_scaled(c, dx, dy, dz, r);
        if (RIGID && (j & kRigidTag)) {
            float s = pj.w * c.rho0 * k_i / rho_i;
            ax += s * g.x; ay += s * g.y; az += s * g.z;
        } else {
            float ks = kr_i + pj.w;
            if (MODE != CORR_DIV || ks > 1e-5f) {
                float s = c.m * ks;
                ax += s * g.x; ay += s * g.y; az += s * g.z;
            }
        }
    };
    struct OperandPS { float4 a; float s; };
    if (QUAD) for_fluid_nbrs_quad<RIGID, false>(nlp, kf, q, fa, P, nullptr, rv, pair);
    else if (staged && (RIGID ? stage_lists16(stage_cnt, blk) : c.nl16 != 0)) for_staged16_nbrs(nlp, kf, s_operand, pair_scaled);
    else if (staged) for_staged_nbrs<RIGID, true>(nlp, kf, s_operand, rv, pair_scaled);
    else if (split)          // a workgroup of a kr_split handle whose set did not fit: two global gathers per neighbour (a tagged entry: the rigid sample)
        walk_list<OperandPS>(nlp, kf, [&](uint32_t j, OperandPS &o) {
                                 const bool rg = RIGID && (j & kRigidTag);
                                 const uint32_t idx = RIGID ? (j & ~kRigidTag) : j;
                                 o.a = rg ? rv.RP[idx] : P[idx]; o.s = rg ? o.a.w : krho[idx];
                             },
                             [&](const OperandPS &o, uint32_t j) { pair(make_float4(o.a.x, o.a.y, o.a.z, o.s), make_float4(0.f, 0.f, 0.f, 0.f), j); });
    else for_fluid_nbrs<RIGID, false>(nlp, kf, P, nullptr, rv, pair);
    float wa[3] = {0.f, 0.f, 0.f};
    float &bx = wa[0], &by = wa[1], &bz = wa[2];
    auto wall = [&](const float4 pj) {
        float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        float r = K::norm3(dx, dy, dz);
        F3 g = K::grad_in(c, dx, dy, dz, r);
        float s = pj.w * k_i / rho_i;                                             // :354 / :390 / :219
        bx += s * g.x; by += s * g.y; bz += s * g.z;
    };
    if (QUAD) for_nbrs_p_quad(nlbp, kb, q, wa, WP, wall);
    else if (wall_gc)
        for_wall_cache(wall_gc + gc_index(ii, 0, c.kbpitch), kb, [&](const float4 gv) {       // (grad W_ib, V_b) as D1 left them
            float s = gv.w * k_i / rho_i;
            bx += s * gv.x; by += s * gv.y; bz += s * gv.z;
        });
    else for_nbrs_p(nlbp, kb, WP, wall);
    if (track) {       // did any lane of this wave apply a correction?  (all sums +-0: v - (+-0) * dt leaves v)
        // (one-column slab handles: a ghost's v* is refreshed from its owner after this sweep, it may change behind this rank's back; two-column
        // handles: the inner ghosts' corrections are computed here like everybody's, the outer ghosts' v* is never read)
        const bool changed = live && ((ghost && !c.ghost_walk) || ax != 0.f || ay != 0.f || az != 0.f || bx != 0.f || by != 0.f || bz != 0.f);
        const unsigned long long any = __ballot(changed);
        if ((threadIdx.x & 63) == 0) wave_dirty[blk * (kBlock / 64) + (threadIdx.x >> 6)] = any != 0ull ? 1 : 0;
        if (live) changed8[i] = changed ? 1 : 0;
        if (flow) {                                                               // every tile that stages a particle of this one must run the next residual sweep
            const int moved = __syncthreads_or(changed ? 1 : 0);
            if (moved && threadIdx.x < 64) flow_push(df, my_nbr);
            if (threadIdx.x == 0) df.worked[tile] = moved ? 1 : 0;
        }
    }
    if (!owner) return;
    float4 v = Vin[i];
    if (MODE == CORR_DIV && sv.v) { sv.v[i] = v; sv.w[i] = warm[i]; }             // this sweep runs AHEAD of the loop decision: what the undo restores
    if (c.boundary_handle) {
        // :322 / :310 ; for D7 :187 then :191 -- same association: (a + b*rho0) * dt
        v.x -= (ax + bx * c.rho0) * dt;
        v.y -= (ay + by * c.rho0) * dt;
        v.z -= (az + bz * c.rho0) * dt;
    } else {
        v.x -= ax * dt; v.y -= ay * dt; v.z -= az * dt;                           // :324 / :312 / :189
    }
    v.w = rho_i;
    Vout[i] = v;
    if (MODE == CORR_WARM) warm[i] = 0.0f;                                        // :325
    if (MODE == CORR_DIV && c.warm_start) warm[i] += src[i] * alpha[i];           // :384 (sum_up_stiff runs under warm_start only, :404-405)
}

// ======================================================================================
// D3 / D6: divergence residual and predicted density.
//   D3 (dfsph_solver.py:252-300): drho_i = cnt<20 ? 0 : max(sum_F m (v_i-v_j).gradW + rho0 sum_B V_b v_i.gradW, 0)
//   D6 (dfsph_solver.py:124-176): rho*_i = max(rho_i + dt (same sums with v*), rho0)
// Writes Pout.w = k/rho for the correction sweep that follows and the block partials of the mean.
// ======================================================================================
template <bool DENS, bool RIGID, int SWEEP, bool RX = false>
__global__ __launch_bounds__(kBlock) void k_residual(Consts c, const float4 *__restrict__ P, const float4 *__restrict__ V,
                                                     const float4 *__restrict__ WP, const uint32_t *__restrict__ nl,
                                                     const uint32_t *__restrict__ nlb, const int *__restrict__ cnt,
                                                     const float *__restrict__ rho, const float *__restrict__ alpha,
                                                     DevScalars *__restrict__ ds, float *__restrict__ out,
                                                     float4 *__restrict__ Pout, double *__restrict__ psum, int *__restrict__ pcnt,
                                                     RigidView rv, const int *__restrict__ ncount, int gate,
                                                     const uint2 *__restrict__ stage_src, const int *__restrict__ stage_cnt, float *__restrict__ krho,
                                                     const int *__restrict__ wave_dirty, const unsigned char *__restrict__ changed8, int force_all,
                                                     const float4 *__restrict__ wall_gc, TilePhase tp, SpecUndo un = SpecUndo{nullptr, nullptr, nullptr, nullptr, 0},
                                                     DensFlow df = kNoFlow)
{
    constexpr bool STAGED = SWEEP == SWEEP_STAGED, QUAD = SWEEP == SWEEP_QUAD;
    using K = KF<RX>;                                        // kernel functions of the sweep's arithmetic (sph_device.h); RX: unstaged handles under SPH_ARITH_RELAXED
    extern __shared__ float4 s_operand[];
    // (see k_correct: round-robin tiles when most of them return at once; the body does not move inside a solver loop, so its terms stand with v*)
    const bool spread = DENS && STAGED && wave_dirty && !force_all;
    const int tile = sweep_tile(tp, spread);
    const bool flow = DENS && STAGED && df.nbr != nullptr;           // the producer says who must run (DensFlow)
    const int n2 = (flow && spread && tp.sparse) ? tp.sparse[tp.ntiles + 1] : 0;      // (see flow_head)
    if (gate_closed(ds, gate)) { spec_undo(c, un, ds, tp); return; }
    if (tile < 0) return;
    StagePre pre = kNoPre;
    bool direct = false;                                             // this tile worked in the last iteration: no per-particle check (see StagePre)
    if (spread) {                                                    // change propagation, see stage_sources_flagged
        bool idle;
        if (flow) {
            const FlowHead fh = flow_head(tp, df, tile, n2, stage_src, stage_cnt);
            pre = fh.pre; direct = fh.direct;
            idle = !fh.need;                                         // ONE word: did the correction sweep before this launch change a velocity this tile stages?
            // its k / rho stands; where that is not zero, the correction sweep behind this launch must still be told
            if (idle && df.nz[tile] != 0 && threadIdx.x < 64) flow_push(df, df.nbr[(size_t)tile * kNbrStride + threadIdx.x]);
            if (idle && direct && threadIdx.x == 0) df.worked[tile] = 0;
        } else {
            const int sw = stage_cnt[tile];
            idle = sw >= 0 && !stage_sources_flagged(stage_src, sw, tile, wave_dirty);
        }
        if (tp.hot && threadIdx.x == 0 && idle) tp.hot[tile] = 0;    // (hot: 0 = left here, 1 = left after the per-particle check, 2 = worked)
        if (idle) return;                                            // rho*, k / rho and the block partial of the last iteration stand
    }
    const int my_nbr = (flow && threadIdx.x < 64) ? df.nbr[(size_t)tile * kNbrStride + threadIdx.x] : 0;      // requested now, used by the push at the end
    SPH_SWEEP_PROLOGUE_B(QUAD, tile)
    // ... and a tile without an owned particle -- ghosts only -- computes no residual (the ghosts' values arrive with the halo): no staging, a zero partial
    if (c.ghost_walk && !__syncthreads_or(live && !ghost)) {
        if (QUAD) block_partial_mean_quad(blk, 0.0, 0, owner, psum, pcnt);
        else block_partial_mean(blk, 0.0, 0, psum, pcnt);
        if (flow && threadIdx.x == 0) df.nz[blk] = 0;               // (the ghosts' k / rho is pushed for by the kernel that unpacks it)
        return;
    }
    float2 *s_v2 = reinterpret_cast<float2 *>(s_operand + c.stage_cap);
    bool staged;
    if (spread && !direct) {       // second level of the change propagation: did the v* of any staged PARTICLE change?  (the flagged waves said "maybe")
        const int verdict = stage_operand_pv_checked<true>(c, s_operand, s_v2, P, V, changed8, stage_src, stage_cnt, blk, pre);
        if (verdict == 2) {                                          // (its k / rho stands like an idle tile's)
            if (flow && df.nz[blk] != 0 && threadIdx.x < 64) flow_push(df, my_nbr);
            if (tp.hot && threadIdx.x == 0) tp.hot[blk] = 1;
            return;
        }
        staged = verdict == 1;
    } else {
        // (a `direct` tile: the bytes of the check ride in the staging batch and say whether it would have passed -- if not, it computes what
        // stands this once and takes the check again next time)
        int would = 1;
        staged = STAGED && stage_operand_pv<true>(c, s_operand, s_v2, P, V, stage_src, stage_cnt, blk, pre, (spread && direct) ? changed8 : nullptr, &would);   // positions * 2^32
        if (spread && flow && threadIdx.x == 0) df.worked[blk] = would ? 1 : 0;
        if (spread && tp.hot && threadIdx.x == 0) tp.hot[blk] = would ? 2 : 1;
    }
    if (spread && !direct) {
        if (tp.hot && threadIdx.x == 0) tp.hot[blk] = 2;
        if (flow && threadIdx.x == 0) df.worked[blk] = 1;
    }
    const float4 vi = V[ii];
    float fa[1] = {0.f};
    float &acc = fa[0];
    const int nq = RIGID ? (live ? ncount[ii] : 0) : kf;                          // ps.get_neighbour_count(i)
    const bool skip = !DENS && nq < 20;                                           // :258-261
    const float dt_r = RIGID ? ds->dt : 0.f;
    auto pair = [&](const float4 pj, const float4 vj, const uint32_t j) {
        float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        float r = K::norm3(dx, dy, dz);
        F3 g = K::grad_in(c, dx, dy, dz, r);
        if (RIGID && (j & kRigidTag)) {
            const F3 w = rigid_velocity(rv, pj, dt_r, DENS);                      // :292-293 / :168-169
            acc += pj.w * c.rho0 * dot3(vi.x - w.x, vi.y - w.y, vi.z - w.z, g.x, g.y, g.z);   // :294 / :170
        } else {
            acc += c.m * dot3(vi.x - vj.x, vi.y - vj.y, vi.z - vj.z, g.x, g.y, g.z);          // :287 / :162
        }
    };
    const float sx_i = pi.x * 0x1p32f, sy_i = pi.y * 0x1p32f, sz_i = pi.z * 0x1p32f;
    auto pair_scaled = [&](const float4 pj, const float4 vj, const uint32_t j) {      // positions carry 2^32 (see k_correct)
        float dx = sx_i - pj.x, dy = sy_i - pj.y, dz = sz_i - pj.z;
        float r = norm3_scaled(dx, dy, dz);
        F3 g = grad_w_scaled(c, dx, dy, dz, r);
        if (RIGID && (j & kRigidTag)) {
            const float4 pu = make_float4(pj.x * 0x1p-32f, pj.y * 0x1p-32f, pj.z * 0x1p-32f, pj.w);
            const F3 w = rigid_velocity(rv, pu, dt_r, DENS);
            acc += pj.w * c.rho0 * dot3(vi.x - w.x, vi.y - w.y, vi.z - w.z, g.x, g.y, g.z);
        } else {
            acc += c.m * dot3(vi.x - vj.x, vi.y - vj.y, vi.z - vj.z, g.x, g.y, g.z);
        }
    };
    if (QUAD) for_fluid_nbrs_quad<RIGID, true>(nlp, skip ? 0 : kf, q, fa, P, V, rv, pair);
    else if (staged && (RIGID ? stage_lists16(stage_cnt, blk) : c.nl16 != 0)) for_staged16_nbrs_pv2(nlp, skip ? 0 : kf, s_operand, s_v2, pair_scaled);
    else if (staged) for_staged_nbrs_pv2<RIGID, true>(nlp, skip ? 0 : kf, s_operand, s_v2, rv, pair_scaled);
    else for_fluid_nbrs<RIGID, true>(nlp, skip ? 0 : kf, P, V, rv, pair);
    float wa[1] = {0.f};
    float &accb = wa[0];
    auto wall = [&](const float4 pj) {
        float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        float r = K::norm3(dx, dy, dz);
        F3 g = K::grad_in(c, dx, dy, dz, r);
        accb += pj.w * dot3(vi.x, vi.y, vi.z, g.x, g.y, g.z);                     // :300 / :176
    };
    if (QUAD) for_nbrs_p_quad(nlbp, skip ? 0 : kb, q, wa, WP, wall);
    else if (wall_gc)
        for_wall_cache(wall_gc + gc_index(ii, 0, c.kbpitch), skip ? 0 : kb, [&](const float4 gv) {
            accb += gv.w * dot3(vi.x, vi.y, vi.z, gv.x, gv.y, gv.z);
        });
    else for_nbrs_p(nlbp, skip ? 0 : kb, WP, wall);
    float val = 0.f, kr = 0.f;
    int flag = 0;
    if (live) {
        const float rho_i = rho[i];
        if (DENS) {
            const float dt = ds->dt;
            if (c.boundary_handle) val = rmax(rho_i + dt * (acc + accb * c.rho0), c.rho0);   // :135
            else val = rmax(rho_i + dt * acc, c.rho0);                                        // :137
            flag = !(val == c.rho0) && !ghost;                                                // :139
            kr = ((val - c.rho0) * alpha[i] / ds->dt2) / rho_i;                               // :199,203
        } else {
            if (skip) val = 0.f;
            else if (c.boundary_handle) val = rmax(acc + accb * c.rho0, 0.0f);                // :267
            else val = rmax(acc, 0.0f);                                                       // :269
            flag = val > 0.f && !ghost;                                                       // :275
            kr = (val * alpha[i] / ds->dt) / rho_i;                                           // :363,367
        }
        if (owner && !(ghost && c.ghost_walk)) {      // (two-column slab handles: a ghost's value and k / rho arrive with the halo, which may overlap this launch)
            out[i] = val;
            if (c.kr_split) krho[i] = kr;
            else Pout[i] = make_float4(pi.x, pi.y, pi.z, kr);
        }
    }
    if (QUAD) block_partial_mean_quad(blk, (double)val, flag, owner, psum, pcnt);
    else block_partial_mean(blk, (double)val, flag, psum, pcnt);
    if (flow) {                                                      // does this tile hold a k / rho != 0?  (a NaN counts)
        const int nzf = __syncthreads_or((live && !ghost && kr != 0.f) ? 1 : 0);
        if (threadIdx.x == 0) df.nz[blk] = nzf ? 1 : 0;
        if (nzf && threadIdx.x < 64) flow_push(df, my_nbr);
    }
}

// ======================================================================================
// D5: tension + viscosity + external force + v* and max |v*|
//     solver_base.py:170-217, dfsph_solver.py:91-103.   V = (vel, rho)
// ======================================================================================
template <bool RIGID, int SWEEP, bool RX = false>
__global__ __launch_bounds__(kBlock) void k_dfsph_ext(Consts c, const float4 *__restrict__ P, const float4 *__restrict__ V,
                                                      const uint32_t *__restrict__ nl, const int *__restrict__ cnt,
                                                      const DevScalars *__restrict__ ds, float4 *__restrict__ VAout,
                                                      float *__restrict__ pmax, RigidView rv,
                                                      const uint2 *__restrict__ stage_src, const int *__restrict__ stage_cnt, TilePhase tp = TilePhase{nullptr, 0, 0})
{
    constexpr bool STAGED = SWEEP == SWEEP_STAGED, QUAD = SWEEP == SWEEP_QUAD;
    using K = KF<RX>;                                        // kernel functions of the sweep's arithmetic (sph_device.h); RX: unstaged handles under SPH_ARITH_RELAXED
    extern __shared__ float4 s_operand[];
    const uint32_t *nlb = nullptr;
    const int tile = tp.phase == 0 ? xcd_block(blockIdx.x, gridDim.x) : sweep_tile(tp, false);
    if (tile < 0) return;
    SPH_SWEEP_PROLOGUE_B(QUAD, tile)
    (void)kb; (void)nlbp;
    uint32_t *s_src = reinterpret_cast<uint32_t *>(s_operand + c.stage_cap);      // (vel, rho) needs 16 B: gathered from memory
    const bool staged = STAGED && stage_operand_src(c, s_operand, s_src, P, stage_src, stage_cnt, blk);
    const float4 vi = V[ii];
    const float rho_i = vi.w;
    float fa[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float &wx = fa[0], &wy = fa[1], &wz = fa[2];
    float &tx = fa[3], &ty = fa[4], &tz = fa[5];
    auto pair = [&](const float4 pj, const float4 vj, const uint32_t j) {
        float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        float r = K::norm3(dx, dy, dz);
        if (RIGID && (j & kRigidTag)) {
            // solver_base.py:190-201: no tension from rigid neighbours; viscosity against the body velocity, with
            // rho[particle_j.index] = the FLUID density at the rigid particle's local index (quirk, :198-199)
            float vx = vi.x - rv.vel[0], vy = vi.y - rv.vel[1], vz = vi.z - rv.vel[2];
            float shear = dot3(vx, vy, vz, dx, dy, dz);
            const int jl = rv.rid[j & ~kRigidTag];
            if (shear < 0.f && jl < rv.n_fluid) {
                F3 g = K::grad_in(c, dx, dy, dz, r);
                float q2 = r * r;
                float nu = c.visc_num / (rho_i + rv.rho_orig[jl]);
                float pi_ = -nu * shear / (q2 + c.visc_eps_h2);
                float sv = -1000.0f * pj.w * pi_;                               // :201
                wx += sv * g.x; wy += sv * g.y; wz += sv * g.z;
            }
            return;
        }
        float st = c.tens_c * K::w_in(c, r);                 // solver_base.py:216
        tx += st * dx; ty += st * dy; tz += st * dz;
        float vx = vi.x - vj.x, vy = vi.y - vj.y, vz = vi.z - vj.z;
        float shear = dot3(vx, vy, vz, dx, dy, dz);          // :183
        if (shear < 0.f) {
            F3 g = K::grad_in(c, dx, dy, dz, r);
            float q2 = r * r;
            float nu = c.visc_num / (rho_i + vj.w);          // :187
            float pi_ = -nu * shear / (q2 + c.visc_eps_h2);  // :188
            float sv = c.neg_m * pi_;                        // :189
            wx += sv * g.x; wy += sv * g.y; wz += sv * g.z;
        }
    };
    if (QUAD) for_fluid_nbrs_quad<RIGID, true>(nlp, kf, q, fa, P, V, rv, pair);
    else if (staged && (RIGID ? stage_lists16(stage_cnt, blk) : c.nl16 != 0)) for_staged16_nbrs_pv(nlp, kf, s_operand, s_src, V, pair);
    else if (staged) for_staged_nbrs_pv<RIGID>(nlp, kf, s_operand, s_src, V, rv, pair);
    else for_fluid_nbrs<RIGID, true>(nlp, kf, P, V, rv, pair);
    float vn = -INFINITY;
    if (live) {
        const float dt = ds->dt;
        float ten[3] = {tx * c.m, ty * c.m, tz * c.m};       // :209
        float vis[3] = {wx * c.m, wy * c.m, wz * c.m};       // :175
        float g[3] = {c.gravity * 0.0f, c.gravity * -1.0f, c.gravity * 0.0f};
        float v[3] = {vi.x, vi.y, vi.z};
        float va[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float f = (g[a] + ten[a]) + vis[a];              // dfsph_solver.py:96
            va[a] = v[a] + dt * f / c.m;                     // :102
        }
        if (owner) VAout[i] = make_float4(va[0], va[1], va[2], rho_i);
        if (!ghost) vn = K::norm3(va[0], va[1], va[2]);         // :103
    }
    block_partial_max(blk, vn, pmax);
}

// D8: compute_all_position                                  dfsph_solver.py:235-250
__global__ __launch_bounds__(kBlock) void k_dfsph_integrate(Consts c, const float4 *__restrict__ P, const float4 *__restrict__ VA,
                                                            const DevScalars *__restrict__ ds, float4 *__restrict__ Pn,
                                                            float4 *__restrict__ Vn)
{
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= c.n) return;
    const float dt = ds->dt;
    float4 p = P[i], va = VA[i];
    float pos[3] = {p.x, p.y, p.z};
    float v[3] = {va.x, va.y, va.z};
    float vel[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        pos[a] = pos[a] + dt * v[a] * 0.9999f;               // :238
        vel[a] = v[a] * 0.9999f;                             // :239
    }
    if (!c.boundary_handle) {                                // :241-250
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (pos[a] <= c.clamp_lo[a]) { pos[a] = c.clamp_lo[a]; vel[a] *= -0.5f; }
            if (pos[a] >= c.clamp_hi[a]) { pos[a] = c.clamp_hi[a]; vel[a] *= -0.5f; }
        }
    }
    Pn[i] = make_float4(pos[0], pos[1], pos[2], 0.f);
    Vn[i] = make_float4(vel[0], vel[1], vel[2], 0.f);
}


// ======================================================================================
// original-order <-> sorted-order transfers for the C-ABI (field.to_numpy / from_numpy)
// ======================================================================================
__global__ __launch_bounds__(kBlock) void k_unsort_vec(int n, const float4 *__restrict__ src, const int *__restrict__ id, float *__restrict__ dst)
{
    int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n) return;
    float4 v = src[s];
    int o = id[s];
    dst[3 * o] = v.x; dst[3 * o + 1] = v.y; dst[3 * o + 2] = v.z;
}
__global__ __launch_bounds__(kBlock) void k_unsort_scalar(int n, const float *__restrict__ src, const int *__restrict__ id, float *__restrict__ dst)
{
    int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n) return;
    dst[id[s]] = src[s];
}
__global__ __launch_bounds__(kBlock) void k_unsort_w(int n, const float4 *__restrict__ src, const int *__restrict__ id, float *__restrict__ dst)
{
    int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n) return;
    dst[id[s]] = src[s].w;
}
__global__ __launch_bounds__(kBlock) void k_unsort_count(int n, const int *__restrict__ cnt, const int *__restrict__ id, float *__restrict__ dst)
{
    int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n) return;
    dst[id[s]] = (float)(cnt[s] & 0xffff);
}
__global__ __launch_bounds__(kBlock) void k_unsort_scalar_int(int n, const int *__restrict__ src, const int *__restrict__ id, float *__restrict__ dst)
{
    int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n) return;
    dst[id[s]] = (float)src[s];
}
__global__ __launch_bounds__(kBlock) void k_sort_in_vec(int n, const float *__restrict__ src, const int *__restrict__ id, float4 *__restrict__ dst)
{
    int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n) return;
    int o = id[s];
    dst[s] = make_float4(src[3 * o], src[3 * o + 1], src[3 * o + 2], 0.f);
}
__global__ __launch_bounds__(kBlock) void k_sort_in_scalar(int n, const float *__restrict__ src, const int *__restrict__ id, float *__restrict__ dst)
{
    int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n) return;
    dst[s] = src[id[s]];
}

// device arithmetic self-test (sph_selftest_math)
__global__ void k_selftest(Consts c, int op, const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b[i];
    float r;
    if (op == 0) r = x / y;
    else if (op == 1) r = sqrtf(x);
    else if (op == 2) { c.h = y; r = cubic_w(c, x); }
    else if (op == 6) r = sqrt_rn(x);
    else {
        float dz = 0.25f * x;
        float rn = norm3(x, y, dz);
        F3 g = grad_w(c, x, y, dz, rn);
        r = op == 3 ? g.x : (op == 4 ? g.y : g.z);
    }
    out[i] = r;
}

// wave primitive self-test (sph_selftest_wave): every lane's result of one primitive over the 64 values of its wave
__global__ void k_selftest_wave(int op, const double *__restrict__ in, double *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const double x = in[i];
    double r;
    if (op == 0) r = wave_sum(x);
    else if (op == 1) r = (double)wave_sum((int)x);
    else if (op == 2) r = (double)wave_max((float)x);
    else if (op == 3) r = (double)wave_max((int)x);
    else r = (double)wave_inclusive_scan((int)x);
    out[i] = r;
}

}  // namespace sph
