// sph_relaxed_kernels.h -- tolerance-grade forms of the DFSPH pair sweeps (SphConfig.arith = SPH_ARITH_RELAXED): the four of the solver loops
// that dominate a step (k_residual / k_correct) and the two per-step ones (density + alpha, external forces).
//
// The EXACT sweeps (sph_kernels.h) evaluate the reference's f32 expressions operation for operation -- a correctly rounded square root
// and three correctly rounded divisions per pair, no contraction, sums in the single-thread order of the reference's cell lists -- so
// that they can be tested bit for bit against oracle/.  north_star's bar is 1e-5 relative on positions / velocities after N steps, and
// the reference itself does not keep an order (its cell lists are appended from a parallel loop, ParticleSystem.py:388-397; its runtime's
// default is fast-math).  These kernels use that latitude for arithmetic only -- same pairs, same lists, same staging plan, same
// loop control, same buffers -- with
//     grad W_ij = s(q) x_ij / (h r)  =  g x_ij,    one scalar per pair:
//         q <= 0.5:  g = (kg6 / h^2) (3 q - 2)            no reciprocal at all
//         q >  0.5:  g = (-kg6 / h) (1 - q)^2 / r         1 / r = v_rsq_f32(r^2), r = r^2 / r
//     FMAs throughout, the particle mass folded into the constants, no 1e-5 gate on q (solver_base.py:97 zeroes the gradient of a pair
//     closer than 1e-6 m, whose term here is g * x_ij with a finite g and x_ij -> 0; r^2 is floored so that coincident particles give 0).
// ~23 VALU instructions per pair instead of ~62 (tools/pair_body_relaxed.hip: 45 -> 19 us and 39 -> 17 us per sweep-equivalent).
// What is NOT relaxed: list membership (decided by the exact k_build_nl), the `ks > 1e-5` gate of the divergence correction
// (dfsph_solver.py:367: semantics, not rounding), the per-particle epilogues, the f64 block partials and the device-side loop decisions.
//
// Used by handles with kr_split lists (single GPU, staged, 16-bit lists, no rigid body: the large scenes BASELINE.json quotes); every
// other handle runs the exact sweeps whatever `arith` says -- relaxed is a permission, not an obligation.
// tests/test_relaxed_gpu.py holds these against the oracle at 1e-5 after N steps and inside the reference's own nondeterminism
// envelope (tools/envelope.py).
#pragma once
#include "sph_kernels.h"

namespace sph {

// m * s / (h r) for the difference vector (dx, dy, dz) of a list member
__device__ __forceinline__ float rx_g(const Consts &c, float dx, float dy, float dz)
{
    const float r2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, __builtin_fmaf(dx, dx, 1e-30f)));
    const float ri = __builtin_amdgcn_rsqf(r2);
    const float q = (r2 * ri) * c.rh;
    const float g1 = __builtin_fmaf(q, c.rx_k1a, c.rx_k1b);
    const float t = 1.0f - q;
    const float g2 = ((t * t) * ri) * c.rx_k2;
    return q <= 0.5f ? g1 : g2;
}

// positions and k / rho of the workgroup's staged set, unscaled (the relaxed counterpart of stage_operand_ps_scaled)
__device__ __forceinline__ bool stage_operand_ps(const Consts &c, float4 *__restrict__ s_A, const float4 *__restrict__ A, const float *__restrict__ S,
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
            if (base + u * kBlock < nst) s_A[base + u * kBlock] = make_float4(a[u].x, a[u].y, a[u].z, sc[u]);
    }
    __syncthreads();
    return true;
}

// Walk of a 16-bit list eight entries at a time WITHOUT tail masks: k_build_nl pads the last group of a list with the particle's own
// local index (NlWriter::flush), and the particle itself contributes x_ij = 0, v_ij = 0: a term that is exactly 0 (g stays finite: r^2 is
// floored in rx_g).  Eight independent pair bodies per trip in one basic block; tools/pair_body_relaxed.hip: 22.1 -> 19.1 us per sweep.
template <class Pair8>
__device__ __forceinline__ void rx_walk8(const uint32_t *__restrict__ base, int cnt, Pair8 pair8)
{
    if (cnt <= 0) return;
    uint4 jn = nl_load(base);
    for (int kk = 0; kk < cnt; kk += 8) {
        const Nl16Group g = {{jn.x, jn.y, jn.z, jn.w}};
        if (kk + 8 < cnt) jn = nl_load(base + (size_t)((kk >> 3) + 1) * 256);
        pair8(g);
    }
}

// W / kw and m s / (h r) of one pair from its difference vector: the two scalars D1 and D5 need (rx_g is the second alone)
struct RxWG { float w, g, r2; };
__device__ __forceinline__ RxWG rx_wg(const Consts &c, float dx, float dy, float dz)
{
    RxWG o;
    o.r2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, __builtin_fmaf(dx, dx, 1e-30f)));
    const float ri = __builtin_amdgcn_rsqf(o.r2);
    const float q = (o.r2 * ri) * c.rh;
    const float t = 1.0f - q, t2 = t * t;
    const float w1 = __builtin_fmaf(6.0f * (q * q), q - 1.0f, 1.0f);          // 6 (q^3 - q^2) + 1          solver_base.py:76-88
    const float w2 = 2.0f * (t2 * t);
    const float g1 = __builtin_fmaf(q, c.rx_k1a, c.rx_k1b);
    const float g2 = (t2 * ri) * c.rx_k2;
    const bool inner = q <= 0.5f;
    o.w = inner ? w1 : w2;
    o.g = inner ? g1 : g2;
    return o;
}

// The wall sums of a step, per fluid particle (walls are static, positions frozen between the grid rebuild and the integrator):
//   G.xyz = sum_B V_b m grad W_ib     D3 / D6 need v_i . G_i, D2 / D4 / D7 (k_i / rho_i) (rho0 / m) G_i, D1 its square
//   G.w   = sum_B V_b W_ib            D1: the walls' share of rho_i                        (solver_base.py:70-71)
//   Gsq   = sum_B |V_b m grad W_ib|^2 D1: the walls' share of alpha's denominator          (dfsph_solver.py:82-89)
__global__ __launch_bounds__(kBlock) void k_rx_wall_grad(Consts c, const float4 *__restrict__ P, const float4 *__restrict__ WP,
                                                         const uint32_t *__restrict__ nlb, const int *__restrict__ cnt, float4 *__restrict__ G,
                                                         float *__restrict__ Gsq)
{
    const uint32_t *nl = nullptr;
    SPH_SWEEP_PROLOGUE_G(false, xcd_block(blockIdx.x, gridDim.x), true)
    (void)nlp; (void)kf;
    float gx = 0.f, gy = 0.f, gz = 0.f, ws = 0.f, sq = 0.f;
    auto wall = [&](const float4 pj) {
        const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        const RxWG k = rx_wg(c, dx, dy, dz);
        const float s = pj.w * k.g;
        const float tx = s * dx, ty = s * dy, tz = s * dz;
        gx += tx; gy += ty; gz += tz;
        sq = __builtin_fmaf(tz, tz, __builtin_fmaf(ty, ty, __builtin_fmaf(tx, tx, sq)));
        ws = __builtin_fmaf(pj.w, k.w, ws);
    };
    for_nbrs_p(nlbp, kb, WP, wall);
    if (live) { G[i] = make_float4(gx, gy, gz, ws * c.kw); Gsq[i] = sq; }
}

// D3 / D6 (k_residual)                                           dfsph_solver.py:252-300, 124-176
// The wall sums come from the per-step array G (k_rx_wall_grad): sum_B V_b v_i . grad W_ib = v_i . G_i / m.
template <bool DENS>
__global__ __launch_bounds__(kBlock) void k_residual_rx(Consts c, const float4 *__restrict__ P, const float4 *__restrict__ V,
                                                        const float4 *__restrict__ G, const uint32_t *__restrict__ nl, const int *__restrict__ cnt,
                                                        const float *__restrict__ rho, const float *__restrict__ alpha,
                                                        DevScalars *__restrict__ ds, float *__restrict__ out,
                                                        double *__restrict__ psum, int *__restrict__ pcnt, int gate,
                                                        const uint2 *__restrict__ stage_src, const int *__restrict__ stage_cnt, float *__restrict__ krho,
                                                        const int *__restrict__ wave_dirty, const unsigned char *__restrict__ changed8, int force_all, TilePhase tp,
                                                        SpecUndo un = SpecUndo{nullptr, nullptr, nullptr, nullptr, 0}, DensFlow df = kNoFlow)
{
    extern __shared__ float4 s_operand[];
    const bool spread = DENS && wave_dirty && !force_all;           // (round-robin tiles when most of them return at once, see k_correct in sph_kernels.h)
    const int tile = sweep_tile(tp, spread);
    const bool flow = DENS && df.nbr != nullptr;                    // the producer says who must run (DensFlow in sph_kernels.h; k_residual is the commented form)
    const int n2 = (flow && spread && tp.sparse) ? tp.sparse[tp.ntiles + 1] : 0;
    if (gate_closed(ds, gate)) { spec_undo(c, un, ds, tp); return; }
    if (tile < 0) return;
    StagePre pre = kNoPre;
    bool direct = false;
    if (spread) {                                                   // change propagation between the sweeps of the density loop (sph_kernels.h)
        bool idle;
        if (flow) {
            const FlowHead fh = flow_head(tp, df, tile, n2, stage_src, stage_cnt);
            pre = fh.pre; direct = fh.direct;
            idle = !fh.need;
            if (idle && df.nz[tile] != 0 && threadIdx.x < 64) flow_push(df, df.nbr[(size_t)tile * kNbrStride + threadIdx.x]);
            if (idle && direct && threadIdx.x == 0) df.worked[tile] = 0;
        } else {
            const int sw = stage_cnt[tile];
            idle = sw >= 0 && !stage_sources_flagged(stage_src, sw, tile, wave_dirty);
        }
        if (tp.hot && threadIdx.x == 0 && idle) tp.hot[tile] = 0;
        if (idle) return;
    }
    const int my_nbr = (flow && threadIdx.x < 64) ? df.nbr[(size_t)tile * kNbrStride + threadIdx.x] : 0;
    const uint32_t *nlb = nullptr;
    SPH_SWEEP_PROLOGUE_B(false, tile)
    (void)nlbp;
    float2 *s_v2 = reinterpret_cast<float2 *>(s_operand + c.stage_cap);
    bool staged;
    if (spread && !direct) {       // second, exact level of the change propagation (sph_kernels.h: stage_operand_pv_checked)
        const int verdict = stage_operand_pv_checked<false>(c, s_operand, s_v2, P, V, changed8, stage_src, stage_cnt, blk, pre);
        if (verdict == 2) {
            if (flow && df.nz[blk] != 0 && threadIdx.x < 64) flow_push(df, my_nbr);
            if (tp.hot && threadIdx.x == 0) tp.hot[blk] = 1;
            return;
        }
        staged = verdict == 1;
    } else {
        int would = 1;                                              // (see k_residual: a `direct` tile learns from the staging batch whether it would have passed the check)
        staged = stage_operand_pv<false>(c, s_operand, s_v2, P, V, stage_src, stage_cnt, blk, pre, (spread && direct) ? changed8 : nullptr, &would);
        if (spread && flow && threadIdx.x == 0) df.worked[blk] = would ? 1 : 0;
        if (spread && tp.hot && threadIdx.x == 0) tp.hot[blk] = would ? 2 : 1;
    }
    if (spread && !direct) {
        if (tp.hot && threadIdx.x == 0) tp.hot[blk] = 2;
        if (flow && threadIdx.x == 0) df.worked[blk] = 1;
    }
    const float4 vi = V[ii];
    float acc = 0.f;
    const bool skip = !DENS && kf < 20;                                           // :258-261
    const int kfx = skip ? 0 : kf;
    auto pair = [&](const float4 pj, const float4 vj, const uint32_t) {
        const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        const float g = rx_g(c, dx, dy, dz);
        const float dot = __builtin_fmaf(vi.z - vj.z, dz, __builtin_fmaf(vi.y - vj.y, dy, (vi.x - vj.x) * dx));
        acc = __builtin_fmaf(g, dot, acc);                                        // :287 / :162
    };
    if (staged)
        rx_walk8(nlp, kfx, [&](const Nl16Group &g) {
            float4 a[8]; float2 b[8];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[2 * u] = s_operand[g.lo(u)]; a[2 * u + 1] = s_operand[g.hi(u)]; b[2 * u] = s_v2[g.lo(u)]; b[2 * u + 1] = s_v2[g.hi(u)]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) pair(a[u], make_float4(a[u].w, b[u].x, b[u].y, 0.f), 0u);
        });
    else for_fluid_nbrs<false, true>(nlp, kfx, P, V, RigidView(), pair);        // a workgroup whose set did not fit: 32-bit global indices, masked tails
    float val = 0.f, kr = 0.f;
    int flag = 0;
    if (live) {
        float sum = acc;
        if (c.boundary_handle && kb > 0 && !skip) {                               // :300 / :176
            const float4 gw = G[i];
            sum = __builtin_fmaf(__builtin_fmaf(vi.z, gw.z, __builtin_fmaf(vi.y, gw.y, vi.x * gw.x)), c.rx_rho0_m, acc);
        }
        const float rho_i = rho[i];
        if (DENS) {
            val = rmax(__builtin_fmaf(ds->dt, sum, rho_i), c.rho0);               // :135 / :137
            flag = !(val == c.rho0) && !ghost;                                    // :139
            kr = ((val - c.rho0) * alpha[i] / ds->dt2) / rho_i;                   // :199,203
        } else {
            val = skip ? 0.f : rmax(sum, 0.0f);                                   // :267 / :269
            flag = val > 0.f && !ghost;                                           // :275
            kr = (val * alpha[i] / ds->dt) / rho_i;                               // :363,367
        }
        if (!(ghost && c.ghost_walk)) {       // (two-column slab handles: a ghost's value and k / rho arrive with the halo)
            out[i] = val;
            krho[i] = kr;
        }
    }
    block_partial_mean(blk, (double)val, flag, psum, pcnt);
    if (flow) {
        const int nzf = __syncthreads_or((live && !ghost && kr != 0.f) ? 1 : 0);
        if (threadIdx.x == 0) df.nz[blk] = nzf ? 1 : 0;
        if (nzf && threadIdx.x < 64) flow_push(df, my_nbr);
    }
}

// D2 / D4 / D7 (k_correct)                                       dfsph_solver.py:314-355, 302-312 + 357-391, 178-219
// wall part: sum_B (V_b k_i / rho_i) grad W_ib = (k_i / rho_i) G_i / m
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_correct_rx(Consts c, const float4 *__restrict__ P, const float4 *__restrict__ G,
                                                       const uint32_t *__restrict__ nl, const int *__restrict__ cnt, const float *__restrict__ rho,
                                                       const float *__restrict__ alpha, const float *__restrict__ src,
                                                       float *__restrict__ warm, const DevScalars *ds,       // (no __restrict__: see k_correct)
                                                       const float4 *Vin, float4 *Vout, int gate,
                                                       const uint2 *__restrict__ stage_src, const int *__restrict__ stage_cnt, const float *__restrict__ krho,
                                                       int *__restrict__ wave_dirty, unsigned char *__restrict__ changed8, TilePhase tp,
                                                       SpecSave sv = SpecSave{nullptr, nullptr}, FinRide fr = kNoRide, DensFlow df = kNoFlow)
{
    extern __shared__ float4 s_operand[];
    if (fr.mode >= 0 && blockIdx.x == 0) { fin_ride_block(fr); return; }          // (see k_correct: the loop decision rides in this launch)
    const uint32_t *nlb = nullptr;
    const int tile = sweep_tile(tp, MODE == CORR_DENS && wave_dirty != nullptr);
    const bool flow = MODE == CORR_DENS && wave_dirty != nullptr && df.nbr != nullptr;      // (DensFlow, see k_correct in sph_kernels.h)
    const int n2 = (flow && tp.sparse) ? tp.sparse[tp.ntiles + 1] : 0;
    if (gate_closed(ds, gate)) return;
    if (tile < 0) return;
    StagePre pre = kNoPre;
    bool direct = false;
    if (flow) {
        const FlowHead fh = flow_head(tp, df, tile, n2, stage_src, stage_cnt);
        pre = fh.pre; direct = fh.direct;
        if (!fh.need) {
            const int i0 = tile * kBlock + (int)threadIdx.x;
            if ((threadIdx.x & 63) == 0) wave_dirty[tile * (kBlock / 64) + (threadIdx.x >> 6)] = 0;
            if (i0 < c.n) changed8[i0] = 0;
            if (direct && threadIdx.x == 0) df.worked[tile] = 0;
            return;
        }
    }
    const int my_nbr = (flow && threadIdx.x < 64) ? df.nbr[(size_t)tile * kNbrStride + threadIdx.x] : 0;
    SPH_SWEEP_PROLOGUE_G(false, tile, true)
    (void)nlbp;
    const bool track = MODE == CORR_DENS && wave_dirty != nullptr;  // change propagation in the density loop (sph_kernels.h: stage_sources_flagged)
    bool staged;
    if (track && !direct) {
        const int verdict = stage_operand_ps_checked<false>(c, s_operand, P, krho, stage_src, stage_cnt, blk, pre);
        if (verdict == 2) {
            const bool foreign = live && ghost && !c.ghost_walk;           // (see k_correct in sph_kernels.h)
            const unsigned long long anyg = __ballot(foreign);
            if ((threadIdx.x & 63) == 0) wave_dirty[blk * (kBlock / 64) + (threadIdx.x >> 6)] = anyg != 0ull ? 1 : 0;
            if (live) changed8[i] = foreign ? 1 : 0;
            return;
        }
        staged = verdict == 1;
    } else {
        staged = stage_operand_ps(c, s_operand, P, krho, stage_src, stage_cnt, blk, pre);
    }
    const float dt = ds->dt;
    const float rho_i = rho[ii];
    float k_i;
    if (MODE == CORR_WARM) k_i = warm[ii] / dt;                                   // :333
    else if (MODE == CORR_DIV) k_i = src[ii] * alpha[ii] / dt;                    // :363
    else k_i = (src[ii] - c.rho0) * alpha[ii] / ds->dt2;                          // :199
    const float kr_i = k_i / rho_i;
    float ax = 0.f, ay = 0.f, az = 0.f;
    auto pair = [&](const float4 pj, const float4, const uint32_t) {
        const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        const float g = rx_g(c, dx, dy, dz);
        float ks = kr_i + pj.w;
        if (MODE == CORR_DIV) ks = ks > 1e-5f ? ks : 0.f;                         // :367
        const float s = ks * g;                                                   // :337 / :369 / :203 (m inside g)
        ax = __builtin_fmaf(s, dx, ax); ay = __builtin_fmaf(s, dy, ay); az = __builtin_fmaf(s, dz, az);
    };
    struct OperandPS { float4 a; float s; };
    const float4 none = make_float4(0.f, 0.f, 0.f, 0.f);
    if (staged)
        rx_walk8(nlp, kf, [&](const Nl16Group &g) {
            float4 a[8];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[2 * u] = s_operand[g.lo(u)]; a[2 * u + 1] = s_operand[g.hi(u)]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) pair(a[u], none, 0u);
        });
    else        // a workgroup whose set did not fit: two global gathers per neighbour, masked tails
        walk_list<OperandPS>(nlp, kf, [&](uint32_t j, OperandPS &o) { o.a = P[j]; o.s = krho[j]; },
                             [&](const OperandPS &o, uint32_t j) { pair(make_float4(o.a.x, o.a.y, o.a.z, o.s), none, j); });
    float gx = 0.f, gy = 0.f, gz = 0.f;
    if (c.boundary_handle && kb > 0 && live) {                                    // :322 / :310 / :187,191
        const float4 gw = G[i];
        const float kb_i = kr_i * c.rx_rho0_m;
        gx = gw.x * kb_i; gy = gw.y * kb_i; gz = gw.z * kb_i;
    }
    if (track) {
        const bool changed = live && ((ghost && !c.ghost_walk) || ax != 0.f || ay != 0.f || az != 0.f || gx != 0.f || gy != 0.f || gz != 0.f);
        const unsigned long long any = __ballot(changed);
        if ((threadIdx.x & 63) == 0) wave_dirty[blk * (kBlock / 64) + (threadIdx.x >> 6)] = any != 0ull ? 1 : 0;
        if (live) changed8[i] = changed ? 1 : 0;
        if (flow) {
            const int moved = __syncthreads_or(changed ? 1 : 0);
            if (moved && threadIdx.x < 64) flow_push(df, my_nbr);
            if (threadIdx.x == 0) df.worked[tile] = moved ? 1 : 0;
        }
    }
    if (!live) return;
    float4 v = Vin[i];
    if (MODE == CORR_DIV && sv.v) { sv.v[i] = v; sv.w[i] = warm[i]; }             // (see k_correct: the undo of a correction that ran ahead of the loop decision)
    v.x -= (ax + gx) * dt; v.y -= (ay + gy) * dt; v.z -= (az + gz) * dt;         // :324 / :312 / :189
    v.w = rho_i;
    Vout[i] = v;
    if (MODE == CORR_WARM) warm[i] = 0.0f;                                        // :325
    if (MODE == CORR_DIV && c.warm_start) warm[i] += src[i] * alpha[i];           // :384 (:404-405)
}


// D1 (k_density<DFSPH>): rho, alpha, k / rho of the warm start         solver_base.py:41-72, dfsph_solver.py:32-89, :333
// The density needs the tail mask the gradient sweeps do without: the particle's own padding entry has W(0) = kw, not 0.
__global__ __launch_bounds__(kBlock) void k_density_rx(Consts c, const float4 *__restrict__ P, const float4 *V, const float4 *__restrict__ G,
                                                       const float *__restrict__ Gsq, const uint32_t *__restrict__ nl, const int *__restrict__ cnt,
                                                       const float *__restrict__ warm, const DevScalars *__restrict__ ds, float *__restrict__ rho_out,
                                                       float *__restrict__ aux_out, float4 *Vout, const uint2 *__restrict__ stage_src,
                                                       const int *__restrict__ stage_cnt, float *__restrict__ krho, TilePhase tp,
                                                       const int *__restrict__ id, float *__restrict__ rho_orig)
{
    extern __shared__ float4 s_operand[];
    const uint32_t *nlb = nullptr;
    const int tile = tp.phase == 0 ? xcd_block(blockIdx.x, gridDim.x) : sweep_tile(tp, false);
    if (tile < 0) return;
    SPH_SWEEP_PROLOGUE_G(false, tile, true)
    (void)nlbp;
    const bool staged = stage_operand<false>(c, s_operand, P, stage_src, stage_cnt, blk);
    float ws = 0.f, sx = 0.f, sy = 0.f, sz = 0.f, sq = 0.f;
    auto pair = [&](const float4 pj, bool valid) {
        const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        const RxWG k = rx_wg(c, dx, dy, dz);
        ws += valid ? k.w : 0.f;                                                  // solver_base.py:62
        const float rx = k.g * dx, ry = k.g * dy, rz = k.g * dz;                  // dfsph_solver.py:58,70 (m inside g)
        sx += rx; sy += ry; sz += rz;
        sq = __builtin_fmaf(rz, rz, __builtin_fmaf(ry, ry, __builtin_fmaf(rx, rx, sq)));     // :71
    };
    if (staged) {
        if (kf > 0) {
            uint4 jn = nl_load(nlp);
            for (int kk = 0; kk < kf; kk += 8) {
                const Nl16Group g = {{jn.x, jn.y, jn.z, jn.w}};
                if (kk + 8 < kf) jn = nl_load(nlp + (size_t)((kk >> 3) + 1) * 256);
                float4 a[8];
#pragma unroll
                for (int u = 0; u < 4; ++u) { a[2 * u] = s_operand[g.lo(u)]; a[2 * u + 1] = s_operand[g.hi(u)]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) pair(a[u], kk + u < kf);
            }
        }
    } else {
        for_fluid_nbrs<false, false>(nlp, kf, P, nullptr, RigidView(), [&](const float4 pj, const float4, const uint32_t) { pair(pj, true); });
    }
    if (!live) return;
    float rho_i = __builtin_fmaf(c.kw * c.m, ws, 0.001f);                         // rho starts at 0.001, solver_base.py:44
    float den = (__builtin_fmaf(sz, sz, __builtin_fmaf(sy, sy, sx * sx))) + sq;   // dfsph_solver.py:47
    if (c.boundary_handle) {
        const float4 gw = kb > 0 ? G[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float gs = kb > 0 ? Gsq[i] : 0.f;
        const float f = c.rx_rho0_m;                                              // V_b rho0 grad W = (rho0 / m) V_b m grad W
        const float bx = gw.x * f, by = gw.y * f, bz = gw.z * f;
        rho_i = __builtin_fmaf(gw.w, c.rho0, rho_i);                              // solver_base.py:49
        den = (den + gs * (f * f)) + __builtin_fmaf(bz, bz, __builtin_fmaf(by, by, bx * bx));    // :45
    }
    rho_out[i] = rho_i;
    if (rho_orig) { const int raw = id[i]; rho_orig[raw < 0 ? ~raw : raw] = rho_i; }                                        // (a coupled body's viscosity reads rho by ORIGINAL id, solver_base.py:198-199)
    const float alpha = fabsf(den) < 1e-6f ? 0.0f : rho_i / den;                  // :48-51
    aux_out[i] = alpha;
    const float4 vi = V[i];
    krho[i] = (warm[i] / ds->dt) / rho_i;                                         // :333
    Vout[i] = make_float4(vi.x, vi.y, vi.z, rho_i);                               // velocity buffers carry rho in .w (read by D5)
}

// D5 (k_dfsph_ext): tension + viscosity + gravity, v*, max |v*|        solver_base.py:170-217, dfsph_solver.py:91-103.   V = (vel, rho)
// The particle's own padding entry contributes x_ij = 0 to the tension and shear = 0 to the viscosity: no tail mask.
__global__ __launch_bounds__(kBlock) void k_dfsph_ext_rx(Consts c, const float4 *__restrict__ P, const float4 *__restrict__ V,
                                                         const uint32_t *__restrict__ nl, const int *__restrict__ cnt,
                                                         const DevScalars *__restrict__ ds, float4 *__restrict__ VAout, float *__restrict__ pmax,
                                                         const uint2 *__restrict__ stage_src, const int *__restrict__ stage_cnt, TilePhase tp)
{
    extern __shared__ float4 s_operand[];
    const uint32_t *nlb = nullptr;
    const int tile = tp.phase == 0 ? xcd_block(blockIdx.x, gridDim.x) : sweep_tile(tp, false);
    if (tile < 0) return;
    SPH_SWEEP_PROLOGUE_B(false, tile)
    (void)kb; (void)nlbp;
    uint32_t *s_src = reinterpret_cast<uint32_t *>(s_operand + c.stage_cap);      // (vel, rho) is gathered from memory through the source list
    const bool staged = stage_operand_src(c, s_operand, s_src, P, stage_src, stage_cnt, blk);
    const float4 vi = V[ii];
    const float rho_i = vi.w;
    const float tk = c.tens_c * c.kw;
    float wx = 0.f, wy = 0.f, wz = 0.f, tx = 0.f, ty = 0.f, tz = 0.f;
    auto pair = [&](const float4 pj, const float4 vj) {
        const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        const RxWG k = rx_wg(c, dx, dy, dz);
        const float st = tk * k.w;                                                // solver_base.py:216
        tx = __builtin_fmaf(st, dx, tx); ty = __builtin_fmaf(st, dy, ty); tz = __builtin_fmaf(st, dz, tz);
        const float shear = __builtin_fmaf(vi.z - vj.z, dz, __builtin_fmaf(vi.y - vj.y, dy, (vi.x - vj.x) * dx));      // :183
        const float nu = c.visc_num * __builtin_amdgcn_rcpf(rho_i + vj.w);        // :187
        const float mp = (nu * shear) * __builtin_amdgcn_rcpf(k.r2 + c.visc_eps_h2);   // -pi_ij, :188
        const float sv = shear < 0.f ? mp * k.g : 0.f;                            // :184, :189 (m inside g)
        wx = __builtin_fmaf(sv, dx, wx); wy = __builtin_fmaf(sv, dy, wy); wz = __builtin_fmaf(sv, dz, wz);
    };
    if (staged) {
        if (kf > 0) {
            uint4 jn = nl_load(nlp);
            for (int kk = 0; kk < kf; kk += 8) {
                const Nl16Group g = {{jn.x, jn.y, jn.z, jn.w}};
                if (kk + 8 < kf) jn = nl_load(nlp + (size_t)((kk >> 3) + 1) * 256);
                float4 a[8], b[8];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a[2 * u] = s_operand[g.lo(u)]; a[2 * u + 1] = s_operand[g.hi(u)];
                    b[2 * u] = V[s_src[g.lo(u)]]; b[2 * u + 1] = V[s_src[g.hi(u)]];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) pair(a[u], b[u]);
            }
        }
    } else {
        for_fluid_nbrs<false, true>(nlp, kf, P, V, RigidView(), [&](const float4 pj, const float4 vj, const uint32_t) { pair(pj, vj); });
    }
    float vn = -INFINITY;
    if (live) {
        const float dt = ds->dt;
        const float ten[3] = {tx * c.m, ty * c.m, tz * c.m};                      // :209
        const float vis[3] = {wx * c.m, wy * c.m, wz * c.m};                      // :175
        const float g[3] = {c.gravity * 0.0f, c.gravity * -1.0f, c.gravity * 0.0f};
        const float v[3] = {vi.x, vi.y, vi.z};
        float va[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float f = (g[a] + ten[a]) + vis[a];                             // dfsph_solver.py:96
            va[a] = v[a] + dt * f / c.m;                                          // :102
        }
        VAout[i] = make_float4(va[0], va[1], va[2], rho_i);
        if (!ghost) vn = norm3(va[0], va[1], va[2]);                              // :103
    }
    block_partial_max(blk, vn, pmax);
}


// ======================================================================================================================================
// WCSPH under the relaxed arithmetic (wcsph_solver.py:25-129, solver_base.py:41-72, 170-217) -- single-GPU handles
// without a rigid body.  Two kernels per step, W and m grad W of a pair from one v_rsq_f32 (rx_wg), FMAs, and
//   * VERLET LISTS: the lists hold every pair within h + skin (cells of edge h + skin, Consts.hcell) and are rebuilt -- sort included -- only
//     when a particle has moved more than skin / 2 since the last build.  The kernel functions are clamped at q = 1, where W and grad W
//     vanish continuously, so a listed pair beyond h contributes exactly 0 and a pair that comes within h between two builds is already listed
//     (both moved < skin / 2).  The integrator raises DevScalars.moved; the next step's sort and list-build kernels, enqueued every step, read
//     it at their first instruction and leave unless it is set (the density kernel, which runs after them, takes it down again): the step stays
//     a fixed launch sequence (hipGraph replay) without any host decision.  At 250 k particles the list build was 44 % of the exact step.
//   * the wall sums (sum_b V_b W, sum_b V_b m grad W) are taken once per step, by the density kernel, and handed to the force kernel.
// Summation order between two builds is the order of the last build, not the canonical one: a different legal execution of the reference,
// whose own envelope for wcsph is 4e-8 after 200 steps; tests/test_relaxed_gpu.py holds this path to 1e-5 of the oracle.
// ======================================================================================================================================
__device__ __forceinline__ RxWG rx_wg_clamped(const Consts &c, float dx, float dy, float dz)
{
    RxWG o;
    o.r2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, __builtin_fmaf(dx, dx, 1e-30f)));
    const float ri = __builtin_amdgcn_rsqf(o.r2);
    const float q = (o.r2 * ri) * c.rh;
    const float t = rmax(1.0f - q, 0.0f), t2 = t * t;                        // q > 1 (a skin entry): W = grad W = 0
    const float w1 = __builtin_fmaf(6.0f * (q * q), q - 1.0f, 1.0f);          // solver_base.py:76-88
    const float w2 = 2.0f * (t2 * t);
    const float g1 = __builtin_fmaf(q, c.rx_k1a, c.rx_k1b);
    const float g2 = (t2 * ri) * c.rx_k2;
    const bool inner = q <= 0.5f;
    o.w = inner ? w1 : w2;
    o.g = inner ? g1 : g2;
    return o;
}


// W1: rho, p, p / rho^2 and the wall sums of the step          solver_base.py:41-72, wcsph_solver.py:66-90
//   writes Pout = (pos, rho), Vout = (vel, p / rho^2), rho[], pressure[], G = sum_b V_b m grad W_ib
__global__ __launch_bounds__(kBlock) void k_wcsph_density_rx(Consts c, const float4 *__restrict__ P, const float4 *__restrict__ V,
                                                             const float4 *__restrict__ WP, const uint32_t *__restrict__ nl,
                                                             const uint32_t *__restrict__ nlb, const int *__restrict__ cnt,
                                                             float *__restrict__ rho_out, float *__restrict__ p_out, float4 *__restrict__ Pout,
                                                             float4 *__restrict__ Vout, float4 *__restrict__ G, DevScalars *__restrict__ ds, int count_build)
{
    // every gated kernel of this step has read DevScalars.moved by now (stream order): take it down for the integrator of this step
    if (blockIdx.x == 0 && threadIdx.x == 0 && ds->moved != 0) { ds->moved = 0; if (count_build) ds->verlet_builds += 1; }
    SPH_SWEEP_PROLOGUE_M(false)
    float ws = 0.f;
    for_fluid_nbrs<false, false>(nlp, kf, P, nullptr, RigidView(), [&](const float4 pj, const float4, const uint32_t) {
        const RxWG k = rx_wg_clamped(c, pi.x - pj.x, pi.y - pj.y, pi.z - pj.z);
        ws += k.w;                                                             // solver_base.py:62 (m kw outside the sum)
    });
    float wb = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
    if (c.boundary_handle)
        for_nbrs_p(nlbp, kb, WP, [&](const float4 pj) {
            const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
            const RxWG k = rx_wg_clamped(c, dx, dy, dz);
            wb = __builtin_fmaf(pj.w, k.w, wb);                                // :70-71
            const float s = pj.w * k.g;
            gx = __builtin_fmaf(s, dx, gx); gy = __builtin_fmaf(s, dy, gy); gz = __builtin_fmaf(s, dz, gz);
        });
    if (!live) return;
    float rho_i = __builtin_fmaf(c.kw * c.m, ws, 0.001f);                     // rho starts at 0.001, :44
    if (c.boundary_handle) rho_i = __builtin_fmaf(wb * c.kw, c.rho0, rho_i);  // :49
    const float p = tait_pressure(rho_i);                                     // wcsph_solver.py:86-90
    rho_out[i] = rho_i;
    p_out[i] = p;
    const float4 vi = V[i];
    Pout[i] = make_float4(pi.x, pi.y, pi.z, rho_i);
    Vout[i] = make_float4(vi.x, vi.y, vi.z, p * __builtin_amdgcn_rcpf(rho_i * rho_i));    // :109,116
    G[i] = make_float4(gx, gy, gz, 0.f);
}

// W2: pressure gradient, wall pressure, viscosity, tension, symplectic Euler      wcsph_solver.py:40-129, solver_base.py:170-217
//   reads P = (pos, rho), V = (vel, p / rho^2), G; writes the next state and raises DevScalars.moved when a particle is more than skin / 2
//   away from X0, its position at the last list build
__global__ __launch_bounds__(kBlock) void k_wcsph_force_rx(Consts c, float dt, const float4 *__restrict__ P, const float4 *__restrict__ V,
                                                           const uint32_t *__restrict__ nl, const int *__restrict__ cnt,
                                                           const float4 *__restrict__ G, const float4 *__restrict__ X0,
                                                           float4 *__restrict__ Pn, float4 *__restrict__ Vn, float4 *__restrict__ acc_out,
                                                           DevScalars *__restrict__ ds)
{
    const uint32_t *nlb = nullptr;
    SPH_SWEEP_PROLOGUE_M(false)
    (void)kb; (void)nlbp;
    const float4 vi = V[ii];
    const float rho_i = pi.w, a_i = vi.w;
    const float tk = c.tens_c * c.kw;
    float px = 0.f, py = 0.f, pz = 0.f, wx = 0.f, wy = 0.f, wz = 0.f, tx = 0.f, ty = 0.f, tz = 0.f;
    for_fluid_nbrs<false, true>(nlp, kf, P, V, RigidView(), [&](const float4 pj, const float4 vj, const uint32_t) {
        const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        const RxWG k = rx_wg_clamped(c, dx, dy, dz);
        const float s = (a_i + vj.w) * k.g;                                    // wcsph_solver.py:116 (m inside g)
        px = __builtin_fmaf(s, dx, px); py = __builtin_fmaf(s, dy, py); pz = __builtin_fmaf(s, dz, pz);
        const float shear = __builtin_fmaf(vi.z - vj.z, dz, __builtin_fmaf(vi.y - vj.y, dy, (vi.x - vj.x) * dx));   // solver_base.py:183
        const float nu = c.visc_num * __builtin_amdgcn_rcpf(rho_i + pj.w);     // :187
        const float mp = (nu * shear) * __builtin_amdgcn_rcpf(k.r2 + c.visc_eps_h2);   // -pi_ij, :188
        const float sv = shear < 0.f ? mp * k.g : 0.f;                         // :184, :189
        wx = __builtin_fmaf(sv, dx, wx); wy = __builtin_fmaf(sv, dy, wy); wz = __builtin_fmaf(sv, dz, wz);
        const float st = tk * k.w;                                             // :216
        tx = __builtin_fmaf(st, dx, tx); ty = __builtin_fmaf(st, dy, ty); tz = __builtin_fmaf(st, dz, tz);
    });
    bool far = false;
    if (live) {
        float acc[3] = {c.gravity * 0.0f, c.gravity * -1.0f, c.gravity * 0.0f};       // solver_base.py:131-133
        const float pg[3] = {-px, -py, -pz};
        const float vis[3] = {wx * c.m, wy * c.m, wz * c.m};                   // :175
        const float ten[3] = {tx * c.m, ty * c.m, tz * c.m};                   // :209
        float bac[3] = {0.f, 0.f, 0.f};
        if (c.boundary_handle) {
            const float4 gw = G[i];
            const float f = -(a_i * c.rx_rho0_m);                              // wcsph_solver.py:83,99: -rho0 (p_i / rho_i^2) sum_b V_b grad W
            bac[0] = f * gw.x; bac[1] = f * gw.y; bac[2] = f * gw.z;
        }
        float pos[3] = {pi.x, pi.y, pi.z}, vel[3] = {vi.x, vi.y, vi.z};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            acc[a] += ((pg[a] + vis[a]) + ten[a]) + bac[a];                    // wcsph_solver.py:44-47
            vel[a] = __builtin_fmaf(acc[a], dt, vel[a]);                       // :50
            vel[a] *= 0.9998f;                                                 // :51
            pos[a] = __builtin_fmaf(vel[a], dt, pos[a]);                       // :52
        }
        if (!c.boundary_handle) {                                              // :54-63
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                if (pos[a] <= c.clamp_lo[a]) { pos[a] = c.clamp_lo[a]; vel[a] *= -0.5f; }
                if (pos[a] >= c.clamp_hi[a]) { pos[a] = c.clamp_hi[a]; vel[a] *= -0.5f; }
            }
        }
        Pn[i] = make_float4(pos[0], pos[1], pos[2], 0.f);
        Vn[i] = make_float4(vel[0], vel[1], vel[2], 0.f);
        acc_out[i] = make_float4(acc[0], acc[1], acc[2], 0.f);
        const float4 x0 = X0[i];
        const float ex = pos[0] - x0.x, ey = pos[1] - x0.y, ez = pos[2] - x0.z;
        far = !(__builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex)) <= c.verlet_thr2);       // (a NaN position counts as moved)
    }
    if (__ballot(far) != 0ull && (threadIdx.x & 63) == 0) ds->moved = 1;       // rare, plain store: the lists are rebuilt before the next step's sweeps
}

}  // namespace sph
