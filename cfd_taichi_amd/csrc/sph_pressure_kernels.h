// sph_pressure_kernels.h -- PCISPH (pcisph_solver.py) and IISPH (iisph_solver.py) sweeps on the same cell-sorted
// arrays, wave-tiled neighbour lists and sweep skeleton as the WCSPH / DFSPH kernels (sph_kernels.h).
//
// Both solvers iterate a pressure field with the positions frozen, so the lists built once per step serve every
// iteration.  The reference drives the iteration from Python (pcisph_solver.py:49-71, iisph_solver.py:85-108); here
// k_finalize_pressure evaluates the loop condition on the device and later sweeps of a finished loop exit at
// their first instruction (same control scheme as DFSPH).
//
// Buffer roles (float4, sorted order):
//   P  = (pos, rho)            written by k_density<false>        static through the step
//   PB[2] = (pos, pressure)    press_iter (PCISPH) / p_iter (IISPH); ping-pong: a sweep reads the pressures of its
//                              neighbours from one and writes its own new pressure to the other
//   PP = (pos_predict, -)      PCISPH            DII = (d_ii, -), DIJ = (d_ij, -)   IISPH
//   EF = (ext_force, -)  PF = (press_force, -)   PCISPH;   VA = (v_adv, -)   IISPH
#pragma once
#include "sph_kernels.h"

namespace sph {

// a / d for a compile-time style constant d with rd = RN(1/d): Markstein's sequence, correctly rounded like `/`
// (same argument as div_by_h; d = 1e6 has a significand that is not all ones)
__device__ __forceinline__ float div_const(float a, float d, float rd)
{
#ifdef SPH_GENERIC_DIV
    return a / d;
#endif
    float q0 = a * rd;
    float e = __builtin_fmaf(-q0, d, a);
    return __builtin_fmaf(e, rd, q0);
}

// fluid-list walkers; with RIGID the list may hold tagged rigid entries (bit 31): their position comes from rv.RP and the
// other operands are undefined.  body(..., j): j & kRigidTag marks a rigid neighbour.
template <bool RIGID, class Body>
__device__ __forceinline__ void for_nbrs_ps(const uint32_t *__restrict__ base, int cnt, const float4 *__restrict__ A,
                                            const float *__restrict__ S, const RigidView &rv, Body body)
{
    struct Op { float4 a; float s; };
    walk_list<Op>(base, cnt, [&](uint32_t j, Op &o) {
        const bool rg = RIGID && (j & kRigidTag);
        const uint32_t idx = RIGID ? (j & ~kRigidTag) : j;
        o.a = rg ? rv.RP[idx] : A[idx];
        o.s = S[rg ? 0u : idx];
    }, [&](const Op &o, uint32_t j) { body(o.a, o.s, j); });
}
template <bool RIGID, class Body>
__device__ __forceinline__ void for_nbrs_3(const uint32_t *__restrict__ base, int cnt, const float4 *__restrict__ A,
                                           const float4 *__restrict__ B, const float4 *__restrict__ C, const RigidView &rv, Body body)
{
    struct Op { float4 a, b, c; };
    walk_list<Op>(base, cnt, [&](uint32_t j, Op &o) {
        const bool rg = RIGID && (j & kRigidTag);
        const uint32_t idx = RIGID ? (j & ~kRigidTag) : j;
        o.a = rg ? rv.RP[idx] : A[idx];
        o.b = B[rg ? 0u : idx];
        o.c = C[rg ? 0u : idx];
    }, [&](const Op &o, uint32_t j) { body(o.a, o.b, o.c, j); });
}

// quad forms (small scenes, four lanes per particle: walk_list_quad in sph_kernels.h)
template <bool RIGID, int N, class Body>
__device__ __forceinline__ void for_nbrs_ps_quad(const uint32_t *__restrict__ base, int cnt, int q, float (&acc)[N], const float4 *__restrict__ A,
                                                 const float *__restrict__ S, const RigidView &rv, Body body)
{
    struct Op { float4 a; float s; };
    walk_list_quad<N, Op>(base, cnt, q, acc, [&](uint32_t j, Op &o) {
        const bool rg = RIGID && (j & kRigidTag);
        const uint32_t idx = RIGID ? (j & ~kRigidTag) : j;
        o.a = rg ? rv.RP[idx] : A[idx];
        o.s = S[rg ? 0u : idx];
    }, [&](const Op &o, uint32_t j) { body(o.a, o.s, j); });
}
template <bool RIGID, int N, class Body>
__device__ __forceinline__ void for_nbrs_3_quad(const uint32_t *__restrict__ base, int cnt, int q, float (&acc)[N], const float4 *__restrict__ A,
                                                const float4 *__restrict__ B, const float4 *__restrict__ C, const RigidView &rv, Body body)
{
    struct Op { float4 a, b, c; };
    walk_list_quad<N, Op>(base, cnt, q, acc, [&](uint32_t j, Op &o) {
        const bool rg = RIGID && (j & kRigidTag);
        const uint32_t idx = RIGID ? (j & ~kRigidTag) : j;
        o.a = rg ? rv.RP[idx] : A[idx];
        o.b = B[rg ? 0u : idx];
        o.c = C[rg ? 0u : idx];
    }, [&](const Op &o, uint32_t j) { body(o.a, o.b, o.c, j); });
}

// staged forms (LDS staging plan of k_build_nl, IISPH on the Morton curve): the first operand comes from LDS, the others are
// gathered from memory through the staged source index
template <bool RIGID, class Body>
__device__ __forceinline__ void for_staged_nbrs_ps(const uint32_t *__restrict__ base, int cnt, const float4 *__restrict__ s_A,
                                                   const uint32_t *__restrict__ s_src, const float *__restrict__ S, const RigidView &rv, Body body)
{
    NlAhead ahead(base);
    for (int kk = 0; kk < cnt; kk += 4) {
        const uint4 jj = ahead.front();
        const uint32_t j[4] = {jj.x, jj.y, jj.z, jj.w};
        float4 a[4]; float sc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool rg = RIGID && (j[u] & kRigidTag);
            const uint32_t idx = RIGID ? (j[u] & ~kRigidTag) : j[u];
            sc[u] = S[rg ? 0u : s_src[idx]];
            a[u] = rg ? rv.RP[idx] : s_A[idx];
        }
        ahead.advance(kk);
        body(a[0], sc[0], j[0]);
        if (kk + 1 < cnt) body(a[1], sc[1], j[1]);
        if (kk + 2 < cnt) body(a[2], sc[2], j[2]);
        if (kk + 3 < cnt) body(a[3], sc[3], j[3]);
    }
}
// first operand and a per-particle scalar both staged (20 B per staged particle)
__device__ __forceinline__ bool stage_operand_scalar(const Consts &c, float4 *__restrict__ s_A, float *__restrict__ s_S, const float4 *__restrict__ A,
                                                     const float *__restrict__ S, const uint2 *__restrict__ stage_runs,
                                                     const int *__restrict__ stage_cnt, int blk)
{
    const int nst = stage_expand(stage_runs, stage_cnt, blk, reinterpret_cast<uint32_t *>(s_A));
    if (nst < 0) return false;
    if (nst == 0) return true;
    const StageIdx x = stage_take(reinterpret_cast<const uint32_t *>(s_A), nst);
#pragma unroll
    for (int t = 0; t < kStageTrips; ++t) {
        const int base = threadIdx.x + t * kStageBatch * kBlock;
        if (t * kStageBatch * kBlock >= nst) break;
#pragma unroll
        for (int u = 0; u < kStageBatch; ++u)
            if (base + u * kBlock < nst) { s_A[base + u * kBlock] = A[x.j[t][u]]; s_S[base + u * kBlock] = S[x.j[t][u]]; }
    }
    __syncthreads();
    return true;
}
// the same, reporting whether any staged A.w is != 0: 0 = not staged, 1 = staged, 2 = staged and every A.w is 0 (k_ii_dij: see there)
__device__ __forceinline__ int stage_operand_scalar_checked(const Consts &c, float4 *__restrict__ s_A, float *__restrict__ s_S, const float4 *__restrict__ A,
                                                            const float *__restrict__ S, const uint2 *__restrict__ stage_runs,
                                                            const int *__restrict__ stage_cnt, int blk)
{
    const int nst = stage_expand(stage_runs, stage_cnt, blk, reinterpret_cast<uint32_t *>(s_A));
    if (nst < 0) return 0;
    if (nst == 0) return 2;
    const StageIdx x = stage_take(reinterpret_cast<const uint32_t *>(s_A), nst);
    int any = 0;
#pragma unroll
    for (int t = 0; t < kStageTrips; ++t) {
        const int base = threadIdx.x + t * kStageBatch * kBlock;
        if (t * kStageBatch * kBlock >= nst) break;
#pragma unroll
        for (int u = 0; u < kStageBatch; ++u)
            if (base + u * kBlock < nst) {
                const float4 a = A[x.j[t][u]];
                any |= a.w != 0.f;
                s_A[base + u * kBlock] = a; s_S[base + u * kBlock] = S[x.j[t][u]];
            }
    }
    return __syncthreads_or(any) ? 1 : 2;
}
template <bool RIGID, class Body>
__device__ __forceinline__ void for_staged_nbrs_ps2(const uint32_t *__restrict__ base, int cnt, const float4 *__restrict__ s_A,
                                                    const float *__restrict__ s_S, const RigidView &rv, Body body)
{
    NlAhead ahead(base);
    for (int kk = 0; kk < cnt; kk += 4) {
        const uint4 jj = ahead.front();
        const uint32_t j[4] = {jj.x, jj.y, jj.z, jj.w};
        float4 a[4]; float sc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool rg = RIGID && (j[u] & kRigidTag);
            const uint32_t idx = RIGID ? (j[u] & ~kRigidTag) : j[u];
            sc[u] = s_S[rg ? 0u : idx];
            a[u] = rg ? rv.RP[idx] : s_A[idx];
        }
        ahead.advance(kk);
        body(a[0], sc[0], j[0]);
        if (kk + 1 < cnt) body(a[1], sc[1], j[1]);
        if (kk + 2 < cnt) body(a[2], sc[2], j[2]);
        if (kk + 3 < cnt) body(a[3], sc[3], j[3]);
    }
}
// update_p: positions + p, the source index and the three components of the third operand staged (32 B per staged particle);
// the second operand is gathered from memory through the source index
template <bool RIGID, class Body>
__device__ __forceinline__ void for_staged_nbrs_3e(const uint32_t *__restrict__ base, int cnt, const float4 *__restrict__ s_A,
                                                   const uint32_t *__restrict__ s_src, const float *__restrict__ s_cx,
                                                   const float *__restrict__ s_cy, const float *__restrict__ s_cz,
                                                   const float4 *__restrict__ B, const RigidView &rv, Body body)
{
    NlAhead ahead(base);
    for (int kk = 0; kk < cnt; kk += 4) {
        const uint4 jj = ahead.front();
        const uint32_t j[4] = {jj.x, jj.y, jj.z, jj.w};
        float4 a[4], b[4], cc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool rg = RIGID && (j[u] & kRigidTag);
            const uint32_t idx = RIGID ? (j[u] & ~kRigidTag) : j[u];
            const uint32_t li = rg ? 0u : idx;
            b[u] = B[s_src[li]];
            cc[u] = make_float4(s_cx[li], s_cy[li], s_cz[li], 0.f);
            a[u] = rg ? rv.RP[idx] : s_A[idx];
        }
        ahead.advance(kk);
        body(a[0], b[0], cc[0], j[0]);
        if (kk + 1 < cnt) body(a[1], b[1], cc[1], j[1]);
        if (kk + 2 < cnt) body(a[2], b[2], cc[2], j[2]);
        if (kk + 3 < cnt) body(a[3], b[3], cc[3], j[3]);
    }
}
template <bool RIGID, class Body>
__device__ __forceinline__ void for_staged_nbrs_3(const uint32_t *__restrict__ base, int cnt, const float4 *__restrict__ s_A,
                                                  const uint32_t *__restrict__ s_src, const float4 *__restrict__ B, const float4 *__restrict__ C,
                                                  const RigidView &rv, Body body)
{
    NlAhead ahead(base);
    for (int kk = 0; kk < cnt; kk += 4) {
        const uint4 jj = ahead.front();
        const uint32_t j[4] = {jj.x, jj.y, jj.z, jj.w};
        float4 a[4], b[4], cc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool rg = RIGID && (j[u] & kRigidTag);
            const uint32_t idx = RIGID ? (j[u] & ~kRigidTag) : j[u];
            const uint32_t src = rg ? 0u : s_src[idx];
            b[u] = B[src];
            cc[u] = C[src];
            a[u] = rg ? rv.RP[idx] : s_A[idx];
        }
        ahead.advance(kk);
        body(a[0], b[0], cc[0], j[0]);
        if (kk + 1 < cnt) body(a[1], b[1], cc[1], j[1]);
        if (kk + 2 < cnt) body(a[2], b[2], cc[2], j[2]);
        if (kk + 3 < cnt) body(a[3], b[3], cc[3], j[3]);
    }
}

// predicted / integrated positions against the clamp walls      pcisph_solver.py:79-89, 234-244; iisph_solver.py:198-207
__device__ __forceinline__ void clamp_walls(const Consts &c, float pos[3], float vel[3])
{
    if (c.boundary_handle) return;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (pos[a] <= c.clamp_lo[a]) { pos[a] = c.clamp_lo[a]; vel[a] *= -0.5f; }
        if (pos[a] >= c.clamp_hi[a]) { pos[a] = c.clamp_hi[a]; vel[a] *= -0.5f; }
    }
}

// ---- loop control ---------------------------------------------------------------------------------------------
// mean of the block partials + the reference's while-condition, evaluated on the device (f64 compares like the Python
// host code).  PCI_FIRST: pcisph_solver.py:56 (the evaluation before the loop); PCI_LOOP: :58-70; II_LOOP:
// iisph_solver.py:89-100.  Iteration counter, residual and the "open" flag live in DevScalars (dens_* fields).
enum { PFIN_PCI_FIRST = 0, PFIN_PCI_LOOP = 1, PFIN_II_LOOP = 2 };

// phase / red: as k_finalize_mean (FINP_ALL on one GPU; FINP_REDUCE -> all-reduce over the slabs -> FINP_DECIDE when sharded)
// group / nparts: as k_finalize_mean (quad sweeps write one partial per 64 particles; a block's partial is the in-order sum of its four)
__global__ __launch_bounds__(kFinBlock) void k_finalize_pressure(const double *__restrict__ psum, const int *__restrict__ pcnt, int nblocks,
                                                                 DevScalars *__restrict__ ds, int mode, int phase, double *__restrict__ red,
                                                                 int group = 1, int nparts = 0, int gather_n = 0)
{
    if (mode != PFIN_PCI_FIRST && ds->dens_active == 0) return;
    __shared__ double s_sum[kFinBlock / 64];
    __shared__ long long s_cnt[kFinBlock / 64];
    if (phase != FINP_DECIDE) fin_reduce(psum, pcnt, nblocks, group, nparts, s_sum, s_cnt);      // the reduction of k_finalize_mean (one batch of loads, one barrier)
    if (threadIdx.x != 0) return;
    if (phase == FINP_REDUCE) { red[0] = s_sum[0]; red[1] = (double)s_cnt[0]; red[2] = (double)ds->overflow; return; }      // (third word: the slab's overflow flags, as in k_finalize_mean)
    if (phase == FINP_DECIDE) {           // gather_n > 0: every slab's (sum, count, flags), four doubles apart, gathered with the last ghost refresh (finalize_mean_block)
        double rs = red[0], rc = red[1], rf = red[2];
        for (int r = 1; r < gather_n; ++r) { rs += red[4 * r]; rc += red[4 * r + 1]; rf += red[4 * r + 2]; }
        s_sum[0] = rs; s_cnt[0] = (long long)rc;
        ds->overflow_any = rf != 0.0 ? 1 : 0;
    }
    ds->sum = s_sum[0]; ds->cnt = s_cnt[0];
    const float res = s_cnt[0] > 0 ? (float)(s_sum[0] / (double)s_cnt[0]) : 0.0f;   // pcisph :136-137, iisph :119-120
    const int cap = ds->dens_cap;
    if (mode == PFIN_PCI_FIRST) {
        ds->dens_avg = res; ds->dens_it = 0; ds->dens_capped = 0; ds->res_diverged = 0;
        ds->dens_active = cap > 0 ? 1 : 0;                                          // iter_cnt (0) < min_iteration (1)
    } else if (mode == PFIN_PCI_LOOP) {
        const int it = ds->dens_it + 1;                                             // :70
        ds->dens_it = it; ds->dens_avg = res;
        const int active = (((double)res > 1000 * 0.1 * 0.01 || it < 1) && it < cap) ? 1 : 0;   // :58
        if (it >= cap) ds->dens_capped = 1;
        ds->dens_active = active;
    } else {
        const int l = ds->dens_it + 1;                                              // iisph :94
        ds->dens_it = l; ds->dens_avg = res;
        int active;
        if (ds->res_have_prev && (double)res - (double)ds->res_prev > 0) {          // :97-99 "Iteration trend to divergence"
            ds->res_diverged = 1;
            active = 0;
        } else {
            ds->res_prev = res; ds->res_have_prev = 1;                              // :100
            active = (((double)res > 0.1 * 1000 * 0.01 || l < 1) && l < cap) ? 1 : 0;   // :88-89
        }
        if (l >= cap) ds->dens_capped = 1;
        ds->dens_active = active;
    }
}

__global__ void k_pressure_ctrl_begin(DevScalars *__restrict__ ds, int cap)
{
    ds->overflow_any = 0;
    ds->dens_active = 1; ds->dens_it = 0; ds->dens_cap = cap; ds->dens_capped = 0; ds->dens_avg = 0.f;
    ds->res_prev = 0.f; ds->res_have_prev = 0; ds->res_diverged = 0;
}

// ======================================================================================
// PCISPH
// ======================================================================================
// compute_ext_force (:237-244: tension, viscosity, gravity) + reset() (:247-250) + the first predict_vel_pos (:73-89)
//   reads P = (pos, rho), V = (vel, -)
template <bool RIGID, int SWEEP, bool RX = false>
__global__ __launch_bounds__(kBlock) void k_pci_ext(Consts c, float dt, const float4 *__restrict__ P, const float4 *__restrict__ V,
                                                    const uint32_t *__restrict__ nl, const int *__restrict__ cnt,
                                                    float4 *__restrict__ EF, float4 *__restrict__ PF, float4 *__restrict__ PB0,
                                                    float4 *__restrict__ PP, RigidView rv, const uint2 *__restrict__ stage_src, const int *__restrict__ stage_cnt)
{
    constexpr bool STAGED = SWEEP == SWEEP_STAGED, QUAD = SWEEP == SWEEP_QUAD;
    using K = KF<RX>;                                        // kernel functions of the handle's arithmetic (sph_device.h)
    extern __shared__ float4 s_operand[];
    const uint32_t *nlb = nullptr;
    SPH_SWEEP_PROLOGUE_M(QUAD)
    (void)kb; (void)nlbp;
    uint32_t *s_src = reinterpret_cast<uint32_t *>(s_operand + c.stage_cap);
    const bool staged = STAGED && stage_operand_src(c, s_operand, s_src, P, stage_src, stage_cnt, blk);
    const float4 vi = V[ii];
    const float rho_i = pi.w;
    float fa[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float &wx = fa[0], &wy = fa[1], &wz = fa[2];
    float &tx = fa[3], &ty = fa[4], &tz = fa[5];
    auto pair = [&](const float4 pj, const float4 vj, const uint32_t j) {
        float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        float r = K::norm3(dx, dy, dz);
        if (RIGID && (j & kRigidTag)) { rigid_viscosity(c, rv, vi, rho_i, pj, j, dx, dy, dz, r, wx, wy, wz); return; }
        float st = c.tens_c * K::w(c, r);                 // solver_base.py:216
        tx += st * dx; ty += st * dy; tz += st * dz;
        float vx = vi.x - vj.x, vy = vi.y - vj.y, vz = vi.z - vj.z;
        float shear = dot3(vx, vy, vz, dx, dy, dz);          // :183
        if (shear < 0.f) {
            F3 g = K::grad(c, dx, dy, dz, r);
            float q2 = r * r;
            float nu = c.visc_num / (rho_i + pj.w);          // :187
            float pi_ = -nu * shear / (q2 + c.visc_eps_h2);  // :188
            float sv = c.neg_m * pi_;                        // :189
            wx += sv * g.x; wy += sv * g.y; wz += sv * g.z;
        }
    };
    if (QUAD) for_fluid_nbrs_quad<RIGID, true>(nlp, kf, q, fa, P, V, rv, pair);
    else if (staged) for_staged_nbrs_pv<RIGID>(nlp, kf, s_operand, s_src, V, rv, pair);
    else for_fluid_nbrs<RIGID, true>(nlp, kf, P, V, rv, pair);
    if (!owner) return;
    float ten[3] = {tx * c.m, ty * c.m, tz * c.m};           // :209
    float vis[3] = {wx * c.m, wy * c.m, wz * c.m};           // :175
    float g[3] = {c.gravity * 0.0f, c.gravity * -1.0f, c.gravity * 0.0f};
    float pos[3] = {pi.x, pi.y, pi.z};
    float v[3] = {vi.x, vi.y, vi.z};
    float ext[3], vp[3], pp[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        ext[a] = (g[a] + ten[a]) + vis[a];                   // pcisph_solver.py:243
        vp[a] = v[a] + dt * (ext[a] + 0.0f) / c.m;           // :76 with press_force = 0 after reset()
        pp[a] = pos[a] + dt * vp[a];                         // :77
    }
    clamp_walls(c, pp, vp);
    EF[i] = make_float4(ext[0], ext[1], ext[2], 0.f);
    PF[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    PB0[i] = make_float4(pi.x, pi.y, pi.z, 0.f);             // press_iter = 0
    PP[i] = make_float4(pp[0], pp[1], pp[2], 0.f);
}

// Tiles without pressure in the PCISPH pressure loop (round 3; the idea of the DFSPH density loop, sph_kernels.h: stage_sources_flagged).
// press_iter = max(0, press_iter + delta * (rho_predict - rho0)) and the reference's density sum has no self term: early in a collapse the
// pressure is 0 nearly everywhere (6 % of the particles of pcisph_1m carry one at step 60, 29 % at step 100: tools/pci_sparsity.py).
// k_pci_press of a tile whose staged pressures are all 0 produces press_force = 0 and the zero-pressure prediction; if its outputs already
// hold those (zero_press[tile]: set by k_pci_ext's values at the start of a step, kept by this sweep) it returns.  Bit-identical
// (SPH_TILE_SKIP=0); single-GPU staged handles without rigid entries.  pcisph_1m steps 21-120: press sweep 68 -> 52 us, 137 -> 152
// Mparticle-steps/s.  The same for predict_rho (skip when no staged predicted position was rewritten, the own pressures are and stay 0 and the
// ping-pong destination already holds the zeros) was built and measured: at tile granularity the rewritten set dilates to everything,
// no tile ever returned, and the check cost 3 us per launch -- not kept.
// predict_rho (:91-103) + compute_residual partials (:126-138) + the iter_press this particle would see next (:105-109).
//   P here is PP = predicted positions: the neighbour SET is the list (current positions), the kernel argument is not.
template <bool RIGID, int SWEEP, bool RX = false>
__global__ __launch_bounds__(kBlock) void k_pci_predict_rho(Consts c, float delta, const float4 *__restrict__ P,
                                                            const float4 *__restrict__ WP, const uint32_t *__restrict__ nl,
                                                            const uint32_t *__restrict__ nlb, const int *__restrict__ cnt,
                                                            const DevScalars *__restrict__ ds, const float4 *__restrict__ PBin,
                                                            float4 *__restrict__ PBout, float *__restrict__ rho_predict,
                                                            double *__restrict__ psum, int *__restrict__ pcnt, int gate, RigidView rv,
                                                            const uint2 *__restrict__ stage_src, const int *__restrict__ stage_cnt)
{
    constexpr bool STAGED = SWEEP == SWEEP_STAGED, QUAD = SWEEP == SWEEP_QUAD;
    using K = KF<RX>;                                        // kernel functions of the handle's arithmetic (sph_device.h)
    extern __shared__ float4 s_operand[];
    if (gate_closed(ds, gate)) return;
    SPH_SWEEP_PROLOGUE_M(QUAD)
    const bool staged = STAGED && stage_operand(c, s_operand, P, stage_src, stage_cnt, blk);
    float fa[1] = {0.f};
    float &rp = fa[0];
    auto pair = [&](const float4 pj, const float4, const uint32_t j) {
        float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;   // rigid entries: the body where it is now (:159-161)
        if (RIGID && (j & kRigidTag)) rp += K::w(c, K::norm3(dx, dy, dz)) * pj.w * c.rho0;
        else rp += K::w(c, K::norm3(dx, dy, dz)) * c.m;      // :155-156
    };
    if (QUAD) for_fluid_nbrs_quad<RIGID, false>(nlp, kf, q, fa, P, nullptr, rv, pair);
    else if (staged) for_staged_nbrs<RIGID>(nlp, kf, s_operand, rv, pair);
    else for_fluid_nbrs<RIGID, false>(nlp, kf, P, nullptr, rv, pair);
    float wa[1] = {0.f};
    float &rb = wa[0];
    auto wall = [&](const float4 pj) {
        float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        rb += K::w(c, K::norm3(dx, dy, dz)) * pj.w;          // :167-168
    };
    if (QUAD) for_nbrs_p_quad(nlbp, kb, q, wa, WP, wall);
    else for_nbrs_p(nlbp, kb, WP, wall);
    float val = 0.f;
    int flag = 0;
    if (live) {
        const float rho_p = c.boundary_handle ? rp + rb * c.rho0 : rp;   // :100 / :102
        const float err = rho_p - c.rho0;                    // :103
        if (owner) rho_predict[i] = rho_p;
        const float4 pb = PBin[i];
        float pr = pb.w + err * delta;                       // :107
        pr = rmax(0.0f, pr);                                 // :108
        if (owner) PBout[i] = make_float4(pb.x, pb.y, pb.z, pr);
        val = rmax(err, 0.0f);                               // :132
        flag = val > 0.0f && !ghost;                         // ghosts (multi-GPU) are counted by their owner
    }
    if (QUAD) block_partial_mean_quad(blk, (double)val, flag, owner, psum, pcnt);
    else block_partial_mean(blk, (double)val, flag, psum, pcnt);
}

// update_press_force (:111-124, :192-224) + predict_vel_pos (:73-89).   P here is PB = (pos, press_iter)
template <bool RIGID, int SWEEP, bool RX = false>
__global__ __launch_bounds__(kBlock) void k_pci_press(Consts c, float dt, const float4 *__restrict__ P, const float4 *__restrict__ WP,
                                                      const uint32_t *__restrict__ nl, const uint32_t *__restrict__ nlb,
                                                      const int *__restrict__ cnt, const float *__restrict__ rho,
                                                      const float4 *__restrict__ V, const float4 *__restrict__ EF,
                                                      const DevScalars *__restrict__ ds, float4 *__restrict__ PF,
                                                      float4 *__restrict__ PP, int gate, RigidView rv, const uint2 *__restrict__ stage_src, const int *__restrict__ stage_cnt,
                                                      int *__restrict__ zero_press)
{
    constexpr bool STAGED = SWEEP == SWEEP_STAGED, QUAD = SWEEP == SWEEP_QUAD;
    using K = KF<RX>;                                        // kernel functions of the handle's arithmetic (sph_device.h)
    extern __shared__ float4 s_operand[];
    if (gate_closed(ds, gate)) return;
    const bool track = STAGED && !RIGID && zero_press != nullptr;          // tiles without pressure, see the note above k_pci_predict_rho
    SPH_SWEEP_PROLOGUE_B(QUAD, track ? (int)blockIdx.x : xcd_block(blockIdx.x, gridDim.x))
    bool staged, all_zero = false;
    if (track) {
        const int was_zero = zero_press[blk];                      // PF / PP of this tile hold the zero-pressure values (read before the barriers below)
        const int verdict = stage_operand_w_checked<false>(c, s_operand, P, stage_src, stage_cnt, blk);
        all_zero = verdict == 2;                                   // every pressure this tile can see is 0: every term below is +-0
        if (all_zero && was_zero) return;                          // ... and its outputs already say so
        if (threadIdx.x == 0) zero_press[blk] = all_zero ? 1 : 0;
        staged = verdict != 0;
    } else {
        staged = STAGED && stage_operand(c, s_operand, P, stage_src, stage_cnt, blk);
    }
    const float p_i = pi.w;
    constexpr float kRho0Sq = 1000000.0f;                    // self.rho_0 ** 2 (Python int)
    constexpr float kRcpRho0Sq = 1.0f / 1000000.0f;
    float fa[3] = {0.f, 0.f, 0.f};
    float &fx = fa[0], &fy = fa[1], &fz = fa[2];
    const float rho_own = (RIGID || c.boundary_handle) ? rho[ii] : 1.0f;
    const Recip rden = recip_prepare(rho_own * rho_own);     // rho_i ** 2, the divisor of every rigid term (:208)
    auto pair = [&](const float4 pj, const float4, const uint32_t j) {
        float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        float r = K::norm3(dx, dy, dz);
        F3 g = K::grad(c, dx, dy, dz, r);
        if (RIGID && (j & kRigidTag)) {
            const float a = pj.w * c.rho0 * p_i;                                  // :208
            fx += div_shared(a * g.x, rden) * c.m; fy += div_shared(a * g.y, rden) * c.m; fz += div_shared(a * g.z, rden) * c.m;   // :210
            return;
        }
        float ps = p_i + pj.w;
        fx += div_const(ps * g.x, kRho0Sq, kRcpRho0Sq) * c.m * c.m;   // :199
        fy += div_const(ps * g.y, kRho0Sq, kRcpRho0Sq) * c.m * c.m;
        fz += div_const(ps * g.z, kRho0Sq, kRcpRho0Sq) * c.m * c.m;
    };
    // (all_zero: the sums would be +0 + (+-0 terms) = +0, the accumulators as they stand)
    if (all_zero) {}
    else if (QUAD) for_fluid_nbrs_quad<RIGID, false>(nlp, kf, q, fa, P, nullptr, rv, pair);
    else if (staged) for_staged_nbrs<RIGID>(nlp, kf, s_operand, rv, pair);
    else for_fluid_nbrs<RIGID, false>(nlp, kf, P, nullptr, rv, pair);
    float wa[3] = {0.f, 0.f, 0.f};
    float &bx = wa[0], &by = wa[1], &bz = wa[2];
    if (c.boundary_handle && !all_zero) {
        const float rho_i_2 = rho_own * rho_own;             // :221
        auto wall = [&](const float4 pj) {
            float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
            float r = K::norm3(dx, dy, dz);
            F3 g = K::grad(c, dx, dy, dz, r);
            float s = pj.w * p_i / rho_i_2;                  // :223
            bx -= s * g.x; by -= s * g.y; bz -= s * g.z;
        };
        if (QUAD) for_nbrs_p_quad(nlbp, kb, q, wa, WP, wall);
        else for_nbrs_p(nlbp, kb, WP, wall);
    }
    if (!owner) return;
    float pf[3];
    if (c.boundary_handle) {
        pf[0] = -fx + bx * c.rho0 * c.m; pf[1] = -fy + by * c.rho0 * c.m; pf[2] = -fz + bz * c.rho0 * c.m;   // :120
    } else {
        pf[0] = -fx; pf[1] = -fy; pf[2] = -fz;               // :122
    }
    const float4 vi = V[i], e = EF[i];
    float ext[3] = {e.x, e.y, e.z};
    float v[3] = {vi.x, vi.y, vi.z};
    float pos[3] = {pi.x, pi.y, pi.z};
    float vp[3], pp[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        vp[a] = v[a] + dt * (ext[a] + pf[a]) / c.m;          // :76
        pp[a] = pos[a] + dt * vp[a];                         // :77
    }
    clamp_walls(c, pp, vp);
    PF[i] = make_float4(pf[0], pf[1], pf[2], 0.f);
    PP[i] = make_float4(pp[0], pp[1], pp[2], 0.f);
}

// integration :226-245
__global__ __launch_bounds__(kBlock) void k_pci_integrate(Consts c, float dt, const float4 *__restrict__ P, const float4 *__restrict__ V,
                                                          const float4 *__restrict__ EF, const float4 *__restrict__ PF,
                                                          float4 *__restrict__ Pn, float4 *__restrict__ Vn)
{
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= c.n) return;
    const float4 p = P[i], vi = V[i], e = EF[i], f = PF[i];
    float pos[3] = {p.x, p.y, p.z};
    float vel[3] = {vi.x, vi.y, vi.z};
    float ext[3] = {e.x, e.y, e.z}, pf[3] = {f.x, f.y, f.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        vel[a] = vel[a] + dt * (ext[a] + pf[a]) / c.m;       // :229-230
        vel[a] *= 0.9999f;                                   // :231
        pos[a] = pos[a] + dt * vel[a];                       // :232
    }
    clamp_walls(c, pos, vel);
    Pn[i] = make_float4(pos[0], pos[1], pos[2], 0.f);
    Vn[i] = make_float4(vel[0], vel[1], vel[2], 0.f);
}

// ======================================================================================
// IISPH
// ======================================================================================
// predict_advection, first half (:43-56): tension, viscosity, f_adv, v_adv, d_ii.   P = (pos, rho), V = (vel, -)
template <bool RIGID, int SWEEP, bool RX = false>
__global__ __launch_bounds__(kBlock) void k_ii_advect(Consts c, float dt, const float4 *__restrict__ P, const float4 *__restrict__ V,
                                                      const float4 *__restrict__ WP, const uint32_t *__restrict__ nl,
                                                      const uint32_t *__restrict__ nlb, const int *__restrict__ cnt,
                                                      float4 *__restrict__ VA, float4 *__restrict__ DII, RigidView rv,
                                                      const uint2 *__restrict__ stage_src, const int *__restrict__ stage_cnt)
{
    constexpr bool STAGED = SWEEP == SWEEP_STAGED, QUAD = SWEEP == SWEEP_QUAD;
    using K = KF<RX>;                                        // kernel functions of the handle's arithmetic (sph_device.h)
    extern __shared__ float4 s_operand[];
    SPH_SWEEP_PROLOGUE_M(QUAD)
    uint32_t *s_src = reinterpret_cast<uint32_t *>(s_operand + c.stage_cap);
    const bool staged = STAGED && stage_operand_src(c, s_operand, s_src, P, stage_src, stage_cnt, blk);
    const float4 vi = V[ii];
    const float rho_i = pi.w;
    const float s_f = c.neg_m / (rho_i * rho_i);             // compute_d_ii :280 (same value for every fluid neighbour)
    float fa[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float &wx = fa[0], &wy = fa[1], &wz = fa[2];
    float &tx = fa[3], &ty = fa[4], &tz = fa[5];
    float &ex = fa[6], &ey = fa[7], &ez = fa[8];
    auto pair = [&](const float4 pj, const float4 vj, const uint32_t j) {
        float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        float r = K::norm3(dx, dy, dz);
        F3 g = K::grad_in(c, dx, dy, dz, r);
        if (RIGID && (j & kRigidTag)) {
            const float sr = -pj.w * c.rho0 / (rho_i * rho_i);             // compute_d_ii :286
            ex += sr * g.x; ey += sr * g.y; ez += sr * g.z;
            rigid_viscosity(c, rv, vi, rho_i, pj, j, dx, dy, dz, r, wx, wy, wz);
            return;
        }
        ex += s_f * g.x; ey += s_f * g.y; ez += s_f * g.z;
        float st = c.tens_c * K::w_in(c, r);                 // solver_base.py:216
        tx += st * dx; ty += st * dy; tz += st * dz;
        float vx = vi.x - vj.x, vy = vi.y - vj.y, vz = vi.z - vj.z;
        float shear = dot3(vx, vy, vz, dx, dy, dz);          // :183
        if (shear < 0.f) {
            float q2 = r * r;
            float nu = c.visc_num / (rho_i + pj.w);          // :187
            float pi_ = -nu * shear / (q2 + c.visc_eps_h2);  // :188
            float sv = c.neg_m * pi_;                        // :189
            wx += sv * g.x; wy += sv * g.y; wz += sv * g.z;
        }
    };
    if (QUAD) for_fluid_nbrs_quad<RIGID, true>(nlp, kf, q, fa, P, V, rv, pair);
    else if (staged) for_staged_nbrs_pv<RIGID>(nlp, kf, s_operand, s_src, V, rv, pair);
    else for_fluid_nbrs<RIGID, true>(nlp, kf, P, V, rv, pair);
    float wa[3] = {0.f, 0.f, 0.f};
    float &bx = wa[0], &by = wa[1], &bz = wa[2];
    if (c.boundary_handle) {
        const float den = rho_i * rho_i;
        auto wall = [&](const float4 pj) {
            float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
            float r = K::norm3(dx, dy, dz);
            F3 g = K::grad_in(c, dx, dy, dz, r);
            float s = -pj.w / den;                           // compute_boundary_d_ii :292
            bx += s * g.x; by += s * g.y; bz += s * g.z;
        };
        if (QUAD) for_nbrs_p_quad(nlbp, kb, q, wa, WP, wall);
        else for_nbrs_p(nlbp, kb, WP, wall);
    }
    if (!owner) return;
    float ten[3] = {tx * c.m, ty * c.m, tz * c.m};
    float vis[3] = {wx * c.m, wy * c.m, wz * c.m};
    float g[3] = {c.gravity * 0.0f, c.gravity * -1.0f, c.gravity * 0.0f};
    float v[3] = {vi.x, vi.y, vi.z};
    float va[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float f = (g[a] + ten[a]) + vis[a];                  // iisph_solver.py:46
        va[a] = v[a] + dt * f / c.m;                         // :48
    }
    float d[3] = {ex, ey, ez}, b[3] = {bx, by, bz};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (c.boundary_handle) d[a] = (d[a] + b[a] * c.rho0) * dt * dt;   // :54
        else d[a] = d[a] * dt * dt;                                        // :56
    }
    VA[i] = make_float4(va[0], va[1], va[2], 0.f);
    DII[i] = make_float4(d[0], d[1], d[2], 0.f);
}

// predict_advection, second half (:58-82): rho_adv, p_iter = 0.5 p_past, a_ii.   P = (pos, rho), V = VA = (v_adv, -)
template <bool RIGID, int SWEEP, bool RX = false>
__global__ __launch_bounds__(kBlock) void k_ii_rho_adv(Consts c, float dt, const float4 *__restrict__ P, const float4 *__restrict__ V,
                                                       const float4 *__restrict__ WP, const uint32_t *__restrict__ nl,
                                                       const uint32_t *__restrict__ nlb, const int *__restrict__ cnt,
                                                       const float4 *__restrict__ DII, const float *__restrict__ p_past,
                                                       float *__restrict__ rho_adv, float *__restrict__ a_ii, float4 *__restrict__ PB0,
                                                       RigidView rv, const uint2 *__restrict__ stage_src, const int *__restrict__ stage_cnt)
{
    constexpr bool STAGED = SWEEP == SWEEP_STAGED, QUAD = SWEEP == SWEEP_QUAD;
    using K = KF<RX>;                                        // kernel functions of the handle's arithmetic (sph_device.h)
    extern __shared__ float4 s_operand[];
    SPH_SWEEP_PROLOGUE_M(QUAD)
    float2 *s_v2 = reinterpret_cast<float2 *>(s_operand + c.stage_cap);
    const bool staged = STAGED && stage_operand_pv(c, s_operand, s_v2, P, V, stage_src, stage_cnt, blk);
    const float4 vi = V[ii], di = DII[ii];
    const float rho_i = pi.w;
    const float cji = -dt * dt * c.m / (rho_i * rho_i);      // scalar prefix of d_ji, compute_a_ii :302-303
    float fa[2] = {0.f, 0.f};
    float &ra = fa[0], &aii = fa[1];
    auto pair = [&](const float4 pj, const float4 vj, const uint32_t j) {
        float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        float r = K::norm3(dx, dy, dz);
        F3 g = K::grad_in(c, dx, dy, dz, r);
        float ex = di.x - cji * -g.x, ey = di.y - cji * -g.y, ez = di.z - cji * -g.z;   // d_ii[i] - d_ji; gradW(-q) = -gradW(q)
        if (RIGID && (j & kRigidTag)) {
            const F3 w = rigid_velocity(rv, pj, dt, true);                             // compute_rho_adv :337-339
            ra += pj.w * dot3(vi.x - w.x, vi.y - w.y, vi.z - w.z, g.x, g.y, g.z) * c.rho0;   // :342
            aii += pj.w * dot3(ex, ey, ez, g.x, g.y, g.z) * c.rho0;                    // compute_a_ii :312
            return;
        }
        ra += c.m * dot3(vi.x - vj.x, vi.y - vj.y, vi.z - vj.z, g.x, g.y, g.z);        // compute_rho_adv :332
        aii += c.m * dot3(ex, ey, ez, g.x, g.y, g.z);                                  // compute_a_ii :304
    };
    if (QUAD) for_fluid_nbrs_quad<RIGID, true>(nlp, kf, q, fa, P, V, rv, pair);
    else if (staged) for_staged_nbrs_pv2<RIGID>(nlp, kf, s_operand, s_v2, rv, pair);
    else for_fluid_nbrs<RIGID, true>(nlp, kf, P, V, rv, pair);
    float wa[2] = {0.f, 0.f};
    float &rb = wa[0], &ab = wa[1];
    if (c.boundary_handle) {
        auto wall = [&](const float4 pj) {
            float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
            float r = K::norm3(dx, dy, dz);
            F3 g = K::grad_in(c, dx, dy, dz, r);
            rb += pj.w * dot3(vi.x, vi.y, vi.z, g.x, g.y, g.z);                        // compute_rho_adv_boundary :349
            float ex = di.x - cji * -g.x, ey = di.y - cji * -g.y, ez = di.z - cji * -g.z;
            ab += pj.w * dot3(ex, ey, ez, g.x, g.y, g.z);                              // compute_a_ii_boundary :322
        };
        if (QUAD) for_nbrs_p_quad(nlbp, kb, q, wa, WP, wall);
        else for_nbrs_p(nlbp, kb, WP, wall);
    }
    if (!owner) return;
    if (c.boundary_handle) {
        rho_adv[i] = (ra + rb * c.rho0) * dt + rho_i;        // :64
        a_ii[i] = aii + ab * c.rho0;                         // :75
    } else {
        rho_adv[i] = ra * dt + rho_i;                        // :67
        a_ii[i] = aii;                                       // :77
    }
    PB0[i] = make_float4(pi.x, pi.y, pi.z, 0.5f * p_past[i]);   // :68
}

// compute_all_d_ij (:130-135, :324-327).   P here is PB = (pos, p_iter)
template <bool RIGID, int SWEEP, bool RX = false>
__global__ __launch_bounds__(kBlock) void k_ii_dij(Consts c, float dt, const float4 *__restrict__ P, const float *__restrict__ rho,
                                                   const uint32_t *__restrict__ nl, const int *__restrict__ cnt,
                                                   const DevScalars *__restrict__ ds, float4 *__restrict__ DIJ, int gate, RigidView rv,
                                                   const uint2 *__restrict__ stage_src, const int *__restrict__ stage_cnt, int *__restrict__ zero_dij)
{
    constexpr bool STAGED = SWEEP == SWEEP_STAGED, QUAD = SWEEP == SWEEP_QUAD;
    using K = KF<RX>;                                        // kernel functions of the handle's arithmetic (sph_device.h)
    extern __shared__ float4 s_operand[];
    if (gate_closed(ds, gate)) return;
    const uint32_t *nlb = nullptr;
    // Tiles without pressure (round 3, as in k_pci_press): d_ij = dt^2 sum_j (-m p_j / rho_j^2) grad W is 0 when every staged pressure is 0
    // (9 % of the particles of iisph_1m carry one at step 60, 32 % at step 100); a tile whose DIJ already holds the zeros (zero_dij[tile],
    // cleared at the start of a step) returns.  Rigid entries are not part of this sum (:319).  Bit-identical (SPH_TILE_SKIP=0).
    const bool track = STAGED && zero_dij != nullptr;
    SPH_SWEEP_PROLOGUE_B(QUAD, track ? (int)blockIdx.x : xcd_block(blockIdx.x, gridDim.x))
    (void)kb; (void)nlbp;
    float *s_rho = reinterpret_cast<float *>(s_operand + c.stage_cap);
    bool staged, all_zero = false;
    if (track) {
        const int was_zero = zero_dij[blk];
        const int verdict = stage_operand_scalar_checked(c, s_operand, s_rho, P, rho, stage_src, stage_cnt, blk);
        all_zero = verdict == 2;
        if (all_zero && was_zero) return;
        if (threadIdx.x == 0) zero_dij[blk] = all_zero ? 1 : 0;
        staged = verdict != 0;
    } else {
        staged = STAGED && stage_operand_scalar(c, s_operand, s_rho, P, rho, stage_src, stage_cnt, blk);
    }
    float fa[3] = {0.f, 0.f, 0.f};
    float &sx = fa[0], &sy = fa[1], &sz = fa[2];
    auto pair = [&](const float4 pj, const float rho_j, const uint32_t j) {
        if (RIGID && (j & kRigidTag)) return;                // compute_d_ij: fluid neighbours only (:319)
        float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        float r = K::norm3(dx, dy, dz);
        F3 g = K::grad_in(c, dx, dy, dz, r);
        const float a = c.neg_m * pj.w;                      // - m * p_iter[j]
        const Recip den = recip_prepare(rho_j * rho_j);
        sx += div_shared(a * g.x, den); sy += div_shared(a * g.y, den); sz += div_shared(a * g.z, den);   // :327
    };
    if (all_zero) {}                                         // (the sums would be +0 + (+-0 terms) = +0)
    else if (QUAD) for_nbrs_ps_quad<RIGID>(nlp, kf, q, fa, P, rho, rv, pair);
    else if (staged) for_staged_nbrs_ps2<RIGID>(nlp, kf, s_operand, s_rho, rv, pair);
    else for_nbrs_ps<RIGID>(nlp, kf, P, rho, rv, pair);
    if (!owner) return;
    DIJ[i] = make_float4(sx * dt * dt, sy * dt * dt, sz * dt * dt, 0.f);   // :135
}

// update_p (:137-157) + compute_residual partials (:110-121).   P = PBin = (pos, p_iter); writes PBout = (pos, new p_iter)
template <bool RIGID, int SWEEP, bool RX = false>
__global__ __launch_bounds__(kBlock) void k_ii_update_p(Consts c, float dt, const float4 *__restrict__ P, const float4 *__restrict__ DII,
                                                        const float4 *__restrict__ DIJ, const float4 *__restrict__ WP,
                                                        const uint32_t *__restrict__ nl, const uint32_t *__restrict__ nlb,
                                                        const int *__restrict__ cnt, const float *__restrict__ rho,
                                                        const float *__restrict__ rho_adv, const float *__restrict__ a_ii,
                                                        const DevScalars *__restrict__ ds, float4 *__restrict__ PBout,
                                                        double *__restrict__ psum, int *__restrict__ pcnt, int gate, RigidView rv,
                                                        const uint2 *__restrict__ stage_src, const int *__restrict__ stage_cnt)
{
    constexpr bool STAGED = SWEEP == SWEEP_STAGED, QUAD = SWEEP == SWEEP_QUAD;
    using K = KF<RX>;                                        // kernel functions of the handle's arithmetic (sph_device.h)
    extern __shared__ float4 s_operand[];
    if (gate_closed(ds, gate)) return;
    SPH_SWEEP_PROLOGUE_M(QUAD)
    uint32_t *s_src = reinterpret_cast<uint32_t *>(s_operand + c.stage_cap);
    float *s_ex = reinterpret_cast<float *>(s_src + c.stage_cap), *s_ey = s_ex + c.stage_cap, *s_ez = s_ey + c.stage_cap;
    const int nst = STAGED ? stage_expand(stage_src, stage_cnt, blk, s_src) : -1;      // the list stays: DII is gathered through it
    const bool staged = nst >= 0;
    if (staged) {
        for (int e = threadIdx.x; e < nst; e += kBlock) {
            const uint32_t j = s_src[e];
            const float4 ev = DIJ[j];
            s_operand[e] = P[j];
            s_ex[e] = ev.x; s_ey[e] = ev.y; s_ez[e] = ev.z;
        }
        __syncthreads();
    }
    const float p_i = pi.w;
    const float rho_i = rho[ii];
    const float cji = -dt * dt * c.m / (rho_i * rho_i);      // :252-253
    const float4 a = DIJ[ii];
    float fa[1] = {0.f};
    float &sum = fa[0];
    auto pair = [&](const float4 pj, const float4 dj, const float4 ej, const uint32_t j) {
        float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        float r = K::norm3(dx, dy, dz);
        F3 g = K::grad_in(c, dx, dy, dz, r);
        if (RIGID && (j & kRigidTag)) {
            sum += dot3(a.x, a.y, a.z, g.x, g.y, g.z) * pj.w * c.rho0;   // sum_factor :261
            return;
        }
        float jx = cji * -g.x * p_i, jy = cji * -g.y * p_i, jz = cji * -g.z * p_i;     // d_ji
        float tx = a.x - dj.x * pj.w - (ej.x - jx);
        float ty = a.y - dj.y * pj.w - (ej.y - jy);
        float tz = a.z - dj.z * pj.w - (ej.z - jz);
        sum += c.m * dot3(tx, ty, tz, g.x, g.y, g.z);        // sum_factor :254
    };
    if (QUAD) for_nbrs_3_quad<RIGID>(nlp, kf, q, fa, P, DII, DIJ, rv, pair);
    else if (staged) for_staged_nbrs_3e<RIGID>(nlp, kf, s_operand, s_src, s_ex, s_ey, s_ez, DII, rv, pair);
    else for_nbrs_3<RIGID>(nlp, kf, P, DII, DIJ, rv, pair);
    float wa[1] = {0.f};
    float &bsum = wa[0];
    if (c.boundary_handle) {
        auto wall = [&](const float4 pj) {
            float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
            float r = K::norm3(dx, dy, dz);
            F3 g = K::grad_in(c, dx, dy, dz, r);
            bsum += dot3(a.x, a.y, a.z, g.x, g.y, g.z) * pj.w * c.rho0;   // sum_factor_boundary :240
        };
        if (QUAD) for_nbrs_p_quad(nlbp, kb, q, wa, WP, wall);
        else for_nbrs_p(nlbp, kb, WP, wall);
    }
    float val = 0.f;
    int flag = 0;
    if (live) {
        const float r_sum = c.boundary_handle ? sum + bsum : sum;        // :145 / :147
        const float aii = a_ii[i], radv = rho_adv[i];
        float p_new;
        if (fabsf(aii) > 1e-7f) p_new = 0.5f * p_i + 0.5f * ((c.rho0 - radv) - r_sum) / aii;   // :150-151 (omega = 0.5)
        else p_new = 0.0f;
        const float p = rmax(p_new, 0.0f);                   // :156
        if (owner) PBout[i] = make_float4(pi.x, pi.y, pi.z, p);
        flag = p > 0.0f && !ghost;                           // :115; ghosts (multi-GPU) are counted by their owner
        val = ((aii * p + r_sum) + radv) - 1000.0f;          // :116
    }
    if (QUAD) block_partial_mean_quad(blk, (double)val, flag, owner, psum, pcnt);
    else block_partial_mean(blk, (double)val, flag, psum, pcnt);
}

// intergation (:189-210) with compute_all_press_force (:172-176)
__global__ __launch_bounds__(kBlock) void k_ii_integrate(Consts c, float dt, const float4 *__restrict__ P, const float4 *__restrict__ VA,
                                                         const float4 *__restrict__ DII, const float4 *__restrict__ DIJ,
                                                         const float4 *__restrict__ PB, float4 *__restrict__ Pn, float4 *__restrict__ Vn,
                                                         float4 *__restrict__ FP, float *__restrict__ p_past)
{
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= c.n) return;
    const float4 p = P[i], va = VA[i], di = DII[i], dj = DIJ[i];
    const float p_it = PB[i].w;
    float pos[3] = {p.x, p.y, p.z};
    float v[3] = {va.x, va.y, va.z};
    float dii[3] = {di.x, di.y, di.z}, dij[3] = {dj.x, dj.y, dj.z};
    float vel[3], fp[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        fp[a] = (dij[a] + dii[a] * p_it) * c.m / (dt * dt);  // :176
        vel[a] = v[a] + dt * fp[a] / c.m;                    // :193
        vel[a] *= 0.9999f;                                   // :194
        pos[a] = pos[a] + dt * vel[a];                       // :195
    }
    clamp_walls(c, pos, vel);
    Pn[i] = make_float4(pos[0], pos[1], pos[2], 0.f);
    Vn[i] = make_float4(vel[0], vel[1], vel[2], 0.f);
    FP[i] = make_float4(fp[0], fp[1], fp[2], 0.f);
    p_past[i] = p_it;                                        // :209-210
}

// Force of the fluid on the rigid sample particles for wcsph / pcisph / iisph, gathered per rigid particle over its fluid
// neighbours in cell-walk order like k_rigid_force (dfsph): no atomics, same serialisation as the oracle.
//   RF_WCSPH  wcsph_solver.py:125-127   force += -ret * m,  ret = -V_j p_i / rho_i^2 * gradW * rho_0        (S = pressure[])
//   RF_PCISPH pcisph_solver.py:208-210  force += ret * m,   ret = V_j rho_0 press_iter_i * gradW / rho_i^2   (PB.w, every iteration)
//   RF_IISPH  iisph_solver.py:166-167   force += f * m,     f = V_j rho_0 / rho_i^2 * gradW * p_iter_i       (PB.w)
enum { RF_WCSPH = 0, RF_PCISPH = 2, RF_IISPH = 3 };

template <int MODE>
__global__ __launch_bounds__(kBlock) void k_rigid_force_p(Consts c, int nr, const float4 *__restrict__ RP, const int *__restrict__ rid,
                                                          const float4 *__restrict__ P, const uint32_t *__restrict__ rnl,
                                                          const int *__restrict__ rcnt, const float *__restrict__ rho,
                                                          const float *__restrict__ S, const float4 *__restrict__ PB,
                                                          const DevScalars *__restrict__ ds, float *__restrict__ force, int gate)
{
    if (gate_closed(ds, gate)) return;
    int r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= nr) return;
    const float4 pr = RP[r];
    float fx = 0.f, fy = 0.f, fz = 0.f;
    struct Op { float4 p; float rho, s; };
    walk_list<Op>(rnl + nl_index(r, 0, c.kpitch), rcnt[r], [&](uint32_t i, Op &o) {      // k_build_rnl (sph_rigid_kernels.h)
        o.p = P[i]; o.rho = rho[i];
        o.s = MODE == RF_WCSPH ? S[i] : PB[i].w;
    }, [&](const Op &o, uint32_t) {
        const float4 pi = o.p;
        float ddx = pi.x - pr.x, ddy = pi.y - pr.y, ddz = pi.z - pr.z;
        float r2 = (ddx * ddx + ddy * ddy) + ddz * ddz;
        F3 g = grad_w(c, ddx, ddy, ddz, sqrtf(r2));
        const float rho_i = o.rho;
        if (MODE == RF_WCSPH) {
            const float s = -pr.w * o.s / (rho_i * rho_i);                               // :125
            fx += -(s * g.x * c.rho0) * c.m; fy += -(s * g.y * c.rho0) * c.m; fz += -(s * g.z * c.rho0) * c.m;   // :127
        } else if (MODE == RF_PCISPH) {
            const float a = pr.w * c.rho0 * o.s;                                         // :208
            const float den = rho_i * rho_i;
            fx += a * g.x / den * c.m; fy += a * g.y / den * c.m; fz += a * g.z / den * c.m;   // :209
        } else {
            const float s = pr.w * c.rho0 / (rho_i * rho_i);                             // :166
            const float p = o.s;
            fx += s * g.x * p * c.m; fy += s * g.y * p * c.m; fz += s * g.z * p * c.m;   // :167
        }
    });
    const int o = rid[r];
    force[3 * o] += fx; force[3 * o + 1] += fy; force[3 * o + 2] += fz;
}

}  // namespace sph
