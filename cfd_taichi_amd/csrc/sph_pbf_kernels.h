// sph_pbf_kernels.h -- PBF (pbf_solver.py) on the cell-sorted arrays and per-step neighbour lists of sph_kernels.h.
//
// pbf_solver.py is stale at the surveyed commit (its fluid callbacks still index particle fields with what for_all_neighbor now passes
// as particle structs, ParticleSystem.py:468-469); it is read the way the other solvers' callbacks were updated (callback argument
// -> the struct's pos / vel / index).  update_all_pos (:66-95) races on pos and vel; the kernels follow the barrier-synchronised
// schedule (every particle's own writes, then every particle's neighbour reads, then the viscosity update), which is one legal
// execution of the reference loop and the one the tests' CPU restatement follows.  No rigid coupling (the callbacks have none).
//
// Three sweeps per step:
//   k_pbf_lambda   rho (poly6), constrain, constrain_derivative, lambda           :32-52, 108-140   (the reference: five walks)
//   k_pbf_delta    delta_pos; externel_force_predict_pos; update_all_pos phase 1   :26-29, 55-64, 66-84
//   k_pbf_xsph     v_i = sum_j (vel_j - vel_i) W(|x_i - x_j|) on the NEW positions through the cell lists of the step's start; vel += c v   :86-98
#pragma once
#include "sph_kernels.h"

namespace sph {

struct PbfConsts {
    float kpoly;      // 315 / (64 pi h^3)                      solver_base.py:128
    float pih4;       // pi * h^4                               solver_base.py:120
    float w_corr;     // poly_kernel(s_corr_factor * kernel_h)  pbf_solver.py:148
    float neg_k;      // -k (tension)                           :151
    float c_visc;     // c                                      :92
    float eps;        // epsilon                                :17
    float lo[3], hi[3];   // clamp walls at particle_radius     :74-81
};

__device__ __forceinline__ float pow3f(float a) { return a * (a * a); }
__device__ __forceinline__ float poly_w(const Consts &c, const PbfConsts &k, float r)       // solver_base.py:123-129
{
    const float q = div_by_h(c, r);
    const float q2 = q * q;
    const float w = k.kpoly * pow3f(1.0f - q2);
    return q <= 1.0f ? w : 0.0f;
}
__device__ __forceinline__ F3 spiky_grad(const Consts &c, const PbfConsts &k, float dx, float dy, float dz, float r)   // solver_base.py:114-121
{
    const float q = div_by_h(c, r);
    const float t = 1.0f - q;
    const float a = -(45.0f * (t * t));
    const float den = k.pih4 * r;
    const bool in = q <= 1.0f && q > 0.0f;
    F3 o;
    o.x = in ? (a * dx) / den : 0.0f; o.y = in ? (a * dy) / den : 0.0f; o.z = in ? (a * dz) / den : 0.0f;
    return o;
}

// rho, constrain, constrain_derivative, lambda.  Pout = (pos, lambda) for the gather of k_pbf_delta.
// QUAD (both list sweeps): four lanes per particle, small scenes (walk_list_quad, sph_kernels.h)
template <bool QUAD>
__global__ __launch_bounds__(kBlock) void k_pbf_lambda(Consts c, PbfConsts k, const float4 *__restrict__ P, const float4 *__restrict__ WP,
                                                       const uint32_t *__restrict__ nl, const uint32_t *__restrict__ nlb,
                                                       const int *__restrict__ cnt, float *__restrict__ rho_out, float *__restrict__ lambda_out,
                                                       float4 *__restrict__ Pout, int rho_only)
{
    SPH_SWEEP_PROLOGUE_M(QUAD)
    float fa[5] = {0.001f, 0.f, 0.f, 0.f, 0.f};                    // rho starts at 0.001, solver_base.py:44
    float &rho = fa[0], &cx = fa[1], &cy = fa[2], &cz = fa[3], &sum = fa[4];
    auto pair = [&](const float4 pj) {
        const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        const float r = norm3(dx, dy, dz);
        rho += c.m * poly_w(c, k, r);                              // :169-170
        const F3 g = spiky_grad(c, k, dx, dy, dz, r);
        const float gx = g.x / c.rho0, gy = g.y / c.rho0, gz = g.z / c.rho0;
        cx += gx; cy += gy; cz += gz;                              // :116-117
        sum += (gx * gx + gy * gy) + gz * gz;                      // :133-134
    };
    if (QUAD) for_nbrs_p_quad(nlp, kf, q, fa, P, pair);
    else for_nbrs_p(nlp, kf, P, pair);
    float wa[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    float &rb = wa[0], &bx = wa[1], &by = wa[2], &bz = wa[3], &sb = wa[4];
    auto wall = [&](const float4 pj) {
        const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        const float r = norm3(dx, dy, dz);
        rb += pj.w * poly_w(c, k, r);                              // :173-176
        const F3 g = spiky_grad(c, k, dx, dy, dz, r);
        const float gx = g.x / c.rho0, gy = g.y / c.rho0, gz = g.z / c.rho0;
        bx += gx; by += gy; bz += gz;                              // :120-122
        sb += (gx * gx + gy * gy) + gz * gz;                       // :139-140
    };
    if (QUAD) for_nbrs_p_quad(nlbp, kb, q, wa, WP, wall);
    else for_nbrs_p(nlbp, kb, WP, wall);
    if (!owner) return;
    const float rho_i = c.boundary_handle ? rho + rb * c.rho0 : rho;
    if (rho_only) { rho_out[i] = rho_i; return; }                  // compute_all_rho alone (solver_base.py:36-50 with :166-174): pbf_lambda keeps its values
    const float con = rmax(rho_i / c.rho0 - 1.0f, 0.0f);          // :127-128
    float dxs = cx, dys = cy, dzs = cz;
    if (c.boundary_handle) { dxs = cx + bx; dys = cy + by; dzs = cz + bz; }   // :112
    const float cdcd = (dxs * dxs + dys * dys) + dzs * dzs;
    const float tot = c.boundary_handle ? (cdcd + sum) + sb : cdcd + sum;     // :48 / :50
    const float lam = con == 0.0f ? 0.0f : -con / (tot + k.eps);  // :39-52
    rho_out[i] = rho_i;
    lambda_out[i] = lam;
    Pout[i] = make_float4(pi.x, pi.y, pi.z, lam);
}

// delta_pos, the prediction and phase 1 of update_all_pos.  PL = (pos, lambda); writes Pn = new position, Vn = phase-1 velocity.
template <bool QUAD>
__global__ __launch_bounds__(kBlock) void k_pbf_delta(Consts c, PbfConsts k, float dt, const float4 *__restrict__ PL, const float4 *__restrict__ V,
                                                      const float4 *__restrict__ WP, const uint32_t *__restrict__ nl,
                                                      const uint32_t *__restrict__ nlb, const int *__restrict__ cnt,
                                                      float4 *__restrict__ dpos_out, float4 *__restrict__ Pn, float4 *__restrict__ Vn)
{
    const float4 *P = PL;
    SPH_SWEEP_PROLOGUE_M(QUAD)
    const float li = pi.w;
    float fa[3] = {0.f, 0.f, 0.f};
    float &ax = fa[0], &ay = fa[1], &az = fa[2];
    auto pair = [&](const float4 pj) {
        const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        const float r = norm3(dx, dy, dz);
        float sc = poly_w(c, k, r) / k.w_corr;                     // :148
        sc *= sc; sc *= sc; sc *= k.neg_k;                         // :149-151
        const F3 g = spiky_grad(c, k, dx, dy, dz, r);
        const float f = (li + pj.w) + sc;                          // :153
        ax += f * g.x; ay += f * g.y; az += f * g.z;
    };
    if (QUAD) for_nbrs_p_quad(nlp, kf, q, fa, PL, pair);
    else for_nbrs_p(nlp, kf, PL, pair);
    float wa[3] = {0.f, 0.f, 0.f};
    float &bx = wa[0], &by = wa[1], &bz = wa[2];
    auto wall = [&](const float4 pj) {
        const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        const float r = norm3(dx, dy, dz);
        float sc = poly_w(c, k, r) / k.w_corr;
        sc *= sc; sc *= sc; sc *= k.neg_k;
        const F3 g = spiky_grad(c, k, dx, dy, dz, r);
        const float f = li + sc;                                   // :164
        bx += f * g.x; by += f * g.y; bz += f * g.z;
    };
    if (QUAD) for_nbrs_p_quad(nlbp, kb, q, wa, WP, wall);
    else for_nbrs_p(nlbp, kb, WP, wall);
    if (!owner) return;
    float dp[3];
    if (c.boundary_handle) { dp[0] = (ax + bx) / c.rho0; dp[1] = (ay + by) / c.rho0; dp[2] = (az + bz) / c.rho0; }   // :62
    else { dp[0] = ax / c.rho0; dp[1] = ay / c.rho0; dp[2] = az / c.rho0; }                                          // :64
    const float4 vi = V[i];
    const float acc[3] = {c.gravity * 0.0f, c.gravity * -1.0f, c.gravity * 0.0f};      // reset(), solver_base.py:131-133
    const float pos[3] = {pi.x, pi.y, pi.z};
    float vel[3] = {vi.x, vi.y, vi.z}, pp[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        vel[a] += dt * acc[a];                                     // :28
        pp[a] = pos[a] + dt * vel[a];                              // :29
        pp[a] += dp[a];                                            // :69
        vel[a] = (pp[a] - pos[a]) / dt;                            // :70
        if (!c.boundary_handle) {                                  // :73-81
            if (pp[a] <= k.lo[a]) { pp[a] = k.lo[a]; vel[a] *= 0.5f; }
            if (pp[a] >= k.hi[a]) { pp[a] = k.hi[a]; vel[a] *= 0.5f; }
        }
    }
    dpos_out[i] = make_float4(dp[0], dp[1], dp[2], 0.f);
    Pn[i] = make_float4(pp[0], pp[1], pp[2], 0.f);
    Vn[i] = make_float4(vel[0], vel[1], vel[2], 0.f);
}

// phases 2 and 3 of update_all_pos: the 27-cell walk of for_all_neighbor over the cell lists of the step's start (belong_grid of the OLD
// position), distances and kernel on the NEW positions.  Writes the final state: Pfin = new position, Vfin = vel + c * v.
// QUAD (small scenes): the four lanes of a quad serve one particle; lane q evaluates candidate j0 + q of every batch of four of a cell and
// all four add the four terms in candidate order (quad_bcast, sph_kernels.h) -- a candidate that is skipped contributes +0, which leaves
// the running sums (never -0) unchanged: the same additions in the same order as the one-lane walk.
template <bool QUAD>
__global__ __launch_bounds__(kBlock) void k_pbf_xsph(Consts c, PbfConsts k, const float4 *__restrict__ Pold, const float4 *__restrict__ Pn,
                                                     const float4 *__restrict__ Vn, const int *__restrict__ cell_start,
                                                     float4 *__restrict__ Pfin, float4 *__restrict__ Vfin)
{
    const int blk = xcd_block(blockIdx.x, gridDim.x);
    const int q = QUAD ? (int)(threadIdx.x & 3) : 0;
    const int i = QUAD ? blk * (kBlock / 4) + (int)(threadIdx.x >> 2) : blk * kBlock + (int)threadIdx.x;
    if (i >= c.n) return;                                          // (a quad leaves together)
    const float4 po = Pold[i], pi = Pn[i], vi = Vn[i];
    int cx, cy, cz;
    cell_id_of(c, po.x, po.y, po.z, cx, cy, cz);                   // belong_grid, set by update_grid at the step's start
    float ax = 0.f, ay = 0.f, az = 0.f;
    auto term = [&](int j, float &tx, float &ty, float &tz) {      // (vel_j - vel_i) W, or nothing
        const float4 pj = Pn[j];
        const float ex = pi.x - pj.x, ey = pi.y - pj.y, ez = pi.z - pj.z;
        const float r = norm3(ex, ey, ez);
        if (r > c.h) return;                                       // :466
        const float4 vj = Vn[j];
        const float w = poly_w(c, k, r);                           // pbf_solver.py:98
        tx = (vj.x - vi.x) * w; ty = (vj.y - vi.y) * w; tz = (vj.z - vi.z) * w;
    };
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dz = -1; dz <= 1; ++dz) {
                const int x = cx + dx, y = cy + dy, z = cz + dz;
                if (x >= c.gx || y >= c.gy || z >= c.gz || x < 0 || y < 0 || z < 0) continue;          // ParticleSystem.py:453-456
                const int slot = cell_slot_xyz(c, x, y, z, x + y * c.sy + z * c.sz);
                if (slot < 0) continue;
                const int a = cell_start[slot], b = cell_start[slot + 1];
                if (QUAD) {
                    for (int j0 = a; j0 < b; j0 += 4) {
                        const int j = j0 + q;
                        float tx = 0.f, ty = 0.f, tz = 0.f;
                        if (j < b && j != i) term(j, tx, ty, tz);  // :461
                        ax += quad_bcast<0>(tx); ax += quad_bcast<1>(tx); ax += quad_bcast<2>(tx); ax += quad_bcast<3>(tx);
                        ay += quad_bcast<0>(ty); ay += quad_bcast<1>(ty); ay += quad_bcast<2>(ty); ay += quad_bcast<3>(ty);
                        az += quad_bcast<0>(tz); az += quad_bcast<1>(tz); az += quad_bcast<2>(tz); az += quad_bcast<3>(tz);
                    }
                } else {
                    for (int j = a; j < b; ++j) {
                        if (j == i) continue;                      // :461
                        float tx = 0.f, ty = 0.f, tz = 0.f;
                        const float4 pj = Pn[j];
                        const float ex = pi.x - pj.x, ey = pi.y - pj.y, ez = pi.z - pj.z;
                        const float r = norm3(ex, ey, ez);
                        if (r > c.h) continue;                     // :466
                        const float4 vj = Vn[j];
                        const float w = poly_w(c, k, r);           // pbf_solver.py:98
                        tx = (vj.x - vi.x) * w; ty = (vj.y - vi.y) * w; tz = (vj.z - vi.z) * w;
                        ax += tx; ay += ty; az += tz;
                    }
                }
            }
    if (q != 0) return;
    Pfin[i] = make_float4(pi.x, pi.y, pi.z, 0.f);
    Vfin[i] = make_float4(vi.x + k.c_visc * ax, vi.y + k.c_visc * ay, vi.z + k.c_visc * az, 0.f);   // :92 / :94
}

}  // namespace sph
