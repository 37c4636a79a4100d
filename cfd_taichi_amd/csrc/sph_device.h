// sph_device.h -- shared structs and device-side SPH math for the gfx950 kernels.
//
// Arithmetic contract: every expression below is written with the association the reference
// source text has (solver_base.py:74-103 and the per-pair callbacks of wcsph_solver.py /
// dfsph_solver.py), evaluated in IEEE f32 with contraction off and correctly rounded
// divide / sqrt (see build flags in cfd_taichi_amd/build.py).  tests/ checks the result
// against the independent CPU restatement in oracle/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sph {

#ifndef SPH_KBLOCK
#define SPH_KBLOCK 256
#endif
constexpr int kBlock = SPH_KBLOCK;   // threads per workgroup of every kernel (tuning knob at build time)

// Launch-invariant constants, passed to kernels by value (lands in SGPRs).
enum { CELL_ORDER_LINEAR = 0, CELL_ORDER_TILED = 1 };

struct Consts {
    float h;           // support radius = kernel_h = 4r            ParticleSystem.py:82, solver_base.py:17
    float m;           // particle_m                                ParticleSystem.py:83
    float d;           // particle_diameter
    float rho0;        // 1000                                      solver_base.py:19
    float gravity;
    float kw;          // 8/(pi h^3)                                solver_base.py:79
    float kg6;         // 48/(pi h^3) * 6                           solver_base.py:95,98
    float neg_kg6;     // -48/(pi h^3) * 6                          solver_base.py:100
    float r2_cut;      // largest f32 t with sqrtf(t) <= h: (|x_ij| > h) <=> (r2 > r2_cut)
    float rh;          // RN(1/h), for the exact division by the constant h
    float rh_s, h_s;   // rh * 2^-32 and h * 2^32: the same division for a numerator carried with a factor 2^32 (grad_w_scaled)
    float visc_num;    // 2*alpha*h*c_s (f64-folded)                solver_base.py:187
    float visc_eps_h2; // eps*h*h (f64-folded)                      solver_base.py:188
    float tens_c;      // -k/m*m (f64-folded)                       solver_base.py:216
    float neg_m;       // -particle_m                               solver_base.py:189
    float dt_cfl_num;  // 0.4*r*2 (f64-folded)                      dfsph_solver.py:112
    float clamp_lo[3]; // clamp walls (boundary_handle == 0)        wcsph_solver.py:54-63, dfsph_solver.py:241-250
    float clamp_hi[3];
    int gx, gy, gz, C; // grid_num, cell count                      ParticleSystem.py:100-103
    int sy, sz;        // _3d_to_1d_tran = (1, gx*gz, gx)           ParticleSystem.py:102
    int boundary_handle;
    int n;             // particles resident in the arrays
    int stride;        // neighbour-list row stride (>= n, multiple of 64)
    int kmax, kbmax;   // neighbour-list rows (fluid, wall)
    int kpitch, kbpitch; // rows between consecutive 64-particle tiles of a list (>= kmax; see build_scene)
    int strict_cells;  // slab handles: a particle with any cell coordinate outside the grid is binned nowhere (see cell_id_of)
    // Storage order of the cells (see cell_slot() in sph_kernels.h): where cell_start[] keeps each reference cell.
    int order;            // CELL_ORDER_LINEAR (the reference's 1-D index) or CELL_ORDER_TILED (Morton curve, see cell_slot_xyz)
    int S;                // slots in cell_start[] (>= C: tiles pad each axis to a multiple of the tile edge); slot S = "outside the grid"
    int tbits, tnx, tnxz; // tiles of 2^tbits cells per axis; tile strides: tiles along x, tiles along x times tiles along z
    const int *tile_rank; // position of every tile along the Morton curve of the tile coordinates
    int stage_cap;        // LDS staging: particles a workgroup may stage (see the plan in k_build_nl); 0 = staging off
    int nl16;             // fluid lists of staged workgroups hold 16-bit local indices, eight per 16-byte group (NlWriter)
    int kr_split;         // dfsph sweeps hand k / rho to the next sweep in a 4-byte array instead of a (pos, k / rho) float4 (k_correct)
    // tolerance-grade sweeps (SphConfig.arith = SPH_ARITH_RELAXED, sph_relaxed_kernels.h): m grad W = g x_ij with
    // g = rx_k1a q + rx_k1b (q <= 0.5) or rx_k2 (1 - q)^2 / r (q > 0.5); constants folded in f64
    float rx_k1a, rx_k1b; // 3 m kg6 / h^2, -2 m kg6 / h^2
    float rx_k2;          // -m kg6 / h
    float rx_rho0_m;      // rho0 / m (wall sums carry m like the fluid sums)
    // slab handles with two ghost columns per side (dfsph): ghosts of the INNER column (cell x == gw_left or gw_right; -1: none) have neighbour
    // lists of their own and run D1 and the correction sweeps like owned particles (k_build_nl: "walker"), so that their v / v* never has to
    // be refreshed inside a solver loop; ghost_walk = 1 on such handles
    int gw_left, gw_right, ghost_walk;
    int nbr_cap;          // DensFlow: tiles a tile's row may name (kNbrStride - 1; SPH_NBR_CAP lowers it so that tests reach the "unknown row" fallback)
    // cell edge of the grid the particles are binned into (cell_id_of).  Equal to h -- the reference's grid, ParticleSystem.py:100-101,490-494 --
    // except on Verlet handles (wcsph under the relaxed arithmetic, sph_relaxed_kernels.h): there the lists hold every pair within
    // h + skin = hcell and are rebuilt only once a particle has moved more than skin / 2 since the last build (verlet_thr2 = (skin / 2)^2)
    float hcell, verlet_thr2;
    int verlet;
    // attributes of dfsph_solver.py:26-29 that Taichi bakes into its kernels when they compile (sph_set_scalar(SPH_P_*) before the first step)
    int warm_start, adaptive_dt;      // :26-27 (read at :396, :404 and :113)
    float max_dt, min_dt;             // :28-29 (:114-117)
};

// Run-time scalars that live in device memory (0-d fields of the reference).
struct DevScalars {
    float dt;          // solver.delta_time[None]
    float dt2;         // solver.delta_time_2[None]
    float ps_dt;       // ps.delta_time[None]
    float mean;        // result of the last mean reduction (divergence error / rho_adv average)
    float vmax;        // result of the last max reduction
    int overflow;      // neighbour list overflow flag
    int max_nbrs;
    int max_wall_nbrs;
    int lost;          // particles outside the grid
    float rigid_vmax;  // max over rigid particles of |vel| + |omega x (x - c)|   dfsph_solver.py:104-110
    // Verlet handles: `moved` is raised by the integrator when a particle is more than skin / 2 away from where the lists were built; the sort +
    // list build kernels of the next step (all enqueued every step) run only if it is set, the density kernel behind them takes it down
    int rebuild, moved;
    double sum;        // last (sum, count) reduction: the host forms mean = sum / cnt (after an all-reduce when sharded)
    long long cnt;
    // device-side control of the reference's host loops (correct_divergence_error dfsph_solver.py:393-416,
    // correct_density_error :221-233) on a single GPU: kernels of an iteration that the loop would not have run exit at once
    int div_active, div_it, div_evals, dens_active;
    int dens_d7_active, dens_it, dens_cap, dens_capped;
    float div_err, div_past, div_first, dens_avg;
    // pcisph / iisph pressure loops reuse dens_active / dens_it / dens_cap / dens_capped / dens_avg; iisph_solver.py:97-100 adds:
    float res_prev;
    int res_have_prev, res_diverged, verlet_builds;
    // Slab handles that hide the residual's all-reduce behind the next correction sweep (step_dfsph_device_loops: "speculation"): the decision of
    // evaluation e is also kept in gate_hist[e & 1], so that a sweep enqueued BEHIND the reduction of evaluation e can still read the decision of
    // e - 1 without racing the kernel that takes e; stop_at = the evaluation whose decision closed the divergence loop (the correction sweep that ran
    // ahead of it is undone by the next residual launch)
    int gate_hist[2], stop_at;
    int overflow_any;  // slab handles: some slab's overflow flags were set at the last density-loop reduction (summed with the residual pair)
    // Per-build maxima of the list lengths, sharded: workgroup w raises shard w % kNoteShards, the host takes the maximum over the
    // shards into max_nbrs / max_wall_nbrs after a read-back.  (Thousands of waves checking ONE word cost 10 us of a 30 k-particle
    // list build: same-address traffic serialises even when it is only loads.)
    int nbr_shard[64], wall_shard[64];
    // dfsph loop parameters: the attributes dfsph_solver.py:21-25 sets and its Python-scope loops read at every step (:225, :400).  Written by
    // sph_create (the reference's values) and sph_set_scalar(SPH_P_*); k_ctrl_begin leaves them alone
    double p_dens_thr;   // density_threshold * rho_0 * 0.01, folded in f64 like the Python expression                 :225
    double p_div_thr;    // density_divergence_threshold                                                            :400
    int p_min_dens, p_min_div, p_max_div, p_pad;       // min_iteration_density, min / max_iteration_density_divergence
};
constexpr int kNoteShards = 64;

enum { GATE_NONE = 0, GATE_DIV = 1, GATE_DENS = 2, GATE_DENS_D7 = 3, GATE_HIST0 = 16, GATE_HIST1 = 17 };
__device__ __forceinline__ bool gate_closed(const DevScalars *ds, int gate)
{
    if (gate >= GATE_HIST0) return ds->gate_hist[gate - GATE_HIST0] == 0;
    if (gate == GATE_DIV) return ds->div_active == 0;
    if (gate == GATE_DENS) return ds->dens_active == 0;
    if (gate == GATE_DENS_D7) return ds->dens_d7_active == 0;
    return false;
}

struct F3 {
    float x, y, z;
};

// ---- correctly rounded f32 division at a fraction of the generic expansion ----------------------
// The sweeps need RN(a/b) (the oracle's IEEE divide), four times per pair.  hipcc's generic expansion is
// ~10 instructions + VCC hazard nops each (v_div_scale x2, v_rcp, 5 fma, v_div_fmas, v_div_fixup).
//
// (1) Division by the constant h: with y = RN(1/h), q0 = RN(a*y), e = a - h*q0 (exact, fma),
//     q = RN(q0 + e*y) is the correctly rounded quotient (Markstein's theorem; h's significand is not all
//     ones).  3 instructions.
// (2) Three numerators over one denominator: exactly the Newton-Raphson sequence LLVM emits for `/`
//     (fma0..fma4 + fmas) minus the v_div_scale / v_div_fixup range handling, with the reciprocal refinement
//     shared.  Operands here are h*r in [1e-7, 1e-2] and s*dx, far inside the range where v_div_scale is
//     the identity, so the result is bit-identical to `/`.  3 + 3*5 instructions instead of 3*10 + nops.
// tests/test_parity_gpu.py checks both against the oracle's plain divisions bit for bit.
__device__ __forceinline__ float div_by_h(const Consts &c, float a)
{
#ifdef SPH_GENERIC_DIV
    return a / c.h;
#endif
    float q0 = a * c.rh;
    float e = __builtin_fmaf(-q0, c.h, a);
    return __builtin_fmaf(e, c.rh, q0);
}
struct Recip {
    float d, y;
};
__device__ __forceinline__ Recip recip_prepare(float d)
{
    float y0 = __builtin_amdgcn_rcpf(d);
    float e = __builtin_fmaf(-d, y0, 1.0f);
    Recip r;
    r.d = d;
    r.y = __builtin_fmaf(e, y0, y0);
    return r;
}
__device__ __forceinline__ float div_shared(float a, const Recip &r)
{
#ifdef SPH_GENERIC_DIV
    return a / r.d;
#endif
    float q0 = a * r.y;
    float r0 = __builtin_fmaf(-r.d, q0, a);
    float q1 = __builtin_fmaf(r0, r.y, q0);
    float r1 = __builtin_fmaf(-r.d, q1, a);
    return __builtin_fmaf(r1, r.y, q1);
}

// solver_base.py:76-88.  Branch-free form: both polynomial pieces are cheap, the select keeps the
// value of the branch the reference would have taken (identical f32 operations per piece).
__device__ __forceinline__ float cubic_w(const Consts &c, float r)
{
    float q = div_by_h(c, r);                             // r / h
    float q2 = q * q;
    float q3 = q2 * q;
    float w1 = c.kw * (6.0f * (q3 - q2) + 1.0f);          // 0 <= q <= 0.5
    float t = 1.0f - q;
    float w2 = 2.0f * c.kw * (t * (t * t));               // 0.5 < q <= 1
    bool in1 = (0.0f <= q) && (q <= 0.5f);
    bool in2 = (0.5f < q) && (q <= 1.0f);
    return in1 ? w1 : (in2 ? w2 : 0.0f);
}

// solver_base.py:90-103 (with the reference's factor 6).  The two branches differ only in the
// scalar s; selecting s first leaves ONE set of three IEEE divides per pair instead of two
// divergent sets.
__device__ __forceinline__ F3 grad_w(const Consts &c, float dx, float dy, float dz, float r_norm)
{
    float q = div_by_h(c, r_norm);                        // r_norm / h
    float q2 = q * q;
    float s1 = c.kg6 * (3.0f * q2 - 2.0f * q);            // 1e-5 < q <= 0.5
    float t = 1.0f - q;
    float s2 = c.neg_kg6 * (t * t);                       // 0.5 < q <= 1
    bool in1 = (1e-5f < q) && (q <= 0.5f);
    bool in2 = (0.5f < q) && (q <= 1.0f);
    float s = in1 ? s1 : s2;
    const Recip den = recip_prepare(c.h * r_norm);
    F3 o;
    float ox = div_shared(s * dx, den), oy = div_shared(s * dy, den), oz = div_shared(s * dz, den);   // s*d / (h*r_norm)
    bool in = in1 || in2;
    o.x = in ? ox : 0.0f;
    o.y = in ? oy : 0.0f;
    o.z = in ? oz : 0.0f;
    return o;
}

// The same two functions for a pair taken from a neighbour list and evaluated at the positions the list was built from: list
// membership is |x_ij| <= h exactly (r2_cut), so q = RN(r/h) <= 1 unless it is NaN, and NaN fails `1e-5 < q` as well: the gradient
// needs two compares instead of four.  (The PCISPH sweeps at PREDICTED positions must use the general forms; IISPH never moves a particle inside a step and uses these.)
__device__ __forceinline__ float cubic_w_in(const Consts &c, float r)
{
    float q = div_by_h(c, r);
    float q2 = q * q;
    float q3 = q2 * q;
    float w1 = c.kw * (6.0f * (q3 - q2) + 1.0f);
    float t = 1.0f - q;
    float w2 = 2.0f * c.kw * (t * (t * t));
    return q <= 0.5f ? w1 : (q <= 1.0f ? w2 : 0.0f);      // 0 <= q holds; the upper compare stays for q = NaN (a particle that blew up
                                                          // passes the reference's `norm > h` skip and must contribute 0, solver_base.py:84-87)
}
// The 1e-5 gate (:97) is applied to the SCALAR s, once, instead of to the three components: on gfx950 a v_cndmask_b32 whose mask
// does not come straight out of the preceding v_cmp costs ~23 cycles (the mask is fetched through the CU's scalar register port,
// tools/valu_issue.hip), so "one compare, three selects" was 55 of the body's 212 cycles.  With s = 0 the numerators are +-0 and so
// are the quotients; every consumer ADDS the components (times a finite factor) to an accumulator that starts at +0, and
// x + (+-0) == x bit for bit for every x but -0, which such a sum never is.  The divisor of a gated pair (r = 0: coincident
// particles) is raised to a finite value so that 0 / den stays 0; for any pair that passes the gate h * r >= 1e-7 and the max is
// the identity.
constexpr float kDenFloor = 1e-30f;
__device__ __forceinline__ F3 grad_w_in(const Consts &c, float dx, float dy, float dz, float r_norm)
{
    float q = div_by_h(c, r_norm);
    float q2 = q * q;
    float s1 = c.kg6 * (3.0f * q2 - 2.0f * q);
    float t = 1.0f - q;
    float s2 = c.neg_kg6 * (t * t);
    float s = q <= 0.5f ? s1 : s2;
    s = 1e-5f < q ? s : 0.0f;                             // :97 (q <= 1 holds for every list member)
    const Recip den = recip_prepare(__builtin_fmaxf(c.h * r_norm, kDenFloor));
    F3 o;
    o.x = div_shared(s * dx, den); o.y = div_shared(s * dy, den); o.z = div_shared(s * dz, den);
    return o;
}

// ti.max(a, b) as the oracle restates it: a > b ? a : b (keeps the sign-of-zero behaviour identical)
__device__ __forceinline__ float rmax(float a, float b) { return a > b ? a : b; }
// RN(sqrt(x)) for 0 <= x < 2^63 in 11 instructions (hipcc's generic correctly rounded expansion: 16 + hazard nops, for its range
// and class handling).  Scaling by 2^64 is exact, lifts denormal inputs into v_sqrt_f32's domain and cannot overflow below 2^63;
// the hardware root is within 1 ulp, so the exact residuals x - y'*y of the two neighbours y' = y -+ 1 ulp (one fma each) decide
// the rounding -- the same correction step the generic expansion uses; unscaling by 2^-32 is exact again.  x = 0 stays 0 (the
// neighbour below is a NaN pattern whose compare fails).  Squared distances between particles of a scene are far inside the range.
__device__ __forceinline__ float sqrt_rn(float x)
{
    const float xs = x * 0x1p64f;
    const float y = __builtin_amdgcn_sqrtf(xs);
    const float ym = __uint_as_float(__float_as_uint(y) - 1u), yp = __uint_as_float(__float_as_uint(y) + 1u);
    const float rm = __builtin_fmaf(-ym, y, xs), rp = __builtin_fmaf(-yp, y, xs);
    float r = 0.0f >= rm ? ym : y;
    r = 0.0f < rp ? yp : r;
    return r * 0x1p-32f;
}
__device__ __forceinline__ float norm3(float x, float y, float z) { return sqrt_rn((x * x + y * y) + z * z); }

// The staged DFSPH sweeps keep the neighbour positions in LDS multiplied by 2^32 (exact), and the particle's own position likewise:
// the difference vector then carries 2^32, its squared length 2^64 -- exactly what sqrt_rn multiplies in -- and the root 2^32.
// Scaling by a power of two commutes with every rounding involved (no overflow: |x| < 2^20 m; no underflow: it only moves values away
// from the denormal range), so the two multiplications of sqrt_rn disappear and nothing else changes:
//   norm3_scaled(d * 2^32)                     = norm3(d) * 2^32
//   grad_w_scaled(c, d * 2^32, r * 2^32)       = grad_w_in(c, d, r)         (q = r/h through h * 2^32; numerator and divisor of the
//                                                                            three divisions both carry 2^32)
__device__ __forceinline__ float norm3_scaled(float xs, float ys, float zs)
{
    const float x = (xs * xs + ys * ys) + zs * zs;                 // |d|^2 * 2^64
    const float y = __builtin_amdgcn_sqrtf(x);
    const float ym = __uint_as_float(__float_as_uint(y) - 1u), yp = __uint_as_float(__float_as_uint(y) + 1u);
    const float rm = __builtin_fmaf(-ym, y, x), rp = __builtin_fmaf(-yp, y, x);
    float r = 0.0f >= rm ? ym : y;
    r = 0.0f < rp ? yp : r;
    return r;
}
__device__ __forceinline__ F3 grad_w_scaled(const Consts &c, float dxs, float dys, float dzs, float rs)
{
    float q0 = rs * c.rh_s;                                        // div_by_h with both sides scaled
    float e = __builtin_fmaf(-q0, c.h_s, rs);
    float q = __builtin_fmaf(e, c.rh_s, q0);
    float q2 = q * q;
    float s1 = c.kg6 * (3.0f * q2 - 2.0f * q);
    float t = 1.0f - q;
    float s2 = c.neg_kg6 * (t * t);
    float s = q <= 0.5f ? s1 : s2;
    s = 1e-5f < q ? s : 0.0f;                                      // the gate on the scalar (see grad_w_in)
    const Recip den = recip_prepare(__builtin_fmaxf(c.h * rs, kDenFloor));   // (h * r) * 2^32
    F3 o;
    o.x = div_shared(s * dxs, den); o.y = div_shared(s * dys, den); o.z = div_shared(s * dzs, den);
    return o;
}
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
    return (ax * bx + ay * by) + az * bz;
}

// ---- kernel functions by arithmetic (round 4): the pcisph / iisph sweeps pick them through a template argument -----------------------------
// KF<false>: the reference's operations in its order (the functions above).  KF<true> (SphConfig.arith = SPH_ARITH_RELAXED): the distance from
// one v_rsq_f32, W as a clamped polynomial with FMAs (q > 1 gives 0: the pcisph sweeps evaluate list pairs at PREDICTED positions), grad W as
// one scalar times the difference vector with v_rcp_f32 in place of the three correctly rounded divisions, no 1e-5 gate (r^2 is floored, a
// coincident pair contributes 0).  Same signatures, so a sweep body is written once.
template <bool RX> struct KF;
template <> struct KF<false> {
    static __device__ __forceinline__ float norm3(float x, float y, float z) { return ::sph::norm3(x, y, z); }
    static __device__ __forceinline__ float w(const Consts &c, float r) { return cubic_w(c, r); }
    static __device__ __forceinline__ float w_in(const Consts &c, float r) { return cubic_w_in(c, r); }
    static __device__ __forceinline__ F3 grad(const Consts &c, float dx, float dy, float dz, float r) { return grad_w(c, dx, dy, dz, r); }
    static __device__ __forceinline__ F3 grad_in(const Consts &c, float dx, float dy, float dz, float r) { return grad_w_in(c, dx, dy, dz, r); }
};
template <> struct KF<true> {
    static __device__ __forceinline__ float norm3(float x, float y, float z)
    {
        const float r2 = __builtin_fmaf(z, z, __builtin_fmaf(y, y, __builtin_fmaf(x, x, 1e-30f)));
        return r2 * __builtin_amdgcn_rsqf(r2);
    }
    static __device__ __forceinline__ float w(const Consts &c, float r)
    {
        const float q = r * c.rh;
        const float t = rmax(1.0f - q, 0.0f);
        const float w1 = __builtin_fmaf(6.0f * (q * q), q - 1.0f, 1.0f);          // solver_base.py:76-88
        const float w2 = 2.0f * ((t * t) * t);
        return c.kw * (q <= 0.5f ? w1 : w2);
    }
    static __device__ __forceinline__ float w_in(const Consts &c, float r) { return w(c, r); }
    static __device__ __forceinline__ F3 grad(const Consts &c, float dx, float dy, float dz, float r)
    {
        const float q = r * c.rh;
        const float t = rmax(1.0f - q, 0.0f);
        const float s1 = (c.kg6 * q) * __builtin_fmaf(3.0f, q, -2.0f);            // solver_base.py:90-103 (with the reference's factor 6)
        const float s2 = c.neg_kg6 * (t * t);
        const float g = (q <= 0.5f ? s1 : s2) * __builtin_amdgcn_rcpf(__builtin_fmaxf(c.h * r, 1e-30f));
        F3 o;
        o.x = g * dx; o.y = g * dy; o.z = g * dz;
        return o;
    }
    static __device__ __forceinline__ F3 grad_in(const Consts &c, float dx, float dy, float dz, float r) { return grad(c, dx, dy, dz, r); }
};

// wcsph_solver.py:86-90, x**7 by squaring
__device__ __forceinline__ float tait_pressure(float rho)
{
    float rho_i = rmax(rho, 1000.0f);
    float a = rho_i / 1000.0f;
    float a2 = a * a;
    float r3 = a * a2;
    float a4 = a2 * a2;
    return 70000.0f * (r3 * a4 - 1.0f);
}

// ---- wave reductions and scans without LDS traffic or address registers (wave64) ------------------
// Cross-lane moves by data-path permutes instead of ds_bpermute (which needs a per-lane address and a round trip through the LDS
// crossbar per step): DPP modifiers inside a 16-lane row (quad_perm, row_ror, row_shr, row_bcast), ds_swizzle across the two rows
// of a 32-lane half, gfx950's v_permlane32_swap across the halves.
//   lane ^ 32   v_permlane32_swap_b32 (swaps lanes 32..63 of one operand with lanes 0..31 of the other)
//   lane ^ 16   ds_swizzle_b32 swizzle(SWAP,16)
//   lane ^  8   DPP row_ror:8
//   lane ^  4   ds_swizzle_b32 swizzle(SWAP,4)
//   lane ^  2   DPP quad_perm:[2,3,0,1]
//   lane ^  1   DPP quad_perm:[1,0,3,2]
// A reduction is the butterfly over 32, 16, 8, 4, 2, 1 IN THAT ORDER: for every lane i < off the partner i ^ off is i + off, so lane 0
// ends up with exactly the tree a `v += shfl_down(v, off)` ladder builds -- ((v0 + v32) + (v16 + v48)) + ... -- and the f64 block
// partials keep their bits whatever the mechanism (tests/test_parity_gpu.py::test_wave_primitives checks the tree against numpy).
template <int XOR>
__device__ __forceinline__ int lane_xor(int v)
{
    static_assert(XOR == 32 || XOR == 16 || XOR == 8 || XOR == 4 || XOR == 2 || XOR == 1, "butterfly step");
    if (XOR == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);     // r[0] = (lo, lo), r[1] = (hi, hi)
        return (threadIdx.x & 32) ? r[0] : r[1];
    }
    if (XOR == 16) return __builtin_amdgcn_ds_swizzle(v, 0x401F);                // bit mode: and 0x1f, or 0, xor 0x10
    if (XOR == 8) return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);   // row_ror:8
    if (XOR == 4) return __builtin_amdgcn_ds_swizzle(v, 0x101F);                 // xor 0x04
    if (XOR == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);    // quad_perm:[2,3,0,1]
    return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);                  // quad_perm:[1,0,3,2]
}
template <int XOR>
__device__ __forceinline__ float lane_xor(float v) { return __int_as_float(lane_xor<XOR>(__float_as_int(v))); }
template <int XOR>
__device__ __forceinline__ double lane_xor(double v)
{
    const int lo = lane_xor<XOR>(__double2loint(v)), hi = lane_xor<XOR>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
#define SPH_BUTTERFLY(v, OP) \
    v = OP(v, lane_xor<32>(v)); v = OP(v, lane_xor<16>(v)); v = OP(v, lane_xor<8>(v)); \
    v = OP(v, lane_xor<4>(v)); v = OP(v, lane_xor<2>(v)); v = OP(v, lane_xor<1>(v));
#define SPH_OP_ADD(a, b) ((a) + (b))
// result in lane 0 (every lane holds a sum of all 64 values; only lane 0's association is the documented one)
__device__ __forceinline__ double wave_sum(double v) { SPH_BUTTERFLY(v, SPH_OP_ADD) return v; }
__device__ __forceinline__ int wave_sum(int v) { SPH_BUTTERFLY(v, SPH_OP_ADD) return v; }
__device__ __forceinline__ float wave_max(float v) { SPH_BUTTERFLY(v, fmaxf) return v; }
__device__ __forceinline__ int wave_max(int v) { SPH_BUTTERFLY(v, max) return v; }
#undef SPH_BUTTERFLY
#undef SPH_OP_ADD

// inclusive prefix sum over the 64 lanes: row_shr:1,2,4,8 inside each row of 16 (lanes without a source add 0), then lane 15 of
// rows 0 and 2 broadcast into rows 1 and 3 (row_bcast:15), then lane 31 into the upper half (row_bcast:31)
__device__ __forceinline__ int wave_inclusive_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}

}  // namespace sph
