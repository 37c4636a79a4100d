// sph_rigid_kernels.h -- device side of the rigid body of config 5: per-step cell sort of the rigid sample particles,
// the fluid -> rigid force accumulation (dfsph_solver.py:212), and the particle-parallel parts of rigid_solver.step
// (rigid_solver.py:33-141).  The 3x3 algebra between them (inertia, impulse) is a handful of flops and runs on the host.
//
// Reductions are single-workgroup, fixed-order, f64 -> deterministic (the reference uses f32 atomics in thread order).
#pragma once
#include "sph_kernels.h"

namespace sph {

// canonical order inside a cell = ascending rigid index (update_grid_rigid_particles, ParticleSystem.py:399-407)
__global__ __launch_bounds__(kBlock) void k_rigid_order(int nr, const int *__restrict__ cell_of, const int *__restrict__ cell_start,
                                                        const int *__restrict__ slot_src, const float4 *__restrict__ RPin,
                                                        float4 *__restrict__ RPout, int *__restrict__ rid)
{
    int d = blockIdx.x * kBlock + threadIdx.x;
    if (d >= nr) return;
    int src = slot_src[d];
    int cell = cell_of[src];
    int a = cell_start[cell], b = cell_start[cell + 1];
    int r = 0;
    for (int e = a; e < b; ++e) r += (slot_src[e] < src) ? 1 : 0;
    RPout[a + r] = RPin[src];
    rid[a + r] = src;
}

// Fluid neighbours of every rigid sample particle, once per step (fluid and body are frozen between the grid rebuild and the
// integrators): the reference's walk order -- cells dx-outermost, ascending fluid index inside a cell (for_all_neighbor,
// ParticleSystem.py:447-469) -- in the wave-tiled layout of the fluid lists.  The force kernels below run once per solver
// iteration (100 times per step at config 5) and used to repeat the 27-cell walk every time.
__global__ __launch_bounds__(kBlock) void k_build_rnl(Consts c, int nr, const float4 *__restrict__ RP, const float4 *__restrict__ P,
                                                      const int *__restrict__ cell_start, uint32_t *__restrict__ rnl, int *__restrict__ rcnt,
                                                      DevScalars *__restrict__ ds)
{
    __shared__ uint32_t s_stage[4 * kBlock];
    const int r = blockIdx.x * kBlock + threadIdx.x;
#pragma unroll
    for (int q = 0; q < 4; ++q) s_stage[q * kBlock + threadIdx.x] = 0;
    if (r >= nr) return;
    const float4 pr = RP[r];
    const f32x2 pr_xy = {pr.x, pr.y};
    int cx, cy, cz;
    cell_id_of(c, pr.x, pr.y, pr.z, cx, cy, cz);
    NlWriter w{&s_stage[threadIdx.x], rnl + nl_index(r, 0, c.kpitch), 0, c.kmax};
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dz = -1; dz <= 1; ++dz) {
                const int x = cx + dx, y = cy + dy, z = cz + dz;
                if (x >= c.gx || y >= c.gy || z >= c.gz) continue;
                if (x < 0 || y < 0 || z < 0) continue;
                const int slot = cell_slot_xyz(c, x, y, z, x + y * c.sy + z * c.sz);
                if (slot < 0) continue;
                const int a = cell_start[slot], b = cell_start[slot + 1];
                for (int j0 = a; j0 < b; j0 += 4) {
                    const float4 *pb = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(P) + (unsigned)j0 * 16u);
                    unsigned m = near_mask4(pr_xy, pr.z, pb, c.r2_cut);        // (x_i - x_r)^2 == (x_r - x_i)^2 term by term
                    m &= (b - j0 >= 4 ? 15u : (1u << (b - j0)) - 1u);
                    while (m) {
                        const int u = __ffs(m) - 1;
                        m &= m - 1;
                        w.push((uint32_t)(j0 + u));
                    }
                }
            }
    w.flush();
    rcnt[r] = w.k < c.kmax ? w.k : c.kmax;
    if (w.k > c.kmax) atomicOr(&ds->overflow, 1);
}

// rigid_particles[j].force += ret * particle_m (dfsph_solver.py:204-212), gathered per rigid particle over its fluid
// neighbours in list (= cell-walk) order: no atomics, and the same serialisation as the oracle.
__global__ __launch_bounds__(kBlock) void k_rigid_force(Consts c, int nr, const float4 *__restrict__ RP, const int *__restrict__ rid,
                                                        const float4 *__restrict__ P, const uint32_t *__restrict__ rnl,
                                                        const int *__restrict__ rcnt, const float *__restrict__ rho,
                                                        const float *__restrict__ rho_adv, const float *__restrict__ alpha,
                                                        const DevScalars *__restrict__ ds, float *__restrict__ force, int gate,
                                                        int col_lo = -0x7fffffff, int col_hi = 0x7fffffff)
{
    if (gate_closed(ds, gate)) return;
    int r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= nr) return;
    const float4 pr = RP[r];
    // slab handles: the body is replicated on every rank and a sample's force is summed WHOLE by the rank that owns the sample's cell column
    // (all its fluid neighbours are resident there: owned particles and inner ghosts, in the canonical order); the per-sample forces of all ranks
    // are added up before rigid_solver.step (x + 0 = x: exact)
    { const int cx = (int)floorf(pr.x / c.hcell); if (cx < col_lo || cx >= col_hi) return; }
    const float dt2 = ds->dt2;
    float fx = 0.f, fy = 0.f, fz = 0.f;
    struct Op { float4 p; float rho, rho_adv, alpha; };
    walk_list<Op>(rnl + nl_index(r, 0, c.kpitch), rcnt[r], [&](uint32_t i, Op &o) {
        o.p = P[i]; o.rho = rho[i]; o.rho_adv = rho_adv[i]; o.alpha = alpha[i];
    }, [&](const Op &o, uint32_t) {
        const float4 pi = o.p;
        float ddx = pi.x - pr.x, ddy = pi.y - pr.y, ddz = pi.z - pr.z;
        float r2 = (ddx * ddx + ddy * ddy) + ddz * ddz;
        float rn = sqrtf(r2);
        float k_i = (o.rho_adv - c.rho0) * o.alpha / dt2;                           // :208
        F3 g = grad_w(c, ddx, ddy, ddz, rn);
        float s = pr.w * c.rho0 * k_i / o.rho;                                      // :211
        fx += s * g.x * c.m; fy += s * g.y * c.m; fz += s * g.z * c.m;              // :212
    });
    const int o = rid[r];
    force[3 * o] += fx; force[3 * o + 1] += fy; force[3 * o + 2] += fz;
}

struct RigidBodyState {
    float c[3];        // rigid_centriod
    float omega[3];    // rigid_solver.omega
    float vel[3];      // candidate velocity of the step (rigid_solver.py:43)
    float ori[3];      // ori_displacement (:46)
    float lo[3], hi[3];   // box_min + d, box_max - d (:56, :65)
};

struct RigidReduce {   // outputs of the single-workgroup reductions
    double torque[3], force[3];
    double cp[3];      // sum of colliding particle positions (:75)
    float dmax[3], dmin[3];
    int cnorm[3];
    int ccount;
    float vmax;        // max |omega x (x - c)|, dfsph_solver.py:110
    int pad;
};

template <class T>
__device__ __forceinline__ T block_sum_fixed(T v, T *sh)
{
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    T r = sh[0];
    __syncthreads();
    return r;
}

// The reductions over the body's sample particles run on kRigidParts workgroups (a body of config 5 has 123 k samples: one workgroup
// walking all of them took 150-250 us per kernel, three kernels per step); workgroup b leaves its partial in out[b] and the host, which
// reads the result back anyway, combines the partials in index order (sums in f64, maxima, flags): fixed order, no atomics.
constexpr int kRigidParts = 64;

// torque = sum (x - c) x F, force = sum F   (compute_attitude :118-123, kinematic :35-38)
__global__ __launch_bounds__(kBlock) void k_rigid_torque_force(int nr, const float4 *__restrict__ RPos, const float *__restrict__ force,
                                                               RigidBodyState st, RigidReduce *__restrict__ out)
{
    __shared__ double sh[kBlock];
    double t[3] = {0, 0, 0}, f[3] = {0, 0, 0};
    out += blockIdx.x;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < nr; i += gridDim.x * kBlock) {
        const float4 p = RPos[i];
        float rx = p.x - st.c[0], ry = p.y - st.c[1], rz = p.z - st.c[2];
        float fx = force[3 * i], fy = force[3 * i + 1], fz = force[3 * i + 2];
        t[0] += (double)(ry * fz - rz * fy);
        t[1] += (double)(rz * fx - rx * fz);
        t[2] += (double)(rx * fy - ry * fx);
        f[0] += (double)fx; f[1] += (double)fy; f[2] += (double)fz;
    }
    for (int a = 0; a < 3; ++a) {
        double s = block_sum_fixed(t[a], sh);
        if (threadIdx.x == 0) out->torque[a] = s;
        s = block_sum_fixed(f[a], sh);
        if (threadIdx.x == 0) out->force[a] = s;
    }
}

// rotation about the centroid (rotation :130-139) for particles (float4, .w kept) or vertices (packed xyz)
struct Mat3 {
    float m[9];
};

__global__ __launch_bounds__(kBlock) void k_rigid_rotate(int n, float4 *__restrict__ p4, float *__restrict__ p3, Mat3 R, RigidBodyState st)
{
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float x, y, z, w = 0.f;
    if (p4) { float4 p = p4[i]; x = p.x; y = p.y; z = p.z; w = p.w; }
    else { x = p3[3 * i]; y = p3[3 * i + 1]; z = p3[3 * i + 2]; }
    float rx = x - st.c[0], ry = y - st.c[1], rz = z - st.c[2];
    float ox = ((R.m[0] * rx + R.m[1] * ry) + R.m[2] * rz) + st.c[0];
    float oy = ((R.m[3] * rx + R.m[4] * ry) + R.m[5] * rz) + st.c[1];
    float oz = ((R.m[6] * rx + R.m[7] * ry) + R.m[8] * rz) + st.c[2];
    if (p4) p4[i] = make_float4(ox, oy, oz, w);
    else { p3[3 * i] = ox; p3[3 * i + 1] = oy; p3[3 * i + 2] = oz; }
}

// wall test of kinematic (:53-76): extreme displacements per axis, collision normals, colliding-point sum (partials per workgroup, see above;
// cnorm of a partial: bit 0 = a lower-wall hit, bit 1 = an upper-wall hit on that axis)
__global__ __launch_bounds__(kBlock) void k_rigid_collide(int nr, const float4 *__restrict__ RPos, RigidBodyState st,
                                                          RigidReduce *__restrict__ out)
{
    out += blockIdx.x;
    __shared__ double shd[kBlock];
    __shared__ float shf[kBlock];
    __shared__ int shi[kBlock];
    float dmax[3] = {-INFINITY, -INFINITY, -INFINITY}, dmin[3] = {INFINITY, INFINITY, INFINITY};
    int lo_hit[3] = {0, 0, 0}, hi_hit[3] = {0, 0, 0};
    double cp[3] = {0, 0, 0};
    int cc = 0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < nr; i += gridDim.x * kBlock) {
        const float4 p4 = RPos[i];
        const float p[3] = {p4.x, p4.y, p4.z};
        float rel[3] = {p[0] + st.ori[0] - st.c[0], p[1] + st.ori[1] - st.c[1], p[2] + st.ori[2] - st.c[2]};
        float wr[3] = {st.omega[1] * rel[2] - st.omega[2] * rel[1], st.omega[2] * rel[0] - st.omega[0] * rel[2],
                       st.omega[0] * rel[1] - st.omega[1] * rel[0]};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            int collision = 0;
            if (p[j] + st.ori[j] <= st.lo[j]) {
                dmax[j] = fmaxf(dmax[j], st.lo[j] - p[j]);                       // :58
                if (st.vel[j] + wr[j] < 0.f) { collision = 1; lo_hit[j] = 1; }
            }
            if (p[j] + st.ori[j] >= st.hi[j]) {
                dmin[j] = fminf(dmin[j], st.hi[j] - p[j]);                       // :67
                if (st.vel[j] + wr[j] > 0.f) { collision = 1; hi_hit[j] = 1; }
            }
            if (collision) { cp[0] += (double)p[0]; cp[1] += (double)p[1]; cp[2] += (double)p[2]; cc += 1; }   // :74-76
        }
    }
    for (int a = 0; a < 3; ++a) {
        shf[threadIdx.x] = dmax[a];
        __syncthreads();
        for (int off = kBlock / 2; off > 0; off >>= 1) {
            if (threadIdx.x < off) shf[threadIdx.x] = fmaxf(shf[threadIdx.x], shf[threadIdx.x + off]);
            __syncthreads();
        }
        if (threadIdx.x == 0) out->dmax[a] = shf[0];
        __syncthreads();
        shf[threadIdx.x] = dmin[a];
        __syncthreads();
        for (int off = kBlock / 2; off > 0; off >>= 1) {
            if (threadIdx.x < off) shf[threadIdx.x] = fminf(shf[threadIdx.x], shf[threadIdx.x + off]);
            __syncthreads();
        }
        if (threadIdx.x == 0) out->dmin[a] = shf[0];
        __syncthreads();
        // collision_norm[j]: -1 from the lower wall, +1 from the upper wall; if both fire in one step the later write wins
        // in the reference (a race); here the upper wall wins, as in the oracle's particle loop order per axis
        int lh = block_sum_fixed(lo_hit[a], shi), hh = block_sum_fixed(hi_hit[a], shi);
        if (threadIdx.x == 0) out->cnorm[a] = (lh > 0 ? 1 : 0) | (hh > 0 ? 2 : 0);
        double s = block_sum_fixed(cp[a], shd);
        if (threadIdx.x == 0) out->cp[a] = s;
    }
    int n = block_sum_fixed(cc, shi);
    if (threadIdx.x == 0) out->ccount = n;
}

// translation (:98-104) and force reset (:38)
__global__ __launch_bounds__(kBlock) void k_rigid_translate(int n, float4 *__restrict__ p4, float *__restrict__ p3, float dx, float dy,
                                                            float dz, float *__restrict__ force_to_zero)
{
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    if (p4) { float4 p = p4[i]; p.x += dx; p.y += dy; p.z += dz; p4[i] = p; }
    else { p3[3 * i] += dx; p3[3 * i + 1] += dy; p3[3 * i + 2] += dz; }
    if (force_to_zero) { force_to_zero[3 * i] = 0.f; force_to_zero[3 * i + 1] = 0.f; force_to_zero[3 * i + 2] = 0.f; }
}

// max_rigid_vel = max_i ( |vel| + |omega x (x_i - c)| )         dfsph_solver.py:104-110: partial maxima per workgroup in part[], the
// second launch (one workgroup, nparts > 0) takes their maximum into ds->rigid_vmax
__global__ __launch_bounds__(kBlock) void k_rigid_vmax(int nr, const float4 *__restrict__ RPos, RigidBodyState st, float vel_norm,
                                                       DevScalars *__restrict__ ds, float *__restrict__ part, int nparts)
{
    __shared__ float shf[kBlock];
    float m = 0.0f;
    if (nparts > 0) {                      // second stage
        for (int i = threadIdx.x; i < nparts; i += kBlock) m = fmaxf(m, part[i]);
        nr = 0;
    }
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < nr; i += gridDim.x * kBlock) {
        const float4 p = RPos[i];
        float rx = p.x - st.c[0], ry = p.y - st.c[1], rz = p.z - st.c[2];
        float cx = st.omega[1] * rz - st.omega[2] * ry, cy = st.omega[2] * rx - st.omega[0] * rz, cz = st.omega[0] * ry - st.omega[1] * rx;
        m = fmaxf(m, vel_norm + sqrtf((cx * cx + cy * cy) + cz * cz));
    }
    shf[threadIdx.x] = m;
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) shf[threadIdx.x] = fmaxf(shf[threadIdx.x], shf[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (nparts > 0) ds->rigid_vmax = shf[0];
        else part[blockIdx.x] = shf[0];
    }
}

}  // namespace sph
