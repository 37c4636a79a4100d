/*
 * sph_oracle.c -- CPU ORACLE (test infrastructure, see sph_oracle.h).  PARITY UNPINNED.
 *
 * Plain-C restatement of the reference's per-step SPH path, sweep by sweep, in the
 * reference's own data flow: per-cell lists rebuilt every step, one 27-cell walk per
 * sweep, separate accumulators per sweep, host-driven DFSPH loops.  Every function cites
 * the reference file:line it restates (paths are relative to /root/reference).
 *
 * Arithmetic rules ([taichi-semantics]):
 *   - fields and kernel locals are f32 (`real`), ints are i32;
 *   - expressions made only of Python scalars are folded in f64, then rounded to f32;
 *   - x ** n with integer n is exponentiation by squaring (r=1; while n: if n&1: r*=a; a*=a; n>>=1);
 *   - vec.norm() = sqrt((x*x + y*y) + z*z); a.dot(b) = (ax*bx + ay*by) + az*bz;
 *   - a % b on floats = a - b*floor(a/b);
 *   - no FMA contraction, no re-association (build with -ffp-contract=off, no -ffast-math);
 *   - cell-list order = single-thread append order: ascending particle index per cell.
 * Deviations, all outside valid runs: a particle whose linear cell id is <0 or >=C is left
 * out of the grid (reference: `> C`, ParticleSystem.py:393, id==C would write out of bounds)
 * and keeps using the cell of its current position as walk centre (reference: stale
 * belong_grid).  Residual means are accumulated in f64 in particle order (reference: f32
 * atomics in nondeterministic order, dfsph_solver.py:140-141,276-277).
 */
#include "sph_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef ORC_REAL
#define ORC_REAL float
#endif
typedef ORC_REAL real;

#define R(x) ((real)(x))

static inline real r_sqrt(real x) { return sizeof(real) == 4 ? (real)sqrtf((float)x) : (real)sqrt((double)x); }
static inline real r_floor(real x) { return sizeof(real) == 4 ? (real)floorf((float)x) : (real)floor((double)x); }
static inline real r_max(real a, real b) { return a > b ? a : b; }
static inline real r_abs(real a) { return a < 0 ? -a : a; }

struct Orc {
    OrcConfig cfg;
    int N, Nb, Nr;
    int g[3], C;
    int stride[3];         /* _3d_to_1d_tran = (1, gx*gz, gx)  ParticleSystem.py:102 */
    /* constants, each rounded to f32 exactly where the reference would */
    real h;                /* support_radius / kernel_h = 4r */
    real m;                /* particle_m */
    real d;                /* particle_diameter */
    real rho0;             /* 1000 */
    real gravity;
    real visc_num;         /* 2*alpha*h*c_s folded in f64  solver_base.py:187 */
    real visc_eps_h2;      /* eps*h*h folded in f64        solver_base.py:188 */
    real tens_c;           /* -k/m*m folded in f64         solver_base.py:216 */
    real neg_m;            /* -particle_m                  solver_base.py:189 */
    real dt, dt2, ps_dt;   /* delta_time, delta_time_2, ps.delta_time (0-d f32 fields) */
    real dt_cfl_num;       /* 0.4*r*2 folded in f64        dfsph_solver.py:112 */
    /* the attributes a caller may edit on the solver object (orc_set_scalar 64..76; the numbering of include/sph_mi355x.h SPH_P_*):
     * dfsph_solver.py:21-29, solver_base.py:23-26, wcsph_solver.py:17-20.  Python scalars: kept as doubles, folded where the reference folds them */
    double p_density_threshold, p_density_divergence_threshold, p_max_dt, p_min_dt;
    int p_min_iteration_density, p_min_iteration_density_divergence, p_max_iteration_density_divergence, p_warm_start, p_adaptive_dt;
    double p_viscosity_c_s, p_viscosity_alpha, p_viscosity_epsilon, p_tension_k;
    int simulate_cnt;
    int nt;
    /* fluid particles */
    real *pos, *vel, *acc;
    int *cell3;            /* belong_grid */
    /* wall particles */
    real *bpos, *bvol;
    int *bcell3;
    /* cell lists: grids (fluid[+rigid] global indices) and boundary_grids (wall local indices) */
    int *cstart, *citems;
    int *bcstart, *bcitems;
    /* solver fields */
    real *rho, *pressure, *pgrad, *bacc, *visc, *tens;
    real *alpha, *rho_adv, *rho_der, *vel_adv, *vel_adv_delta, *force_ext, *warm_k;
    int *nbr_cnt;
    long lost;
    /* PCISPH (pcisph_solver.py:8-26) and IISPH (iisph_solver.py:10-29) fields */
    real *pos_predict, *vel_predict, *press_force, *rho_err, *press_iter;   /* press_iter doubles as IISPH p_iter, press_force as f_press */
    real *d_ii, *d_ij, *a_ii, *p_past, *p_new, *r_sum;
    real pci_delta, pci_beta;
    int pci_max_index, pci_max_count;
    /* PBF (pbf_solver.py:12-16): constrain, constrain_derivative, pbf_lambda, delta_pos (pos_predict is shared with PCISPH) */
    real *pbf_c, *pbf_cd, *pbf_lambda, *pbf_dpos;
    /* rigid body (config 5): particles sampled from the mesh, ParticleSystem.py:41-64 */
    int Nv;                /* mesh vertices */
    int exist_rigid, active_rigid;
    real rigid_rho;
    real *rpos, *rvol, *rmass, *rforce, *rvert;
    int *rcell3;
    real centroid[3], inertia_inv[9];
    real r_vel[3], r_acc[3], r_omega[3], r_alpha[3];   /* rigid_particles.vel/acc/omega/alpha: always filled uniformly */
    /* rigid_solver state, rigid_solver.py:6-31 */
    real rs_dt, rs_omega[3], rs_attitude[3], rs_mass;
    int rs_run_once, rs_cnt;
    /* "legal schedule" mode (orc_set_schedule; 0 = the canonical single-thread order every parity test uses).  The reference appends to the
     * cell lists from a parallel loop (ParticleSystem.py:388-397: the order inside a cell -- and with it the order of every neighbour sum --
     * is whatever the thread schedule produced) and reduces the residual means with f32 atomics (dfsph_solver.py:139-141, 275-279).  With a
     * seed the restatement draws ONE legal execution: every cell's entries in a seeded random order, redrawn at every rebuild, and the means
     * accumulated in f32: `sched_chunk` consecutive particles summed in order (a thread-local partial), the partials added to the total in a
     * seeded random order (chunk = 1: one atomic per particle).  tools/envelope.py measures how far such executions drift apart. */
    unsigned long long sched_seed;
    int sched_chunk;
    unsigned long long sched_epoch;
};

static inline unsigned long long sched_mix(unsigned long long x)      /* splitmix64 */
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
static void sched_shuffle(int *a, int n, unsigned long long key)
{
    for (int k = n - 1; k > 0; --k) {
        key = sched_mix(key);
        int r = (int)(key % (unsigned long long)(k + 1));
        int t = a[k]; a[k] = a[r]; a[r] = t;
    }
}
/* sum of v[i] over the particles with take[i] != 0, the way the reference's `avg += x` inside a parallel loop may come out */
static double sched_sum(Orc *o, const real *v, const unsigned char *take, int n)
{
    if (o->sched_seed == 0) {                 /* canonical: ascending, f64 (see the header: a documented deviation) */
        double s = 0;
        for (int i = 0; i < n; ++i) if (take[i]) s += (double)v[i];
        return s;
    }
    const int chunk = o->sched_chunk > 0 ? o->sched_chunk : 1;
    const int nch = (n + chunk - 1) / chunk;
    float *part = (float *)malloc(sizeof(float) * (size_t)(nch > 0 ? nch : 1));
    int *ord = (int *)malloc(sizeof(int) * (size_t)(nch > 0 ? nch : 1));
    for (int c = 0; c < nch; ++c) {
        float ps = 0.0f;
        const int hi = (c + 1) * chunk < n ? (c + 1) * chunk : n;
        for (int i = c * chunk; i < hi; ++i) if (take[i]) ps += (float)v[i];
        part[c] = ps; ord[c] = c;
    }
    sched_shuffle(ord, nch, sched_mix(o->sched_seed ^ (0xA5A5ULL + (o->sched_epoch++ << 20))));
    float tot = 0.0f;
    for (int c = 0; c < nch; ++c) tot += part[ord[c]];
    free(part); free(ord);
    return (double)tot;
}
static void fold_params(Orc *o);

void orc_set_schedule(Orc *o, unsigned long long seed, int chunk)
{
    o->sched_seed = seed;
    o->sched_chunk = chunk;
    o->sched_epoch = 0;
    if (seed != 0 && o->Nb > 0)      /* the wall lists are filled once, by the same kind of parallel loop (ParticleSystem.py:372-381) */
        for (int c = 0; c < o->C; ++c)
            sched_shuffle(o->bcitems + o->bcstart[c], o->bcstart[c + 1] - o->bcstart[c], sched_mix(seed ^ 0xB0B0ULL ^ ((unsigned long long)c << 24)));
}

/* ---------------------------------------------------------------------------------------
 * SPH kernels                                                         solver_base.py:74-103
 * ------------------------------------------------------------------------------------- */
static inline real pow3(real a) { return a * (a * a); }           /* r=a; a2=a*a; r=r*a2 */
static inline real pow2(real a) { return a * a; }
static inline real pow7(real a) { real a2 = a * a; real r3 = a * a2; real a4 = a2 * a2; return r3 * a4; }

static const double ORC_PI = 3.141592653589793;

/* solver_base.py:76-88 */
static inline real cubic_kernel(real r, real h)
{
    real ret = 0;
    real q = r / h;
    real k = R(8) / (R(ORC_PI) * pow3(h));
    if (R(0) <= q && q <= R(0.5)) {
        real q2 = q * q;
        real q3 = q2 * q;
        ret = k * (R(6) * (q3 - q2) + R(1));
    } else if (R(0.5) < q && q <= R(1)) {
        ret = R(2) * k * pow3(R(1) - q);
    } else {
        ret = 0;
    }
    return ret;
}

/* solver_base.py:90-103 (keeps the reference's extra factor 6) */
static inline void cubic_kernel_derivative(real rx, real ry, real rz, real h, real out[3])
{
    real r_norm = r_sqrt((rx * rx + ry * ry) + rz * rz);
    real q = r_norm / h;
    real k = R(48) / (R(ORC_PI) * pow3(h));
    out[0] = out[1] = out[2] = 0;
    if (R(1e-5) < q && q <= R(0.5)) {
        real q2 = q * q;
        real s = k * R(6) * (R(3) * q2 - R(2) * q);
        real den = h * r_norm;
        out[0] = s * rx / den;
        out[1] = s * ry / den;
        out[2] = s * rz / den;
    } else if (R(0.5) < q && q <= R(1)) {
        real s = -k * R(6) * pow2(R(1) - q);
        real den = h * r_norm;
        out[0] = s * rx / den;
        out[1] = s * ry / den;
        out[2] = s * rz / den;
    }
}

float orc_cubic_kernel(float r, float h) { return (float)cubic_kernel(R(r), R(h)); }
void orc_cubic_kernel_derivative(const float r[3], float h, float out[3])
{
    real o[3];
    cubic_kernel_derivative(R(r[0]), R(r[1]), R(r[2]), R(h), o);
    out[0] = (float)o[0]; out[1] = (float)o[1]; out[2] = (float)o[2];
}
/* wcsph_solver.py:86-90  (B=70000, gamma=7, rho_0=1000) */
static inline real tait_pressure(real rho)
{
    real rho_i = r_max(rho, R(1000));
    return R(70000) * (pow7(rho_i / R(1000)) - R(1.0));
}
float orc_tait_pressure(float rho) { return (float)tait_pressure(R(rho)); }

/* ---------------------------------------------------------------------------------------
 * grid helpers                                                    ParticleSystem.py:486-494
 * ------------------------------------------------------------------------------------- */
static inline void cell_of(const Orc *o, const real *p, int c[3])
{
    c[0] = (int)r_floor(p[0] / o->h);
    c[1] = (int)r_floor(p[1] / o->h);
    c[2] = (int)r_floor(p[2] / o->h);
}
static inline int cell_1d(const Orc *o, const int c[3])
{
    return c[0] * o->stride[0] + c[1] * o->stride[1] + c[2] * o->stride[2];
}
static inline int cell_valid(const Orc *o, int cx, int cy, int cz)
{
    /* ParticleSystem.py:453-456 */
    if (cx >= o->g[0] || cy >= o->g[1] || cz >= o->g[2]) return 0;
    if (cx < 0 || cy < 0 || cz < 0) return 0;
    return 1;
}

/* counting sort by cell id, ascending index inside a cell (= single-thread append order) */
static long build_lists(const Orc *o, int n, const real *pos, int *cell3, int *cstart, int *citems)
{
    int C = o->C;
    long lost = 0;
    memset(cstart, 0, sizeof(int) * (size_t)(C + 1));
    int *ids = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) {
        int c[3];
        cell_of(o, pos + 3 * i, c);
        cell3[3 * i + 0] = c[0]; cell3[3 * i + 1] = c[1]; cell3[3 * i + 2] = c[2];
        int id = cell_1d(o, c);
        if (id < 0 || id >= C) { ids[i] = -1; ++lost; continue; }   /* ParticleSystem.py:393-395 */
        ids[i] = id;
        cstart[id + 1]++;
    }
    for (int c = 0; c < C; ++c) cstart[c + 1] += cstart[c];
    int *fill = (int *)malloc(sizeof(int) * (size_t)C);
    memcpy(fill, cstart, sizeof(int) * (size_t)C);
    for (int i = 0; i < n; ++i)
        if (ids[i] >= 0) citems[fill[ids[i]]++] = i;
    free(fill);
    free(ids);
    return lost;
}
/* legal-schedule mode: every cell's segment [lo, hi) of a list in a seeded random order */
static void sched_shuffle_cells(Orc *o, int *citems, const int *cstart, int first_is_fluid_only)
{
    const unsigned long long ep = o->sched_epoch++;
    _Pragma("omp parallel for schedule(static) num_threads(o->nt)")
    for (int c = 0; c < o->C; ++c) {
        int lo = cstart[c], hi = cstart[c + 1];
        if (hi - lo < 2) continue;
        if (first_is_fluid_only) {            /* fluid entries were appended by one kernel, rigid entries by the next: two segments */
            int mid = lo;
            while (mid < hi && citems[mid] < o->N) ++mid;
            sched_shuffle(citems + lo, mid - lo, sched_mix(o->sched_seed ^ (ep << 32) ^ (unsigned long long)c));
            sched_shuffle(citems + mid, hi - mid, sched_mix(o->sched_seed ^ (ep << 32) ^ (unsigned long long)c ^ 0x7777ULL));
        } else {
            sched_shuffle(citems + lo, hi - lo, sched_mix(o->sched_seed ^ (ep << 32) ^ (unsigned long long)c));
        }
    }
}

/* reset_grid + update_grid                               ParticleSystem.py:368-397 */
void orc_build_grid(Orc *o)
{
    if (!(o->exist_rigid && o->active_rigid)) {
        o->lost = build_lists(o, o->N, o->pos, o->cell3, o->cstart, o->citems);
        if (o->sched_seed) sched_shuffle_cells(o, o->citems, o->cstart, 0);
        return;
    }
    /* fluid entries first (ascending), then rigid entries (ascending, global index i + N + Nb): the order of
     * update_grid_fluid_particles followed by update_grid_rigid_particles */
    int C = o->C, N = o->N, Nr = o->Nr;
    long lost = 0;
    memset(o->cstart, 0, sizeof(int) * (size_t)(C + 1));
    int *ids = (int *)malloc(sizeof(int) * (size_t)(N + Nr + 1));
    for (int i = 0; i < N + Nr; ++i) {
        int c[3];
        const real *p = i < N ? o->pos + 3 * i : o->rpos + 3 * (i - N);
        int *c3 = i < N ? o->cell3 + 3 * i : o->rcell3 + 3 * (i - N);
        cell_of(o, p, c);
        c3[0] = c[0]; c3[1] = c[1]; c3[2] = c[2];
        int id = cell_1d(o, c);
        if (id < 0 || id >= C) { ids[i] = -1; ++lost; continue; }
        ids[i] = id;
        o->cstart[id + 1]++;
    }
    for (int c = 0; c < C; ++c) o->cstart[c + 1] += o->cstart[c];
    int *fill = (int *)malloc(sizeof(int) * (size_t)C);
    memcpy(fill, o->cstart, sizeof(int) * (size_t)C);
    for (int i = 0; i < N + Nr; ++i)
        if (ids[i] >= 0) o->citems[fill[ids[i]]++] = i < N ? i : i + o->Nb;
    free(fill);
    free(ids);
    o->lost = lost;
    if (o->sched_seed) sched_shuffle_cells(o, o->citems, o->cstart, 1);
}

/* ---------------------------------------------------------------------------------------
 * neighbour iteration macros.  They expand to the 27-cell walk of
 * for_all_neighbor (ParticleSystem.py:447-469) / for_all_boundary_neighbor (:337-366):
 * dx outermost, dz innermost, skip invalid cells, skip self, skip if |x_ij| > h.
 * Inside BODY: j = neighbour index, (xij,yij,zij) = x_i - x_j.
 * ------------------------------------------------------------------------------------- */
#define FOR_FLUID_NEIGHBORS(o, i, ...)                                                         \
    do {                                                                                       \
        const real pix_ = (o)->pos[3 * (i)], piy_ = (o)->pos[3 * (i) + 1], piz_ = (o)->pos[3 * (i) + 2]; \
        const int *cc_ = (o)->cell3 + 3 * (i);                                                 \
        for (int dx_ = -1; dx_ <= 1; ++dx_)                                                    \
            for (int dy_ = -1; dy_ <= 1; ++dy_)                                                \
                for (int dz_ = -1; dz_ <= 1; ++dz_) {                                          \
                    int cx_ = cc_[0] + dx_, cy_ = cc_[1] + dy_, cz_ = cc_[2] + dz_;            \
                    if (!cell_valid((o), cx_, cy_, cz_)) continue;                             \
                    int c1_ = cx_ * (o)->stride[0] + cy_ * (o)->stride[1] + cz_ * (o)->stride[2]; \
                    for (int e_ = (o)->cstart[c1_]; e_ < (o)->cstart[c1_ + 1]; ++e_) {         \
                        int j = (o)->citems[e_];                                               \
                        if (j == (i)) continue;                                                \
                        /* get_particle(): fluid [0,N), rigid [N+Nb, ...)  ParticleSystem.py:496-507 */ \
                        const int jm_ = j < (o)->N ? 0 : 2;                                    \
                        const int jl = jm_ == 0 ? j : j - (o)->N - (o)->Nb;                    \
                        const real *pj_ = jm_ == 0 ? (o)->pos + 3 * j : (o)->rpos + 3 * jl;    \
                        real xij = pix_ - pj_[0];                                              \
                        real yij = piy_ - pj_[1];                                              \
                        real zij = piz_ - pj_[2];                                              \
                        if (r_sqrt((xij * xij + yij * yij) + zij * zij) > (o)->h) continue;    \
                        (void)jl;                                                              \
                        __VA_ARGS__                                                            \
                    }                                                                          \
                }                                                                              \
    } while (0)

/* centre = fluid particle i (is_same_material = 0) */
#define FOR_WALL_NEIGHBORS_OF_FLUID(o, i, ...)                                                 \
    do {                                                                                       \
        const real pix_ = (o)->pos[3 * (i)], piy_ = (o)->pos[3 * (i) + 1], piz_ = (o)->pos[3 * (i) + 2]; \
        const int *cc_ = (o)->cell3 + 3 * (i);                                                 \
        for (int dx_ = -1; dx_ <= 1; ++dx_)                                                    \
            for (int dy_ = -1; dy_ <= 1; ++dy_)                                                \
                for (int dz_ = -1; dz_ <= 1; ++dz_) {                                          \
                    int cx_ = cc_[0] + dx_, cy_ = cc_[1] + dy_, cz_ = cc_[2] + dz_;            \
                    if (!cell_valid((o), cx_, cy_, cz_)) continue;                             \
                    int c1_ = cx_ * (o)->stride[0] + cy_ * (o)->stride[1] + cz_ * (o)->stride[2]; \
                    for (int e_ = (o)->bcstart[c1_]; e_ < (o)->bcstart[c1_ + 1]; ++e_) {       \
                        int j = (o)->bcitems[e_];                                              \
                        real xij = pix_ - (o)->bpos[3 * j];                                    \
                        real yij = piy_ - (o)->bpos[3 * j + 1];                                \
                        real zij = piz_ - (o)->bpos[3 * j + 2];                                \
                        if (r_sqrt((xij * xij + yij * yij) + zij * zij) > (o)->h) continue;    \
                        __VA_ARGS__                                                            \
                    }                                                                          \
                }                                                                              \
    } while (0)

#define PARFOR _Pragma("omp parallel for schedule(static) num_threads(o->nt)")

/* ---------------------------------------------------------------------------------------
 * initialisation                                         ParticleSystem.py:78-103,129-195
 * ------------------------------------------------------------------------------------- */
static int boundary_particles_count(const OrcConfig *c)
{
    /* ParticleSystem.py:129-137, Python f64 */
    double d = c->particle_radius * 2;
    double bx = c->box_max[0] - c->box_min[0];
    double by = c->box_max[1] - c->box_min[1];
    double bz = c->box_max[2] - c->box_min[2];
    int x_cnt = (int)(bx / d + 1);
    int z_cnt = (int)(bz / d + 1);
    int bottom = x_cnt * z_cnt;
    int ring = x_cnt * z_cnt - (x_cnt - 2) * (z_cnt - 2);
    int layer = (int)ceil((by - d) / d);
    return layer * ring + bottom * 2;
}

static inline real fmod_py(real a, real b) { return a - b * r_floor(a / b); }

/* ParticleSystem.py:139-195 */
static void init_particle_pos(Orc *o)
{
    const OrcConfig *c = &o->cfg;
    double dd = c->particle_radius * 2;
    /* compile-time constants folded in f64, then f32 */
    real x_num = R(c->water_size[0] / dd);
    real z_num = R(c->water_size[2] / dd);
    real xz_num = R((c->water_size[0] / dd) * (c->water_size[2] / dd));
    real radius = R(c->particle_radius);
    real sp[3] = { R(c->start_pos[0]), R(c->start_pos[1]), R(c->start_pos[2]) };
    for (int i = 0; i < o->N; ++i) {
        real x, z; int y;
        if (i < (1 << 24) || sizeof(real) == 8) {
            real fi = R(i);
            x = fmod_py(fi, x_num);                        /* :147 */
            z = fmod_py(r_floor(fi / x_num), z_num);       /* :148 */
            y = (int)(fi / xz_num);                        /* :149 */
        } else {    /* beyond 2^24 the f32 index is no longer exact and the reference's lattice collapses: continue in f64 (same rule as the library) */
            double di = (double)i, xn = (double)x_num, zn = (double)z_num, row = floor(di / xn);
            x = R(di - xn * floor(di / xn));
            z = R(row - zn * floor(row / zn));
            y = (int)(di / (double)xz_num);
        }
        o->pos[3 * i + 0] = x * radius * R(2) + sp[0];     /* :150 */
        o->pos[3 * i + 1] = R(y) * radius * R(2) + sp[1];
        o->pos[3 * i + 2] = z * radius * R(2) + sp[2];
    }
    /* walls: kernel-local f32 arithmetic on box                 :155-161 */
    real d = o->d;
    real boxx = R(c->box_max[0]) - R(c->box_min[0]);
    real boxz = R(c->box_max[2]) - R(c->box_min[2]);
    int x_cnt = (int)(boxx / d + R(1));
    int z_cnt = (int)(boxz / d + R(1));
    int xr = x_cnt - 1, zr = z_cnt - 1;
    int bottom = x_cnt * z_cnt;
    int ring = x_cnt * z_cnt - (x_cnt - 2) * (z_cnt - 2);
    int Nb = o->Nb;
    for (int i = 0; i < Nb; ++i) {
        real x = 0, y = 0, z = 0;
        if (i < bottom) {                                   /* :164-168 */
            x = R(i % x_cnt) * d;
            y = 0;
            z = r_floor(R(i) / R(x_cnt)) * d;
        } else if (i < Nb - bottom) {                       /* :169-189 */
            int index = i - bottom;
            int layer = (int)r_floor(R(index) / R(ring));
            y = d * R(layer + 1);
            index -= layer * ring;
            index += 1;
            if (index <= xr) {
                x = R(index % xr) * d; z = 0;
            } else if (index <= xr + zr) {
                x = R(xr) * d; z = R((index - x_cnt) % zr) * d;
            } else if (index <= 2 * xr + zr) {
                x = R((2 * xr + zr - index) % xr + 1) * d; z = R(zr) * d;
            } else if (index <= 2 * (xr + zr)) {
                x = 0; z = R((2 * (xr + zr) - index) % zr + 1) * d;
            }
        } else {                                            /* :190-195 */
            int index = i - (Nb - bottom);
            x = R(index % x_cnt) * d;
            y = R(c->box_max[1]);
            z = R((int)(R(index) / R(x_cnt))) * d;
        }
        o->bpos[3 * i] = x; o->bpos[3 * i + 1] = y; o->bpos[3 * i + 2] = z;
    }
}

/* compute_all_boundary_volume                            ParticleSystem.py:309-320 */
static void compute_all_boundary_volume(Orc *o)
{
    PARFOR
    for (int i = 0; i < o->Nb; ++i) {
        real volume = 0;
        const real pix = o->bpos[3 * i], piy = o->bpos[3 * i + 1], piz = o->bpos[3 * i + 2];
        const int *cc = o->bcell3 + 3 * i;
        for (int dx = -1; dx <= 1; ++dx)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dz = -1; dz <= 1; ++dz) {
                    int cx = cc[0] + dx, cy = cc[1] + dy, cz = cc[2] + dz;
                    if (!cell_valid(o, cx, cy, cz)) continue;
                    int c1 = cx * o->stride[0] + cy * o->stride[1] + cz * o->stride[2];
                    for (int e = o->bcstart[c1]; e < o->bcstart[c1 + 1]; ++e) {
                        int j = o->bcitems[e];
                        if (j == i) continue;                                   /* :362 */
                        real xij = pix - o->bpos[3 * j], yij = piy - o->bpos[3 * j + 1], zij = piz - o->bpos[3 * j + 2];
                        real q = r_sqrt((xij * xij + yij * yij) + zij * zij);
                        if (q > o->h) continue;
                        volume += cubic_kernel(q, o->h);                        /* :317-320 */
                    }
                }
        o->bvol[i] = R(1.0) / volume;                                           /* :314 */
    }
}

static void pcisph_init(Orc *o);
static void iisph_init(Orc *o);
void orc_compute_nbr_count(Orc *o);

Orc *orc_create(const OrcConfig *cfg)
{
    Orc *o = (Orc *)calloc(1, sizeof(Orc));
    o->cfg = *cfg;
    const OrcConfig *c = &o->cfg;
    double r = c->particle_radius;
    double d = r * 2;                       /* ParticleSystem.py:81 */
    double support = 4 * r;                 /* :82 */
    double m = 1000 * (r * r * r) * 8;      /* :83 */
    /* :85-86 left-to-right f64 */
    o->N = (int)(c->water_size[0] / d * c->water_size[1] / d * c->water_size[2] / d);
    o->Nb = boundary_particles_count(c);    /* :95 */
    o->Nr = 0;
    for (int a = 0; a < 3; ++a)             /* :100-101 */
        o->g[a] = (int)ceil((c->box_max[a] - c->box_min[a]) / support) + 1;
    o->stride[0] = 1; o->stride[1] = o->g[0] * o->g[2]; o->stride[2] = o->g[0];   /* :102 */
    o->C = o->g[0] * o->g[1] * o->g[2];
    o->h = R(support);
    o->m = R(m);
    o->d = R(d);
    o->rho0 = R(1000);
    o->gravity = R(c->gravity);
    /* per-solver constants: wcsph_solver.py:17-22 vs solver_base.py:23-26 */
    o->p_viscosity_c_s = c->solver == 0 ? 10 : 13;
    o->p_tension_k = c->solver == 0 ? 0.2 : 0.5;
    o->p_viscosity_alpha = 0.08; o->p_viscosity_epsilon = 0.01;
    o->p_density_threshold = 0.1; o->p_density_divergence_threshold = 10; o->p_max_dt = 1e-3; o->p_min_dt = 1e-5;       /* dfsph_solver.py:21-29 */
    o->p_min_iteration_density = 2; o->p_min_iteration_density_divergence = 1; o->p_max_iteration_density_divergence = 15;
    o->p_warm_start = 1; o->p_adaptive_dt = 1;
    fold_params(o);
    o->neg_m = R(-m);
    o->dt = R(c->delta_time);
    o->dt2 = R((real)R(c->delta_time) * (real)R(c->delta_time));   /* dfsph_solver.py:20: f32 field ** 2 */
    o->ps_dt = 0;
    o->dt_cfl_num = R(0.4 * r * 2);
    o->nt = c->num_threads > 0 ? c->num_threads : 1;

    size_t N = (size_t)(o->N > 0 ? o->N : 1), Nb = (size_t)(o->Nb > 0 ? o->Nb : 1);
#define ALLOC(ptr, type, n) ptr = (type *)calloc((n), sizeof(type))
    ALLOC(o->pos, real, 3 * N); ALLOC(o->vel, real, 3 * N); ALLOC(o->acc, real, 3 * N);
    ALLOC(o->cell3, int, 3 * N);
    ALLOC(o->bpos, real, 3 * Nb); ALLOC(o->bvol, real, Nb); ALLOC(o->bcell3, int, 3 * Nb);
    ALLOC(o->cstart, int, (size_t)o->C + 1); ALLOC(o->citems, int, N);
    ALLOC(o->bcstart, int, (size_t)o->C + 1); ALLOC(o->bcitems, int, Nb);
    ALLOC(o->rho, real, N); ALLOC(o->pressure, real, N); ALLOC(o->pgrad, real, 3 * N);
    ALLOC(o->bacc, real, 3 * N); ALLOC(o->visc, real, 3 * N); ALLOC(o->tens, real, 3 * N);
    ALLOC(o->alpha, real, N); ALLOC(o->rho_adv, real, N); ALLOC(o->rho_der, real, N);
    ALLOC(o->vel_adv, real, 3 * N); ALLOC(o->vel_adv_delta, real, 3 * N);
    ALLOC(o->force_ext, real, 3 * N); ALLOC(o->warm_k, real, N);
    ALLOC(o->nbr_cnt, int, N);
    if (c->solver == 4) {
        ALLOC(o->pbf_c, real, N); ALLOC(o->pbf_cd, real, 3 * N); ALLOC(o->pbf_lambda, real, N); ALLOC(o->pbf_dpos, real, 3 * N);
    }
    if (c->solver >= 2) {
        ALLOC(o->pos_predict, real, 3 * N); ALLOC(o->vel_predict, real, 3 * N); ALLOC(o->press_force, real, 3 * N);
        ALLOC(o->rho_err, real, N); ALLOC(o->press_iter, real, N);
        ALLOC(o->d_ii, real, 3 * N); ALLOC(o->d_ij, real, 3 * N); ALLOC(o->a_ii, real, N);
        ALLOC(o->p_past, real, N); ALLOC(o->p_new, real, N); ALLOC(o->r_sum, real, N);
    }
#undef ALLOC
    init_particle_pos(o);                                       /* ParticleSystem.py:119 */
    /* init_particles_data                                         :225-247 */
    build_lists(o, o->Nb, o->bpos, o->bcell3, o->bcstart, o->bcitems);   /* :237-238 */
    orc_build_grid(o);                                                   /* :240-241 */
    compute_all_boundary_volume(o);                                      /* :243 */
    if (c->solver == 2) pcisph_init(o);                                  /* pcisph_solver.__init__ */
    if (c->solver == 3) iisph_init(o);                                   /* iisph_solver.__init__ */
    return o;
}

void orc_destroy(Orc *o)
{
    if (!o) return;
    free(o->pos); free(o->vel); free(o->acc); free(o->cell3);
    free(o->bpos); free(o->bvol); free(o->bcell3);
    free(o->cstart); free(o->citems); free(o->bcstart); free(o->bcitems);
    free(o->rho); free(o->pressure); free(o->pgrad); free(o->bacc); free(o->visc); free(o->tens);
    free(o->alpha); free(o->rho_adv); free(o->rho_der); free(o->vel_adv); free(o->vel_adv_delta);
    free(o->force_ext); free(o->warm_k); free(o->nbr_cnt);
    free(o->pos_predict); free(o->vel_predict); free(o->press_force); free(o->rho_err); free(o->press_iter);
    free(o->d_ii); free(o->d_ij); free(o->a_ii); free(o->p_past); free(o->p_new); free(o->r_sum);
    free(o->rpos); free(o->rvol); free(o->rmass); free(o->rforce); free(o->rvert); free(o->rcell3);
    free(o->pbf_c); free(o->pbf_cd); free(o->pbf_lambda); free(o->pbf_dpos);
    free(o);
}

void orc_sizes(const Orc *o, int *out)
{
    out[0] = o->N; out[1] = o->Nb; out[2] = o->Nr;
    out[3] = o->g[0]; out[4] = o->g[1]; out[5] = o->g[2]; out[6] = o->C;
}

static real *field_ptr(Orc *o, int field, long *count, int *is_int)
{
    *is_int = 0;
    switch (field) {
    case ORC_F_POS: *count = 3L * o->N; return o->pos;
    case ORC_F_VEL: *count = 3L * o->N; return o->vel;
    case ORC_F_ACC: *count = 3L * o->N; return o->acc;
    case ORC_F_RHO: *count = o->N; return o->rho;
    case ORC_F_PRESSURE: *count = o->N; return o->pressure;
    case ORC_F_ALPHA: *count = o->N; return o->alpha;
    case ORC_F_WARM_K: *count = o->N; return o->warm_k;
    case ORC_F_RHO_ADV: *count = o->N; return o->rho_adv;
    case ORC_F_RHO_DER: *count = o->N; return o->rho_der;
    case ORC_F_VEL_ADV: *count = 3L * o->N; return o->vel_adv;
    case ORC_F_VISCOSITY: *count = 3L * o->N; return o->visc;
    case ORC_F_TENSION: *count = 3L * o->N; return o->tens;
    case ORC_F_PGRAD: *count = 3L * o->N; return o->pgrad;
    case ORC_F_BACC: *count = 3L * o->N; return o->bacc;
    case ORC_F_FORCE_EXT: *count = 3L * o->N; return o->force_ext;
    case ORC_F_PRESS_ITER: *count = o->press_iter ? o->N : -1; return o->press_iter;
    case ORC_F_PRESS_FORCE: *count = o->press_force ? 3L * o->N : -1; return o->press_force;
    case ORC_F_POS_PREDICT: *count = o->pos_predict ? 3L * o->N : -1; return o->pos_predict;
    case ORC_F_D_II: *count = o->d_ii ? 3L * o->N : -1; return o->d_ii;
    case ORC_F_A_II: *count = o->a_ii ? o->N : -1; return o->a_ii;
    case ORC_F_D_IJ: *count = o->d_ij ? 3L * o->N : -1; return o->d_ij;
    case ORC_F_PBF_LAMBDA: *count = o->pbf_lambda ? o->N : -1; return o->pbf_lambda;
    case ORC_F_PBF_DELTA_POS: *count = o->pbf_dpos ? 3L * o->N : -1; return o->pbf_dpos;
    case ORC_F_P_PAST: *count = o->p_past ? o->N : -1; return o->p_past;
    case ORC_F_WALL_POS: *count = 3L * o->Nb; return o->bpos;
    case ORC_F_WALL_VOL: *count = o->Nb; return o->bvol;
    case ORC_F_NBR_COUNT: *count = o->N; *is_int = 1; return NULL;
    case ORC_F_RIGID_POS: *count = 3L * o->Nr; return o->rpos;
    case ORC_F_RIGID_VOL: *count = o->Nr; return o->rvol;
    case ORC_F_RIGID_FORCE: *count = 3L * o->Nr; return o->rforce;
    case ORC_F_RIGID_MASS: *count = o->Nr; return o->rmass;
    case ORC_F_RIGID_VERT: *count = 3L * o->Nv; return o->rvert;
    default: *count = -1; return NULL;
    }
}

long orc_field_floats(Orc *o, int field)          /* element count of a field, -1 if it does not exist on this handle */
{
    long n; int is_int;
    (void)field_ptr(o, field, &n, &is_int);
    return n;
}

long orc_get(Orc *o, int field, float *out)
{
    long n; int is_int;
    real *p = field_ptr(o, field, &n, &is_int);
    if (n < 0) return -1;
    if (is_int) { for (long i = 0; i < n; ++i) out[i] = (float)o->nbr_cnt[i]; return n; }
    for (long i = 0; i < n; ++i) out[i] = (float)p[i];
    return n;
}

long orc_set(Orc *o, int field, const float *in)
{
    long n; int is_int;
    real *p = field_ptr(o, field, &n, &is_int);
    if (n < 0 || is_int) return -1;
    for (long i = 0; i < n; ++i) p[i] = R(in[i]);
    return n;
}

/* the Python scalars of the viscosity / tension terms, folded in f64 and rounded once (solver_base.py:187-188, :216) */
static void fold_params(Orc *o)
{
    double r = o->cfg.particle_radius, m = 1000 * (r * r * r) * 8, kernel_h = r * 4;     /* ParticleSystem.py:83, solver_base.py:17 */
    o->visc_num = R(2 * o->p_viscosity_alpha * kernel_h * o->p_viscosity_c_s);
    o->visc_eps_h2 = R(o->p_viscosity_epsilon * kernel_h * kernel_h);
    o->tens_c = R(-o->p_tension_k / m * m);
}

/* which 0: delta_time (and delta_time_2 = dt * dt, ps.delta_time = dt as compute_all_vel_adv leaves them, dfsph_solver.py:118-119):
 * lets a test or the CPU baseline continue from a state produced elsewhere (positions, velocities, warm_start_k through orc_set);
 * which 64..76: `solver.<attribute> = value` (see struct Orc) */
int orc_set_scalar(Orc *o, int which, double value)
{
    switch (which) {
    case 64: o->p_density_threshold = value; return 0;
    case 65: o->p_min_iteration_density = (int)value; return 0;
    case 66: o->p_min_iteration_density_divergence = (int)value; return 0;
    case 67: o->p_max_iteration_density_divergence = (int)value; return 0;
    case 68: o->p_density_divergence_threshold = value; return 0;
    case 69: o->p_warm_start = value != 0.0; return 0;
    case 70: o->p_adaptive_dt = value != 0.0; return 0;
    case 71: o->p_max_dt = value; return 0;
    case 72: o->p_min_dt = value; return 0;
    case 73: o->p_viscosity_c_s = value; fold_params(o); return 0;
    case 74: o->p_viscosity_alpha = value; fold_params(o); return 0;
    case 75: o->p_viscosity_epsilon = value; fold_params(o); return 0;
    case 76: o->p_tension_k = value; fold_params(o); return 0;
    default: break;
    }
    if (which != 0) return -1;
    o->dt = R(value);
    o->dt2 = o->dt * o->dt;
    o->ps_dt = o->dt;
    return 0;
}

double orc_get_scalar(const Orc *o, int which)
{
    switch (which) {
    case 0: return (double)o->dt;
    case 1: return (double)o->simulate_cnt;
    case 2: return (double)o->m;
    case 3: return (double)o->h;
    case 4: return (double)o->lost;
    case 5: return (double)o->pci_delta;
    case 6: return (double)o->pci_beta;
    case 7: return (double)o->pci_max_index;
    case 8: return (double)o->pci_max_count;
    case 9: return (double)o->ps_dt;
    case 10: case 11: case 12: return (double)o->centroid[which - 10];
    case 13: case 14: case 15: return (double)o->rs_omega[which - 13];
    case 16: case 17: case 18: return (double)o->r_vel[which - 16];
    case 19: return (double)o->rs_mass;
    case 20: case 21: case 22: case 23: case 24: case 25: case 26: case 27: case 28: return (double)o->inertia_inv[which - 20];
    case 64: return o->p_density_threshold;
    case 65: return o->p_min_iteration_density;
    case 66: return o->p_min_iteration_density_divergence;
    case 67: return o->p_max_iteration_density_divergence;
    case 68: return o->p_density_divergence_threshold;
    case 69: return o->p_warm_start;
    case 70: return o->p_adaptive_dt;
    case 71: return o->p_max_dt;
    case 72: return o->p_min_dt;
    case 73: return o->p_viscosity_c_s;
    case 74: return o->p_viscosity_alpha;
    case 75: return o->p_viscosity_epsilon;
    case 76: return o->p_tension_k;
    default: return 0;
    }
}

/* ---------------------------------------------------------------------------------------
 * rigid body helpers (config 5)
 * ------------------------------------------------------------------------------------- */
static inline void cross3(const real a[3], const real b[3], real out[3])
{
    out[0] = a[1] * b[2] - a[2] * b[1];
    out[1] = a[2] * b[0] - a[0] * b[2];
    out[2] = a[0] * b[1] - a[1] * b[0];
}
static inline void matvec3(const real m[9], const real v[3], real out[3])
{
    for (int r = 0; r < 3; ++r) out[r] = (m[3 * r] * v[0] + m[3 * r + 1] * v[1]) + m[3 * r + 2] * v[2];
}
static inline void matmul3(const real a[9], const real b[9], real out[9])
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) out[3 * r + c] = (a[3 * r] * b[c] + a[3 * r + 1] * b[3 + c]) + a[3 * r + 2] * b[6 + c];
}
/* ti.math.inverse for 3x3 (Taichi matrix.py, recalled; [taichi-semantics, unverifiable here]):
 * inv[j][i] = (1/det) * (E(i+1,j+1) E(i+2,j+2) - E(i+2,j+1) E(i+1,j+2)), indices mod 3 */
static void inverse3(const real m[9], real out[9])
{
#define E_(x, y) m[3 * ((x) % 3) + ((y) % 3)]
    real det = (m[0] * (m[4] * m[8] - m[7] * m[5]) - m[3] * (m[1] * m[8] - m[7] * m[2])) + m[6] * (m[1] * m[5] - m[4] * m[2]);
    real inv_det = R(1.0) / det;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            out[3 * j + i] = inv_det * (E_(i + 1, j + 1) * E_(i + 2, j + 2) - E_(i + 2, j + 1) * E_(i + 1, j + 2));
#undef E_
}
/* ti.math.rotation3d(ang_x, ang_y, ang_z), upper-left 3x3 (Taichi mathimpl.py, recalled; unverifiable here:
 * the rigid body's initial orientation is 'parity unpinned') */
static void rotation3d(real ang_x, real ang_y, real ang_z, real m[9])
{
    real ca = sizeof(real) == 4 ? (real)cosf((float)ang_x) : (real)cos((double)ang_x);
    real sa = sizeof(real) == 4 ? (real)sinf((float)ang_x) : (real)sin((double)ang_x);
    real cb = sizeof(real) == 4 ? (real)cosf((float)ang_z) : (real)cos((double)ang_z);
    real sb = sizeof(real) == 4 ? (real)sinf((float)ang_z) : (real)sin((double)ang_z);
    real cy = sizeof(real) == 4 ? (real)cosf((float)ang_y) : (real)cos((double)ang_y);
    real sy = sizeof(real) == 4 ? (real)sinf((float)ang_y) : (real)sin((double)ang_y);
    m[0] = cb * cy + sb * sa * sy; m[1] = sb * ca; m[2] = -cb * sy + sb * sa * cy;
    m[3] = -sb * cy + cb * sa * sy; m[4] = cb * ca; m[5] = sb * sy + cb * sa * cy;
    m[6] = ca * sy; m[7] = -sa; m[8] = ca * cy;
}

/* predicted velocity of rigid particle j as the fluid sweeps see it:
 * v_j = vel + acc*dt + cross(omega [+ alpha*dt], x_j - c)     dfsph_solver.py:292-293 / :168-169 */
static inline void rigid_predicted_velocity(const Orc *o, int jl, int with_alpha, real out[3])
{
    real w[3], rel[3], vo[3];
    for (int a = 0; a < 3; ++a) {
        w[a] = with_alpha ? o->r_omega[a] + o->r_alpha[a] * o->dt : o->r_omega[a];
        rel[a] = o->rpos[3 * jl + a] - o->centroid[a];
    }
    cross3(w, rel, vo);
    for (int a = 0; a < 3; ++a) out[a] = (o->r_vel[a] + o->r_acc[a] * o->dt) + vo[a];
}

/* Force of the fluid on the rigid sample particles, accumulated per rigid particle over its fluid neighbours in cell-walk
 * order (the reference uses atomics in thread order from inside the fluid sweeps: any serialisation is valid).
 *   mode 0 wcsph   wcsph_solver.py:125-127   force += -ret * m,  ret = -V_j p_i / rho_i^2 * gradW * rho_0
 *   mode 1 dfsph   dfsph_solver.py:204-212   force += ret * m,   ret = V_j rho_0 k_i / rho_i * gradW
 *   mode 2 pcisph  pcisph_solver.py:208-210  force += ret * m,   ret = V_j rho_0 press_iter_i * gradW / rho_i^2   (every iteration)
 *   mode 3 iisph   iisph_solver.py:166-167   force += force * m, force = V_j rho_0 / rho_i^2 * gradW * p_iter_i */
static void rigid_accumulate_force_mode(Orc *o, int mode)
{
    PARFOR
    for (int r = 0; r < o->Nr; ++r) {
        real fx = 0, fy = 0, fz = 0;
        const real *pr = o->rpos + 3 * r;
        const int *cc = o->rcell3 + 3 * r;
        for (int dx = -1; dx <= 1; ++dx)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dz = -1; dz <= 1; ++dz) {
                    int cx = cc[0] + dx, cy = cc[1] + dy, cz = cc[2] + dz;
                    if (!cell_valid(o, cx, cy, cz)) continue;
                    int c1 = cx * o->stride[0] + cy * o->stride[1] + cz * o->stride[2];
                    for (int e = o->cstart[c1]; e < o->cstart[c1 + 1]; ++e) {
                        int i = o->citems[e];
                        if (i >= o->N) continue;                       /* fluid particles exert the force */
                        real xij = o->pos[3 * i] - pr[0], yij = o->pos[3 * i + 1] - pr[1], zij = o->pos[3 * i + 2] - pr[2];
                        if (r_sqrt((xij * xij + yij * yij) + zij * zij) > o->h) continue;
                        real gw[3];
                        cubic_kernel_derivative(xij, yij, zij, o->h, gw);
                        if (mode == 1) {
                            real k_i = (o->rho_adv[i] - o->rho0) * o->alpha[i] / o->dt2;             /* :208 */
                            real s = o->rvol[r] * o->rho0 * k_i / o->rho[i];                         /* :211 */
                            fx += s * gw[0] * o->m; fy += s * gw[1] * o->m; fz += s * gw[2] * o->m;   /* :212 */
                        } else if (mode == 0) {
                            real rho_i_2 = o->rho[i] * o->rho[i];
                            real s = -o->rvol[r] * o->pressure[i] / rho_i_2;                         /* wcsph :125 */
                            fx += -(s * gw[0] * o->rho0) * o->m; fy += -(s * gw[1] * o->rho0) * o->m; fz += -(s * gw[2] * o->rho0) * o->m;   /* :127 */
                        } else if (mode == 2) {
                            real a = o->rvol[r] * o->rho0 * o->press_iter[i];                        /* pcisph :208 */
                            real den = o->rho[i] * o->rho[i];
                            fx += a * gw[0] / den * o->m; fy += a * gw[1] / den * o->m; fz += a * gw[2] / den * o->m;   /* :209 */
                        } else {
                            real s = o->rvol[r] * o->rho0 / (o->rho[i] * o->rho[i]);                  /* iisph :166 */
                            fx += s * gw[0] * o->press_iter[i] * o->m; fy += s * gw[1] * o->press_iter[i] * o->m;
                            fz += s * gw[2] * o->press_iter[i] * o->m;                               /* :167 */
                        }
                    }
                }
        o->rforce[3 * r] += fx; o->rforce[3 * r + 1] += fy; o->rforce[3 * r + 2] += fz;
    }
}
static void rigid_accumulate_force(Orc *o) { rigid_accumulate_force_mode(o, 1); }
static inline int rigid_coupled(const Orc *o) { return o->exist_rigid && o->active_rigid && o->cfg.fs_couple; }

/* init_rigid_particles_pos + init_rigid_particles_data          ParticleSystem.py:198-223, 249-295 */
static void init_rigid(Orc *o, const OrcRigid *rg)
{
    const double pi = 3.141592653589793;
    real att[3];
    for (int a = 0; a < 3; ++a) att[a] = R(rg->attitude_offset_deg[a] / 180.0 * pi);            /* :52 */
    real m[9];
    rotation3d(att[0], att[2], att[1], m);                                                       /* :200 */
    real off[3] = { R(rg->pos_offset[0]), R(rg->pos_offset[1]), R(rg->pos_offset[2]) };
    for (int pass = 0; pass < 2; ++pass) {
        int n = pass == 0 ? o->Nr : o->Nv;
        real *dst = pass == 0 ? o->rpos : o->rvert;
        const float *src = pass == 0 ? rg->points : rg->vertices;
        for (int i = 0; i < n; ++i) {
            real p[3] = { R(src[3 * i]), R(src[3 * i + 1]), R(src[3 * i + 2]) };
            for (int r = 0; r < 3; ++r) {
                real v = ((m[3 * r] * p[0] + m[3 * r + 1] * p[1]) + m[3 * r + 2] * p[2]) + R(0) * R(1);   /* mat4 @ (p, 1) :205-207 */
                dst[3 * i + r] = v + off[r];                                                     /* :218, :223 */
            }
        }
    }
}

static void init_rigid_data(Orc *o)
{
    /* volume = 1 / sum of W over SOLID neighbours of the fluid/rigid grid              :252-259, 301-307 */
    for (int i = 0; i < o->Nr; ++i) {
        real volume = 0;
        if (o->active_rigid) {
            const int gi = i + o->N + o->Nb;
            const real *pi_ = o->rpos + 3 * i;
            const int *cc = o->rcell3 + 3 * i;
            for (int dx = -1; dx <= 1; ++dx)
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dz = -1; dz <= 1; ++dz) {
                        int cx = cc[0] + dx, cy = cc[1] + dy, cz = cc[2] + dz;
                        if (!cell_valid(o, cx, cy, cz)) continue;
                        int c1 = cx * o->stride[0] + cy * o->stride[1] + cz * o->stride[2];
                        for (int e = o->cstart[c1]; e < o->cstart[c1 + 1]; ++e) {
                            int j = o->citems[e];
                            if (j == gi || j < o->N) continue;          /* self; fluid neighbours add 0 */
                            const real *pj = o->rpos + 3 * (j - o->N - o->Nb);
                            real x = pi_[0] - pj[0], y = pi_[1] - pj[1], z = pi_[2] - pj[2];
                            real q = r_sqrt((x * x + y * y) + z * z);
                            if (q > o->h) continue;
                            volume += cubic_kernel(q, o->h);
                        }
                    }
        }
        o->rvol[i] = volume < R(1e-6) ? R(0.0) : R(1.0) / volume;                                /* :255-259 */
    }
    for (int i = 0; i < o->Nr; ++i) o->rmass[i] = o->rigid_rho * o->rvol[i];                     /* :262-263 */
    real c[3] = {0, 0, 0}, sum_mass = 0;                                                         /* :266-271 */
    for (int i = 0; i < o->Nr; ++i) {
        for (int a = 0; a < 3; ++a) c[a] += o->rpos[3 * i + a] * o->rmass[i];
        sum_mass += o->rmass[i];
    }
    for (int a = 0; a < 3; ++a) o->centroid[a] = c[a] / sum_mass;
    real Ixx = 0, Iyy = 0, Izz = 0, Ixy = 0, Ixz = 0, Iyz = 0;                                   /* :275-288 */
    for (int i = 0; i < o->Nr; ++i) {
        real x = o->rpos[3 * i] - o->centroid[0], y = o->rpos[3 * i + 1] - o->centroid[1], z = o->rpos[3 * i + 2] - o->centroid[2];
        real mi = o->rmass[i];
        Ixx += mi * (y * y + z * z);
        Iyy += mi * (x * x + z * z);
        Izz += mi * (x * x + y * y);
        Ixy += -mi * (x * y);
        Ixz += -mi * (x * z);
        Iyz += -mi * (z * y);
    }
    real I[9] = { Ixx, Ixy, Ixz, Ixy, Iyy, Iyz, Ixz, Iyz, Izz };                                 /* :290 */
    inverse3(I, o->inertia_inv);                                                                 /* :291 */
}

Orc *orc_create_rigid(const OrcConfig *cfg, const OrcRigid *rg)
{
    if (!rg || rg->n_particles <= 0) return NULL;
    Orc *o = orc_create(cfg);
    o->exist_rigid = 1;
    o->active_rigid = rg->active ? 1 : 0;
    o->Nr = rg->n_particles;
    o->Nv = rg->n_vertices;
    o->rigid_rho = R(rg->rho_0);
    size_t Nr = (size_t)o->Nr, Nv = (size_t)(o->Nv > 0 ? o->Nv : 1);
    o->rpos = (real *)calloc(3 * Nr, sizeof(real)); o->rvol = (real *)calloc(Nr, sizeof(real));
    o->rmass = (real *)calloc(Nr, sizeof(real)); o->rforce = (real *)calloc(3 * Nr, sizeof(real));
    o->rvert = (real *)calloc(3 * Nv, sizeof(real)); o->rcell3 = (int *)calloc(3 * Nr, sizeof(int));
    free(o->citems);
    o->citems = (int *)calloc((size_t)o->N + Nr + 1, sizeof(int));
    init_rigid(o, rg);                                                  /* ParticleSystem.py:121 */
    orc_build_grid(o);                                                  /* :240-241 (now with the rigid particles) */
    init_rigid_data(o);                                                 /* :247 */
    o->rs_dt = R(cfg->delta_time);                                      /* rigid_solver.py:13 */
    if (cfg->solver == 2) pcisph_init(o);                               /* the solver is constructed after the ParticleSystem: the grid holds the body */
    return o;
}

/* rigid_solver.step                                                   rigid_solver.py:216-232 */
void orc_rigid_step(Orc *o)
{
    if (!o->exist_rigid) return;
    if (!o->rs_run_once) {                                              /* compute_sum_mass :156-162 */
        real sm = 0;
        for (int i = 0; i < o->Nr; ++i) sm += o->rmass[i];
        o->rs_mass = sm;
        o->rs_run_once = 1;
    }
    o->rs_cnt += 1;
    if (o->ps_dt > R(0.0)) o->rs_dt = o->ps_dt;                         /* :223-224 */
    const real dt = o->rs_dt;
    /* compute_attitude :118-128 (sums in f64, ascending particle order; reference: f32 atomics) */
    {
        double t[3] = {0, 0, 0};
        for (int i = 0; i < o->Nr; ++i) {
            real rel[3] = { o->rpos[3 * i] - o->centroid[0], o->rpos[3 * i + 1] - o->centroid[1], o->rpos[3 * i + 2] - o->centroid[2] };
            real tq[3];
            cross3(rel, o->rforce + 3 * i, tq);
            t[0] += (double)tq[0]; t[1] += (double)tq[1]; t[2] += (double)tq[2];
        }
        real torque[3] = { R(t[0]), R(t[1]), R(t[2]) }, alpha[3];
        matvec3(o->inertia_inv, torque, alpha);
        for (int a = 0; a < 3; ++a) {
            o->rs_omega[a] += alpha[a] * dt;
            o->rs_attitude[a] = o->rs_omega[a] * dt;
            o->r_alpha[a] = alpha[a];                                   /* rigid_particles.alpha.fill */
        }
    }
    /* rotation :130-141 */
    {
        real m[9], mt[9], tmp[9];
        rotation3d(-o->rs_attitude[0], -o->rs_attitude[2], -o->rs_attitude[1], m);
        for (int pass = 0; pass < 2; ++pass) {
            int n = pass == 0 ? o->Nr : o->Nv;
            real *arr = pass == 0 ? o->rpos : o->rvert;
            for (int i = 0; i < n; ++i) {
                real rel[3] = { arr[3 * i] - o->centroid[0], arr[3 * i + 1] - o->centroid[1], arr[3 * i + 2] - o->centroid[2] }, rot[3];
                matvec3(m, rel, rot);
                for (int a = 0; a < 3; ++a) arr[3 * i + a] = rot[a] + o->centroid[a];
            }
        }
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) mt[3 * r + c] = m[3 * c + r];
        matmul3(m, o->inertia_inv, tmp);
        matmul3(tmp, mt, o->inertia_inv);                               /* :141 */
    }
    /* kinematic :33-104 */
    {
        double f[3] = {0, 0, 0};
        for (int i = 0; i < o->Nr; ++i) {                               /* :36-38 */
            f[0] += (double)o->rforce[3 * i]; f[1] += (double)o->rforce[3 * i + 1]; f[2] += (double)o->rforce[3 * i + 2];
            o->rforce[3 * i] = o->rforce[3 * i + 1] = o->rforce[3 * i + 2] = 0;
        }
        real force[3] = { R(f[0]), R(f[1]), R(f[2]) };
        const real g[3] = { o->gravity * R(0.0), o->gravity * R(-1.0), o->gravity * R(0.0) };
        real vel[3], disp[3], ori[3];
        for (int a = 0; a < 3; ++a) {
            o->r_acc[a] = force[a] / o->rs_mass + g[a];                 /* :40-41 */
            vel[a] = o->r_acc[a] * dt + o->r_vel[a];                    /* :43 */
            disp[a] = vel[a] * dt;                                      /* :45 */
            ori[a] = disp[a];
        }
        int ccount = 0;
        double cp[3] = {0, 0, 0};
        real cnorm[3] = {0, 0, 0};
        /* ti.atomic_max / atomic_min on displacement[j] (:58, :67) run in thread order in the reference; restated as
         * "all lower-wall maxima first, then all upper-wall minima" (identical unless one step hits both walls of an axis) */
        real dmax[3] = {-INFINITY, -INFINITY, -INFINITY}, dmin[3] = {INFINITY, INFINITY, INFINITY};
        int lo_hit[3] = {0, 0, 0}, hi_hit[3] = {0, 0, 0};   /* collision_norm[j] is written racily (:63, :72): upper wall wins here */
        const real lo[3] = { R(o->cfg.box_min[0]) + o->d, R(o->cfg.box_min[1]) + o->d, R(o->cfg.box_min[2]) + o->d };
        const real hi[3] = { R(o->cfg.box_max[0]) - o->d, R(o->cfg.box_max[1]) - o->d, R(o->cfg.box_max[2]) - o->d };
        for (int i = 0; i < o->Nr; ++i) {                               /* :53-76 */
            const real *p = o->rpos + 3 * i;
            real rel[3] = { p[0] + ori[0] - o->centroid[0], p[1] + ori[1] - o->centroid[1], p[2] + ori[2] - o->centroid[2] }, wr[3];
            cross3(o->rs_omega, rel, wr);
            for (int j = 0; j < 3; ++j) {
                int collision = 0;
                if (p[j] + ori[j] <= lo[j]) {
                    dmax[j] = r_max(dmax[j], lo[j] - p[j]);             /* :58 */
                    if (vel[j] + wr[j] < 0) { collision = 1; lo_hit[j] = 1; }
                }
                if (p[j] + ori[j] >= hi[j]) {
                    real cand = hi[j] - p[j];
                    dmin[j] = cand < dmin[j] ? cand : dmin[j];          /* :67 */
                    if (vel[j] + wr[j] > 0) { collision = 1; hi_hit[j] = 1; }
                }
                if (collision == 1) { cp[0] += (double)p[0]; cp[1] += (double)p[1]; cp[2] += (double)p[2]; ccount += 1; }   /* :74-76 */
            }
        }
        for (int j = 0; j < 3; ++j) {
            cnorm[j] = hi_hit[j] ? R(1) : (lo_hit[j] ? R(-1) : R(0));
            disp[j] = r_max(disp[j], dmax[j]);
            disp[j] = dmin[j] < disp[j] ? dmin[j] : disp[j];
        }
        if (ccount > 0) {                                               /* :80-94 */
            real cpt[3], cv[3], wr[3];
            for (int a = 0; a < 3; ++a) cpt[a] = (R(cp[a]) + ori[a]) / R(ccount) - o->centroid[a];
            cross3(o->rs_omega, cpt, wr);
            for (int a = 0; a < 3; ++a) cv[a] = vel[a] + wr[a];
            /* compute_new_vel :106-116 */
            const real mu_n = R(0.1);
            const real mu_c = R(0.8 * (1 + 0.1));               /* mu_t * (1 + mu_n): Python scalars, folded in f64 */
            real vdn = (cv[0] * cnorm[0] + cv[1] * cnorm[1]) + cv[2] * cnorm[2];
            real vn[3], vt[3], vnew[3];
            for (int a = 0; a < 3; ++a) { vn[a] = vdn * cnorm[a]; vt[a] = cv[a] - vn[a]; }
            real nvn = r_sqrt((vn[0] * vn[0] + vn[1] * vn[1]) + vn[2] * vn[2]);
            real nvt = r_sqrt((vt[0] * vt[0] + vt[1] * vt[1]) + vt[2] * vt[2]);
            real a_ = r_max(R(1) - mu_c * nvn / nvt, R(0.0));
            for (int a = 0; a < 3; ++a) vnew[a] = a_ * vt[a] + (-mu_n * vn[a]);
            /* K = I/M - [r]x I^-1 [r]x ; j = K^-1 (v_new - v)        :88-94 */
            real rx[9] = { 0, -cpt[2], cpt[1], cpt[2], 0, -cpt[0], -cpt[1], cpt[0], 0 };
            real t1[9], t2[9], K[9], Kinv[9], dv[3], jimp[3], cj[3], dw[3];
            matmul3(rx, o->inertia_inv, t1);
            matmul3(t1, rx, t2);
            for (int q = 0; q < 9; ++q) K[q] = ((q % 4 == 0) ? R(1) / o->rs_mass : R(0) / o->rs_mass) - t2[q];
            inverse3(K, Kinv);
            for (int a = 0; a < 3; ++a) dv[a] = vnew[a] - cv[a];
            matvec3(Kinv, dv, jimp);
            for (int a = 0; a < 3; ++a) vel[a] += jimp[a] / o->rs_mass;
            cross3(cpt, jimp, cj);
            matvec3(o->inertia_inv, cj, dw);
            for (int a = 0; a < 3; ++a) o->rs_omega[a] += dw[a];
        }
        for (int a = 0; a < 3; ++a) { o->r_omega[a] = o->rs_omega[a]; o->r_vel[a] = vel[a]; }     /* :96-97 */
        for (int i = 0; i < o->Nr; ++i) for (int a = 0; a < 3; ++a) o->rpos[3 * i + a] += disp[a];   /* :98-99 */
        for (int i = 0; i < o->Nv; ++i) for (int a = 0; a < 3; ++a) o->rvert[3 * i + a] += disp[a];  /* :101-102 */
        for (int a = 0; a < 3; ++a) o->centroid[a] += disp[a];                                    /* :104 */
    }
}

/* ---------------------------------------------------------------------------------------
 * shared sweeps                                                       solver_base.py:41-217
 * ------------------------------------------------------------------------------------- */
/* compute_all_rho                                                     solver_base.py:41-72 */
static void pbf_compute_rho(Orc *o);
void orc_compute_rho(Orc *o)
{
    if (o->cfg.solver == 4) { pbf_compute_rho(o); return; }   /* pbf_solver.py:166-174 overrides compute_rho / compute_rho_from_boundary with the poly6 kernel */
    PARFOR
    for (int i = 0; i < o->N; ++i) {
        real rho = R(0.001);                                           /* :44 */
        FOR_FLUID_NEIGHBORS(o, i, {
            if (jm_ == 0)
                rho += o->m * cubic_kernel(r_sqrt((xij * xij + yij * yij) + zij * zij), o->h);   /* :62 */
            else if (o->cfg.fs_couple)
                rho += o->rvol[jl] * cubic_kernel(r_sqrt((xij * xij + yij * yij) + zij * zij), o->h) * o->rho0;   /* :65 */
        });
        if (o->cfg.boundary_handle) {
            real rho_boundary = 0;
            FOR_WALL_NEIGHBORS_OF_FLUID(o, i, {
                real q = r_sqrt((xij * xij + yij * yij) + zij * zij);
                rho_boundary += o->bvol[j] * cubic_kernel(q, o->h);    /* :70-71 */
            });
            o->rho[i] = rho + rho_boundary * o->rho0;                  /* :49 */
        } else {
            o->rho[i] = rho;
        }
    }
}

/* solve_all_viscosity                                                solver_base.py:170-202 */
static void solve_all_viscosity(Orc *o)
{
    PARFOR
    for (int i = 0; i < o->N; ++i) {
        real vx = 0, vy = 0, vz = 0;
        const real vix = o->vel[3 * i], viy = o->vel[3 * i + 1], viz = o->vel[3 * i + 2];
        FOR_FLUID_NEIGHBORS(o, i, {
            if (jm_ != 0) {
                if (o->cfg.fs_couple) {
                    real vijx = vix - o->r_vel[0], vijy = viy - o->r_vel[1], vijz = viz - o->r_vel[2];   /* :192 */
                    real shear = (vijx * xij + vijy * yij) + vijz * zij;
                    if (shear < 0 && jl < o->N) {     /* jl >= N would read rho[] out of bounds in the reference */
                        real q = r_sqrt((xij * xij + yij * yij) + zij * zij);
                        real q2 = q * q;
                        /* quirk: rho[particle_j.index] reads the FLUID density at the rigid particle's local index (:198-199) */
                        real nu = o->visc_num / (o->rho[i] + o->rho[jl]);
                        real pi = -nu * shear / (q2 + o->visc_eps_h2);
                        real gw[3];
                        cubic_kernel_derivative(xij, yij, zij, o->h, gw);
                        real s = R(-1000) * o->rvol[jl] * pi;                   /* :201 */
                        vx += s * gw[0]; vy += s * gw[1]; vz += s * gw[2];
                    }
                }
                continue;
            }
            real vijx = vix - o->vel[3 * j], vijy = viy - o->vel[3 * j + 1], vijz = viz - o->vel[3 * j + 2];
            real shear = (vijx * xij + vijy * yij) + vijz * zij;        /* :183 */
            if (shear < 0) {
                real q = r_sqrt((xij * xij + yij * yij) + zij * zij);
                real q2 = q * q;
                real nu = o->visc_num / (o->rho[i] + o->rho[j]);        /* :187 */
                real pi = -nu * shear / (q2 + o->visc_eps_h2);          /* :188 */
                real gw[3];
                cubic_kernel_derivative(xij, yij, zij, o->h, gw);
                real s = o->neg_m * pi;                                 /* :189 */
                vx += s * gw[0]; vy += s * gw[1]; vz += s * gw[2];
            }
        });
        o->visc[3 * i] = vx * o->m; o->visc[3 * i + 1] = vy * o->m; o->visc[3 * i + 2] = vz * o->m;   /* :175 */
    }
}

/* solve_all_tension                                                  solver_base.py:204-217 */
static void solve_all_tension(Orc *o)
{
    PARFOR
    for (int i = 0; i < o->N; ++i) {
        real tx = 0, ty = 0, tz = 0;
        FOR_FLUID_NEIGHBORS(o, i, {
            if (jm_ != 0) continue;                                     /* :214: fluid neighbours only */
            real w = cubic_kernel(r_sqrt((xij * xij + yij * yij) + zij * zij), o->h);
            real s = o->tens_c * w;                                     /* :216 */
            tx += s * xij; ty += s * yij; tz += s * zij;
        });
        o->tens[3 * i] = tx * o->m; o->tens[3 * i + 1] = ty * o->m; o->tens[3 * i + 2] = tz * o->m;   /* :209 */
    }
}

/* ---------------------------------------------------------------------------------------
 * WCSPH                                                                    wcsph_solver.py
 * ------------------------------------------------------------------------------------- */
/* solve_all_pressure_gradient                                        wcsph_solver.py:70-129 */
static void solve_all_pressure_gradient(Orc *o)
{
    PARFOR
    for (int i = 0; i < o->N; ++i) {
        real rx = 0, ry = 0, rz = 0;
        const real rho_i = o->rho[i];
        const real rho_i_2 = rho_i * rho_i;                             /* :109 */
        const real p_i = o->pressure[i];
        FOR_FLUID_NEIGHBORS(o, i, {
            real gw[3];
            if (jm_ != 0) {
                if (o->cfg.fs_couple) {
                    cubic_kernel_derivative(xij, yij, zij, o->h, gw);
                    real sr = -o->rvol[jl] * p_i / rho_i_2;             /* :125 */
                    rx += sr * gw[0] * o->rho0; ry += sr * gw[1] * o->rho0; rz += sr * gw[2] * o->rho0;
                }
                continue;
            }
            real p_j = o->pressure[j];
            real rho_j = o->rho[j];
            cubic_kernel_derivative(xij, yij, zij, o->h, gw);
            real s = o->m * (p_i / rho_i_2 + p_j / (rho_j * rho_j));    /* :116 */
            rx -= s * gw[0]; ry -= s * gw[1]; rz -= s * gw[2];
        });
        if (o->cfg.boundary_handle) {
            real bx = 0, by = 0, bz = 0;
            FOR_WALL_NEIGHBORS_OF_FLUID(o, i, {
                real gw[3];
                cubic_kernel_derivative(xij, yij, zij, o->h, gw);
                real s = o->bvol[j] * p_i / rho_i_2;                    /* :99 */
                bx -= s * gw[0]; by -= s * gw[1]; bz -= s * gw[2];
            });
            o->bacc[3 * i] = bx * o->rho0; o->bacc[3 * i + 1] = by * o->rho0; o->bacc[3 * i + 2] = bz * o->rho0;  /* :83 */
        }
        o->pgrad[3 * i] = rx; o->pgrad[3 * i + 1] = ry; o->pgrad[3 * i + 2] = rz;      /* :84 */
    }
}

/* kinematic_phase                                                    wcsph_solver.py:40-63 */
static void wcsph_kinematic_phase(Orc *o)
{
    real box_lo[3], box_hi[3];
    for (int a = 0; a < 3; ++a) {
        box_lo[a] = R(o->cfg.box_min[a]) + o->d;
        box_hi[a] = R(o->cfg.box_max[a]) - o->d;
    }
    PARFOR
    for (int i = 0; i < o->N; ++i) {
        for (int a = 0; a < 3; ++a) {
            int k = 3 * i + a;
            if (o->cfg.boundary_handle)
                o->acc[k] += ((o->pgrad[k] + o->visc[k]) + o->tens[k]) + o->bacc[k];   /* :44-45 */
            else
                o->acc[k] += (o->pgrad[k] + o->visc[k]) + o->tens[k];                  /* :47 */
            o->vel[k] += o->acc[k] * o->dt;                                            /* :50 */
            o->vel[k] *= R(0.9998);                                                    /* :51 */
            o->pos[k] += o->vel[k] * o->dt;                                            /* :52 */
        }
        if (!o->cfg.boundary_handle) {                                                 /* :54-63 */
            for (int a = 0; a < 3; ++a) {
                int k = 3 * i + a;
                if (o->pos[k] <= box_lo[a]) { o->pos[k] = box_lo[a]; o->vel[k] *= R(-0.5); }
                if (o->pos[k] >= box_hi[a]) { o->pos[k] = box_hi[a]; o->vel[k] *= R(-0.5); }
            }
        }
    }
}

/* wcsph_solver.step                                                  wcsph_solver.py:25-38 */
int orc_step_wcsph(Orc *o, int nsteps)
{
    for (int s = 0; s < nsteps; ++s) {
        o->simulate_cnt += 1;                                          /* solver_base.py:137 */
        orc_build_grid(o);                                             /* :139-141 */
        PARFOR
        for (int i = 0; i < o->N; ++i) {                               /* reset(): solver_base.py:131-133 */
            o->acc[3 * i] = o->gravity * R(0.0);
            o->acc[3 * i + 1] = o->gravity * R(-1.0);
            o->acc[3 * i + 2] = o->gravity * R(0.0);
        }
        orc_compute_rho(o);                                            /* wcsph_solver.py:34 */
        PARFOR
        for (int i = 0; i < o->N; ++i) o->pressure[i] = tait_pressure(o->rho[i]);   /* :66-68 */
        solve_all_pressure_gradient(o);                                /* :36 */
        if (rigid_coupled(o)) rigid_accumulate_force_mode(o, 0);       /* :127 */
        solve_all_viscosity(o);                                        /* :37 */
        solve_all_tension(o);                                          /* :38 */
        wcsph_kinematic_phase(o);                                      /* :30 */
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------
 * DFSPH                                                                    dfsph_solver.py
 * ------------------------------------------------------------------------------------- */
/* get_neighbour_count for every fluid particle                   ParticleSystem.py:424-445 */
void orc_compute_nbr_count(Orc *o)
{
    PARFOR
    for (int i = 0; i < o->N; ++i) {
        int cnt = 0;
        if (!(o->exist_rigid && o->active_rigid)) {
            FOR_FLUID_NEIGHBORS(o, i, { (void)xij; (void)yij; (void)zij; cnt += 1; });
        } else {
            /* literal: skip when particle_j.index == i (LOCAL index), distance to fluid_particles.pos[particle_j.index] */
            const int *cc = o->cell3 + 3 * i;
            for (int dx = -1; dx <= 1; ++dx)
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dz = -1; dz <= 1; ++dz) {
                        int cx = cc[0] + dx, cy = cc[1] + dy, cz = cc[2] + dz;
                        if (!cell_valid(o, cx, cy, cz)) continue;
                        int c1 = cx * o->stride[0] + cy * o->stride[1] + cz * o->stride[2];
                        for (int e = o->cstart[c1]; e < o->cstart[c1 + 1]; ++e) {
                            int j = o->citems[e];
                            int jl = j < o->N ? j : j - o->N - o->Nb;
                            if (jl == i) continue;                                  /* :440 */
                            if (jl >= o->N) continue;     /* would index fluid_particles out of bounds in the reference */
                            real x = o->pos[3 * i] - o->pos[3 * jl], y = o->pos[3 * i + 1] - o->pos[3 * jl + 1], z = o->pos[3 * i + 2] - o->pos[3 * jl + 2];
                            if (r_sqrt((x * x + y * y) + z * z) > o->h) continue;   /* :442 */
                            cnt += 1;
                        }
                    }
        }
        o->nbr_cnt[i] = cnt;
    }
}

/* compute_all_alpha                                                   dfsph_solver.py:32-89 */
void orc_compute_alpha(Orc *o)
{
    PARFOR
    for (int i = 0; i < o->N; ++i) {
        real sx = 0, sy = 0, sz = 0;       /* sum_square   (:38, compute_sum)        */
        real square_sum = 0;               /* square_sum   (:39, compute_square_sum) */
        FOR_FLUID_NEIGHBORS(o, i, {
            real gw[3];
            if (jm_ != 0 && !o->cfg.fs_couple) continue;
            cubic_kernel_derivative(xij, yij, zij, o->h, gw);
            const real cm = jm_ == 0 ? o->m : o->rvol[jl] * o->rho0;                /* :58,:70 / :62,:75 */
            real rx = cm * gw[0], ry = cm * gw[1], rz = cm * gw[2];
            sx += rx; sy += ry; sz += rz;
            square_sum += (rx * rx + ry * ry) + rz * rz;                            /* :71 */
        });
        real denominator;
        if (o->cfg.boundary_handle) {
            real bx = 0, by = 0, bz = 0, bsq = 0;
            FOR_WALL_NEIGHBORS_OF_FLUID(o, i, {
                real gw[3];
                cubic_kernel_derivative(xij, yij, zij, o->h, gw);
                real c = o->bvol[j] * o->rho0;                                      /* :82, :88 */
                real rx = c * gw[0], ry = c * gw[1], rz = c * gw[2];
                bx += rx; by += ry; bz += rz;
                bsq += (rx * rx + ry * ry) + rz * rz;
            });
            denominator = ((((sx * sx + sy * sy) + sz * sz) + square_sum) + bsq) + ((bx * bx + by * by) + bz * bz);  /* :45 */
        } else {
            denominator = ((sx * sx + sy * sy) + sz * sz) + square_sum;             /* :47 */
        }
        if (r_abs(denominator) < R(1e-6)) o->alpha[i] = 0;                          /* :48-51 */
        else o->alpha[i] = o->rho[i] / denominator;
    }
}

/* divergence_warm_start                                            dfsph_solver.py:314-355 */
static void divergence_warm_start(Orc *o)
{
    real *nv = o->vel_adv_delta;   /* scratch: callbacks never read vel, so in-place == buffered */
    PARFOR
    for (int i = 0; i < o->N; ++i) {
        real ax = 0, ay = 0, az = 0;
        const real k_i = o->warm_k[i] / o->dt;                                      /* :333 */
        FOR_FLUID_NEIGHBORS(o, i, {
            real gw[3];
            if (jm_ != 0) {
                if (o->cfg.fs_couple) {
                    cubic_kernel_derivative(xij, yij, zij, o->h, gw);
                    real s = o->rvol[jl] * o->rho0 * k_i / o->rho[i];               /* :345 */
                    ax += s * gw[0]; ay += s * gw[1]; az += s * gw[2];
                }
                continue;
            }
            real k_j = o->warm_k[j] / o->dt;                                        /* :334 */
            cubic_kernel_derivative(xij, yij, zij, o->h, gw);
            real s = o->m * (k_i / o->rho[i] + k_j / o->rho[j]);                    /* :337 */
            ax += s * gw[0]; ay += s * gw[1]; az += s * gw[2];
        });
        if (o->cfg.boundary_handle) {
            real bx = 0, by = 0, bz = 0;
            FOR_WALL_NEIGHBORS_OF_FLUID(o, i, {
                real gw[3];
                cubic_kernel_derivative(xij, yij, zij, o->h, gw);
                real s = o->bvol[j] * k_i / o->rho[i];                              /* :354 */
                bx += s * gw[0]; by += s * gw[1]; bz += s * gw[2];
            });
            nv[3 * i] = o->vel[3 * i] - (ax + bx * o->rho0) * o->dt;                /* :322 */
            nv[3 * i + 1] = o->vel[3 * i + 1] - (ay + by * o->rho0) * o->dt;
            nv[3 * i + 2] = o->vel[3 * i + 2] - (az + bz * o->rho0) * o->dt;
        } else {
            nv[3 * i] = o->vel[3 * i] - ax * o->dt;                                 /* :324 */
            nv[3 * i + 1] = o->vel[3 * i + 1] - ay * o->dt;
            nv[3 * i + 2] = o->vel[3 * i + 2] - az * o->dt;
        }
    }
    memcpy(o->vel, nv, sizeof(real) * 3 * (size_t)o->N);
    memset(o->warm_k, 0, sizeof(real) * (size_t)o->N);                              /* :325 */
}

/* derivative_iter_all_rho                                          dfsph_solver.py:252-300 */
static real derivative_iter_all_rho(Orc *o)
{
    orc_compute_nbr_count(o);                                                       /* :258 */
    PARFOR
    for (int i = 0; i < o->N; ++i) {
        if (o->nbr_cnt[i] < 20) { o->rho_der[i] = 0; continue; }                    /* :259-261 */
        real rd = 0;
        const real vix = o->vel[3 * i], viy = o->vel[3 * i + 1], viz = o->vel[3 * i + 2];
        FOR_FLUID_NEIGHBORS(o, i, {
            real gw[3];
            if (jm_ != 0) {
                if (o->cfg.fs_couple) {
                    cubic_kernel_derivative(xij, yij, zij, o->h, gw);
                    real vj[3];
                    rigid_predicted_velocity(o, jl, 0, vj);                         /* :292-293 */
                    real dvx = vix - vj[0], dvy = viy - vj[1], dvz = viz - vj[2];
                    rd += o->rvol[jl] * o->rho0 * ((dvx * gw[0] + dvy * gw[1]) + dvz * gw[2]);   /* :294 */
                }
                continue;
            }
            cubic_kernel_derivative(xij, yij, zij, o->h, gw);
            real dvx = vix - o->vel[3 * j], dvy = viy - o->vel[3 * j + 1], dvz = viz - o->vel[3 * j + 2];
            rd += o->m * ((dvx * gw[0] + dvy * gw[1]) + dvz * gw[2]);               /* :287 */
        });
        if (o->cfg.boundary_handle) {
            real rb = 0;
            FOR_WALL_NEIGHBORS_OF_FLUID(o, i, {
                real gw[3];
                cubic_kernel_derivative(xij, yij, zij, o->h, gw);
                rb += o->bvol[j] * ((vix * gw[0] + viy * gw[1]) + viz * gw[2]);     /* :300 */
            });
            o->rho_der[i] = r_max(rd + rb * o->rho0, R(0.0));                       /* :267 */
        } else {
            o->rho_der[i] = r_max(rd, R(0.0));                                      /* :269 */
        }
    }
    long cnt = 0;                                                                   /* :275-280 */
    unsigned char *take = (unsigned char *)malloc((size_t)(o->N > 0 ? o->N : 1));
    for (int i = 0; i < o->N; ++i) { take[i] = o->rho_der[i] > 0; cnt += take[i]; }
    const double avg = sched_sum(o, o->rho_der, take, o->N);
    free(take);
    real ret = 0;
    if (cnt > 0) ret = o->sched_seed ? (real)((float)avg / (float)cnt) : R(avg / (double)cnt);
    return ret;
}

/* divergence_iter_all_vel_adv + sum_up_stiff                  dfsph_solver.py:302-312,357-391 */
static void divergence_iter_all_vel_adv(Orc *o)
{
    real *nv = o->vel_adv_delta;
    PARFOR
    for (int i = 0; i < o->N; ++i) {
        real ax = 0, ay = 0, az = 0;
        const real k_i = o->rho_der[i] * o->alpha[i] / o->dt;                       /* :363 */
        FOR_FLUID_NEIGHBORS(o, i, {
            real gw[3];
            if (jm_ != 0) {
                if (o->cfg.fs_couple) {
                    cubic_kernel_derivative(xij, yij, zij, o->h, gw);
                    real s = o->rvol[jl] * o->rho0 * k_i / o->rho[i];               /* :377 */
                    ax += s * gw[0]; ay += s * gw[1]; az += s * gw[2];
                }
                continue;
            }
            real k_j = o->rho_der[j] * o->alpha[j] / o->dt;                         /* :364 */
            cubic_kernel_derivative(xij, yij, zij, o->h, gw);
            real ks = k_i / o->rho[i] + k_j / o->rho[j];
            if (ks > R(1e-5)) {                                                     /* :367 */
                real s = o->m * ks;                                                 /* :369 */
                ax += s * gw[0]; ay += s * gw[1]; az += s * gw[2];
            }
        });
        if (o->cfg.boundary_handle) {
            real bx = 0, by = 0, bz = 0;
            FOR_WALL_NEIGHBORS_OF_FLUID(o, i, {
                real gw[3];
                cubic_kernel_derivative(xij, yij, zij, o->h, gw);
                real s = o->bvol[j] * k_i / o->rho[i];                              /* :390 */
                bx += s * gw[0]; by += s * gw[1]; bz += s * gw[2];
            });
            nv[3 * i] = o->vel[3 * i] - (ax + bx * o->rho0) * o->dt;                /* :310 */
            nv[3 * i + 1] = o->vel[3 * i + 1] - (ay + by * o->rho0) * o->dt;
            nv[3 * i + 2] = o->vel[3 * i + 2] - (az + bz * o->rho0) * o->dt;
        } else {
            nv[3 * i] = o->vel[3 * i] - ax * o->dt;                                 /* :312 */
            nv[3 * i + 1] = o->vel[3 * i + 1] - ay * o->dt;
            nv[3 * i + 2] = o->vel[3 * i + 2] - az * o->dt;
        }
    }
    memcpy(o->vel, nv, sizeof(real) * 3 * (size_t)o->N);
}

static void sum_up_stiff(Orc *o)                                     /* dfsph_solver.py:381-384 */
{
    PARFOR
    for (int i = 0; i < o->N; ++i) o->warm_k[i] += o->rho_der[i] * o->alpha[i];
}

/* correct_divergence_error                                         dfsph_solver.py:393-416 */
static void correct_divergence_error(Orc *o, OrcStepStats *st)
{
    real past = 0;
    int iter_cnt = 0;
    if (o->p_warm_start) divergence_warm_start(o);                                  /* :396-397 */
    real err = derivative_iter_all_rho(o);                                          /* :398 */
    st->n_div_evals = 1;
    st->div_first_err = (float)err;
    while ((iter_cnt < o->p_min_iteration_density_divergence || (double)err > o->p_density_divergence_threshold)
           && iter_cnt < o->p_max_iteration_density_divergence) {                   /* :400 (host f64 compare) */
        divergence_iter_all_vel_adv(o);
        if (o->p_warm_start) sum_up_stiff(o);                                       /* :404-405 */
        past = err;
        err = derivative_iter_all_rho(o);
        st->n_div_evals += 1;
        if (fabs((double)err - (double)past) < 1e-5) break;                         /* :410-412 (host f64) */
        iter_cnt += 1;
    }
    st->n_div = iter_cnt;
    st->div_err = (float)err;
}

/* compute_all_ext_force                                              dfsph_solver.py:91-96 */
static void compute_all_ext_force(Orc *o)
{
    solve_all_tension(o);
    solve_all_viscosity(o);
    const real g[3] = { o->gravity * R(0), o->gravity * R(-1), o->gravity * R(0) };
    PARFOR
    for (int i = 0; i < o->N; ++i)
        for (int a = 0; a < 3; ++a)
            o->force_ext[3 * i + a] = (g[a] + o->tens[3 * i + a]) + o->visc[3 * i + a];   /* :96 */
}

/* compute_all_vel_adv                                               dfsph_solver.py:98-122 */
static void compute_all_vel_adv(Orc *o)
{
    real max_vel = -INFINITY;
    PARFOR
    for (int i = 0; i < o->N; ++i)
        for (int a = 0; a < 3; ++a)
            o->vel_adv[3 * i + a] = o->vel[3 * i + a] + o->dt * o->force_ext[3 * i + a] / o->m;   /* :102 */
    for (int i = 0; i < o->N; ++i) {
        real x = o->vel_adv[3 * i], y = o->vel_adv[3 * i + 1], z = o->vel_adv[3 * i + 2];
        real n = r_sqrt((x * x + y * y) + z * z);
        if (n > max_vel) max_vel = n;                                               /* :103 */
    }
    real max_rigid_vel = 0;                                                         /* :104-110 */
    for (int i = 0; i < o->Nr; ++i) {
        real px = o->rpos[3 * i] - o->centroid[0], py = o->rpos[3 * i + 1] - o->centroid[1], pz = o->rpos[3 * i + 2] - o->centroid[2];
        real cx = o->r_omega[1] * pz - o->r_omega[2] * py;
        real cy = o->r_omega[2] * px - o->r_omega[0] * pz;
        real cz = o->r_omega[0] * py - o->r_omega[1] * px;
        real vn = r_sqrt((o->r_vel[0] * o->r_vel[0] + o->r_vel[1] * o->r_vel[1]) + o->r_vel[2] * o->r_vel[2]);
        real v = vn + r_sqrt((cx * cx + cy * cy) + cz * cz);
        if (v > max_rigid_vel) max_rigid_vel = v;
    }
    max_vel += max_rigid_vel;
    real max_delta_time = o->dt_cfl_num / max_vel * R(0.2);                         /* :112 */
    if (o->p_adaptive_dt) {                                                         /* :113 */
        if (max_delta_time > R(o->p_max_dt)) o->dt = R(o->p_max_dt);                /* :114-117 */
        else o->dt = r_max(max_delta_time, R(o->p_min_dt));
        o->dt2 = o->dt * o->dt;                                                     /* :118 */
        o->ps_dt = o->dt;                                                           /* :119 */
    }
}

/* compute_all_rho_adv                                              dfsph_solver.py:124-176 */
static real compute_all_rho_adv(Orc *o)
{
    PARFOR
    for (int i = 0; i < o->N; ++i) {
        real delta = 0;
        const real vix = o->vel_adv[3 * i], viy = o->vel_adv[3 * i + 1], viz = o->vel_adv[3 * i + 2];
        FOR_FLUID_NEIGHBORS(o, i, {
            real gw[3];
            if (jm_ != 0) {
                if (o->cfg.fs_couple) {
                    cubic_kernel_derivative(xij, yij, zij, o->h, gw);
                    real vj[3];
                    rigid_predicted_velocity(o, jl, 1, vj);                         /* :168-169 (omega + alpha*dt) */
                    real dvx = vix - vj[0], dvy = viy - vj[1], dvz = viz - vj[2];
                    delta += o->rvol[jl] * o->rho0 * ((dvx * gw[0] + dvy * gw[1]) + dvz * gw[2]);   /* :170 */
                }
                continue;
            }
            cubic_kernel_derivative(xij, yij, zij, o->h, gw);
            real dvx = vix - o->vel_adv[3 * j], dvy = viy - o->vel_adv[3 * j + 1], dvz = viz - o->vel_adv[3 * j + 2];
            delta += o->m * ((dvx * gw[0] + dvy * gw[1]) + dvz * gw[2]);            /* :162 */
        });
        if (o->cfg.boundary_handle) {
            real db = 0;
            FOR_WALL_NEIGHBORS_OF_FLUID(o, i, {
                real gw[3];
                cubic_kernel_derivative(xij, yij, zij, o->h, gw);
                db += o->bvol[j] * ((vix * gw[0] + viy * gw[1]) + viz * gw[2]);     /* :176 */
            });
            o->rho_adv[i] = r_max(o->rho[i] + o->dt * (delta + db * o->rho0), o->rho0);   /* :135 */
        } else {
            o->rho_adv[i] = r_max(o->rho[i] + o->dt * delta, o->rho0);              /* :137 */
        }
    }
    long cnt = 0;                                                                   /* :139-141 */
    unsigned char *take = (unsigned char *)malloc((size_t)(o->N > 0 ? o->N : 1));
    for (int i = 0; i < o->N; ++i) { take[i] = !(o->rho_adv[i] == o->rho0); cnt += take[i]; }
    const double rho_avg = sched_sum(o, o->rho_adv, take, o->N);
    free(take);
    real ret = R(1000.0);
    if (cnt > 0) ret = o->sched_seed ? (real)((float)rho_avg / (float)cnt) : R(rho_avg / (double)cnt);   /* :148-149 */
    return ret;
}

/* iter_all_vel_adv                                                 dfsph_solver.py:178-219 */
static void iter_all_vel_adv(Orc *o)
{
    PARFOR
    for (int i = 0; i < o->N; ++i) {
        real ax = 0, ay = 0, az = 0;
        const real k_i = (o->rho_adv[i] - o->rho0) * o->alpha[i] / o->dt2;          /* :199 */
        FOR_FLUID_NEIGHBORS(o, i, {
            real gw[3];
            if (jm_ != 0) {
                if (o->cfg.fs_couple) {
                    cubic_kernel_derivative(xij, yij, zij, o->h, gw);
                    real s = o->rvol[jl] * o->rho0 * k_i / o->rho[i];               /* :211 */
                    ax += s * gw[0]; ay += s * gw[1]; az += s * gw[2];
                }
                continue;
            }
            real k_j = (o->rho_adv[j] - o->rho0) * o->alpha[j] / o->dt2;            /* :200 */
            cubic_kernel_derivative(xij, yij, zij, o->h, gw);
            real s = o->m * (k_i / o->rho[i] + k_j / o->rho[j]);                    /* :203 */
            ax += s * gw[0]; ay += s * gw[1]; az += s * gw[2];
        });
        if (o->cfg.boundary_handle) {
            real bx = 0, by = 0, bz = 0;
            FOR_WALL_NEIGHBORS_OF_FLUID(o, i, {
                real gw[3];
                cubic_kernel_derivative(xij, yij, zij, o->h, gw);
                real s = o->bvol[j] * k_i / o->rho[i];                              /* :219 */
                bx += s * gw[0]; by += s * gw[1]; bz += s * gw[2];
            });
            o->vel_adv_delta[3 * i] = ax + bx * o->rho0;                            /* :187 */
            o->vel_adv_delta[3 * i + 1] = ay + by * o->rho0;
            o->vel_adv_delta[3 * i + 2] = az + bz * o->rho0;
        } else {
            o->vel_adv_delta[3 * i] = ax; o->vel_adv_delta[3 * i + 1] = ay; o->vel_adv_delta[3 * i + 2] = az;
        }
    }
    if (o->exist_rigid && o->active_rigid && o->cfg.fs_couple) rigid_accumulate_force(o);   /* :212 */
    PARFOR
    for (int i = 0; i < 3 * o->N; ++i) o->vel_adv[i] -= o->vel_adv_delta[i] * o->dt;   /* :190-191 */
}

/* compute_all_position                                             dfsph_solver.py:235-250 */
static void compute_all_position(Orc *o)
{
    real lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = R(o->cfg.box_min[a]) + R(o->cfg.particle_radius);
        hi[a] = R(o->cfg.box_max[a]) - R(o->cfg.particle_radius);
    }
    PARFOR
    for (int i = 0; i < o->N; ++i) {
        for (int a = 0; a < 3; ++a) {
            int k = 3 * i + a;
            o->pos[k] = o->pos[k] + o->dt * o->vel_adv[k] * R(0.9999);              /* :238 */
            o->vel[k] = o->vel_adv[k] * R(0.9999);                                  /* :239 */
        }
        if (!o->cfg.boundary_handle) {                                              /* :241-250 */
            for (int a = 0; a < 3; ++a) {
                int k = 3 * i + a;
                if (o->pos[k] <= lo[a]) { o->pos[k] = lo[a]; o->vel[k] *= R(-0.5); }
                if (o->pos[k] >= hi[a]) { o->pos[k] = hi[a]; o->vel[k] *= R(-0.5); }
            }
        }
    }
}

/* dfsph_solver.step                                                dfsph_solver.py:440-445 */
int orc_step_dfsph(Orc *o, int nsteps, int max_dens_iter, OrcStepStats *last)
{
    int capped = 0;
    for (int s = 0; s < nsteps; ++s) {
        OrcStepStats st;
        memset(&st, 0, sizeof(st));
        o->simulate_cnt += 1;                                          /* solver_base.py:137 */
        orc_build_grid(o);                                             /* :139-141; reset() is a no-op, dfsph:418-421 */
        orc_compute_rho(o);                                            /* initialize(): dfsph:423-426 */
        orc_compute_alpha(o);
        correct_divergence_error(o, &st);                              /* iterate(): :428-438 */
        compute_all_ext_force(o);
        compute_all_vel_adv(o);
        {                                                              /* correct_density_error :221-233 */
            real rho_avg = INFINITY;
            int iter_cnt = 0;
            while (iter_cnt < o->p_min_iteration_density || (double)rho_avg - 1000.0 > o->p_density_threshold * 1000 * 0.01) {      /* :225 (host f64) */
                if (max_dens_iter > 0 && iter_cnt >= max_dens_iter) { capped = 1; break; }
                rho_avg = compute_all_rho_adv(o);
                iter_all_vel_adv(o);
                iter_cnt += 1;
            }
            st.n_dens = iter_cnt;
            st.dens_err = (float)(rho_avg - o->rho0);
        }
        compute_all_position(o);
        st.dt = (float)o->dt;
        if (last) *last = st;
    }
    return capped;
}

/* ---------------------------------------------------------------------------------------
 * PCISPH                                                                  pcisph_solver.py
 * press_iter / press_force / rho_err / pos_predict as in the reference; ext_force lives in force_ext.
 * The neighbour SET always comes from the current positions (for_all_neighbor tests particle_j.pos,
 * ParticleSystem.py:462-466); only the kernel argument uses the predicted positions (:155, :167).
 * ------------------------------------------------------------------------------------- */
static void clamp_bounds(const Orc *o, real lo[3], real hi[3])
{
    for (int a = 0; a < 3; ++a) {
        lo[a] = R(o->cfg.box_min[a]) + R(o->cfg.particle_radius);
        hi[a] = R(o->cfg.box_max[a]) - R(o->cfg.particle_radius);
    }
}

/* __init__ :8-26 and pre_compute :28-47 */
static void pcisph_init(Orc *o)
{
    const OrcConfig *c = &o->cfg;
    double r = c->particle_radius;
    double m = 1000 * (r * r * r) * 8;
    double dtf = (double)o->dt;                                        /* self.delta_time[None]: the f32 field read back as a Python float */
    double beta = dtf * dtf * m * m * 2 / (double)(1000 * 1000);       /* :23, Python f64, left to right */
    o->pci_beta = R(beta);
    for (int i = 0; i < o->N; ++i) {                                   /* :13 */
        o->force_ext[3 * i] = o->gravity * R(0); o->force_ext[3 * i + 1] = o->gravity * R(-1); o->force_ext[3 * i + 2] = o->gravity * R(0);
    }
    /* get_max_neighbor_particle_index, ParticleSystem.py:410-422, in single-thread order: atomic_max returns the OLD maximum,
     * so max_index is the last particle whose count ties the running maximum */
    orc_compute_nbr_count(o);
    int max_count = -1, max_index = -1;
    for (int i = 0; i < o->N; ++i) {
        int cnt = o->nbr_cnt[i];
        int old = max_count;
        if (cnt > max_count) max_count = cnt;
        if (old == cnt) max_index = i;
    }
    o->pci_max_index = max_index; o->pci_max_count = max_count;
    real sx = 0, sy = 0, sz = 0, sq = 0;                               /* pre_compute_delta :41-47 */
    if (max_index >= 0) {
        const int i = max_index;
        FOR_FLUID_NEIGHBORS(o, i, {
            real gw[3];
            cubic_kernel_derivative(xij, yij, zij, o->h, gw);           /* compute_sum :179-183 */
            sx += gw[0]; sy += gw[1]; sz += gw[2];
        });
        FOR_FLUID_NEIGHBORS(o, i, {
            real gw[3];
            cubic_kernel_derivative(xij, yij, zij, o->h, gw);           /* compute_square_sum :185-190 */
            sq += (gw[0] * gw[0] + gw[1] * gw[1]) + gw[2] * gw[2];
        });
    }
    o->pci_delta = R(1) / ((((sx * sx + sy * sy) + sz * sz) + sq) * o->pci_beta);   /* :47 */
}

/* predict_vel_pos :73-89 */
static void pci_predict_vel_pos(Orc *o)
{
    real lo[3], hi[3];
    clamp_bounds(o, lo, hi);
    PARFOR
    for (int i = 0; i < o->N; ++i) {
        for (int a = 0; a < 3; ++a) {
            int k = 3 * i + a;
            o->vel_predict[k] = o->vel[k] + o->dt * (o->force_ext[k] + o->press_force[k]) / o->m;   /* :76 */
            o->pos_predict[k] = o->pos[k] + o->dt * o->vel_predict[k];                              /* :77 */
        }
        if (!o->cfg.boundary_handle)
            for (int a = 0; a < 3; ++a) {                                                           /* :79-89 */
                int k = 3 * i + a;
                if (o->pos_predict[k] <= lo[a]) { o->pos_predict[k] = lo[a]; o->vel_predict[k] *= R(-0.5); }
                if (o->pos_predict[k] >= hi[a]) { o->pos_predict[k] = hi[a]; o->vel_predict[k] *= R(-0.5); }
            }
    }
}

/* predict_rho :91-103 */
static void pci_predict_rho(Orc *o)
{
    PARFOR
    for (int i = 0; i < o->N; ++i) {
        const real *pp = o->pos_predict + 3 * i;
        real rho_predict = 0;
        FOR_FLUID_NEIGHBORS(o, i, {
            (void)xij; (void)yij; (void)zij;
            if (jm_ != 0) {
                if (o->cfg.fs_couple) {                                 /* :157-161: predicted fluid position against the body where it is */
                    real x = pp[0] - o->rpos[3 * jl], y = pp[1] - o->rpos[3 * jl + 1], z = pp[2] - o->rpos[3 * jl + 2];
                    real q = r_sqrt((x * x + y * y) + z * z);
                    rho_predict += cubic_kernel(q, o->h) * o->rvol[jl] * o->rho0;
                }
                continue;
            }
            real x = pp[0] - o->pos_predict[3 * j], y = pp[1] - o->pos_predict[3 * j + 1], z = pp[2] - o->pos_predict[3 * j + 2];
            real q = r_sqrt((x * x + y * y) + z * z);                   /* :155 */
            rho_predict += cubic_kernel(q, o->h) * o->m;                /* :156 */
        });
        if (o->cfg.boundary_handle) {
            real rho_boundary = 0;
            FOR_WALL_NEIGHBORS_OF_FLUID(o, i, {
                (void)xij; (void)yij; (void)zij;
                real x = pp[0] - o->bpos[3 * j], y = pp[1] - o->bpos[3 * j + 1], z = pp[2] - o->bpos[3 * j + 2];
                real q = r_sqrt((x * x + y * y) + z * z);               /* :167 */
                rho_boundary += cubic_kernel(q, o->h) * o->bvol[j];     /* :168 */
            });
            o->rho_adv[i] = rho_predict + rho_boundary * o->rho0;       /* :100  (rho_predict field kept in rho_adv) */
        } else {
            o->rho_adv[i] = rho_predict;
        }
        o->rho_err[i] = o->rho_adv[i] - o->rho0;                        /* :103 */
    }
}

/* compute_residual :126-138 (mean accumulated in f64: the reference's f32 atomic order is unspecified) */
static real pci_compute_residual(Orc *o)
{
    double sum = 0; long cnt = 0;
    for (int i = 0; i < o->N; ++i) {
        real err = r_max(o->rho_err[i], R(0.0));
        if (err > 0) { sum += (double)err; cnt += 1; }
    }
    return cnt > 0 ? R(sum / (double)cnt) : R(0);
}

/* iter_press :105-109 */
static void pci_iter_press(Orc *o)
{
    PARFOR
    for (int i = 0; i < o->N; ++i) {
        o->press_iter[i] += o->rho_err[i] * o->pci_delta;
        o->press_iter[i] = r_max(R(0.0), o->press_iter[i]);
    }
}

/* update_press_force :111-124 */
static void pci_update_press_force(Orc *o)
{
    const real rho0_sq = R(1000 * 1000);                                /* self.rho_0 ** 2, Python int */
    PARFOR
    for (int i = 0; i < o->N; ++i) {
        real fx = 0, fy = 0, fz = 0;
        const real p_i = o->press_iter[i];
        FOR_FLUID_NEIGHBORS(o, i, {
            real gw[3];
            if (jm_ != 0) {
                if (o->cfg.fs_couple) {                                 /* :200-211 */
                    cubic_kernel_derivative(xij, yij, zij, o->h, gw);
                    real a = o->rvol[jl] * o->rho0 * p_i;
                    real den = o->rho[i] * o->rho[i];
                    fx += a * gw[0] / den * o->m; fy += a * gw[1] / den * o->m; fz += a * gw[2] / den * o->m;
                }
                continue;
            }
            cubic_kernel_derivative(xij, yij, zij, o->h, gw);
            real ps = p_i + o->press_iter[j];
            fx += ps * gw[0] / rho0_sq * o->m * o->m;                   /* :199 */
            fy += ps * gw[1] / rho0_sq * o->m * o->m;
            fz += ps * gw[2] / rho0_sq * o->m * o->m;
        });
        if (o->cfg.boundary_handle) {
            real bx = 0, by = 0, bz = 0;
            const real rho_i = o->rho[i];
            const real rho_i_2 = rho_i * rho_i;                         /* :221 */
            FOR_WALL_NEIGHBORS_OF_FLUID(o, i, {
                real gw[3];
                cubic_kernel_derivative(xij, yij, zij, o->h, gw);
                real s = o->bvol[j] * p_i / rho_i_2;                    /* :223 */
                bx -= s * gw[0]; by -= s * gw[1]; bz -= s * gw[2];
            });
            o->press_force[3 * i] = -fx + bx * o->rho0 * o->m;          /* :120 */
            o->press_force[3 * i + 1] = -fy + by * o->rho0 * o->m;
            o->press_force[3 * i + 2] = -fz + bz * o->rho0 * o->m;
        } else {
            o->press_force[3 * i] = -fx; o->press_force[3 * i + 1] = -fy; o->press_force[3 * i + 2] = -fz;   /* :122 */
        }
    }
}

/* integration :226-245 */
static void pci_integration(Orc *o)
{
    real lo[3], hi[3];
    clamp_bounds(o, lo, hi);
    PARFOR
    for (int i = 0; i < o->N; ++i) {
        for (int a = 0; a < 3; ++a) {
            int k = 3 * i + a;
            o->vel[k] = o->vel[k] + o->dt * (o->force_ext[k] + o->press_force[k]) / o->m;   /* :229-230 */
            o->vel[k] *= R(0.9999);                                                         /* :231 */
            o->pos[k] = o->pos[k] + o->dt * o->vel[k];                                      /* :232 */
        }
        if (!o->cfg.boundary_handle)
            for (int a = 0; a < 3; ++a) {
                int k = 3 * i + a;
                if (o->pos[k] <= lo[a]) { o->pos[k] = lo[a]; o->vel[k] *= R(-0.5); }
                if (o->pos[k] >= hi[a]) { o->pos[k] = hi[a]; o->vel[k] *= R(-0.5); }
            }
    }
}

/* pcisph_solver.step :252-259 */
int orc_step_pcisph(Orc *o, int nsteps, OrcStepStats *last)
{
    if (o->cfg.solver != 2) return -1;
    int capped = 0;
    for (int s = 0; s < nsteps; ++s) {
        OrcStepStats st;
        memset(&st, 0, sizeof(st));
        o->simulate_cnt += 1;                                          /* solver_base.py:137 */
        orc_build_grid(o);                                             /* :139-141 */
        memset(o->press_iter, 0, sizeof(real) * (size_t)o->N);         /* reset() :247-250 */
        memset(o->press_force, 0, sizeof(real) * 3 * (size_t)o->N);
        orc_compute_rho(o);                                            /* compute_ext_force :237-244 */
        solve_all_tension(o);
        solve_all_viscosity(o);
        {
            const real g[3] = { o->gravity * R(0), o->gravity * R(-1), o->gravity * R(0) };
            PARFOR
            for (int i = 0; i < o->N; ++i)
                for (int a = 0; a < 3; ++a)
                    o->force_ext[3 * i + a] = (g[a] + o->tens[3 * i + a]) + o->visc[3 * i + a];   /* :243 */
        }
        int iter_cnt = 0;                                              /* iteration() :49-71 */
        pci_predict_vel_pos(o);
        pci_predict_rho(o);
        real rho_err_avg = pci_compute_residual(o);
        while (((double)rho_err_avg > 1000 * 0.1 * 0.01 || iter_cnt < 1) && iter_cnt < 80) {   /* :58 (host f64) */
            pci_iter_press(o);
            pci_update_press_force(o);
            if (rigid_coupled(o)) rigid_accumulate_force_mode(o, 2);   /* :209, once per iteration: the reset at :60 is commented out */
            pci_predict_vel_pos(o);
            pci_predict_rho(o);
            rho_err_avg = pci_compute_residual(o);
            iter_cnt += 1;
        }
        if (iter_cnt >= 80) capped = 1;
        st.n_dens = iter_cnt;
        st.dens_err = (float)rho_err_avg;
        pci_integration(o);
        st.dt = (float)o->dt;
        if (last) *last = st;
    }
    return capped;
}

/* ---------------------------------------------------------------------------------------
 * IISPH                                                                    iisph_solver.py
 * v_adv -> vel_adv, f_adv -> force_ext, p_iter -> press_iter, f_press -> press_force.
 * cubic_kernel_derivative(-q) == -cubic_kernel_derivative(q) bit for bit (the norm is even, s*(-x) == -(s*x)).
 * ------------------------------------------------------------------------------------- */
static void iisph_init(Orc *o)
{
    for (int i = 0; i < o->N; ++i) {                                   /* :18: gravity * (0,-1,0) * m */
        o->force_ext[3 * i] = o->gravity * R(0.0) * o->m; o->force_ext[3 * i + 1] = o->gravity * R(-1.0) * o->m; o->force_ext[3 * i + 2] = o->gravity * R(0.0) * o->m;
    }
}

/* predict_advection :36-82 */
static void iisph_predict_advection(Orc *o)
{
    orc_compute_rho(o);                                                /* :38 */
    solve_all_tension(o);                                              /* :43 */
    solve_all_viscosity(o);                                            /* :44 */
    const real g[3] = { o->gravity * R(0), o->gravity * R(-1), o->gravity * R(0) };
    PARFOR
    for (int i = 0; i < o->N; ++i)
        for (int a = 0; a < 3; ++a)
            o->force_ext[3 * i + a] = (g[a] + o->tens[3 * i + a]) + o->visc[3 * i + a];       /* :46 */
    PARFOR
    for (int i = 0; i < o->N; ++i) {                                                          /* :47-56 */
        for (int a = 0; a < 3; ++a)
            o->vel_adv[3 * i + a] = o->vel[3 * i + a] + o->dt * o->force_ext[3 * i + a] / o->m;   /* :48 */
        const real rho_i = o->rho[i];
        real dx = 0, dy = 0, dz = 0;
        FOR_FLUID_NEIGHBORS(o, i, {
            real gw[3];
            if (jm_ != 0) {
                if (o->cfg.fs_couple) {
                    cubic_kernel_derivative(xij, yij, zij, o->h, gw);
                    real sr = -o->rvol[jl] * o->rho0 / (rho_i * rho_i);  /* compute_d_ii :286 */
                    dx += sr * gw[0]; dy += sr * gw[1]; dz += sr * gw[2];
                }
                continue;
            }
            cubic_kernel_derivative(xij, yij, zij, o->h, gw);
            real s = -o->m / (rho_i * rho_i);                           /* compute_d_ii :280 */
            dx += s * gw[0]; dy += s * gw[1]; dz += s * gw[2];
        });
        if (o->cfg.boundary_handle) {
            real bx = 0, by = 0, bz = 0;
            FOR_WALL_NEIGHBORS_OF_FLUID(o, i, {
                real gw[3];
                cubic_kernel_derivative(xij, yij, zij, o->h, gw);
                real s = -o->bvol[j] / (rho_i * rho_i);                 /* compute_boundary_d_ii :292 */
                bx += s * gw[0]; by += s * gw[1]; bz += s * gw[2];
            });
            o->d_ii[3 * i] = (dx + bx * o->rho0) * o->dt * o->dt;       /* :54 */
            o->d_ii[3 * i + 1] = (dy + by * o->rho0) * o->dt * o->dt;
            o->d_ii[3 * i + 2] = (dz + bz * o->rho0) * o->dt * o->dt;
        } else {
            o->d_ii[3 * i] = dx * o->dt * o->dt; o->d_ii[3 * i + 1] = dy * o->dt * o->dt; o->d_ii[3 * i + 2] = dz * o->dt * o->dt;   /* :56 */
        }
    }
    PARFOR
    for (int i = 0; i < o->N; ++i) {                                                          /* :58-82 */
        const real rho_i = o->rho[i];
        const real vx = o->vel_adv[3 * i], vy = o->vel_adv[3 * i + 1], vz = o->vel_adv[3 * i + 2];
        const real dix = o->d_ii[3 * i], diy = o->d_ii[3 * i + 1], diz = o->d_ii[3 * i + 2];
        const real cji = -o->dt * o->dt * o->m / (rho_i * rho_i);       /* scalar prefix of d_ji :302-303 */
        real ra = 0, aii = 0;
        FOR_FLUID_NEIGHBORS(o, i, {
            real gw[3];
            if (jm_ != 0) {
                if (o->cfg.fs_couple) {                                 /* compute_rho_adv :333-342 */
                    cubic_kernel_derivative(xij, yij, zij, o->h, gw);
                    real vj[3];
                    rigid_predicted_velocity(o, jl, 1, vj);
                    real ux = vx - vj[0], uy = vy - vj[1], uz = vz - vj[2];
                    ra += o->rvol[jl] * ((ux * gw[0] + uy * gw[1]) + uz * gw[2]) * o->rho0;
                }
                continue;
            }
            cubic_kernel_derivative(xij, yij, zij, o->h, gw);
            real ux = vx - o->vel_adv[3 * j], uy = vy - o->vel_adv[3 * j + 1], uz = vz - o->vel_adv[3 * j + 2];
            ra += o->m * ((ux * gw[0] + uy * gw[1]) + uz * gw[2]);      /* compute_rho_adv :332 */
        });
        FOR_FLUID_NEIGHBORS(o, i, {
            real gw[3];
            cubic_kernel_derivative(xij, yij, zij, o->h, gw);
            real ex = dix - cji * -gw[0], ey = diy - cji * -gw[1], ez = diz - cji * -gw[2];
            if (jm_ != 0) {
                if (o->cfg.fs_couple) aii += o->rvol[jl] * ((ex * gw[0] + ey * gw[1]) + ez * gw[2]) * o->rho0;   /* compute_a_ii :305-312 */
                continue;
            }
            aii += o->m * ((ex * gw[0] + ey * gw[1]) + ez * gw[2]);     /* compute_a_ii :304 */
        });
        if (o->cfg.boundary_handle) {
            real rb = 0, ab = 0;
            FOR_WALL_NEIGHBORS_OF_FLUID(o, i, {
                real gw[3];
                cubic_kernel_derivative(xij, yij, zij, o->h, gw);
                rb += o->bvol[j] * ((vx * gw[0] + vy * gw[1]) + vz * gw[2]);                  /* :349 */
            });
            FOR_WALL_NEIGHBORS_OF_FLUID(o, i, {
                real gw[3];
                cubic_kernel_derivative(xij, yij, zij, o->h, gw);
                real ex = dix - cji * -gw[0], ey = diy - cji * -gw[1], ez = diz - cji * -gw[2];
                ab += o->bvol[j] * ((ex * gw[0] + ey * gw[1]) + ez * gw[2]);                  /* compute_a_ii_boundary :322 */
            });
            o->rho_adv[i] = (ra + rb * o->rho0) * o->dt + rho_i;        /* :64 */
            o->a_ii[i] = aii + ab * o->rho0;                            /* :75 */
        } else {
            o->rho_adv[i] = ra * o->dt + rho_i;                         /* :67 */
            o->a_ii[i] = aii;
        }
        o->press_iter[i] = R(0.5) * o->p_past[i];                       /* :68 */
    }
}

/* compute_all_d_ij :130-135 */
static void iisph_compute_all_d_ij(Orc *o)
{
    PARFOR
    for (int i = 0; i < o->N; ++i) {
        real dx = 0, dy = 0, dz = 0;
        FOR_FLUID_NEIGHBORS(o, i, {
            if (jm_ != 0) continue;
            real gw[3];
            cubic_kernel_derivative(xij, yij, zij, o->h, gw);
            real a = -o->m * o->press_iter[j];
            real den = o->rho[j] * o->rho[j];
            dx += a * gw[0] / den; dy += a * gw[1] / den; dz += a * gw[2] / den;   /* compute_d_ij :327 */
        });
        o->d_ij[3 * i] = dx * o->dt * o->dt; o->d_ij[3 * i + 1] = dy * o->dt * o->dt; o->d_ij[3 * i + 2] = dz * o->dt * o->dt;   /* :135 */
    }
}

/* update_p :137-157 */
static void iisph_update_p(Orc *o)
{
    PARFOR
    for (int i = 0; i < o->N; ++i) {
        const real rho_i = o->rho[i];
        const real p_i = o->press_iter[i];
        const real cji = -o->dt * o->dt * o->m / (rho_i * rho_i);       /* :252-253 */
        const real ax = o->d_ij[3 * i], ay = o->d_ij[3 * i + 1], az = o->d_ij[3 * i + 2];
        real sum = 0, bsum = 0;
        FOR_FLUID_NEIGHBORS(o, i, {
            real gw[3];
            cubic_kernel_derivative(xij, yij, zij, o->h, gw);
            if (jm_ != 0) {
                if (o->cfg.fs_couple) sum += ((ax * gw[0] + ay * gw[1]) + az * gw[2]) * o->rvol[jl] * o->rho0;   /* sum_factor :256-261 */
                continue;
            }
            const real p_j = o->press_iter[j];
            real djx = cji * -gw[0] * p_i, djy = cji * -gw[1] * p_i, djz = cji * -gw[2] * p_i;   /* d_ji */
            real tx = ax - o->d_ii[3 * j] * p_j - (o->d_ij[3 * j] - djx);
            real ty = ay - o->d_ii[3 * j + 1] * p_j - (o->d_ij[3 * j + 1] - djy);
            real tz = az - o->d_ii[3 * j + 2] * p_j - (o->d_ij[3 * j + 2] - djz);
            sum += o->m * ((tx * gw[0] + ty * gw[1]) + tz * gw[2]);     /* sum_factor :254 */
        });
        if (o->cfg.boundary_handle) {
            FOR_WALL_NEIGHBORS_OF_FLUID(o, i, {
                real gw[3];
                cubic_kernel_derivative(xij, yij, zij, o->h, gw);
                bsum += ((ax * gw[0] + ay * gw[1]) + az * gw[2]) * o->bvol[j] * o->rho0;   /* sum_factor_boundary :240 */
            });
            o->r_sum[i] = sum + bsum;                                   /* :145 */
        } else {
            o->r_sum[i] = sum;
        }
    }
    PARFOR
    for (int i = 0; i < o->N; ++i) {                                    /* :148-153 */
        if (r_abs(o->a_ii[i]) > R(1e-7))
            o->p_new[i] = R(0.5) * o->press_iter[i] + R(0.5) * ((o->rho0 - o->rho_adv[i]) - o->r_sum[i]) / o->a_ii[i];
        else
            o->p_new[i] = R(0.0);
    }
    PARFOR
    for (int i = 0; i < o->N; ++i) o->press_iter[i] = r_max(o->p_new[i], R(0.0));   /* :155-156 */
}

/* compute_residual :110-121 */
static real iisph_compute_residual(Orc *o)
{
    double sum = 0; long cnt = 0;
    for (int i = 0; i < o->N; ++i)
        if (o->press_iter[i] > 0) {
            sum += (double)(((o->a_ii[i] * o->press_iter[i] + o->r_sum[i]) + o->rho_adv[i]) - R(1000));   /* :116 */
            cnt += 1;
        }
    return cnt > 0 ? R(sum / (double)cnt) : R(0);
}

/* intergation :189-210 with compute_all_press_force :172-185 inlined */
static void iisph_integration(Orc *o)
{
    real lo[3], hi[3];
    clamp_bounds(o, lo, hi);
    PARFOR
    for (int i = 0; i < o->N; ++i) {
        for (int a = 0; a < 3; ++a) {
            int k = 3 * i + a;
            o->press_force[k] = (o->d_ij[k] + o->d_ii[k] * o->press_iter[i]) * o->m / (o->dt * o->dt);   /* :176 */
            o->vel[k] = o->vel_adv[k] + o->dt * o->press_force[k] / o->m;                                /* :193 */
            o->vel[k] *= R(0.9999);                                                                      /* :194 */
            o->pos[k] = o->pos[k] + o->dt * o->vel[k];                                                   /* :195 */
        }
        if (!o->cfg.boundary_handle)
            for (int a = 0; a < 3; ++a) {                                                                /* :198-207 */
                int k = 3 * i + a;
                if (o->pos[k] <= lo[a]) { o->pos[k] = lo[a]; o->vel[k] *= R(-0.5); }
                if (o->pos[k] >= hi[a]) { o->pos[k] = hi[a]; o->vel[k] *= R(-0.5); }
            }
        o->p_past[i] = o->press_iter[i];                                                                 /* :209-210 */
    }
}

/* iisph_solver.step :340-347 */
int orc_step_iisph(Orc *o, int nsteps, OrcStepStats *last)
{
    if (o->cfg.solver != 3) return -1;
    int capped = 0;
    for (int s = 0; s < nsteps; ++s) {
        OrcStepStats st;
        memset(&st, 0, sizeof(st));
        o->simulate_cnt += 1;                                          /* solver_base.py:137; reset() is a no-op :31-33 */
        orc_build_grid(o);
        iisph_predict_advection(o);
        int l = 0;                                                     /* pressure_solve :85-108 */
        double residual = INFINITY, prev = 0;
        int have_prev = 0;
        const double err = 0.1 * 1000 * 0.01;                          /* :88 */
        while ((residual > err || l < 1) && l < 180) {
            iisph_compute_all_d_ij(o);
            iisph_update_p(o);
            l += 1;
            residual = (double)iisph_compute_residual(o);
            if (have_prev && residual - prev > 0) { st.n_div = 1; break; }   /* "Iteration trend to divergence" :97-99 */
            prev = residual; have_prev = 1;
        }
        if (l >= 180) capped = 1;
        st.n_dens = l;
        st.dens_err = (float)residual;
        if (rigid_coupled(o)) rigid_accumulate_force_mode(o, 3);       /* compute_all_press_force :172-179, before the positions move */
        iisph_integration(o);
        st.dt = (float)o->dt;
        if (last) *last = st;
    }
    return capped;
}

/* ---------------------------------------------------------------------------------------
 * PBF                                                                        pbf_solver.py
 *
 * The file is STALE at the surveyed commit: its fluid callbacks are written against an older ParticleSystem.for_all_neighbor that
 * passed particle INDICES (compute_rho(self, i, j) reads fluid_particles.pos[i], :169-170), while for_all_neighbor now passes particle
 * STRUCTS (ParticleSystem.py:469; the index form is the commented-out line :468) -- as committed the solver does not compile.  The
 * restatement reads every callback the way the other solvers' callbacks were updated: `fluid_particles.X[i]` -> particle_i.X,
 * `fluid_particles.X[j]` -> particle_j.X, `pbf_lambda[j]` -> pbf_lambda[particle_j.index]; the boundary callbacks still take indices
 * (for_all_boundary_neighbor, ParticleSystem.py:337-366) and are restated as written.
 *
 * update_all_pos (:63-95) is one parallel loop in which every particle first WRITES its own pos_predict / vel / pos and then READS
 * its neighbours' pos and vel, and finally adds the viscosity term to its own vel: a data race, the outcome depends on the thread
 * schedule.  The schedule restated here (and by the HIP kernels) is the one every barrier-synchronised implementation produces:
 *   phase 1  all particles: pos_predict += delta_pos, vel = (pos_predict - pos) / dt, clamp, pos = pos_predict
 *   phase 2  all particles: v_i = sum_j (vel_j - vel_i) W_poly6(|pos_i - pos_j|)   on the NEW positions and phase-1 velocities,
 *            the candidates being those of the cell lists of the step's start (belong_grid is not updated inside a step)
 *   phase 3  all particles: vel_i += c * v_i
 * i.e. the interleaving "every write of phase 1 before any read of phase 2, every read of phase 2 before any write of phase 3",
 * which is one legal execution of the reference loop.  The wall part of the viscosity sum is computed and discarded by the
 * reference (:91 is commented out); it is not restated.  No rigid coupling (the callbacks have no material branches).
 * ------------------------------------------------------------------------------------- */
static inline real poly_kernel(real r, real h)                          /* solver_base.py:123-129 */
{
    real q = r / h;
    real q2 = q * q;
    real ret = R(0.0);
    if (q <= R(1.0)) ret = R(315.0) / (R(64 * ORC_PI) * pow3(h)) * pow3(R(1.0) - q2);
    return ret;
}
static inline void spiky_kernel_derivative(real rx, real ry, real rz, real h, real out[3])   /* solver_base.py:114-121 */
{
    real r_norm = r_sqrt((rx * rx + ry * ry) + rz * rz);
    real q = r_norm / h;
    out[0] = out[1] = out[2] = R(0.0);
    if (q <= R(1.0) && q > R(0.0)) {
        real a = -(R(45.0) * pow2(R(1.0) - q));
        real den = R(ORC_PI) * pow2(pow2(h)) * r_norm;
        out[0] = a * rx / den; out[1] = a * ry / den; out[2] = a * rz / den;
    }
}

/* compute_all_rho (solver_base.py:36-50) with pbf_solver's callbacks (pbf_solver.py:166-174): rho only, nothing else touched */
static void pbf_compute_rho(Orc *o)
{
    PARFOR
    for (int i = 0; i < o->N; ++i) {
        real rho = R(0.001);                                                            /* solver_base.py:44 */
        FOR_FLUID_NEIGHBORS(o, i, { if (jm_ == 0) rho += o->m * poly_kernel(r_sqrt((xij * xij + yij * yij) + zij * zij), o->h); });   /* :169-170 */
        if (o->cfg.boundary_handle) {
            real rb = R(0.0);
            FOR_WALL_NEIGHBORS_OF_FLUID(o, i, { rb += o->bvol[j] * poly_kernel(r_sqrt((xij * xij + yij * yij) + zij * zij), o->h); });   /* :173-176 */
            o->rho[i] = rho + rb * o->rho0;
        } else {
            o->rho[i] = rho;
        }
    }
}

int orc_step_pbf(Orc *o, int nsteps)
{
    if (o->cfg.solver != 4) return -1;
    const real eps = R(1.0e-6), k_tension = R(1e-7), c_visc = R(9e-6);                  /* :17-21 */
    const real corr_r = R(0.3 * (o->cfg.particle_radius * 4));                          /* s_corr_factor * kernel_h, both Python floats */
    real lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {                                                       /* :74-81: clamp at particle_radius */
        lo[a] = R(o->cfg.box_min[a]) + R(o->cfg.particle_radius);
        hi[a] = R(o->cfg.box_max[a]) - R(o->cfg.particle_radius);
    }
    for (int s = 0; s < nsteps; ++s) {
        o->simulate_cnt += 1;                                                           /* solver_base.py:137 */
        orc_build_grid(o);                                                              /* :139-141 */
        PARFOR
        for (int i = 0; i < o->N; ++i) {
            o->acc[3 * i] = o->gravity * R(0.0);                                        /* reset(), solver_base.py:131-133 */
            o->acc[3 * i + 1] = o->gravity * R(-1.0);
            o->acc[3 * i + 2] = o->gravity * R(0.0);
            for (int a = 0; a < 3; ++a) {                                               /* externel_force_predict_pos :26-29 */
                o->vel[3 * i + a] += o->dt * o->acc[3 * i + a];
                o->pos_predict[3 * i + a] = o->pos[3 * i + a] + o->dt * o->vel[3 * i + a];
            }
        }
        /* compute_all_lambda :32-52: rho (poly6), constrain, constrain_derivative, lambda -- all on the CURRENT positions */
        PARFOR
        for (int i = 0; i < o->N; ++i) {
            real rho = R(0.001);                                                        /* solver_base.py:44 */
            FOR_FLUID_NEIGHBORS(o, i, { if (jm_ == 0) rho += o->m * poly_kernel(r_sqrt((xij * xij + yij * yij) + zij * zij), o->h); });   /* :169-170 */
            if (o->cfg.boundary_handle) {
                real rb = R(0.0);
                FOR_WALL_NEIGHBORS_OF_FLUID(o, i, { rb += o->bvol[j] * poly_kernel(r_sqrt((xij * xij + yij * yij) + zij * zij), o->h); });   /* :173-176 */
                o->rho[i] = rho + rb * o->rho0;
            } else {
                o->rho[i] = rho;
            }
            o->pbf_c[i] = r_max(o->rho[i] / o->rho0 - R(1.0), R(0.0));                  /* :127-128 */
        }
        PARFOR
        for (int i = 0; i < o->N; ++i) {
            real cd[3] = {0, 0, 0}, cdb[3] = {0, 0, 0}, g[3];
            FOR_FLUID_NEIGHBORS(o, i, { if (jm_ == 0) { spiky_kernel_derivative(xij, yij, zij, o->h, g);
                                                        cd[0] += g[0] / o->rho0; cd[1] += g[1] / o->rho0; cd[2] += g[2] / o->rho0; } });   /* :116-117 */
            if (o->cfg.boundary_handle) {
                FOR_WALL_NEIGHBORS_OF_FLUID(o, i, { spiky_kernel_derivative(xij, yij, zij, o->h, g);
                                                    cdb[0] += g[0] / o->rho0; cdb[1] += g[1] / o->rho0; cdb[2] += g[2] / o->rho0; });    /* :120-122 */
                for (int a = 0; a < 3; ++a) o->pbf_cd[3 * i + a] = cd[a] + cdb[a];      /* :112 */
            } else {
                for (int a = 0; a < 3; ++a) o->pbf_cd[3 * i + a] = cd[a];
            }
        }
        PARFOR
        for (int i = 0; i < o->N; ++i) {
            if (o->pbf_c[i] == R(0.0)) { o->pbf_lambda[i] = R(0.0); continue; }         /* :39-40 */
            real sum = R(0.0), g[3];
            FOR_FLUID_NEIGHBORS(o, i, { if (jm_ == 0) { spiky_kernel_derivative(xij, yij, zij, o->h, g);
                                                        real gx = g[0] / o->rho0, gy = g[1] / o->rho0, gz = g[2] / o->rho0;
                                                        sum += (gx * gx + gy * gy) + gz * gz; } });                                       /* :133-134 */
            const real *cd = o->pbf_cd + 3 * i;
            const real cdcd = (cd[0] * cd[0] + cd[1] * cd[1]) + cd[2] * cd[2];
            if (o->cfg.boundary_handle) {
                real sb = R(0.0);
                FOR_WALL_NEIGHBORS_OF_FLUID(o, i, { spiky_kernel_derivative(xij, yij, zij, o->h, g);
                                                    real gx = g[0] / o->rho0, gy = g[1] / o->rho0, gz = g[2] / o->rho0;
                                                    sb += (gx * gx + gy * gy) + gz * gz; });                                              /* :139-140 */
                sum = (cdcd + sum) + sb;                                                /* :48 */
            } else {
                sum = cdcd + sum;                                                       /* :50 */
            }
            o->pbf_lambda[i] = -o->pbf_c[i] / (sum + eps);                              /* :52 */
        }
        /* compute_all_delta_pos :55-64 */
        const real w_corr = poly_kernel(corr_r, o->h);
        PARFOR
        for (int i = 0; i < o->N; ++i) {
            real dp[3] = {0, 0, 0}, dpb[3] = {0, 0, 0}, g[3];
            const real li = o->pbf_lambda[i];
            FOR_FLUID_NEIGHBORS(o, i, { if (jm_ == 0) {
                real sc = poly_kernel(r_sqrt((xij * xij + yij * yij) + zij * zij), o->h) / w_corr;                                        /* :148 */
                sc *= sc; sc *= sc; sc *= -k_tension;                                                                                     /* :149-151 */
                spiky_kernel_derivative(xij, yij, zij, o->h, g);
                const real f = (li + o->pbf_lambda[j]) + sc;                                                                              /* :153 */
                dp[0] += f * g[0]; dp[1] += f * g[1]; dp[2] += f * g[2]; } });
            if (o->cfg.boundary_handle) {
                FOR_WALL_NEIGHBORS_OF_FLUID(o, i, {
                    real sc = poly_kernel(r_sqrt((xij * xij + yij * yij) + zij * zij), o->h) / w_corr;
                    sc *= sc; sc *= sc; sc *= -k_tension;
                    spiky_kernel_derivative(xij, yij, zij, o->h, g);
                    const real f = li + sc;                                                                                               /* :164 */
                    dpb[0] += f * g[0]; dpb[1] += f * g[1]; dpb[2] += f * g[2]; });
                for (int a = 0; a < 3; ++a) o->pbf_dpos[3 * i + a] = (dp[a] + dpb[a]) / o->rho0;                                          /* :62 */
            } else {
                for (int a = 0; a < 3; ++a) o->pbf_dpos[3 * i + a] = dp[a] / o->rho0;                                                     /* :64 */
            }
        }
        /* update_all_pos :66-95, phase 1 */
        PARFOR
        for (int i = 0; i < o->N; ++i) {
            for (int a = 0; a < 3; ++a) {
                const int k = 3 * i + a;
                o->pos_predict[k] += o->pbf_dpos[k];                                    /* :69 */
                o->vel[k] = (o->pos_predict[k] - o->pos[k]) / o->dt;                    /* :70 */
                if (!o->cfg.boundary_handle) {                                          /* :73-81 */
                    if (o->pos_predict[k] <= lo[a]) { o->pos_predict[k] = lo[a]; o->vel[k] *= R(0.5); }
                    if (o->pos_predict[k] >= hi[a]) { o->pos_predict[k] = hi[a]; o->vel[k] *= R(0.5); }
                }
                o->pos[k] = o->pos_predict[k];                                          /* :84 */
            }
        }
        /* phases 2 and 3: new positions, the step's cell lists (cell3 is not rebuilt), phase-1 velocities */
        real *v = o->vel_predict;
        PARFOR
        for (int i = 0; i < o->N; ++i) {
            real acc[3] = {0, 0, 0};
            FOR_FLUID_NEIGHBORS(o, i, { if (jm_ == 0) {
                const real w = poly_kernel(r_sqrt((xij * xij + yij * yij) + zij * zij), o->h);                                            /* :98 */
                acc[0] += (o->vel[3 * j] - o->vel[3 * i]) * w; acc[1] += (o->vel[3 * j + 1] - o->vel[3 * i + 1]) * w;
                acc[2] += (o->vel[3 * j + 2] - o->vel[3 * i + 2]) * w; } });
            v[3 * i] = acc[0]; v[3 * i + 1] = acc[1]; v[3 * i + 2] = acc[2];
        }
        PARFOR
        for (int i = 0; i < 3 * o->N; ++i) o->vel[i] += c_visc * v[i];                  /* :92 / :94 */
    }
    return 0;
}
