/*
 * sph_oracle.h -- CPU ORACLE for the SPH per-step path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the arithmetic in the reference
 * (Jukgei/CFD_Taichi @ 2024_08_07): ParticleSystem.py, solver_base.py,
 * wcsph_solver.py, dfsph_solver.py, pcisph_solver.py, iisph_solver.py, rigid_solver.py.  It is NOT Taichi and NOT the reference itself.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this
 * path, and its runtime (taichi==1.6.0) is not installed in this image (plain
 * ModuleNotFoundError; nothing was refused), so the restatement is pinned only by the
 * analytic known-answer tests in tests/golden/kats.json.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (cfd_taichi_amd/) never links, imports or calls it.
 *
 * Conventions: all field arithmetic is f32 (ORC_REAL=float; build
 * with -DORC_REAL=double for the f64 error-attribution variant), FP contraction off,
 * Python-scalar sub-expressions are folded in f64 and then rounded to f32; cell lists
 * are in the single-thread Taichi order: ascending particle index inside a cell.
 */
#ifndef SPH_ORACLE_H
#define SPH_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OrcConfig {
    double box_min[3];
    double box_max[3];
    double particle_radius;
    double gravity;
    double delta_time;
    double start_pos[3];
    double water_size[3];
    int boundary_handle; /* 1 = Akinci wall particles (default), 0 = clamp walls */
    int fs_couple;       /* unused until rigid coupling is restated */
    int solver;          /* 0 = wcsph, 1 = dfsph, 2 = pcisph, 3 = iisph, 4 = pbf */
    int num_threads;     /* OpenMP threads for the sweeps; results do not depend on it */
} OrcConfig;

typedef struct OrcStepStats {
    int n_div;          /* divergence-solver iteration count printed at dfsph_solver.py:416 */
    int n_dens;         /* density-solver iteration count printed at dfsph_solver.py:233 */
    int n_div_evals;    /* number of derivative_iter_all_rho evaluations this step */
    float div_first_err;
    float div_err;
    float dens_err;     /* rho_avg - rho_0 */
    float dt;           /* delta_time after the step */
} OrcStepStats;

/* field ids for orc_get / orc_set (fluid particles unless stated) */
enum {
    ORC_F_POS = 0, ORC_F_VEL = 1, ORC_F_ACC = 2, ORC_F_RHO = 3, ORC_F_PRESSURE = 4,
    ORC_F_ALPHA = 5, ORC_F_WARM_K = 6, ORC_F_RHO_ADV = 7, ORC_F_RHO_DER = 8,
    ORC_F_VEL_ADV = 9, ORC_F_VISCOSITY = 10, ORC_F_TENSION = 11, ORC_F_PGRAD = 12,
    ORC_F_BACC = 13, ORC_F_NBR_COUNT = 14, ORC_F_FORCE_EXT = 15,
    ORC_F_PRESS_ITER = 16,   /* pcisph press_iter / iisph p_iter */
    ORC_F_PRESS_FORCE = 17,  /* pcisph press_force / iisph f_press */
    ORC_F_POS_PREDICT = 18, ORC_F_D_II = 19, ORC_F_A_II = 20, ORC_F_D_IJ = 21,
    ORC_F_PBF_LAMBDA = 22, ORC_F_PBF_DELTA_POS = 23,
    ORC_F_P_PAST = 24,            /* iisph: last step's pressure (iisph_solver.py:209-210), what a hand-over has to carry */
    ORC_F_WALL_POS = 32, ORC_F_WALL_VOL = 33,
    ORC_F_RIGID_POS = 48, ORC_F_RIGID_VOL = 49, ORC_F_RIGID_FORCE = 50, ORC_F_RIGID_MASS = 51, ORC_F_RIGID_VERT = 52
};

/* rigid body of config 5 (ParticleSystem.py:41-64): sample points and mesh vertices in the mesh frame */
typedef struct OrcRigid {
    int n_particles, n_vertices;
    const float *points, *vertices;
    double rho_0;
    double pos_offset[3];
    double attitude_offset_deg[3];
    int active;
} OrcRigid;

typedef struct Orc Orc;

Orc *orc_create(const OrcConfig *cfg);
Orc *orc_create_rigid(const OrcConfig *cfg, const OrcRigid *rigid);   /* dfsph only */
void orc_rigid_step(Orc *o);                                          /* rigid_solver.step, rigid_solver.py:216-232 */
void orc_destroy(Orc *o);
/* out[0]=N fluid, out[1]=Nb wall, out[2]=Nr rigid, out[3..5]=grid_num, out[6]=C */
void orc_sizes(const Orc *o, int *out7);
/* copies the whole field as float (3 floats per particle for vectors); returns element count or -1 */
long orc_field_floats(Orc *o, int field);
long orc_get(Orc *o, int field, float *out);
long orc_set(Orc *o, int field, const float *in);
int orc_set_scalar(Orc *o, int which, double value);   /* 0: delta_time (with delta_time_2 and ps.delta_time) */
double orc_get_scalar(const Orc *o, int which); /* 0 dt, 1 simulate_cnt, 2 particle_m, 3 h, 4 lost, 5 pcisph delta, 6 pcisph beta, 7/8 pcisph max-neighbour index/count, 9 ps.delta_time */

void orc_build_grid(Orc *o);               /* reset_grid + update_grid */
void orc_compute_rho(Orc *o);              /* solver_base.compute_all_rho */
void orc_compute_alpha(Orc *o);            /* dfsph compute_all_alpha (needs rho) */
void orc_compute_nbr_count(Orc *o);        /* get_neighbour_count for all i -> ORC_F_NBR_COUNT */
int orc_step_wcsph(Orc *o, int nsteps);
/* max_dens_iter <= 0 means "no cap" like the reference */
int orc_step_dfsph(Orc *o, int nsteps, int max_dens_iter, OrcStepStats *last);
/* pcisph_solver.step / iisph_solver.step: last->n_dens = pressure iterations, last->dens_err = the printed residual */
int orc_step_pcisph(Orc *o, int nsteps, OrcStepStats *last);
int orc_step_iisph(Orc *o, int nsteps, OrcStepStats *last);
/* "Legal schedule" mode: seed != 0 draws one of the executions the reference's racy cell-list append (ParticleSystem.py:388-397) and f32
 * atomic means (dfsph_solver.py:139-141, 275-279) allow; chunk = particles per thread-local partial of a mean (1 = one atomic per particle).
 * seed = 0 (default) is the canonical order of every parity test.  tools/envelope.py. */
void orc_set_schedule(Orc *o, unsigned long long seed, int chunk);

/* pbf_solver.step (pbf_solver.py:176-187) under the schedule stated in sph_oracle.c */
int orc_step_pbf(Orc *o, int nsteps);

/* scalar kernels, for the known-answer tests */
float orc_cubic_kernel(float r, float h);
void orc_cubic_kernel_derivative(const float r[3], float h, float out[3]);
float orc_tait_pressure(float rho);

#ifdef __cplusplus
}
#endif
#endif
