/*
 * sph_mi355x.h -- C-ABI of the MI355X-native SPH step library (libsph_mi355x.so).
 *
 * The reference (Jukgei/CFD_Taichi @ 2024_08_07) has no FFI: its per-step path is Python
 * driving Taichi-JIT kernels, and the only caller contract is main.py:64-71,165-173
 * (`ParticleSystem(config)`, `<name>_solver(ps, config)`, `solver.step()`,
 * `solver.delta_time[None]`, `ps.fluid_particles.pos.to_numpy()`).  This header is what a
 * ctypes binding for that path binds instead; each entry point names the reference
 * interface it replaces.  Plain pointers and sizes only, no torch / HIP types.
 *
 * Conventions: every call returns SPH_OK (0) or a negative SPH_E_* code and never throws;
 * sph_last_error() gives the message of the last failing call on that handle (or of the
 * last failing sph_create when handle == NULL).  One host thread per handle, one HIP
 * stream per handle, no global state.  Host pointers are borrowed for the call only.
 * Particle data crosses the ABI in ORIGINAL particle order (the order of
 * ParticleSystem.init_particle_pos, ParticleSystem.py:142-151), f32, 3 floats per vector.
 */
#ifndef SPH_MI355X_H
#define SPH_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Version of this interface.  It changes whenever a struct grows or a signature changes (round 4: sph_replan_slabs gained ghost_layers, SphComm
 * gained exchange_counts_n / reduce_capacity; round 5: SPH_P_* scalars).  A binding checks sph_abi_version() == SPH_ABI_VERSION when it loads the
 * library (cfd_taichi_amd/_native.py does) instead of finding out through a misread struct. */
#define SPH_ABI_VERSION 5
int32_t sph_abi_version(void);

#define SPH_OK 0
#define SPH_E_INVALID (-1)   /* bad argument / size mismatch */
#define SPH_E_HIP (-2)       /* HIP runtime error (message has hipGetErrorString) */
#define SPH_E_NO_DEVICE (-3) /* no gfx950 device visible: the library has no CPU fallback */
#define SPH_E_OVERFLOW (-4)  /* a neighbour list overflowed its capacity.  Detected at the step's first read-back (wcsph: after the nsteps of the
                              * call), i.e. after sweeps have already run on truncated lists: the handle's particle state is UNDEFINED
                              * afterwards -- recreate the handle with a larger max_neighbors, or re-upload positions and velocities */
#define SPH_E_STATE (-5)     /* call not valid for this handle (e.g. dfsph step on a wcsph handle) */

#define SPH_SOLVER_WCSPH 0
#define SPH_SOLVER_DFSPH 1
#define SPH_SOLVER_PCISPH 2   /* pcisph_solver.py */
#define SPH_SOLVER_IISPH 3    /* iisph_solver.py */
#define SPH_SOLVER_PBF 4      /* pbf_solver.py (stale in the reference: read as csrc/sph_pbf_kernels.h states; single GPU, no rigid body) */

/* config/X.json of the reference, flattened (utils.py:3-11 reads it,
 * ParticleSystem.py:31-103 and solver_base.py:7-39 consume it).  Doubles carry the Python
 * scalars unrounded; the library rounds to f32 exactly where Taichi would. */
typedef struct SphConfig {
    double box_min[3];          /* scene.box_min */
    double box_max[3];          /* scene.box_max */
    double particle_radius;     /* scene.particle_radius */
    double gravity;             /* scene.gravity */
    double delta_time;          /* solver.delta_time */
    double start_pos[3];        /* fluid.start_pos */
    double water_size[3];       /* fluid.water_size */
    int32_t boundary_handle;    /* solver.boundary_handle (default 1: Akinci wall particles) */
    int32_t fs_couple;          /* solver.fs_couple (default 1) */
    int32_t solver;             /* SPH_SOLVER_* (solver.name) */
    int32_t device;             /* HIP device ordinal */
    int32_t max_neighbors;      /* fluid neighbour-list rows per particle; 0 = default (64) */
    int32_t max_wall_neighbors; /* wall neighbour-list rows per particle; 0 = default (64) */
    int32_t max_density_iters;  /* cap on correct_density_error (reference has none, dfsph_solver.py:225); 0 = default (100); reported in SphStepStats.capped */
    int32_t slab_rank;          /* multi-GPU x-slab rank, 0 for single GPU */
    int32_t slab_count;         /* number of slabs (world size), 0 or 1 for single GPU */
    int32_t slab_capacity;      /* particles (owned + ghosts) a slab handle can hold; 0 = default (1.75 N / slab_count + 256k) */
    int32_t slab_rebalance_every; /* re-cut the slabs from the current particle distribution every M steps; 0 = static cuts */
    int32_t arith;              /* SPH_ARITH_EXACT (0, default): every f32 operation of the reference in its order, bit-equal to oracle/;
                                   SPH_ARITH_RELAXED (1): the dfsph pair sweeps of large single-GPU scenes may use approximate
                                   reciprocal square roots and FMA contraction (north_star's 1e-5 bar; see csrc/sph_relaxed_kernels.h).
                                   A permission: handles the relaxed sweeps do not cover run the exact ones. */
    int32_t slab_ghost_layers;  /* slab handles: ghost cell columns per side.  0 = default: 2 for dfsph (the correction sweeps run on the inner
                                   ghost column too, so a solver iteration needs ONE halo refresh -- the residual's -- instead of two), 1 for
                                   the other solvers; 1 forces the one-column protocol.  Results do not depend on it. */
    int32_t slab_overlap;       /* slab handles, dfsph: the OVERLAPPED protocol runs the residual sweeps' edge tiles first, the halo of their results on a
                                   second stream under the interior tiles, and the residual's all-reduce + loop decision on a third under the next
                                   correction sweep; the IN-ORDER one runs everything on the handle's stream (same bits).  0 = default: the handle can do
                                   both and starts with the one that measured faster for its transport (native RCCL: in order, with the residual's
                                   triple riding in the halo's transfers; a synchronous callback transport: overlapped) -- sph_slab_set_overlap switches
                                   between steps; 1 = in order only (no second / third stream); 2 = start overlapped.  What "can do both" costs: two more
                                   HIP streams, four events, a tile order and 20 bytes per particle of slab capacity (what a divergence correction that
                                   runs ahead of its loop decision overwrites) -- allocated with the handle; pass 1 where that memory matters */
    int32_t reserved[2];
} SphConfig;

#define SPH_ARITH_EXACT 0
#define SPH_ARITH_RELAXED 1

typedef struct SphSizes {
    int32_t n_fluid;            /* ps.particle_num                 ParticleSystem.py:85 */
    int32_t n_wall;             /* ps.boundary_particles_num       ParticleSystem.py:95 */
    int32_t n_rigid;            /* ps.rigid_particles_num */
    int32_t grid[3];            /* ps.grid_num                     ParticleSystem.py:101 */
    int32_t n_cells;
    int32_t max_neighbors;
    int32_t max_wall_neighbors;
} SphSizes;

/* what dfsph_solver.py prints at :233 and :416, plus health counters */
typedef struct SphStepStats {
    int32_t n_div;              /* divergence iterations */
    int32_t n_dens;             /* density iterations */
    int32_t n_div_evals;        /* derivative_iter_all_rho evaluations */
    int32_t capped;             /* 1 if max_density_iters stopped the density loop */
    float div_first_err;
    float div_err;
    float dens_err;             /* rho_avg - rho_0 */
    float dt;                   /* delta_time after the step */
    int32_t max_nbrs;           /* largest fluid-neighbour count seen in the last list build */
    int32_t max_wall_nbrs;
    int32_t lost;               /* particles outside the grid (reference prints an error, ParticleSystem.py:393-395) */
    int32_t reserved;
} SphStepStats;

/* the `solid` block of the reference's config (ParticleSystem.py:41-64) after mesh loading: sample points
 * (trimesh voxelized(pitch).fill().points in the reference; cfd_taichi_amd/mesh.py here) and mesh vertices, both in the
 * mesh frame, before attitude_offset / pos_offset are applied */
typedef struct SphRigid {
    int32_t n_particles;
    int32_t n_vertices;
    const float *points;         /* 3 * n_particles */
    const float *vertices;       /* 3 * n_vertices */
    double rho_0;                /* solid.rho_0 */
    double pos_offset[3];        /* solid.pos_offset */
    double attitude_offset[3];   /* solid.attitude_offset, degrees */
    int32_t active;              /* solid.active */
    int32_t reserved;
} SphRigid;

/* species for upload / download */
#define SPH_SPECIES_FLUID 0
#define SPH_SPECIES_WALL 1
#define SPH_SPECIES_RIGID 2

/* fields (fluid unless stated).  Vectors are 3 floats per particle. */
#define SPH_F_POS 0        /* fluid_particles.pos   (upload + download) */
#define SPH_F_VEL 1        /* fluid_particles.vel   (upload + download) */
#define SPH_F_ACC 2        /* fluid_particles.acc   (wcsph only, download) */
#define SPH_F_RHO 3        /* solver.rho */
#define SPH_F_PRESSURE 4   /* wcsph solver.pressure */
#define SPH_F_ALPHA 5      /* dfsph solver.alpha */
#define SPH_F_WARM_K 6     /* dfsph solver.warm_start_k (upload + download) */
#define SPH_F_RHO_ADV 7    /* dfsph solver.rho_adv */
#define SPH_F_RHO_DER 8    /* dfsph solver.rho_derivative */
#define SPH_F_VEL_ADV 9    /* dfsph solver.vel_adv */
#define SPH_F_NBR_COUNT 14 /* ps.get_neighbour_count(i), as float */
#define SPH_F_PRESS_ITER 16  /* pcisph solver.press_iter / iisph solver.p_iter after the last step */
#define SPH_F_PRESS_FORCE 17 /* pcisph solver.press_force / iisph solver.f_press */
#define SPH_F_POS_PREDICT 18 /* pcisph solver.pos_predict (pbf: equals pos after a step, pbf_solver.py:84) */
#define SPH_F_D_II 19        /* iisph solver.d_ii */
#define SPH_F_A_II 20        /* iisph solver.a_ii */
#define SPH_F_D_IJ 21        /* iisph solver.d_ij */
#define SPH_F_PBF_LAMBDA 22  /* pbf solver.pbf_lambda of the last step */
#define SPH_F_PBF_DELTA_POS 23 /* pbf solver.delta_pos of the last step */
#define SPH_F_WALL_POS 32  /* boundary_particles.pos    (species WALL) */
#define SPH_F_WALL_VOL 33  /* boundary_particles.volume (species WALL) */
#define SPH_F_RIGID_POS 48    /* rigid_particles.pos    (species RIGID) */
#define SPH_F_RIGID_VOL 49    /* rigid_particles.volume */
#define SPH_F_RIGID_FORCE 50  /* rigid_particles.force */
#define SPH_F_RIGID_MASS 51   /* rigid_particles.mass */
#define SPH_F_RIGID_VERT 52   /* ps.rigid_vertices (mesh vertices, for OBJ export) */

/* scalars for sph_get_scalar */
#define SPH_S_DELTA_TIME 0     /* solver.delta_time[None] */
#define SPH_S_SIMULATE_CNT 1   /* solver.simulate_cnt[None] */
#define SPH_S_PARTICLE_M 2     /* ps.particle_m */
#define SPH_S_SUPPORT_RADIUS 3 /* ps.support_radius */
#define SPH_S_PS_DELTA_TIME 4  /* ps.delta_time[None] */
#define SPH_S_GRAPH_LAUNCHES 5 /* diagnostics: hipGraph replays issued by sph_step_wcsph (each replays two steps) */
#define SPH_S_PCISPH_DELTA 6   /* pcisph solver.delta[None]         pcisph_solver.py:47 */
#define SPH_S_PCISPH_BETA 7    /* pcisph solver.beta                :23 */
#define SPH_S_PCISPH_MAX_INDEX 8 /* ps.get_max_neighbor_particle_index()  ParticleSystem.py:410-422 (single-thread reading) */
#define SPH_S_PCISPH_MAX_COUNT 9
#define SPH_S_RIGID_CENTROID 10   /* +0,1,2: ps.rigid_centriod[None] */
#define SPH_S_RIGID_OMEGA 13      /* +0,1,2: rigid_solver.omega[None] */
#define SPH_S_RIGID_VEL 16        /* +0,1,2: rigid_particles.vel (uniform over the body) */
#define SPH_S_RIGID_MASS 19       /* rigid_solver.mass[None] */
#define SPH_S_RIGID_INERTIA_INV 20 /* +0..8: ps.rigid_inertia_tensor_inv[None], row major */
#define SPH_S_VERLET_BUILDS 31    /* diagnostics: list builds so far on a Verlet handle (wcsph under the relaxed arithmetic: the lists carry a skin and are rebuilt on demand) */
#define SPH_S_ARITH_RELAXED 30    /* diagnostics: 1 if this handle's dfsph sweeps run the tolerance-grade kernels (SphConfig.arith asked AND the handle qualifies) */

/* Solver attributes a caller of the reference edits on the solver object after constructing it -- sph_set_scalar / sph_get_scalar.
 * The defaults are the reference's.  The dfsph loop attributes (64-68) are read by Python-scope loops at every step (dfsph_solver.py:225, :400)
 * and may be written between steps; the others are baked into Taichi kernels when they first compile, so the mirror classes
 * (cfd_taichi_amd/solver_base.py) forward them once, at the first step().  On slab handles every rank must write the same values. */
#define SPH_P_DENSITY_THRESHOLD 64                 /* dfsph_solver.density_threshold (percent of rho_0)      dfsph_solver.py:22, :225 */
#define SPH_P_MIN_ITERATION_DENSITY 65             /* dfsph_solver.min_iteration_density                     :21, :225 */
#define SPH_P_MIN_ITERATION_DENSITY_DIVERGENCE 66  /* dfsph_solver.min_iteration_density_divergence          :23, :400 */
#define SPH_P_MAX_ITERATION_DENSITY_DIVERGENCE 67  /* dfsph_solver.max_iteration_density_divergence          :24, :400 */
#define SPH_P_DENSITY_DIVERGENCE_THRESHOLD 68      /* dfsph_solver.density_divergence_threshold              :25, :400 */
#define SPH_P_WARM_START 69                        /* dfsph_solver.warm_start (0 / 1)                        :26, :396, :404 */
#define SPH_P_ADAPTIVE_DT 70                       /* dfsph_solver.adaptive_dt (0 / 1)                       :27, :113 */
#define SPH_P_MAX_DT 71                            /* dfsph_solver.max_dt                                    :28, :114-115 */
#define SPH_P_MIN_DT 72                            /* dfsph_solver.min_dt                                    :29, :117 */
#define SPH_P_VISCOSITY_C_S 73                     /* solver.viscosity_c_s      solver_base.py:24 (13), wcsph_solver.py:18 (10); :187 */
#define SPH_P_VISCOSITY_ALPHA 74                   /* solver.viscosity_alpha    solver_base.py:25, :187 */
#define SPH_P_VISCOSITY_EPSILON 75                 /* solver.viscosity_epsilon  solver_base.py:23, :188 */
#define SPH_P_TENSION_K 76                         /* solver.tension_k          solver_base.py:26 (0.5), wcsph_solver.py:20 (0.2); :216 */

typedef struct SphHandle SphHandle;

/* replaces ParticleSystem(config) + <name>_solver(ps, config)   main.py:64-68.
 * Builds the fluid lattice and the wall particles (ParticleSystem.py:139-195), the static
 * wall cell list and wall volumes (:309-335), and allocates every device buffer. */
int sph_create(const SphConfig *cfg, SphHandle **out);
/* the same with a rigid body: replaces ParticleSystem(config) with a `solid` block + rigid_solver(ps, config)   main.py:69-71.
 * All four solvers couple to the body (wcsph_solver.py:118-127, dfsph_solver.py:204-212, pcisph_solver.py:200-211, iisph_solver.py:159-168).
 * On slab handles: dfsph with two ghost columns only; the body is replicated on every rank, three small all-reduces per step (the fluid positions and
 * densities the reference's index quirks read, the per-sample forces) keep every rank's copy bit-identical to the one-GPU run. */
int sph_create_rigid(const SphConfig *cfg, const SphRigid *rigid, SphHandle **out);
/* replaces rigid_solver.step()   rigid_solver.py:216-232 */
int sph_rigid_step(SphHandle *h);
void sph_destroy(SphHandle *h);
int sph_get_sizes(SphHandle *h, SphSizes *out);
const char *sph_last_error(SphHandle *h);

/* replaces field.from_numpy / field.to_numpy on ps.fluid_particles.* and solver.*   main.py:159,190.
 * n_floats must equal the field's float count. */
int sph_upload(SphHandle *h, int species, int field, const float *host, size_t n_floats);
int sph_download(SphHandle *h, int species, int field, float *host, size_t n_floats);

/* replaces wcsph_solver.step() x nsteps   wcsph_solver.py:25-30 (asynchronous; sph_download /
 * sph_synchronize wait for it) */
int sph_step_wcsph(SphHandle *h, int nsteps);
/* replaces dfsph_solver.step() x nsteps   dfsph_solver.py:440-445; `last` (may be NULL) gets the
 * last step's statistics */
int sph_step_dfsph(SphHandle *h, int nsteps, SphStepStats *last);
/* replaces pcisph_solver.step() x nsteps  pcisph_solver.py:252-259.  last->n_dens = iter_cnt and last->dens_err = rho_err_avg as
 * printed at :71; last->capped = 1 when max_iteration (80) ended the loop */
int sph_step_pcisph(SphHandle *h, int nsteps, SphStepStats *last);
/* replaces iisph_solver.step() x nsteps   iisph_solver.py:340-347.  last->n_dens = l and last->dens_err = residual as printed at :102;
 * last->n_div = 1 when the loop left on "Iteration trend to divergence" (:97-99); last->capped = 1 at max_iter_cnt (180) */
int sph_step_iisph(SphHandle *h, int nsteps, SphStepStats *last);
/* replaces pbf_solver.step() x nsteps     pbf_solver.py:176-187 (update_all_pos under the barrier-synchronised schedule, csrc/sph_pbf_kernels.h) */
int sph_step_pbf(SphHandle *h, int nsteps);
/* stages of the step, for parity tests against the oracle's stages:
 * ps.reset_grid()+update_grid() (+ neighbour-list build), solver.compute_all_rho(), dfsph compute_all_alpha() */
int sph_build_neighbors(SphHandle *h);
int sph_compute_density(SphHandle *h);
int sph_compute_alpha(SphHandle *h);

int sph_get_scalar(SphHandle *h, int which, double *out);
/* `solver.delta_time[None] = value` (the reference's 0-d field is writable, main.py:111; dfsph re-derives it every step from the CFL
 * rule, dfsph_solver.py:112-119, so a written value lasts one step there): which = SPH_S_DELTA_TIME; with it a device state
 * (pos, vel, warm_start_k, delta_time) can be moved into another handle, or into the oracle, completely.
 * `solver.<attribute> = value`: which = SPH_P_* (above). */
int sph_set_scalar(SphHandle *h, int which, double value);
int sph_synchronize(SphHandle *h);
/* Development overrides in force on this handle: "NAME=value;NAME=value" (empty string: none).  The SPH_* environment knobs of the
 * library (layout / arithmetic switches used by tests and tools for A/B runs) are read only when SPH_DEV=1 is set; every one that took
 * effect is listed here, so a measurement can name the switches it ran under -- bench.py refuses to print a line otherwise. */
const char *sph_overrides(SphHandle *h);

/* Per-kernel timing with HIP events on the handle's stream (bench.py's roofline leg).
 * Enabling it records an event pair around every launch; totals are read back per kernel id. */
int sph_profile_enable(SphHandle *h, int on);
int sph_profile_reset(SphHandle *h);
int sph_profile_kernel_count(void);
const char *sph_profile_kernel_name(int kernel_id);
int sph_profile_get(SphHandle *h, int kernel_id, double *total_ms, int64_t *launches);

/* ---- multi-GPU: x-slab decomposition (new capability, the reference is single-device) ----
 * One process per GPU.  A handle created with slab_count > 1 owns the particles whose cell x-index lies in its slab
 * [x_lo, x_hi) plus one ghost cell layer on each side.  Every step it (1) migrates particles that left the slab,
 * (2) re-sends its two edge layers as ghosts, and after every sweep whose output the neighbours read (3) refreshes that
 * field on the ghosts; residual sums and the CFL maximum are all-reduced.  The library packs / unpacks on the device and
 * calls back into the host for the transport, so the same code runs over RCCL (device buffers) or any host transport.
 *
 * Buffers: four byte buffers of `capacity` bytes each, owned by the caller (e.g. torch tensors): send/recv x left/right.
 * With on_host = 0 they are device pointers (RCCL over xGMI reads/writes them directly); with on_host = 1 they are host
 * pointers and the library stages through them with hipMemcpy (used by the gloo tests).
 * Callbacks return 0 on success and are invoked on the calling thread.
 *
 * Two transport disciplines:
 *   - synchronous (stream_ordered = 0): the library synchronises its stream before exchange_buffers (the packed data is complete) and
 *     the callback returns when the receive buffers are filled.  Host transports (on_host = 1) are always of this kind.
 *   - stream-ordered (stream_ordered = 1, device buffers): the callbacks ENQUEUE the transfer on the handle's own stream
 *     (sph_get_stream) -- e.g. RCCL send/recv issued with that stream current -- and return at once; the library never blocks the
 *     host around them, so a whole chunk of solver iterations, halo refreshes and residual all-reduces is in flight at a time.
 * With allreduce_stream set, DFSPH runs its loops with the device-side control of the single-GPU path: the (sum, count) pair of a
 * residual is written to reduce_buf, all-reduced in place by allreduce_stream, and a second kernel takes the reference's loop
 * decision from it -- identical on every slab because all ranks see the same reduced values. */
typedef struct SphComm {
    void *user;
    /* send my counts to the left / right neighbour and receive theirs (absent neighbour: recv 0) */
    int (*exchange_counts)(void *user, int32_t send_left, int32_t send_right, int32_t *recv_left, int32_t *recv_right);
    /* send the first send_*_bytes of the send buffers, receive exactly recv_*_bytes into the recv buffers */
    int (*exchange_buffers)(void *user, size_t send_left_bytes, size_t send_right_bytes, size_t recv_left_bytes, size_t recv_right_bytes);
    /* in-place all-reduce of n doubles over all slabs; op 0 = sum, 1 = max */
    int (*allreduce)(void *user, double *values, int32_t n, int32_t op);
    void *send_left, *send_right, *recv_left, *recv_right;
    size_t capacity;
    int32_t on_host;
    int32_t stream_ordered;     /* 1: exchange_buffers / allreduce_stream enqueue on the handle's stream and return (device buffers only) */
    /* optional: in-place all-reduce of the first n doubles of reduce_buf (op 0 = sum, 1 = max), ordered like exchange_buffers */
    int (*allreduce_stream)(void *user, int32_t n, int32_t op);
    double *reduce_buf;         /* >= 4 doubles: device memory with on_host = 0, host memory (the library stages) with on_host = 1 */
    /* optional: exchange_counts with n (<= 8) ints per neighbour in one round trip -- the particle exchange of a step sends
     * (records, ghosts per column, migrants the sender keeps as ghosts per column) in ONE message; NULL: the library calls exchange_counts n times */
    int (*exchange_counts_n)(void *user, int32_t n, const int32_t *send_left, const int32_t *send_right, int32_t *recv_left, int32_t *recv_right);
    size_t reduce_capacity;     /* doubles reduce_buf holds; 0 = 4.  A slab handle with a rigid body sums arrays of 4 x (rigid sample count) doubles through it */
} SphComm;

int sph_set_comm(SphHandle *h, const SphComm *comm);
/* the same for a caller built against an older, SHORTER SphComm: comm_size = that caller's sizeof(SphComm); the fields it does not have read as 0 / NULL */
int sph_set_comm_sized(SphHandle *h, const SphComm *comm, size_t comm_size);
/* Native transport: instead of callbacks, the library itself issues ncclSend / ncclRecv to the left and right slab neighbour and
 * ncclAllReduce of the residual pair on its own stream (librccl is dlopen'ed).  Rank 0 obtains a 128-byte id with
 * sph_rccl_unique_id and the application broadcasts it by any means; every rank then calls sph_rccl_attach (collective:
 * ncclCommInitRank with rank = slab_rank, world = slab_count) instead of sph_set_comm.  sph_rccl_selftest all-reduces n doubles and
 * runs an empty neighbour exchange (a single-GPU handle attaches as a communicator of one rank for it). */
int sph_rccl_unique_id(void *id128);
int sph_rccl_attach(SphHandle *h, const void *id128, size_t capacity_bytes);
int sph_rccl_selftest(SphHandle *h, double *inout, int32_t n, int32_t op);
/* the handle's HIP stream (a hipStream_t), for stream-ordered transports */
int sph_get_stream(SphHandle *h, void **stream);
/* host-only planning (no device needed): cuts[0..slab_count] = cell-column boundaries of the slabs of the scene's initial lattice,
 * counts[k] = particles slab k owns at t = 0.  The cuts balance what a slab costs: its particles plus the ghosts of each cut it has
 * (the particles of the solver's ghost columns beyond the cut) -- the largest such load is minimised */
int sph_plan_slabs(const SphConfig *cfg, int32_t *cuts, int32_t *counts);
/* host-only: the re-balancing rule.  new_cuts = cuts of `column_histogram` (grid_x counts) that minimise the largest load (particles + the
 * particles of `ghost_layers` columns beyond each cut; ghost_layers = 0: plain equal counts), every slab >= 3 columns wide and
 * old_cuts[k-1] < new_cuts[k] < old_cuts[k+1] (a particle's new owner is its rank or a direct neighbour) */
int sph_replan_slabs(const int64_t *column_histogram, int32_t grid_x, int32_t slab_count, const int32_t *old_cuts, int32_t ghost_layers, int32_t *new_cuts);
/* between steps, on every slab alike: 1 = the dfsph solver loops run their halo and reductions on their own streams (the default where the
 * handle and its transport can), 0 = in order on the handle's stream.  Results are the same bits; which is faster depends on the link */
int sph_slab_set_overlap(SphHandle *h, int32_t on);
/* slab bookkeeping: out[0] = owned particles, out[1] = ghosts, out[2] = x_lo, out[3] = x_hi (cell units), out[4] = capacity,
 * out[5] = number of re-balancings that moved a cut, out[6] = slab_rebalance_every, out[7] = the halo protocol in force: ghost columns per side (1 or 2)
 * | 16 if the dfsph residual sweeps run their edge tiles first with the halo on its own stream | 32 if the residual's all-reduce and loop decision run
 * on a third stream under the next correction sweep (both need the native transport or a synchronous one) */
int sph_slab_info(SphHandle *h, int32_t *out8);
/* What the halo transport was asked to do since the last reset (whichever transport drives it: native RCCL, callbacks over RCCL or gloo):
 * out[0] point-to-point groups (a send / recv pair with each slab neighbour), out[1] bytes sent, out[2] bytes received, out[3] count
 * exchanges (one host round trip each), out[4] all-reduces ordered on the handle's stream, out[5] all-reduces through the host,
 * out[6] steps, out[7] 0.  bench.py reports them per step in config.rank0_comm so that a measured N > 1 line can be read against the
 * cost model of DESIGN.md section 6. */
int sph_comm_stats(SphHandle *h, int64_t *out8, int reset);
/* local (device-order) access for slab handles: all resident particles, owned and ghost; ids < 0 mark ghosts (~id) */
int sph_download_local(SphHandle *h, int field, float *host, size_t n_floats);
int sph_download_ids(SphHandle *h, int32_t *host, size_t n);

/* device arithmetic self-test: out[i] = op(a[i], b[i]) evaluated on the GPU with the same
 * compiler flags as the sweeps (op 0: a/b, 1: sqrt(a), 2: cubic_kernel(a, h=b), 3..5:
 * component op-3 of cubic_kernel_derivative((a, b, 0.25*a), h=0.1)).  Used by tests to prove
 * the device's f32 divide/sqrt are correctly rounded like the oracle's. */
/* tuning aid: mean microseconds of `reps` launches of one dfsph sweep (0 divergence residual, 1 divergence correction, 2 density
 * residual, 3 sort + list build) with `lds_bytes` of dynamic LDS per block; see tools/tune_sweeps.py */
int sph_tune_time(SphHandle *h, int which, unsigned lds_bytes, int reps, double *avg_us);
int sph_selftest_math(int device, int op, const float *a, const float *b, float *out, size_t n);
/* wave primitive self-test: out[i] = what lane i % 64 holds after one cross-lane primitive over the 64 values in[64*(i/64) ..]
 * (op 0: f64 sum, 1: i32 sum, 2: f32 max, 3: i32 max -- reductions, documented result in lane 0; 4: i32 inclusive prefix sum).
 * The primitives are DPP / ds_swizzle / v_permlane32_swap butterflies (csrc/sph_device.h); n must be a multiple of 256. */
int sph_selftest_wave(int device, int op, const double *in, double *out, size_t n);

#ifdef __cplusplus
}
#endif
#endif
