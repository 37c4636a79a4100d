/*
 * sph_oracle_abi.c -- the per-step entry points of include/sph_mi355x.h implemented on the CPU ORACLE, so that a test written
 * against the C-ABI can run unchanged on either backend ("same entry points in the CPU oracle library so tests can
 * swap backends").  TEST INFRASTRUCTURE ONLY, like the rest of oracle/: the product never loads this library.
 *
 * Exports the path itself -- create / destroy / sizes / upload / download / the five step functions / rigid step / the stage
 * functions / scalars.  The device-specific entry points of the header (slabs and RCCL, profiling, tuning, device self-tests) have
 * no CPU counterpart and are not exported.  PARITY UNPINNED, see sph_oracle.h.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/sph_mi355x.h"
#include "sph_oracle.h"

struct SphHandle {
    Orc *o;
    SphConfig cfg;
    int has_rigid;
    char err[256];
};
static char g_err[256];

static int fail(SphHandle *h, int code, const char *msg)
{
    snprintf(h ? h->err : g_err, 256, "%s", msg);
    return code;
}
const char *sph_last_error(SphHandle *h) { return h ? h->err : g_err; }

static OrcConfig to_orc(const SphConfig *c)
{
    OrcConfig o;
    memset(&o, 0, sizeof(o));
    memcpy(o.box_min, c->box_min, sizeof(o.box_min)); memcpy(o.box_max, c->box_max, sizeof(o.box_max));
    o.particle_radius = c->particle_radius; o.gravity = c->gravity; o.delta_time = c->delta_time;
    memcpy(o.start_pos, c->start_pos, sizeof(o.start_pos)); memcpy(o.water_size, c->water_size, sizeof(o.water_size));
    o.boundary_handle = c->boundary_handle; o.fs_couple = c->fs_couple; o.solver = c->solver;
    const char *t = getenv("ORC_THREADS");
    o.num_threads = t ? atoi(t) : 4;
    return o;
}

int sph_create(const SphConfig *cfg, SphHandle **out)
{
    if (!cfg || !out) return fail(NULL, SPH_E_INVALID, "null argument");
    if (cfg->solver < SPH_SOLVER_WCSPH || cfg->solver > SPH_SOLVER_PBF) return fail(NULL, SPH_E_INVALID, "unknown solver");
    if (cfg->slab_count > 1) return fail(NULL, SPH_E_INVALID, "the oracle has no slab handles");
    SphHandle *h = (SphHandle *)calloc(1, sizeof(SphHandle));
    h->cfg = *cfg;
    OrcConfig oc = to_orc(cfg);
    h->o = orc_create(&oc);
    *out = h;
    return SPH_OK;
}

int sph_create_rigid(const SphConfig *cfg, const SphRigid *rigid, SphHandle **out)
{
    if (!cfg || !rigid || !out) return fail(NULL, SPH_E_INVALID, "null argument");
    if (cfg->solver == SPH_SOLVER_PBF) return fail(NULL, SPH_E_INVALID, "pbf has no rigid coupling");
    SphHandle *h = (SphHandle *)calloc(1, sizeof(SphHandle));
    h->cfg = *cfg;
    OrcConfig oc = to_orc(cfg);
    OrcRigid rg;
    memset(&rg, 0, sizeof(rg));
    rg.n_particles = rigid->n_particles; rg.n_vertices = rigid->n_vertices; rg.points = rigid->points; rg.vertices = rigid->vertices;
    rg.rho_0 = rigid->rho_0; rg.active = rigid->active;
    memcpy(rg.pos_offset, rigid->pos_offset, sizeof(rg.pos_offset));
    memcpy(rg.attitude_offset_deg, rigid->attitude_offset, sizeof(rg.attitude_offset_deg));
    h->o = orc_create_rigid(&oc, &rg);
    if (!h->o) { free(h); return fail(NULL, SPH_E_INVALID, "orc_create_rigid failed"); }
    h->has_rigid = 1;
    *out = h;
    return SPH_OK;
}

void sph_destroy(SphHandle *h)
{
    if (!h) return;
    orc_destroy(h->o);
    free(h);
}

int sph_get_sizes(SphHandle *h, SphSizes *out)
{
    if (!h || !out) return SPH_E_INVALID;
    int s[7];
    orc_sizes(h->o, s);
    memset(out, 0, sizeof(*out));
    out->n_fluid = s[0]; out->n_wall = s[1]; out->n_rigid = s[2];
    out->grid[0] = s[3]; out->grid[1] = s[4]; out->grid[2] = s[5]; out->n_cells = s[6];
    out->max_neighbors = h->cfg.max_neighbors > 0 ? h->cfg.max_neighbors : 64;
    out->max_wall_neighbors = h->cfg.max_wall_neighbors > 0 ? h->cfg.max_wall_neighbors : 64;
    return SPH_OK;
}

/* field ids of the two headers coincide by construction (SPH_F_* == ORC_F_*) */
int sph_upload(SphHandle *h, int species, int field, const float *host, size_t n_floats)
{
    if (!h || !host) return SPH_E_INVALID;
    if (species != SPH_SPECIES_FLUID || !(field == SPH_F_POS || field == SPH_F_VEL || field == SPH_F_WARM_K)) return fail(h, SPH_E_INVALID, "field is read-only");
    int s[7];
    orc_sizes(h->o, s);
    if (n_floats != (size_t)s[0] * (field == SPH_F_WARM_K ? 1 : 3)) return fail(h, SPH_E_INVALID, "size mismatch");
    return orc_set(h->o, field, host) < 0 ? fail(h, SPH_E_INVALID, "orc_set failed") : SPH_OK;
}

int sph_download(SphHandle *h, int species, int field, float *host, size_t n_floats)
{
    if (!h || !host) return SPH_E_INVALID;
    (void)species;
    if (h->cfg.solver == SPH_SOLVER_PBF && field == SPH_F_POS_PREDICT) field = SPH_F_POS;     /* pos = pos_predict after a step, pbf_solver.py:84 */
    const long want = orc_field_floats(h->o, field);
    if (want < 0) return fail(h, SPH_E_INVALID, "field cannot be downloaded");
    if ((size_t)want != n_floats) return fail(h, SPH_E_INVALID, "size mismatch");
    return orc_get(h->o, field, host) == want ? SPH_OK : fail(h, SPH_E_INVALID, "orc_get failed");
}

int sph_step_wcsph(SphHandle *h, int nsteps)
{
    if (!h || h->cfg.solver != SPH_SOLVER_WCSPH) return h ? fail(h, SPH_E_STATE, "not a wcsph handle") : SPH_E_INVALID;
    orc_step_wcsph(h->o, nsteps);
    return SPH_OK;
}

static void stats_out(SphHandle *h, const OrcStepStats *st, int capped, SphStepStats *out)
{
    if (!out) return;
    memset(out, 0, sizeof(*out));
    out->n_div = st->n_div; out->n_dens = st->n_dens; out->n_div_evals = st->n_div_evals; out->capped = capped;
    out->div_first_err = st->div_first_err; out->div_err = st->div_err; out->dens_err = st->dens_err; out->dt = st->dt;
    out->lost = (int32_t)orc_get_scalar(h->o, 4);
}

int sph_step_dfsph(SphHandle *h, int nsteps, SphStepStats *last)
{
    if (!h || h->cfg.solver != SPH_SOLVER_DFSPH) return h ? fail(h, SPH_E_STATE, "not a dfsph handle") : SPH_E_INVALID;
    OrcStepStats st;
    memset(&st, 0, sizeof(st));
    const int cap = h->cfg.max_density_iters > 0 ? h->cfg.max_density_iters : 100;
    int capped = 0;
    for (int k = 0; k < nsteps; ++k) capped = orc_step_dfsph(h->o, 1, cap, &st);
    stats_out(h, &st, capped, last);
    return SPH_OK;
}

int sph_step_pcisph(SphHandle *h, int nsteps, SphStepStats *last)
{
    if (!h || h->cfg.solver != SPH_SOLVER_PCISPH) return h ? fail(h, SPH_E_STATE, "not a pcisph handle") : SPH_E_INVALID;
    OrcStepStats st;
    memset(&st, 0, sizeof(st));
    int capped = 0;
    for (int k = 0; k < nsteps; ++k) capped = orc_step_pcisph(h->o, 1, &st);
    stats_out(h, &st, capped, last);
    return SPH_OK;
}

int sph_step_iisph(SphHandle *h, int nsteps, SphStepStats *last)
{
    if (!h || h->cfg.solver != SPH_SOLVER_IISPH) return h ? fail(h, SPH_E_STATE, "not an iisph handle") : SPH_E_INVALID;
    OrcStepStats st;
    memset(&st, 0, sizeof(st));
    int capped = 0;
    for (int k = 0; k < nsteps; ++k) capped = orc_step_iisph(h->o, 1, &st);
    stats_out(h, &st, capped, last);
    return SPH_OK;
}

int sph_step_pbf(SphHandle *h, int nsteps)
{
    if (!h || h->cfg.solver != SPH_SOLVER_PBF) return h ? fail(h, SPH_E_STATE, "not a pbf handle") : SPH_E_INVALID;
    return orc_step_pbf(h->o, nsteps) == 0 ? SPH_OK : fail(h, SPH_E_STATE, "orc_step_pbf failed");
}

int sph_rigid_step(SphHandle *h)
{
    if (!h) return SPH_E_INVALID;
    if (!h->has_rigid) return fail(h, SPH_E_STATE, "handle has no rigid body");
    orc_rigid_step(h->o);
    return SPH_OK;
}

int sph_build_neighbors(SphHandle *h) { if (!h) return SPH_E_INVALID; orc_build_grid(h->o); orc_compute_nbr_count(h->o); return SPH_OK; }
int sph_compute_density(SphHandle *h) { if (!h) return SPH_E_INVALID; orc_build_grid(h->o); orc_compute_rho(h->o); return SPH_OK; }
int sph_compute_alpha(SphHandle *h)
{
    if (!h) return SPH_E_INVALID;
    if (h->cfg.solver != SPH_SOLVER_DFSPH) return fail(h, SPH_E_STATE, "alpha is a dfsph stage");
    orc_build_grid(h->o); orc_compute_rho(h->o); orc_compute_alpha(h->o); orc_compute_nbr_count(h->o);
    return SPH_OK;
}

int sph_get_scalar(SphHandle *h, int which, double *out)
{
    if (!h || !out) return SPH_E_INVALID;
    switch (which) {
    case SPH_S_DELTA_TIME: *out = orc_get_scalar(h->o, 0); return SPH_OK;
    case SPH_S_SIMULATE_CNT: *out = orc_get_scalar(h->o, 1); return SPH_OK;
    case SPH_S_PARTICLE_M: *out = orc_get_scalar(h->o, 2); return SPH_OK;
    case SPH_S_SUPPORT_RADIUS: *out = orc_get_scalar(h->o, 3); return SPH_OK;
    case SPH_S_PS_DELTA_TIME: *out = orc_get_scalar(h->o, 9); return SPH_OK;
    case SPH_S_GRAPH_LAUNCHES: *out = 0.0; return SPH_OK;
    case SPH_S_ARITH_RELAXED: *out = 0.0; return SPH_OK;          /* the restatement has one arithmetic */
    case SPH_S_PCISPH_DELTA: *out = orc_get_scalar(h->o, 5); return SPH_OK;
    case SPH_S_PCISPH_BETA: *out = orc_get_scalar(h->o, 6); return SPH_OK;
    case SPH_S_PCISPH_MAX_INDEX: *out = orc_get_scalar(h->o, 7); return SPH_OK;
    case SPH_S_PCISPH_MAX_COUNT: *out = orc_get_scalar(h->o, 8); return SPH_OK;
    default:
        if (which >= SPH_P_DENSITY_THRESHOLD && which <= SPH_P_TENSION_K) { *out = orc_get_scalar(h->o, which); return SPH_OK; }
        if (h->has_rigid && which >= SPH_S_RIGID_CENTROID && which < SPH_S_RIGID_INERTIA_INV + 9) { *out = orc_get_scalar(h->o, which); return SPH_OK; }
        return fail(h, SPH_E_INVALID, "unknown scalar");
    }
}

int sph_set_scalar(SphHandle *h, int which, double value)
{
    if (!h) return SPH_E_INVALID;
    if (which >= SPH_P_DENSITY_THRESHOLD && which <= SPH_P_TENSION_K) { orc_set_scalar(h->o, which, value); return SPH_OK; }
    if (which != SPH_S_DELTA_TIME || !(value > 0.0)) return fail(h, SPH_E_INVALID, "sph_set_scalar: SPH_S_DELTA_TIME > 0 or a solver attribute SPH_P_* can be written");
    orc_set_scalar(h->o, 0, value);
    return SPH_OK;
}

int sph_synchronize(SphHandle *h) { return h ? SPH_OK : SPH_E_INVALID; }
int32_t sph_abi_version(void) { return SPH_ABI_VERSION; }
