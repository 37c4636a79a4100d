// sph_mi355x.hip -- host side of libsph_mi355x.so: scene construction, buffer ownership, the per-step
// launch sequences for WCSPH and DFSPH, and the C-ABI of include/sph_mi355x.h.
//
// There is no CPU fallback: without a HIP device sph_create fails with SPH_E_NO_DEVICE.
#include "../../include/sph_mi355x.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sph_kernels.h"
#include <dlfcn.h>
#include <rccl/rccl.h>      // types only: librccl is dlopen'ed when a handle attaches the native transport
#include "sph_slab_kernels.h"
#include "sph_pressure_kernels.h"
#include "sph_rigid_kernels.h"
#include "sph_pbf_kernels.h"
#include "sph_relaxed_kernels.h"

using namespace sph;

namespace {

enum KernelId {
    K_HASH = 0, K_SCAN, K_SCATTER, K_ORDER_GATHER, K_BUILD_NL, K_W_DENSITY, K_W_FORCE, K_D_DENSITY_ALPHA,
    K_D_WARM, K_D_DIV_RESIDUAL, K_D_DIV_CORRECT, K_D_EXT, K_D_DENS_RESIDUAL, K_D_DENS_CORRECT, K_D_INTEGRATE,
    K_FINALIZE, K_TRANSFER, K_SLAB, K_RIGID, K_P_EXT, K_P_PREDICT_RHO, K_P_PRESS, K_P_INTEGRATE, K_I_ADVECT, K_I_RHO_ADV, K_I_DIJ,
    K_I_UPDATE_P, K_I_INTEGRATE, K_B_LAMBDA, K_B_DELTA, K_B_XSPH, K_COUNT
};
const char *kKernelNames[K_COUNT] = {
    "hash_count", "scan", "scatter", "order_gather", "build_nl", "wcsph_density", "wcsph_force", "dfsph_density_alpha",
    "dfsph_warm_start", "dfsph_div_residual", "dfsph_div_correct", "dfsph_ext_force", "dfsph_dens_residual",
    "dfsph_dens_correct", "dfsph_integrate", "finalize", "transfer", "slab_exchange", "rigid",
    "pcisph_ext_force", "pcisph_predict_rho", "pcisph_press_force", "pcisph_integrate", "iisph_advect", "iisph_rho_adv", "iisph_d_ij",
    "iisph_update_p", "iisph_integrate", "pbf_lambda", "pbf_delta_pos", "pbf_xsph"};

thread_local std::string g_create_error;
thread_local bool g_creating_with_rigid = false;     // sph_create_rigid builds the fluid handle first: no Verlet lists there (the body moves through the grid)

// Development overrides.  The SPH_* environment knobs (layout and arithmetic switches for A/B runs, tests and tools) take effect only
// when SPH_DEV=1 is set as well; without it a set knob is ignored with one line on stderr.  Every override that did take effect is
// recorded and reported by sph_overrides(), so that a measurement can name (or refuse) the switches it ran under.
const char *dev_env(std::string *record, const char *name)
{
    const char *e = getenv(name);
    if (!e) return nullptr;
    const char *dev = getenv("SPH_DEV");
    if (!(dev && dev[0] == '1' && dev[1] == 0)) {
        static thread_local std::string warned;
        if (warned.find(std::string(";") + name + ";") == std::string::npos) {
            warned += std::string(";") + name + ";";
            fprintf(stderr, "libsph_mi355x: %s is set but ignored (development overrides need SPH_DEV=1)\n", name);
        }
        return nullptr;
    }
    if (record && record->find(std::string(name) + "=") == std::string::npos) {
        if (!record->empty()) *record += ";";
        *record += std::string(name) + "=" + e;
    }
    return e;
}

}  // namespace

struct SphHandle {
    SphConfig cfg;
    Consts c;
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    std::string overrides;       // development overrides in force on this handle (dev_env)

    // the solver attributes a caller of the reference may edit before (or, for the Python-scope loops, between) steps: solver_base.py:23-26,
    // wcsph_solver.py:17-20, dfsph_solver.py:21-29; sph_set_scalar(SPH_P_*) writes them, apply_params() folds them into Consts / DevScalars
    struct Params {
        double density_threshold = 0.1, density_divergence_threshold = 10, max_dt = 1e-3, min_dt = 1e-5;
        int min_iteration_density = 2, min_iteration_density_divergence = 1, max_iteration_density_divergence = 15;
        int warm_start = 1, adaptive_dt = 1;
        double viscosity_c_s = 13, viscosity_alpha = 0.08, viscosity_epsilon = 0.01, tension_k = 0.5;
    } p;

    int N = 0, Nb = 0, Nr = 0;
    int nblocks = 0;
    float dt_wcsph = 0.f;
    int simulate_cnt = 0;
    bool nl_valid = false;      // neighbour list matches the current positions
    bool density_valid = false;

    // device state (sorted order); index [cur] is the live one
    float4 *P[2] = {nullptr, nullptr};
    float4 *V[2] = {nullptr, nullptr};
    float4 *VA[2] = {nullptr, nullptr};   // dfsph v* ping-pong; VA[0] doubles as wcsph acc
    float *warm[2] = {nullptr, nullptr};
    int *id[2] = {nullptr, nullptr};
    int pcur = 0, vcur = 0, vacur = 0, wcur = 0, icur = 0;

    float *rho = nullptr, *aux = nullptr /* pressure | alpha | a_ii */, *drho = nullptr, *rho_adv = nullptr, *krho = nullptr /* k / rho (kr_split) */;
    float4 *X[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // pcisph: EF, PF, PP, PB0, PB1; iisph: DII, DIJ, f_press, PB0, PB1
    int pb_final = 0;            // which PB holds press_iter / p_iter after the last step
    unsigned sweep_lds = 0;      // sph_tune_time: dynamic LDS bytes per block on the unstaged DFSPH sweeps (caps the waves per CU)
    int last_iters = 2;          // iteration count of the last step's density / pressure loop: size of the next step's first chunk
    float pci_delta = 0.f, pci_beta = 0.f;   // pcisph_solver.py:23-24, :47
    int pci_max_index = -1, pci_max_count = -1;
    std::vector<float> pci_fluid_pos, rigid_pos_host;   // initial lattice / placed rigid samples, for pre_compute's 27-cell walk on the host
    int *cnt = nullptr;
    uint32_t *nl = nullptr, *nlb = nullptr;
    int *cell_of = nullptr, *rank = nullptr, *slot_src = nullptr;
    int *cell_count = nullptr, *cell_start = nullptr, *tile_sums = nullptr;
    int ntiles = 0;
    float4 *WP = nullptr;        // wall particles, cell-sorted: (x, y, z, V_b)
    int *wcell_start = nullptr;
    std::vector<std::pair<void **, size_t>> plan;   // dalloc() requests not yet committed
    std::vector<char *> arenas;                      // dcommit() allocations
    int *tile_rank = nullptr;            // Consts.tile_rank
    // slab handles on the Morton curve keep the cell slots of THEIR columns only (slab_local_grid): the whole grid's tile ranks, and the slot count the
    // cell arrays were allocated for
    std::vector<int> tile_rank_full, tile_rank_local;
    int S_full = 0;
    bool staged = false;                 // LDS staging of the sweeps' gather operand (k_build_nl plan)
    int quad_below = 65536;              // quad sweeps (four lanes per particle) for unstaged single-GPU handles of up to this many particles
    bool opt_quad = true;                // SPH_QUAD=0 at sph_create: small scenes keep one lane per particle in the sweeps (A/B, tests)
    int opt_bnl_split = -1;              // SPH_BNL_SPLIT=0 | 3 | 9 at sph_create: never / always k_build_nl_split with that many waves (A/B, tests); -1: by size
    bool opt_nl16 = true, opt_kr_split = true;   // SPH_NL16=0 / SPH_KR_SPLIT=0 at sph_create (A/B, tests)
    bool relaxed = false;                        // SphConfig.arith == SPH_ARITH_RELAXED (or SPH_ARITH=relaxed in the environment: tools)
    float4 *wall_grad = nullptr;                 // relaxed handles: per-step wall sums (k_rx_wall_grad)
    float *wall_gsq = nullptr;                   //   ... and the walls' share of alpha's denominator
    float4 *wall_gc = nullptr;                   // exact dfsph sweeps: (grad W_ib, V_b) per wall-list entry, written by D1 (for_wall_cache)
    bool opt_wall_cache = true;                  // SPH_WALL_CACHE=0 at sph_create: D2-D7 walk the wall lists themselves (A/B, tests)
    // change propagation between the sweeps of the density loop (sph_kernels.h: stage_sources_flagged); SPH_TILE_SKIP=0 turns it off
    int *wave_dirty = nullptr;                   // per 64-particle wave: did the last density correction change a velocity there?
    unsigned char *changed8 = nullptr;           // ... and per particle (the second, exact level of the residual sweep's check)
    int *pci_zero_press = nullptr;               // pcisph: per tile, "press_force / pos_predict hold the zero-pressure values" (k_pci_press); iisph: "d_ij holds zeros" (k_ii_dij)
    bool opt_tile_skip = true, dens_first = true, tune_all = false;
    bool verlet = false;                         // wcsph under the relaxed arithmetic: lists with a skin, rebuilt on demand (sph_relaxed_kernels.h)
    float4 *x0 = nullptr;                        //   ... positions at the last list build
    // slab handles: what the transport was asked to do since the last sph_comm_stats(reset): [0] point-to-point groups (a send / recv
    // pair with each neighbour), [1] bytes sent, [2] bytes received, [3] count exchanges (one host round trip each), [4] all-reduces
    // ordered on the stream, [5] all-reduces through the host, [6] steps
    long long comm_stat[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint2 *stage_src = nullptr;          // cell runs of every workgroup's staged set (kStageMaxCells per workgroup)
    int *stage_cnt = nullptr;
    double *psum = nullptr; int *pcnt = nullptr; float *pmax = nullptr;
    DevScalars *ds = nullptr;    // device
    DevScalars *ds_host = nullptr;   // pinned mirror
    DevScalarsPub *pub_host = nullptr, *pub_dev = nullptr;      // the same block as the device publishes it itself (k_publish_scalars): pinned, mapped
    unsigned long long pub_seq = 0;
    int pub_late = 0;               // read_scalars_fast: consecutive reads whose word arrived only with the stream's drain
    float *staging = nullptr;    // 3*max(N,Nb) floats, device
    // host copies of the wall particles in original order (for download)
    std::vector<float> wall_pos_host, wall_vol_host;

    // multi-GPU x-slab state (slab_count > 1)
    bool slab = false;
    int slab_rank = 0, nslab = 1;
    SlabGeom geom = {0, 0, 0, 0, 1, 0, 0};
    int ncap = 0;                 // capacity (particles) of every per-particle array
    int n_owned = 0, n_ghost = 0, n_dead = 0;
    bool comm_set = false;
    SphComm comm = {};
    void *dsend[2] = {nullptr, nullptr}, *drecv[2] = {nullptr, nullptr};   // device-side message buffers
    bool own_dev_comm = false;
    int *dead = nullptr;
    // ordered edge lists, one per (side, direction): 0 ghost-left, 1 send-left, 2 send-right, 3 ghost-right.  With two ghost columns per side a
    // list holds the column next to the cut first (edge_n[k][0] entries), then the second one (edge_n[k][1]); edge_off[2 k + l] are the per-cell
    // offsets of column l of list k
    int *edge_off[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int *edge_list[4] = {nullptr, nullptr, nullptr, nullptr};
    int edge_n[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    bool cuts_moved = false;          // this step's re-balancing moved a cut: the particle exchange runs its two-round form
    bool overlap = false;             // dfsph, two ghost columns: edge tiles of the residual sweeps first, halo on xstream under the interior tiles
    hipStream_t xstream = nullptr;    // the halo's stream when overlap is on
    hipEvent_t ev_edge = nullptr, ev_halo = nullptr;
    // ... and the residual's all-reduce + loop decision on a third stream, under the next correction sweep (slab_can_overlap; step_dfsph_device_loops)
    hipStream_t rstream = nullptr;
    hipEvent_t ev_red = nullptr, ev_dec = nullptr;
    float4 *spec_v = nullptr; float *spec_w = nullptr;      // what a divergence correction that ran ahead of its loop decision overwrote (SpecSave / SpecUndo)
    int *tile_flag = nullptr, *tile_order = nullptr;
    bool overlap_on = true;       // sph_slab_set_overlap: the split + the hidden all-reduce may be switched off between steps (same bits either way)
    std::vector<int> cuts;        // all slabs' cell-column cuts (identical on every rank)
    double *red_dev = nullptr;    // (sum, count) / max of this slab on its way through allreduce_stream
    // native transport (sph_rccl_attach): the library drives RCCL itself on its stream
    bool native = false;
    ncclComm_t nccl = nullptr;
    int red_cap = 4;              // doubles in red_dev
    int *cnt_dev = nullptr, *cnt_host = nullptr;     // neighbour count exchange: [send_left, send_right, recv_left, recv_right]
    double *red_host = nullptr;   // pinned staging for host-side all-reduces
    double *gath_dev = nullptr;   // native transport, in-order protocol: every slab's (sum, count, flags), four doubles per slab (native_exchange)
    bool opt_gather = true;       // SPH_SLAB_GATHER=0: the residual pair is all-reduced instead (A/B)
    // one GPU, density loop: working tiles first in the change-propagated launches (TilePhase.sparse / hot in sph_kernels.h)
    FinRide pending_div = kNoRide;      // one GPU: the divergence loop's last decision, taken by k_finalize_max's launch
    int *dens_hot = nullptr, *dens_order = nullptr;
    bool dens_sparse = false;
    // ... and the producer says who must run (DensFlow in sph_kernels.h): the tiles that stage each tile's particles (k_build_nl), the stamps the
    // density loop's sweeps push to them, per tile "its k / rho holds a nonzero", the two broadcast words.  SPH_DENS_PUSH=0 turns it off (A/B, tests)
    int *tile_nbr = nullptr, *need6 = nullptr, *need7 = nullptr, *tile_nz = nullptr, *dens_bcast = nullptr, *worked6 = nullptr, *worked7 = nullptr;
    int flow_stamp = 0, flow_last = 0;          // launch counter of the density loop's sweeps; the stamp of the sweep enqueued last
    DensFlow flow_d6 = kNoFlow;                 // ... and the residual sweep's whole block: a split sweep's second launch and the kernel that unpacks the ghosts' k / rho push with it
    bool opt_dens_push = true;
    bool opt_layer_generic = false;  // SPH_LAYER_GENERIC=1: k_layer_offsets in the form columns of more than 18 432 cells take (tests)
    bool opt_slab_check = false;     // SPH_SLAB_CHECK=1: the host's edge-column bookkeeping against the sorted arrays, every step (tests, soak runs)
    bool own_red = false;
    int rebalance_every = 0, steps_since_rebalance = 0, n_recuts = 0;
    int *col_hist = nullptr, *col_hist_host = nullptr;
    int *counters = nullptr, *counters_host = nullptr;
    int *class_cnt = nullptr;     // k_classify_*: per-workgroup counts / offsets, kSlabCounted arrays of (capacity / 256) ints
    std::vector<int> init_ids;    // original ids of the particles this handle owns at t = 0

    // rigid body (config 5)
    bool rigid = false;
    int rigid_active = 0;
    int Nv = 0;
    float rigid_rho = 0.f;
    float4 *RPos = nullptr;       // [Nr] rigid particles in their own index order: (x, y, z, V_r)
    float4 *RPs = nullptr;        // [Nr] cell-sorted copy, rebuilt every step
    uint32_t *rnl = nullptr;             // fluid neighbours of the rigid sample particles (k_build_rnl), rcnt = their number
    int *rcnt = nullptr;
    int *rid = nullptr, *rcell_of = nullptr, *rrank = nullptr, *rslot = nullptr, *rcell_count = nullptr, *rcell_start = nullptr;
    float *rforce = nullptr;      // [3 Nr] rigid_particles.force
    float *rvert = nullptr;       // [3 Nv] mesh vertices
    float4 *pos_orig = nullptr;   // fluid positions by original id   (get_neighbour_count quirk)
    float *rho_orig = nullptr;    // fluid densities by original id   (viscosity quirk)
    int *ncount = nullptr;        // ps.get_neighbour_count(i) with rigid entries
    RigidReduce *rred = nullptr, *rred_host = nullptr;      // kRigidParts partials on the device, combined into rred_host[0] (read_rigid_reduce)
    float *rvmax_part = nullptr;
    std::vector<float> rvol_host, rmass_host;
    float centroid[3] = {0, 0, 0}, inertia_inv[9] = {0}, r_vel[3] = {0, 0, 0}, r_acc[3] = {0, 0, 0}, r_omega[3] = {0, 0, 0},
          r_alpha[3] = {0, 0, 0};
    float rs_dt = 0.f, rs_omega[3] = {0, 0, 0}, rs_attitude[3] = {0, 0, 0}, rs_mass = 0.f;
    bool rs_run_once = false;
    int rs_cnt = 0;

    // hipGraph replay of WCSPH step pairs (launch-bound at small N): one executable graph per buffer parity
    hipGraphExec_t wcsph_graph[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool graphs_enabled = true;
    long long graph_launches = 0;

    // profiling
    bool profiling = false;
    struct Ev { hipEvent_t a, b; int kid; };
    std::vector<Ev> ev_pending;
    std::vector<hipEvent_t> ev_pool;
    double prof_ms[K_COUNT] = {0};
    int64_t prof_n[K_COUNT] = {0};
};

namespace {

int fail(SphHandle *h, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(h, expr)                                                                        \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return fail(h, SPH_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

hipEvent_t take_event(SphHandle *h)
{
    if (!h->ev_pool.empty()) { hipEvent_t e = h->ev_pool.back(); h->ev_pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}

struct ProfScope {
    SphHandle *h; int kid; hipEvent_t a = nullptr, b = nullptr; hipStream_t st;
    ProfScope(SphHandle *h_, int kid_, hipStream_t on = nullptr) : h(h_), kid(kid_), st(on ? on : h_->stream)
    {
        if (h->profiling) { a = take_event(h); b = take_event(h); (void)hipEventRecord(a, st); }
    }
    ~ProfScope()
    {
        if (h->profiling) { (void)hipEventRecord(b, st); h->ev_pending.push_back({a, b, kid}); }
    }
};

void drain_profile(SphHandle *h)
{
    if (h->ev_pending.empty()) return;
    (void)hipStreamSynchronize(h->stream);
    if (h->xstream) (void)hipStreamSynchronize(h->xstream);
    if (h->rstream) (void)hipStreamSynchronize(h->rstream);
    for (auto &e : h->ev_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) { h->prof_ms[e.kid] += ms; h->prof_n[e.kid] += 1; }
        h->ev_pool.push_back(e.a); h->ev_pool.push_back(e.b);
    }
    h->ev_pending.clear();
}

inline dim3 grid_for(int n) { return dim3((unsigned)std::max(1, (n + kBlock - 1) / kBlock)); }   // an empty slab still launches one (idle) block

#include "sph_host_scene.h"
#include "sph_host_transport.h"
#include "sph_host_rigid.h"
#include "sph_host_dfsph.h"
#include "sph_host_pressure.h"

}  // namespace

// =============================================================================================
// C-ABI
// =============================================================================================
extern "C" {

const char *sph_last_error(SphHandle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int sph_create(const SphConfig *cfg, SphHandle **out)
{
    if (!cfg || !out) return fail(nullptr, SPH_E_INVALID, "null argument");
    *out = nullptr;
    if (cfg->solver < SPH_SOLVER_WCSPH || cfg->solver > SPH_SOLVER_PBF)
        return fail(nullptr, SPH_E_INVALID, "unknown solver %d", cfg->solver);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, SPH_E_NO_DEVICE, "no HIP device available (%s); libsph_mi355x has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, SPH_E_INVALID, "device %d out of range [0,%d)", cfg->device, ndev);
    SphHandle *h = new SphHandle();
    h->cfg = *cfg;
    h->device = cfg->device;
    { const char *e = dev_env(&h->overrides, "SPH_SLAB_GATHER"); h->opt_gather = !(e && atoi(e) == 0); }
    { const char *e = dev_env(&h->overrides, "SPH_LAYER_GENERIC"); h->opt_layer_generic = e && atoi(e) != 0; }
    { const char *e = dev_env(&h->overrides, "SPH_SLAB_CHECK"); h->opt_slab_check = e && atoi(e) != 0; }       // ("0" is off: eight blocking read-backs per step otherwise)
    { const char *e = dev_env(&h->overrides, "SPH_NL16"); h->opt_nl16 = !(e && atoi(e) == 0); }
    { const char *e = dev_env(&h->overrides, "SPH_KR_SPLIT"); h->opt_kr_split = !(e && atoi(e) == 0); }
    { const char *e = dev_env(&h->overrides, "SPH_TILE_SKIP"); h->opt_tile_skip = !(e && atoi(e) == 0); }
    { const char *e = dev_env(&h->overrides, "SPH_DENS_PUSH"); h->opt_dens_push = !(e && atoi(e) == 0); }
    { const char *e = dev_env(&h->overrides, "SPH_WALL_CACHE"); h->opt_wall_cache = !(e && atoi(e) == 0); }
    { const char *e = dev_env(&h->overrides, "SPH_ARITH"); h->relaxed = cfg->arith == SPH_ARITH_RELAXED || (e && (e[0] == 'r' || e[0] == '1')); }
    { const char *e = dev_env(&h->overrides, "SPH_QUAD"); h->opt_quad = !(e && atoi(e) == 0); }
    { const char *e = dev_env(&h->overrides, "SPH_BNL_SPLIT"); const int v = e ? atoi(e) : -1; h->opt_bnl_split = (v == 0 || v == 3 || v == 9) ? v : -1; }
    int rc = SPH_OK;
    do {
        if (hipSetDevice(h->device) != hipSuccess) { rc = fail(h, SPH_E_HIP, "hipSetDevice(%d) failed", h->device); break; }
        if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { rc = fail(h, SPH_E_HIP, "hipStreamCreate failed"); break; }
        HostScene sc;
        if ((rc = build_scene(h, sc))) break;
        if ((rc = alloc_device(h, sc))) break;
        if (h->cfg.solver == SPH_SOLVER_PCISPH) {
            if (!h->slab) h->pci_fluid_pos = sc.fluid_pos;
            if ((rc = pcisph_precompute(h))) break;
        }
        h->wall_pos_host = sc.wall_pos;
        h->wall_vol_host = sc.wall_vol;
    } while (0);
    if (rc) {
        g_create_error = h->err;
        sph_destroy(h);
        return rc;
    }
    *out = h;
    return SPH_OK;
}

int sph_create_rigid(const SphConfig *cfg, const SphRigid *rigid, SphHandle **out)
{
    if (!cfg || !rigid || !out) return fail(nullptr, SPH_E_INVALID, "null argument");
    if (cfg->slab_count > 1 && (cfg->solver != SPH_SOLVER_DFSPH || cfg->slab_ghost_layers == 1))
        return fail(nullptr, SPH_E_INVALID, "a rigid body on slab handles needs dfsph with two ghost columns (the body's fluid neighbours must be resident on the rank that owns its column)");
    if (rigid->n_particles <= 0 || !rigid->points) return fail(nullptr, SPH_E_INVALID, "rigid body has no sample points");
    if (cfg->solver == SPH_SOLVER_PBF) return fail(nullptr, SPH_E_INVALID, "pbf has no rigid coupling (pbf_solver.py has no material branches)");
    g_creating_with_rigid = true;
    int rc = sph_create(cfg, out);
    g_creating_with_rigid = false;
    if (rc) return rc;
    SphHandle *h = *out;
    rc = build_rigid(h, rigid);
    if (!rc && h->cfg.solver == SPH_SOLVER_PCISPH) rc = pcisph_precompute(h);   // the solver is constructed after the ParticleSystem: the grid holds the body
    if (rc) {
        g_create_error = h->err;
        sph_destroy(h);
        *out = nullptr;
    }
    return rc;
}

int sph_rigid_step(SphHandle *h)
{
    if (!h) return SPH_E_INVALID;
    if (!h->rigid) return fail(h, SPH_E_STATE, "handle has no rigid body");
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->slab) {       // every sample's force was summed whole by the rank that owns its column (k_rigid_force): add the ranks' arrays up, x + 0 = x
        const int n3 = 3 * h->Nr;
        hipLaunchKernelGGL(k_floats_to_doubles, grid_for(n3), dim3(kBlock), 0, h->stream, n3, h->rforce, h->red_dev);
        int rc = slab_allreduce_stream(h, n3, 0);
        if (rc) return rc;
        hipLaunchKernelGGL(k_doubles_to_floats, grid_for(n3), dim3(kBlock), 0, h->stream, n3, h->red_dev, h->rforce);
    }
    return rigid_step(h);
}

void sph_destroy(SphHandle *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (auto &e : h->ev_pending) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto &e : h->ev_pool) (void)hipEventDestroy(e);
    for (int k = 0; k < 8; ++k) if (h->wcsph_graph[k]) (void)hipGraphExecDestroy(h->wcsph_graph[k]);
    for (char *arena : h->arenas) (void)hipFree(arena);        // every dalloc'd array
    if (h->rred_host) (void)hipHostFree(h->rred_host);
    if (h->own_dev_comm) { (void)hipFree(h->dsend[0]); (void)hipFree(h->dsend[1]); (void)hipFree(h->drecv[0]); (void)hipFree(h->drecv[1]); }
    if (h->own_red) (void)hipFree(h->red_dev);
    if (h->nccl && rccl().ok) (void)rccl().CommDestroy(h->nccl);
    if (h->cnt_dev) (void)hipFree(h->cnt_dev);
    if (h->cnt_host) (void)hipHostFree(h->cnt_host);
    if (h->red_host) (void)hipHostFree(h->red_host);
    if (h->gath_dev) (void)hipFree(h->gath_dev);
    if (h->counters_host) (void)hipHostFree(h->counters_host);
    if (h->col_hist_host) (void)hipHostFree(h->col_hist_host);
    if (h->ds_host) (void)hipHostFree(h->ds_host);
    if (h->pub_host) (void)hipHostFree(h->pub_host);
    if (h->ev_edge) (void)hipEventDestroy(h->ev_edge);
    if (h->ev_halo) (void)hipEventDestroy(h->ev_halo);
    if (h->xstream) { (void)hipStreamSynchronize(h->xstream); (void)hipStreamDestroy(h->xstream); }
    if (h->ev_red) (void)hipEventDestroy(h->ev_red);
    if (h->ev_dec) (void)hipEventDestroy(h->ev_dec);
    if (h->rstream) { (void)hipStreamSynchronize(h->rstream); (void)hipStreamDestroy(h->rstream); }
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int sph_get_sizes(SphHandle *h, SphSizes *out)
{
    if (!h || !out) return SPH_E_INVALID;
    out->n_fluid = h->N; out->n_wall = h->Nb; out->n_rigid = h->rigid ? h->Nr : 0;
    out->grid[0] = h->c.gx; out->grid[1] = h->c.gy; out->grid[2] = h->c.gz;
    out->n_cells = h->c.C;
    out->max_neighbors = h->c.kmax; out->max_wall_neighbors = h->c.kbmax;
    return SPH_OK;
}

int sph_upload(SphHandle *h, int species, int field, const float *host, size_t n_floats)
{
    if (!h || !host) return SPH_E_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    size_t count; bool vec;
    int rc = field_floats(h, species, field, &count, &vec);
    if (rc) return rc;
    if (count != n_floats) return fail(h, SPH_E_INVALID, "field %d holds %zu floats, got %zu", field, count, n_floats);
    if (species != SPH_SPECIES_FLUID || !(field == SPH_F_POS || field == SPH_F_VEL || field == SPH_F_WARM_K))
        return fail(h, SPH_E_INVALID, "field %d is read-only", field);
    if (h->slab) return fail(h, SPH_E_STATE, "sph_upload is not available on a slab handle");
    if (field == SPH_F_WARM_K && h->cfg.solver != SPH_SOLVER_DFSPH) return fail(h, SPH_E_STATE, "warm_start_k needs a dfsph handle");
    hipStream_t s = h->stream;
    HIP_TRY(h, hipMemcpyAsync(h->staging, host, sizeof(float) * count, hipMemcpyHostToDevice, s));
    ProfScope ps(h, K_TRANSFER);
    const dim3 g = grid_for(h->N), b(kBlock);
    if (field == SPH_F_POS) hipLaunchKernelGGL(k_sort_in_vec, g, b, 0, s, h->N, h->staging, h->id[h->icur], h->P[h->pcur]);
    else if (field == SPH_F_VEL) hipLaunchKernelGGL(k_sort_in_vec, g, b, 0, s, h->N, h->staging, h->id[h->icur], h->V[h->vcur]);
    else hipLaunchKernelGGL(k_sort_in_scalar, g, b, 0, s, h->N, h->staging, h->id[h->icur], h->warm[h->wcur]);
    HIP_TRY(h, hipGetLastError());
    if (h->verlet) {              // new positions: the Verlet lists of the last build no longer apply
        h->ds_host->moved = 1;
        HIP_TRY(h, hipMemcpyAsync(&h->ds->moved, &h->ds_host->moved, sizeof(int), hipMemcpyHostToDevice, s));
    }
    HIP_TRY(h, hipStreamSynchronize(s));
    h->nl_valid = false;
    h->density_valid = false;
    return SPH_OK;
}

int sph_download(SphHandle *h, int species, int field, float *host, size_t n_floats)
{
    if (!h || !host) return SPH_E_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    size_t count; bool vec;
    int rc = field_floats(h, species, field, &count, &vec);
    if (rc) return rc;
    if (count != n_floats) return fail(h, SPH_E_INVALID, "field %d holds %zu floats, got %zu", field, count, n_floats);
    if (species == SPH_SPECIES_WALL) {
        const std::vector<float> &src = field == SPH_F_WALL_POS ? h->wall_pos_host : h->wall_vol_host;
        memcpy(host, src.data(), sizeof(float) * count);
        return SPH_OK;
    }
    if (species == SPH_SPECIES_RIGID) {
        if (field == SPH_F_RIGID_VOL || field == SPH_F_RIGID_MASS) {
            memcpy(host, (field == SPH_F_RIGID_VOL ? h->rvol_host : h->rmass_host).data(), sizeof(float) * count);
            return SPH_OK;
        }
        hipStream_t s = h->stream;
        if (field == SPH_F_RIGID_POS) {
            hipLaunchKernelGGL(k_copy_vec_local, grid_for(h->Nr), dim3(kBlock), 0, s, h->Nr, h->RPos, h->staging);
            HIP_TRY(h, hipGetLastError());
            HIP_TRY(h, hipMemcpyAsync(host, h->staging, sizeof(float) * count, hipMemcpyDeviceToHost, s));
        } else {
            HIP_TRY(h, hipMemcpyAsync(host, field == SPH_F_RIGID_FORCE ? h->rforce : h->rvert, sizeof(float) * count, hipMemcpyDeviceToHost, s));
        }
        HIP_TRY(h, hipStreamSynchronize(s));
        return SPH_OK;
    }
    if (h->slab) return fail(h, SPH_E_STATE, "slab handle: use sph_download_local + sph_download_ids (device order, owned and ghost particles)");
    const bool dfsph = h->cfg.solver == SPH_SOLVER_DFSPH;
    const bool pcisph = h->cfg.solver == SPH_SOLVER_PCISPH, iisph = h->cfg.solver == SPH_SOLVER_IISPH;
    hipStream_t s = h->stream;
    const dim3 g = grid_for(h->N), b(kBlock);
    const int *id = h->id[h->icur];
    {
        ProfScope ps(h, K_TRANSFER);
        switch (field) {
        case SPH_F_POS: hipLaunchKernelGGL(k_unsort_vec, g, b, 0, s, h->N, h->P[h->pcur], id, h->staging); break;
        case SPH_F_VEL: hipLaunchKernelGGL(k_unsort_vec, g, b, 0, s, h->N, h->V[h->vcur], id, h->staging); break;
        case SPH_F_ACC:
            if (h->cfg.solver != SPH_SOLVER_WCSPH) return fail(h, SPH_E_STATE, "acc is a wcsph field (the other solvers never fill it, dfsph_solver.py:418-421)");
            hipLaunchKernelGGL(k_unsort_vec, g, b, 0, s, h->N, h->VA[0], id, h->staging); break;
        case SPH_F_PRESS_ITER:
            if (!is_pressure_solver(h)) return fail(h, SPH_E_STATE, "press_iter / p_iter is a pcisph / iisph field");
            hipLaunchKernelGGL(k_unsort_w, g, b, 0, s, h->N, h->X[3 + h->pb_final], id, h->staging); break;
        case SPH_F_PRESS_FORCE:
            if (!is_pressure_solver(h)) return fail(h, SPH_E_STATE, "press_force / f_press is a pcisph / iisph field");
            hipLaunchKernelGGL(k_unsort_vec, g, b, 0, s, h->N, h->X[pcisph ? 1 : 2], id, h->staging); break;
        case SPH_F_POS_PREDICT:
            if (h->cfg.solver == SPH_SOLVER_PBF) { hipLaunchKernelGGL(k_unsort_vec, g, b, 0, s, h->N, h->P[h->pcur], id, h->staging); break; }   // pos = pos_predict after a step (:84)
            if (!pcisph) return fail(h, SPH_E_STATE, "pos_predict is a pcisph / pbf field");
            hipLaunchKernelGGL(k_unsort_vec, g, b, 0, s, h->N, h->X[2], id, h->staging); break;
        case SPH_F_PBF_LAMBDA: case SPH_F_PBF_DELTA_POS:
            if (h->cfg.solver != SPH_SOLVER_PBF) return fail(h, SPH_E_STATE, "pbf_lambda / delta_pos are pbf fields");
            if (h->simulate_cnt == 0) return fail(h, SPH_E_STATE, "pbf_lambda / delta_pos exist after the first step");
            // (the device order does not change inside a step: the ids of the current generation apply)
            if (field == SPH_F_PBF_LAMBDA) hipLaunchKernelGGL(k_unsort_scalar, g, b, 0, s, h->N, h->aux, id, h->staging);
            else hipLaunchKernelGGL(k_unsort_vec, g, b, 0, s, h->N, h->X[0], id, h->staging);
            break;
        case SPH_F_D_II: case SPH_F_D_IJ:
            if (!iisph) return fail(h, SPH_E_STATE, "d_ii / d_ij are iisph fields");
            hipLaunchKernelGGL(k_unsort_vec, g, b, 0, s, h->N, h->X[field == SPH_F_D_II ? 0 : 1], id, h->staging); break;
        case SPH_F_A_II:
            if (!iisph) return fail(h, SPH_E_STATE, "a_ii is an iisph field");
            hipLaunchKernelGGL(k_unsort_scalar, g, b, 0, s, h->N, h->aux, id, h->staging); break;
        case SPH_F_VEL_ADV:
            if (iisph) { hipLaunchKernelGGL(k_unsort_vec, g, b, 0, s, h->N, h->VA[0], id, h->staging); break; }   // iisph v_adv
            if (!dfsph) return fail(h, SPH_E_STATE, "vel_adv is a dfsph / iisph field");
            hipLaunchKernelGGL(k_unsort_vec, g, b, 0, s, h->N, h->VA[0], id, h->staging); break;
        case SPH_F_RHO: hipLaunchKernelGGL(k_unsort_scalar, g, b, 0, s, h->N, h->rho, id, h->staging); break;
        case SPH_F_PRESSURE:
            if (h->cfg.solver != SPH_SOLVER_WCSPH) return fail(h, SPH_E_STATE, "pressure is a wcsph field");
            hipLaunchKernelGGL(k_unsort_scalar, g, b, 0, s, h->N, h->aux, id, h->staging); break;
        case SPH_F_ALPHA:
            if (!dfsph) return fail(h, SPH_E_STATE, "alpha is a dfsph field");
            hipLaunchKernelGGL(k_unsort_scalar, g, b, 0, s, h->N, h->aux, id, h->staging); break;
        case SPH_F_WARM_K:
            if (!dfsph) return fail(h, SPH_E_STATE, "warm_start_k is a dfsph field");
            hipLaunchKernelGGL(k_unsort_scalar, g, b, 0, s, h->N, h->warm[h->wcur], id, h->staging); break;
        case SPH_F_RHO_ADV: hipLaunchKernelGGL(k_unsort_scalar, g, b, 0, s, h->N, h->rho_adv, id, h->staging); break;
        case SPH_F_RHO_DER: hipLaunchKernelGGL(k_unsort_scalar, g, b, 0, s, h->N, h->drho, id, h->staging); break;
        case SPH_F_NBR_COUNT:
            if (rigid_coupled(h)) hipLaunchKernelGGL(k_unsort_scalar_int, g, b, 0, s, h->N, h->ncount, id, h->staging);
            else hipLaunchKernelGGL(k_unsort_count, g, b, 0, s, h->N, h->cnt, id, h->staging);
            break;
        default: return fail(h, SPH_E_INVALID, "field %d cannot be downloaded", field);
        }
    }
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(host, h->staging, sizeof(float) * count, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return SPH_OK;
}

int sph_plan_slabs(const SphConfig *cfg, int32_t *cuts, int32_t *counts)
{
    // host-only: runs the same scene construction as sph_create on a throw-away handle, no device needed
    if (!cfg || !cuts || !counts || cfg->slab_count < 1) return fail(nullptr, SPH_E_INVALID, "bad argument");
    SphHandle tmp;
    tmp.cfg = *cfg;
    HostScene sc;
    const int nslab = cfg->slab_count;
    tmp.cfg.slab_count = 1;                  // build the full lattice
    tmp.cfg.slab_rank = 0;
    int rc = build_scene(&tmp, sc);
    if (rc) { g_create_error = tmp.err; return rc; }
    std::vector<int> col, cut;
    std::string why;
    if (!plan_slab_cuts(sc.fluid_pos, tmp.N, tmp.c.h, tmp.c.gx, nslab, col, cut, why, slab_layers_of(*cfg))) return fail(nullptr, SPH_E_INVALID, "%s", why.c_str());
    for (int k = 0; k <= nslab; ++k) cuts[k] = cut[k];
    for (int k = 0; k < nslab; ++k) counts[k] = 0;
    for (int i = 0; i < tmp.N; ++i)
        for (int k = 0; k < nslab; ++k)
            if (col[i] >= cut[k] && col[i] < cut[k + 1]) { counts[k]++; break; }
    return SPH_OK;
}

int sph_replan_slabs(const int64_t *column_histogram, int32_t grid_x, int32_t slab_count, const int32_t *old_cuts, int32_t ghost_layers, int32_t *new_cuts)
{
    if (!column_histogram || !old_cuts || !new_cuts || grid_x < 2 || slab_count < 1 || grid_x < kMinSlabColumns * slab_count || ghost_layers < 0 || ghost_layers > 2)
        return fail(nullptr, SPH_E_INVALID, "bad argument");
    for (int k = 0; k < slab_count; ++k)
        if (old_cuts[k + 1] < old_cuts[k] + kMinSlabColumns || old_cuts[0] != 0 || old_cuts[slab_count] != grid_x)
            return fail(nullptr, SPH_E_INVALID, "old cuts must start at 0, end at grid_x and leave every slab >= %d columns", kMinSlabColumns);
    std::vector<long long> hist(column_histogram, column_histogram + grid_x);
    std::vector<int> oldc(old_cuts, old_cuts + slab_count + 1), cut;
    replan_slab_cuts(hist, grid_x, slab_count, oldc, cut, ghost_layers);
    for (int k = 0; k <= slab_count; ++k) new_cuts[k] = cut[k];
    return SPH_OK;
}

int sph_rccl_unique_id(void *id128)
{
    if (!id128) return SPH_E_INVALID;
    RcclApi &n = rccl();
    if (!n.ok) return fail(nullptr, SPH_E_STATE, "RCCL is not available: %s", n.why.c_str());
    ncclUniqueId id;
    ncclResult_t r = n.GetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, SPH_E_HIP, "ncclGetUniqueId failed: %s", n.GetErrorString(r));
    memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return SPH_OK;
}

int sph_rccl_attach(SphHandle *h, const void *id128, size_t capacity_bytes)
{
    if (!h || !id128) return SPH_E_INVALID;
    RcclApi &n = rccl();
    if (!n.ok) return fail(h, SPH_E_STATE, "RCCL is not available: %s", n.why.c_str());
    (void)dev_env(&h->overrides, "SPH_RCCL_LIB");       // reported by sph_overrides() when a stand-in transport library is in force
    if (h->native) return fail(h, SPH_E_STATE, "the native transport is already attached");
    if (capacity_bytes < 4096) return fail(h, SPH_E_INVALID, "halo buffers smaller than 4 KiB");
    HIP_TRY(h, hipSetDevice(h->device));
    // a single-GPU handle may attach as a communicator of one rank: that is all a 1-GPU box can exercise (tests), it never exchanges
    const int rank = h->slab ? h->slab_rank : 0, world = h->slab ? h->nslab : 1;
    ncclUniqueId id;
    memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    NCCL_TRY(h, n.CommInitRank(&h->nccl, world, id, rank));
    if (h->own_dev_comm) { (void)hipFree(h->dsend[0]); (void)hipFree(h->dsend[1]); (void)hipFree(h->drecv[0]); (void)hipFree(h->drecv[1]); }
    for (int k = 0; k < 2; ++k) {
        HIP_TRY(h, hipMalloc(&h->dsend[k], capacity_bytes));
        HIP_TRY(h, hipMalloc(&h->drecv[k], capacity_bytes));
    }
    h->own_dev_comm = true;
    if (h->own_red) (void)hipFree(h->red_dev);
    h->red_cap = (int)std::max<size_t>(std::max(1024, h->c.gx + 8), slab_reduce_need(h));      // the re-balancing histogram (gx counts) and a rigid body's by-id sums go through it too
    HIP_TRY(h, hipMalloc((void **)&h->red_dev, sizeof(double) * (size_t)h->red_cap));
    h->own_red = true;
    HIP_TRY(h, hipHostMalloc((void **)&h->red_host, sizeof(double) * (size_t)h->red_cap, hipHostMallocDefault));
    if (h->gath_dev) (void)hipFree(h->gath_dev);
    HIP_TRY(h, hipMalloc((void **)&h->gath_dev, sizeof(double) * 4 * (size_t)std::max(world, 1)));
    HIP_TRY(h, hipMemsetAsync(h->gath_dev, 0, sizeof(double) * 4 * (size_t)std::max(world, 1), h->stream));
    HIP_TRY(h, hipMalloc((void **)&h->cnt_dev, sizeof(int) * 4 * kCountInts));
    HIP_TRY(h, hipHostMalloc((void **)&h->cnt_host, sizeof(int) * 4 * kCountInts, hipHostMallocDefault));
    memset(&h->comm, 0, sizeof(h->comm));
    h->comm.capacity = capacity_bytes;
    h->comm.stream_ordered = 1;
    // the native transport starts in order (one stream; the residual's triple rides with the halo): with transfers and collectives that take
    // 0-50 us to start it is the faster protocol in every replay (profiles/r04/loopback/link_latency_sweep.txt); slab_overlap = 2 starts overlapped,
    // sph_slab_set_overlap switches between steps (bench.py times both)
    h->overlap_on = h->cfg.slab_overlap == 2;
    h->native = true;
    h->comm_set = true;
    return SPH_OK;
}

int sph_rccl_selftest(SphHandle *h, double *inout, int32_t n, int32_t op)
{
    if (!h || !inout || n < 1) return SPH_E_INVALID;
    if (!h->native) return fail(h, SPH_E_STATE, "attach the native transport first (sph_rccl_attach)");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = slab_allreduce_host(h, inout, n, op);       // H2D, ncclAllReduce on the handle's stream, D2H
    if (rc) return rc;
    int32_t rl = -1, rr = -1;
    if ((rc = native_exchange_counts(h, 11, 22, &rl, &rr))) return rc;
    if (!h->slab && (rl != 0 || rr != 0)) return fail(h, SPH_E_STATE, "a one-rank communicator has no neighbours");
    return native_exchange(h, 0, 0, 0, 0);
}

int32_t sph_abi_version(void) { return SPH_ABI_VERSION; }

int sph_set_comm_sized(SphHandle *h, const SphComm *comm, size_t comm_size)
{
    if (!h || !comm) return SPH_E_INVALID;
    if (comm_size < offsetof(SphComm, allreduce_stream)) return fail(h, SPH_E_INVALID, "SphComm of %zu bytes is older than any this library knows", comm_size);
    SphComm full;
    memset(&full, 0, sizeof(full));
    memcpy(&full, comm, std::min(comm_size, sizeof(full)));
    return sph_set_comm(h, &full);
}

int sph_set_comm(SphHandle *h, const SphComm *comm)
{
    if (!h || !comm) return SPH_E_INVALID;
    if (!h->slab) return fail(h, SPH_E_STATE, "sph_set_comm needs a handle created with slab_count > 1");
    if (h->native) return fail(h, SPH_E_STATE, "the native RCCL transport is attached to this handle");
    if (!comm->exchange_counts || !comm->exchange_buffers || !comm->allreduce) return fail(h, SPH_E_INVALID, "SphComm callbacks must all be set");
    if (!comm->send_left || !comm->send_right || !comm->recv_left || !comm->recv_right || comm->capacity < 4096)
        return fail(h, SPH_E_INVALID, "SphComm buffers missing or smaller than 4 KiB");
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->own_dev_comm) { (void)hipFree(h->dsend[0]); (void)hipFree(h->dsend[1]); (void)hipFree(h->drecv[0]); (void)hipFree(h->drecv[1]); h->own_dev_comm = false; }
    h->comm = *comm;
    if (comm->on_host) {
        for (int k = 0; k < 2; ++k) {
            HIP_TRY(h, hipMalloc(&h->dsend[k], comm->capacity));
            HIP_TRY(h, hipMalloc(&h->drecv[k], comm->capacity));
        }
        h->own_dev_comm = true;
    } else {
        h->dsend[0] = comm->send_left; h->dsend[1] = comm->send_right;
        h->drecv[0] = comm->recv_left; h->drecv[1] = comm->recv_right;
    }
    if (h->own_red) { (void)hipFree(h->red_dev); h->own_red = false; }
    h->red_dev = nullptr;
    if (comm->allreduce_stream) {
        if (!comm->reduce_buf) return fail(h, SPH_E_INVALID, "SphComm.allreduce_stream needs reduce_buf");
        const size_t have = comm->reduce_capacity ? comm->reduce_capacity : 4;
        if (have < slab_reduce_need(h))
            return fail(h, SPH_E_INVALID, "SphComm.reduce_buf holds %zu doubles, this handle (rigid body of %d samples) needs %zu: set reduce_capacity", have, h->Nr, slab_reduce_need(h));
        h->red_cap = (int)std::min<size_t>(have, 0x7fffffff);
        if (comm->on_host) { HIP_TRY(h, hipMalloc((void **)&h->red_dev, sizeof(double) * have)); h->own_red = true; }
        else h->red_dev = comm->reduce_buf;
    } else if (h->rigid) {
        return fail(h, SPH_E_INVALID, "a rigid body on a slab handle needs a transport with allreduce_stream");
    }
    if (comm->stream_ordered && comm->on_host) return fail(h, SPH_E_INVALID, "a stream-ordered transport needs device buffers (on_host = 0)");
    h->overlap_on = true;          // (a synchronous transport is slow: the overlapped protocol wins there, profiles/r04/rehearsal_2ranks.json)
    h->comm_set = true;
    return SPH_OK;
}

int sph_get_stream(SphHandle *h, void **stream)
{
    if (!h || !stream) return SPH_E_INVALID;
    *stream = (void *)h->stream;
    return SPH_OK;
}

int sph_comm_stats(SphHandle *h, int64_t *out, int reset)
{
    if (!h || !out) return SPH_E_INVALID;
    for (int k = 0; k < 8; ++k) out[k] = (int64_t)h->comm_stat[k];
    if (reset) for (int k = 0; k < 8; ++k) h->comm_stat[k] = 0;
    return SPH_OK;
}

int sph_slab_info(SphHandle *h, int32_t *out)
{
    if (!h || !out) return SPH_E_INVALID;
    out[0] = h->n_owned; out[1] = h->slab ? h->c.n - h->n_owned : 0;
    out[2] = h->geom.x_lo; out[3] = h->slab ? h->geom.x_hi : h->c.gx; out[4] = h->ncap;
    out[5] = h->n_recuts; out[6] = h->rebalance_every;
    // the halo protocol in force: ghost columns per side | 16 if the residual sweeps run edge tiles first with the halo on its own stream | 32 if the
    // residual's all-reduce and loop decision run on a third stream under the next correction sweep (both need a transport that can: slab_can_overlap)
    out[7] = !h->slab ? 0 : h->geom.layers | ((h->geom.layers == 2 && h->comm_set && slab_can_overlap(h)) ? 16 : 0) |
                            ((h->geom.layers == 2 && h->comm_set && slab_can_overlap(h) && slab_async(h) && h->rstream && is_dfsph(h)) ? 32 : 0);
    return SPH_OK;
}

// Between steps, on every slab alike: run the dfsph loops with the halo and the reductions on their own streams (1; needs a handle created with
// slab_overlap != 1) or in order on the handle's stream (0).  The bits do not change; which is faster depends on the link and on the slab's
// size -- with the link time at zero a rank of 1.2 M particles steps 12 % faster in order, on a slow link the overlap wins:
// bench.py times both on the node it runs on.
int sph_slab_set_overlap(SphHandle *h, int32_t on)
{
    if (!h) return SPH_E_INVALID;
    if (!h->slab) return fail(h, SPH_E_STATE, "not a slab handle");
    if (on && !h->overlap) return fail(h, SPH_E_STATE, "this handle was created without the overlapped protocol (slab_overlap = 1, or one ghost column)");
    h->overlap_on = on != 0;
    return SPH_OK;
}

int sph_download_ids(SphHandle *h, int32_t *host, size_t n)
{
    if (!h || !host) return SPH_E_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    if (n != (size_t)h->c.n) return fail(h, SPH_E_INVALID, "%d particles resident, got room for %zu", h->c.n, n);
    HIP_TRY(h, hipMemcpyAsync(host, h->id[h->icur], sizeof(int) * n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SPH_OK;
}

int sph_download_local(SphHandle *h, int field, float *host, size_t n_floats)
{
    if (!h || !host) return SPH_E_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    const bool dfsph = h->cfg.solver == SPH_SOLVER_DFSPH;
    const size_t n = (size_t)h->c.n;
    const float4 *vec = nullptr; const float *sca = nullptr;
    switch (field) {
    case SPH_F_POS: vec = h->P[h->pcur]; break;
    case SPH_F_VEL: vec = h->V[h->vcur]; break;
    case SPH_F_VEL_ADV: if (dfsph) vec = h->VA[0]; break;
    case SPH_F_ACC: if (!dfsph) vec = h->VA[0]; break;
    case SPH_F_RHO: sca = h->rho; break;
    case SPH_F_PRESSURE: if (!dfsph) sca = h->aux; break;
    case SPH_F_ALPHA: if (dfsph) sca = h->aux; break;
    case SPH_F_WARM_K: if (dfsph) sca = h->warm[h->wcur]; break;
    case SPH_F_RHO_ADV: sca = h->rho_adv; break;
    case SPH_F_RHO_DER: sca = h->drho; break;
    default: break;
    }
    if (!vec && !sca) return fail(h, SPH_E_INVALID, "field %d cannot be downloaded from this handle", field);
    const size_t want = vec ? 3 * n : n;
    if (n_floats != want) return fail(h, SPH_E_INVALID, "field %d holds %zu floats locally, got %zu", field, want, n_floats);
    if (vec) {
        ProfScope ps(h, K_TRANSFER);
        hipLaunchKernelGGL(k_copy_vec_local, grid_for((int)n), dim3(kBlock), 0, h->stream, (int)n, vec, h->staging);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipMemcpyAsync(host, h->staging, sizeof(float) * want, hipMemcpyDeviceToHost, h->stream));
    } else {
        HIP_TRY(h, hipMemcpyAsync(host, sca, sizeof(float) * want, hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SPH_OK;
}

int sph_build_neighbors(SphHandle *h)
{
    if (!h) return SPH_E_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = stage_sort_and_lists(h);
    if (rc) return rc;
    if ((rc = read_scalars(h))) return rc;
    return check_overflow_all(h);
}

int sph_compute_density(SphHandle *h)
{
    if (!h) return SPH_E_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    int rc;
    if (!h->nl_valid) {
        // pbf_lambda is a per-particle field of the solver that compute_all_rho does not touch; it lives in device order (aux), which the
        // re-sort below changes: carry it through in API order
        const bool keep = h->cfg.solver == SPH_SOLVER_PBF && h->simulate_cnt > 0;
        if (keep) hipLaunchKernelGGL(k_unsort_scalar, grid_for(h->N), dim3(kBlock), 0, h->stream, h->N, h->aux, h->id[h->icur], h->staging);
        if ((rc = sph_build_neighbors(h))) return rc;
        if (keep) hipLaunchKernelGGL(k_sort_in_scalar, grid_for(h->N), dim3(kBlock), 0, h->stream, h->N, h->staging, h->id[h->icur], h->aux);
    }
    if (h->density_valid) return SPH_OK;
    if ((rc = stage_density(h))) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SPH_OK;
}

int sph_compute_alpha(SphHandle *h)
{
    if (!h) return SPH_E_INVALID;
    if (h->cfg.solver != SPH_SOLVER_DFSPH) return fail(h, SPH_E_STATE, "alpha needs a dfsph handle");
    return sph_compute_density(h);   // the fused sweep produces rho and alpha together
}

int sph_step_wcsph(SphHandle *h, int nsteps)
{
    if (!h) return SPH_E_INVALID;
    if (h->cfg.solver != SPH_SOLVER_WCSPH) return fail(h, SPH_E_STATE, "handle was not created for wcsph");
    HIP_TRY(h, hipSetDevice(h->device));
    int k = 0;
    // The WCSPH step is a fixed launch sequence with no host decision in it, so two steps (after which the ping-pong
    // buffers are back in the same roles) are captured once into a hipGraph and replayed: at 30k particles the step is
    // launch-bound and replay halves it.  Eager launches remain for odd remainders, profiling and slab handles.
    if (h->graphs_enabled && !h->profiling && !h->slab && !h->rigid && nsteps >= 2) {
        while (nsteps - k >= 2) {
            const int key = h->pcur | (h->vcur << 1) | (h->icur << 2);
            if (!h->wcsph_graph[key]) {
                hipGraph_t graph = nullptr;
                const int sim_cnt = h->simulate_cnt;
                if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { h->graphs_enabled = false; break; }
                int rc = step_wcsph_once(h);
                if (!rc) rc = step_wcsph_once(h);
                hipError_t e = hipStreamEndCapture(h->stream, &graph);
                h->simulate_cnt = sim_cnt;             // the capture launched nothing
                if (rc || e != hipSuccess || !graph) { if (graph) (void)hipGraphDestroy(graph); h->graphs_enabled = false; (void)hipGetLastError(); break; }
                e = hipGraphInstantiate(&h->wcsph_graph[key], graph, nullptr, nullptr, 0);
                (void)hipGraphDestroy(graph);
                if (e != hipSuccess) { h->wcsph_graph[key] = nullptr; h->graphs_enabled = false; (void)hipGetLastError(); break; }
                if ((h->pcur | (h->vcur << 1) | (h->icur << 2)) != key) { h->graphs_enabled = false; break; }   // roles must return after 2 steps
            }
            HIP_TRY(h, hipGraphLaunch(h->wcsph_graph[key], h->stream));
            h->graph_launches += 1;
            h->simulate_cnt += 2;
            h->nl_valid = false; h->density_valid = false;
            k += 2;
        }
    }
    for (; k < nsteps; ++k) {
        int rc = step_wcsph_once(h);
        if (rc) return rc;
    }
    // overflow is sticky within a call: one read-back per call keeps the steps asynchronous
    int rc = read_scalars(h);
    if (rc) return rc;
    return check_overflow_all(h);
}

int sph_step_dfsph(SphHandle *h, int nsteps, SphStepStats *last)
{
    if (!h) return SPH_E_INVALID;
    if (h->cfg.solver != SPH_SOLVER_DFSPH) return fail(h, SPH_E_STATE, "handle was not created for dfsph");
    HIP_TRY(h, hipSetDevice(h->device));
    SphStepStats st;
    memset(&st, 0, sizeof(st));
    for (int k = 0; k < nsteps; ++k) {
        int rc = step_dfsph_once(h, &st);
        if (rc) return rc;
    }
    if (last) *last = st;
    return SPH_OK;
}

int sph_step_pcisph(SphHandle *h, int nsteps, SphStepStats *last)
{
    if (!h) return SPH_E_INVALID;
    if (h->cfg.solver != SPH_SOLVER_PCISPH) return fail(h, SPH_E_STATE, "handle was not created for pcisph");
    HIP_TRY(h, hipSetDevice(h->device));
    SphStepStats st;
    memset(&st, 0, sizeof(st));
    for (int k = 0; k < nsteps; ++k) {
        int rc = step_pcisph_once(h, &st);
        if (rc) return rc;
    }
    if (last) *last = st;
    return SPH_OK;
}

int sph_step_iisph(SphHandle *h, int nsteps, SphStepStats *last)
{
    if (!h) return SPH_E_INVALID;
    if (h->cfg.solver != SPH_SOLVER_IISPH) return fail(h, SPH_E_STATE, "handle was not created for iisph");
    HIP_TRY(h, hipSetDevice(h->device));
    SphStepStats st;
    memset(&st, 0, sizeof(st));
    for (int k = 0; k < nsteps; ++k) {
        int rc = step_iisph_once(h, &st);
        if (rc) return rc;
    }
    if (last) *last = st;
    return SPH_OK;
}

int sph_step_pbf(SphHandle *h, int nsteps)
{
    if (!h || nsteps < 0) return SPH_E_INVALID;
    if (h->cfg.solver != SPH_SOLVER_PBF) return fail(h, SPH_E_STATE, "handle was not created with solver = SPH_SOLVER_PBF");
    if (h->slab) return fail(h, SPH_E_STATE, "pbf is not available on slab handles");
    HIP_TRY(h, hipSetDevice(h->device));
    for (int k = 0; k < nsteps; ++k) {
        int rc = step_pbf_once(h);
        if (rc) return rc;
    }
    if (nsteps > 0) {
        int rc = read_scalars(h);
        if (rc) return rc;
        if ((rc = check_overflow(h))) return rc;
    }
    return SPH_OK;
}

int sph_get_scalar(SphHandle *h, int which, double *out)
{
    if (!h || !out) return SPH_E_INVALID;
    switch (which) {
    case SPH_S_DELTA_TIME:
        if (h->cfg.solver != SPH_SOLVER_DFSPH) { *out = (double)h->dt_wcsph; return SPH_OK; }   // only dfsph adapts delta_time
        else { int rc = read_scalars(h); if (rc) return rc; *out = (double)h->ds_host->dt; return SPH_OK; }
    case SPH_S_SIMULATE_CNT: *out = (double)h->simulate_cnt; return SPH_OK;
    case SPH_S_PARTICLE_M: *out = (double)h->c.m; return SPH_OK;
    case SPH_S_SUPPORT_RADIUS: *out = (double)h->c.h; return SPH_OK;
    case SPH_S_GRAPH_LAUNCHES: *out = (double)h->graph_launches; return SPH_OK;
    case SPH_S_PCISPH_DELTA: *out = (double)h->pci_delta; return SPH_OK;
    case SPH_S_PCISPH_BETA: *out = (double)h->pci_beta; return SPH_OK;
    case SPH_S_PCISPH_MAX_INDEX: *out = (double)h->pci_max_index; return SPH_OK;
    case SPH_S_PCISPH_MAX_COUNT: *out = (double)h->pci_max_count; return SPH_OK;
    case SPH_S_PS_DELTA_TIME: { int rc = read_scalars(h); if (rc) return rc; *out = (double)h->ds_host->ps_dt; return SPH_OK; }
    case SPH_S_ARITH_RELAXED: *out = (use_relaxed(h) || h->verlet || relaxed_pressure(h) || relaxed_unstaged(h)) ? 1.0 : 0.0; return SPH_OK;
    case SPH_S_VERLET_BUILDS: { int rc = read_scalars(h); if (rc) return rc; *out = (double)h->ds_host->verlet_builds; return SPH_OK; }      // kr_split is settled by the first list build
    case SPH_P_DENSITY_THRESHOLD: *out = h->p.density_threshold; return SPH_OK;
    case SPH_P_MIN_ITERATION_DENSITY: *out = h->p.min_iteration_density; return SPH_OK;
    case SPH_P_MIN_ITERATION_DENSITY_DIVERGENCE: *out = h->p.min_iteration_density_divergence; return SPH_OK;
    case SPH_P_MAX_ITERATION_DENSITY_DIVERGENCE: *out = h->p.max_iteration_density_divergence; return SPH_OK;
    case SPH_P_DENSITY_DIVERGENCE_THRESHOLD: *out = h->p.density_divergence_threshold; return SPH_OK;
    case SPH_P_WARM_START: *out = h->p.warm_start; return SPH_OK;
    case SPH_P_ADAPTIVE_DT: *out = h->p.adaptive_dt; return SPH_OK;
    case SPH_P_MAX_DT: *out = h->p.max_dt; return SPH_OK;
    case SPH_P_MIN_DT: *out = h->p.min_dt; return SPH_OK;
    case SPH_P_VISCOSITY_C_S: *out = h->p.viscosity_c_s; return SPH_OK;
    case SPH_P_VISCOSITY_ALPHA: *out = h->p.viscosity_alpha; return SPH_OK;
    case SPH_P_VISCOSITY_EPSILON: *out = h->p.viscosity_epsilon; return SPH_OK;
    case SPH_P_TENSION_K: *out = h->p.tension_k; return SPH_OK;
    default:
        if (h->rigid && which >= SPH_S_RIGID_CENTROID && which < SPH_S_RIGID_INERTIA_INV + 9) {
            if (which < SPH_S_RIGID_OMEGA) *out = (double)h->centroid[which - SPH_S_RIGID_CENTROID];
            else if (which < SPH_S_RIGID_VEL) *out = (double)h->rs_omega[which - SPH_S_RIGID_OMEGA];
            else if (which < SPH_S_RIGID_MASS) *out = (double)h->r_vel[which - SPH_S_RIGID_VEL];
            else if (which == SPH_S_RIGID_MASS) *out = (double)h->rs_mass;
            else *out = (double)h->inertia_inv[which - SPH_S_RIGID_INERTIA_INV];
            return SPH_OK;
        }
        return fail(h, SPH_E_INVALID, "unknown scalar %d", which);
    }
}

// the solver attributes a caller of the reference edits on the solver object (SPH_P_*): validated, kept in h->p, folded into the launch constants
// and the device's loop-control block
static int set_param(SphHandle *h, int which, double value)
{
    // validated into a COPY and committed at the end: a rejected call leaves the handle -- and what sph_get_scalar reports -- as it was
    SphHandle::Params p = h->p;
    const bool dfsph = h->cfg.solver == SPH_SOLVER_DFSPH;
    auto count = [&](int *dst, int lo) -> int {
        if (!(value >= lo && value <= 100000.0) || value != std::floor(value)) return fail(h, SPH_E_INVALID, "sph_set_scalar(%d): an integer >= %d expected, got %g", which, lo, value);
        *dst = (int)value; return SPH_OK;
    };
    // the reference takes whatever the caller assigns; what is refused here is what cannot be computed with: a value that is not finite, a time step bound <= 0
    auto finite = [&](double *dst) -> int {
        if (!std::isfinite(value)) return fail(h, SPH_E_INVALID, "sph_set_scalar(%d): a finite value expected, got %g", which, value);
        *dst = value; return SPH_OK;
    };
    auto positive = [&](double *dst) -> int {
        if (!(value > 0.0) || !std::isfinite(value)) return fail(h, SPH_E_INVALID, "sph_set_scalar(%d): a finite value > 0 expected, got %g", which, value);
        *dst = value; return SPH_OK;
    };
    int rc = SPH_OK;
    if (which >= SPH_P_DENSITY_THRESHOLD && which <= SPH_P_MIN_DT && !dfsph) return fail(h, SPH_E_STATE, "sph_set_scalar(%d): a dfsph_solver attribute on a handle of another solver", which);
    if (h->cfg.solver == SPH_SOLVER_PBF && which >= SPH_P_VISCOSITY_C_S && which <= SPH_P_TENSION_K) return fail(h, SPH_E_STATE, "sph_set_scalar(%d): pbf_solver has no such attribute", which);
    switch (which) {
    case SPH_P_DENSITY_THRESHOLD: rc = finite(&p.density_threshold); break;
    case SPH_P_MIN_ITERATION_DENSITY: rc = count(&p.min_iteration_density, 0); break;
    case SPH_P_MIN_ITERATION_DENSITY_DIVERGENCE: rc = count(&p.min_iteration_density_divergence, 0); break;
    case SPH_P_MAX_ITERATION_DENSITY_DIVERGENCE: rc = count(&p.max_iteration_density_divergence, 0); break;
    case SPH_P_DENSITY_DIVERGENCE_THRESHOLD: rc = finite(&p.density_divergence_threshold); break;
    case SPH_P_WARM_START: p.warm_start = value != 0.0; break;
    case SPH_P_ADAPTIVE_DT: p.adaptive_dt = value != 0.0; break;
    case SPH_P_MAX_DT: rc = positive(&p.max_dt); break;
    case SPH_P_MIN_DT: rc = positive(&p.min_dt); break;
    case SPH_P_VISCOSITY_C_S: rc = finite(&p.viscosity_c_s); break;
    case SPH_P_VISCOSITY_ALPHA: rc = finite(&p.viscosity_alpha); break;
    case SPH_P_VISCOSITY_EPSILON: rc = finite(&p.viscosity_epsilon); break;
    case SPH_P_TENSION_K: rc = finite(&p.tension_k); break;
    default: return fail(h, SPH_E_INVALID, "unknown scalar %d", which);
    }
    if (rc) return rc;
    h->p = p;
    HIP_TRY(h, hipSetDevice(h->device));
    fold_params(h);
    if (dfsph) {          // (the mirror's other words are whatever the last read-back left: only the p_* block is written)
        loop_params(h, h->ds_host);
        HIP_TRY(h, hipMemcpyAsync(&h->ds->p_dens_thr, &h->ds_host->p_dens_thr, sizeof(DevScalars) - offsetof(DevScalars, p_dens_thr), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    for (int k = 0; k < 8; ++k)                                      // captured wcsph step pairs carry the old constants as launch arguments
        if (h->wcsph_graph[k]) { (void)hipGraphExecDestroy(h->wcsph_graph[k]); h->wcsph_graph[k] = nullptr; }
    return SPH_OK;
}

int sph_set_scalar(SphHandle *h, int which, double value)
{
    if (!h) return SPH_E_INVALID;
    if (which >= SPH_P_DENSITY_THRESHOLD && which <= SPH_P_TENSION_K) return set_param(h, which, value);      // (on slab handles too: every rank sets the same)
    if (which != SPH_S_DELTA_TIME || !(value > 0.0)) return fail(h, SPH_E_INVALID, "sph_set_scalar: SPH_S_DELTA_TIME > 0 or a solver attribute SPH_P_* can be written");
    if (h->slab) return fail(h, SPH_E_STATE, "sph_set_scalar is not available on slab handles");
    HIP_TRY(h, hipSetDevice(h->device));
    h->dt_wcsph = (float)value;                                      // the launch argument of the fixed-dt solvers
    h->cfg.delta_time = value;
    if (h->cfg.solver == SPH_SOLVER_DFSPH) {                         // dfsph keeps delta_time, delta_time_2 on the device (dfsph_solver.py:20, :118)
        int rc = read_scalars(h);
        if (rc) return rc;
        h->ds_host->dt = (float)value;
        h->ds_host->dt2 = h->ds_host->dt * h->ds_host->dt;
        HIP_TRY(h, hipMemcpyAsync(h->ds, h->ds_host, offsetof(DevScalars, ps_dt), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    for (int k = 0; k < 8; ++k)                                      // captured wcsph step pairs carry the old delta_time as a launch argument
        if (h->wcsph_graph[k]) { (void)hipGraphExecDestroy(h->wcsph_graph[k]); h->wcsph_graph[k] = nullptr; }
    return SPH_OK;
}

const char *sph_overrides(SphHandle *h) { return h ? h->overrides.c_str() : ""; }

int sph_synchronize(SphHandle *h)
{
    if (!h) return SPH_E_INVALID;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SPH_OK;
}

int sph_profile_enable(SphHandle *h, int on)
{
    if (!h) return SPH_E_INVALID;
    drain_profile(h);
    h->profiling = on != 0;
    return SPH_OK;
}

int sph_profile_reset(SphHandle *h)
{
    if (!h) return SPH_E_INVALID;
    drain_profile(h);
    for (int k = 0; k < K_COUNT; ++k) { h->prof_ms[k] = 0; h->prof_n[k] = 0; }
    return SPH_OK;
}

int sph_profile_kernel_count(void) { return K_COUNT; }
const char *sph_profile_kernel_name(int kid) { return (kid >= 0 && kid < K_COUNT) ? kKernelNames[kid] : ""; }

int sph_profile_get(SphHandle *h, int kid, double *total_ms, int64_t *launches)
{
    if (!h || kid < 0 || kid >= K_COUNT) return SPH_E_INVALID;
    drain_profile(h);
    if (total_ms) *total_ms = h->prof_ms[kid];
    if (launches) *launches = h->prof_n[kid];
    return SPH_OK;
}

// Tuning aid (not part of the reference's surface): mean duration in microseconds of `reps` back-to-back launches of one DFSPH sweep
// on the handle's current state with `lds_bytes` of dynamic LDS, bracketed by one HIP event pair.  which: 0 = divergence residual
// (idempotent), 1 = divergence correction (advances the velocities: use a throw-away handle), 2 = density residual, 3 = sort + list build.
int sph_tune_time(SphHandle *h, int which, unsigned lds_bytes, int reps, double *avg_us)
{
    if (!h || !avg_us || reps < 1) return SPH_E_INVALID;
    if (h->cfg.solver != SPH_SOLVER_DFSPH || h->slab) return fail(h, SPH_E_STATE, "sph_tune_time needs a single-GPU dfsph handle");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc;
    if (!h->nl_valid && (rc = stage_sort_and_lists(h))) return rc;
    if (!h->density_valid && (rc = stage_density(h))) return rc;
    const unsigned saved = h->sweep_lds;
    h->sweep_lds = lds_bytes;
    h->tune_all = true;                    // repeated launches on one state: every tile computes (no change propagation)
    hipEvent_t a, b;
    HIP_TRY(h, hipEventCreate(&a));
    HIP_TRY(h, hipEventCreate(&b));
    HIP_TRY(h, hipEventRecord(a, h->stream));
    for (int k = 0; k < reps; ++k) {
        if (which == 0) launch_div_residual(h, GATE_NONE);
        else if (which == 1) launch_correct<CORR_DIV>(h, K_D_DIV_CORRECT, h->drho, h->V[h->vcur], GATE_NONE);
        else if (which == 2) launch_dens_residual(h, GATE_NONE);
        else if (which == 4) {      // a sweep followed by the single-workgroup reduction of its block partials, as in the solver loops
            launch_div_residual(h, GATE_NONE);
            hipLaunchKernelGGL(k_finalize_mean, dim3(1), dim3(kFinBlock), 0, h->stream, h->psum, h->pcnt, h->nblocks, h->ds, (int)FIN_PLAIN, (int)FINP_ALL, (double *)nullptr, partial_group(h), partial_count(h));
        } else if (which == 5) {    // the reduction alone
            hipLaunchKernelGGL(k_finalize_mean, dim3(1), dim3(kFinBlock), 0, h->stream, h->psum, h->pcnt, h->nblocks, h->ds, (int)FIN_PLAIN, (int)FINP_ALL, (double *)nullptr, partial_group(h), partial_count(h));
        } else if (which == 6) {    // two different sweeps alternating (residual, correct), no reduction between them
            launch_div_residual(h, GATE_NONE);
            launch_correct<CORR_DIV>(h, K_D_DIV_CORRECT, h->drho, h->V[h->vcur], GATE_NONE);
        } else if (which == 7) {    // the same with the reduction after the residual: one solver iteration
            launch_div_residual(h, GATE_NONE);
            hipLaunchKernelGGL(k_finalize_mean, dim3(1), dim3(kFinBlock), 0, h->stream, h->psum, h->pcnt, h->nblocks, h->ds, (int)FIN_PLAIN, (int)FINP_ALL, (double *)nullptr, partial_group(h), partial_count(h));
            launch_correct<CORR_DIV>(h, K_D_DIV_CORRECT, h->drho, h->V[h->vcur], GATE_NONE);
        }
        else if ((rc = stage_sort_and_lists(h))) break;
    }
    HIP_TRY(h, hipEventRecord(b, h->stream));
    HIP_TRY(h, hipEventSynchronize(b));
    float ms = 0.f;
    HIP_TRY(h, hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    h->sweep_lds = saved;
    h->tune_all = false;
    if (which == 3) h->density_valid = false;
    *avg_us = (double)ms * 1000.0 / reps;
    return SPH_OK;
}

int sph_selftest_math(int device, int op, const float *a, const float *b, float *out, size_t n)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, SPH_E_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev || !a || !b || !out) return fail(nullptr, SPH_E_INVALID, "bad argument");
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, SPH_E_HIP, "hipSetDevice failed");
    float *da = nullptr, *db = nullptr, *dout = nullptr;
    int rc = SPH_OK;
    if (hipMalloc((void **)&da, n * 4) != hipSuccess || hipMalloc((void **)&db, n * 4) != hipSuccess || hipMalloc((void **)&dout, n * 4) != hipSuccess)
        rc = fail(nullptr, SPH_E_HIP, "hipMalloc failed");
    if (!rc) {
        (void)hipMemcpy(da, a, n * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(db, b, n * 4, hipMemcpyHostToDevice);
        Consts c;
        memset(&c, 0, sizeof(c));
        c.h = 0.1f;
        const float pi_f = (float)3.141592653589793;
        const float h3 = c.h * (c.h * c.h);
        c.kw = 8.0f / (pi_f * h3);
        c.rh = 1.0f / c.h;
        const float kg = 48.0f / (pi_f * h3);
        c.kg6 = kg * 6.0f; c.neg_kg6 = -kg * 6.0f;
        hipLaunchKernelGGL(k_selftest, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, c, op, da, db, dout, n);
        if (hipDeviceSynchronize() != hipSuccess) rc = fail(nullptr, SPH_E_HIP, "selftest kernel failed");
        else (void)hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost);
    }
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
    return rc;
}

int sph_selftest_wave(int device, int op, const double *in, double *out, size_t n)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, SPH_E_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev || !in || !out || n == 0 || n % 256 != 0 || op < 0 || op > 4) return fail(nullptr, SPH_E_INVALID, "bad argument (n must be a multiple of 256)");
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, SPH_E_HIP, "hipSetDevice failed");
    double *din = nullptr, *dout = nullptr;
    int rc = SPH_OK;
    if (hipMalloc((void **)&din, n * 8) != hipSuccess || hipMalloc((void **)&dout, n * 8) != hipSuccess) rc = fail(nullptr, SPH_E_HIP, "hipMalloc failed");
    if (!rc) {
        (void)hipMemcpy(din, in, n * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_selftest_wave, dim3((unsigned)(n / 256)), dim3(256), 0, 0, op, din, dout);
        if (hipDeviceSynchronize() != hipSuccess) rc = fail(nullptr, SPH_E_HIP, "selftest kernel failed");
        else (void)hipMemcpy(out, dout, n * 8, hipMemcpyDeviceToHost);
    }
    (void)hipFree(din); (void)hipFree(dout);
    return rc;
}

}  // extern "C"
