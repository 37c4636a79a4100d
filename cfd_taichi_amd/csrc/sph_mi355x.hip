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

// ---------------------------------------------------------------------------------------------
// host-side scene construction (one-time; mirrors ParticleSystem.__init__)
// ---------------------------------------------------------------------------------------------
inline float fmod_py(float a, float b) { return a - b * floorf(a / b); }   // Taichi float %: a - b*floor(a/b)

// host twin of cubic_w for the one-time wall volumes (ParticleSystem.py:309-320)
inline float host_cubic_w(float r, float h, float kw)
{
    float ret = 0.0f;
    float q = r / h;
    if (0.0f <= q && q <= 0.5f) {
        float q2 = q * q;
        float q3 = q2 * q;
        ret = kw * (6.0f * (q3 - q2) + 1.0f);
    } else if (0.5f < q && q <= 1.0f) {
        float t = 1.0f - q;
        ret = 2.0f * kw * (t * (t * t));
    }
    return ret;
}

// ghost cell columns per side: two by default for dfsph (one halo refresh per solver iteration, see step_dfsph_device_loops); the other solvers
// keep the one-column protocol
inline int slab_layers_of(const SphConfig &cf) { return cf.slab_ghost_layers == 1 ? 1 : (cf.slab_ghost_layers == 2 || cf.solver == SPH_SOLVER_DFSPH) ? 2 : 1; }

// Every slab is at least three columns wide: with two ghost columns per side the merged particle exchange needs ghost layers + 1 (a particle
// that arrives from one neighbour must not land in the columns copied to the other, see k_classify_slab).
constexpr int kMinSlabColumns = 3;
// cut[k] = first column x with (particles in columns < x) >= k N / nslab
void cuts_from_histogram(const std::vector<long long> &hist, long long N, int gx, int nslab, std::vector<int> &cut)
{
    cut.assign((size_t)nslab + 1, 0);
    cut[nslab] = gx;
    long long pre = 0;      // particles with column < x
    int k = 1;
    for (int x = 0; x < gx && k < nslab; ++x) {
        while (k < nslab && pre >= (long long)k * N / nslab) { cut[k] = x; ++k; }
        pre += hist[x];
    }
    for (; k < nslab; ++k) cut[k] = gx;
}

// Cuts that balance what a slab COSTS, not what it owns.  A rank's step is its owned particles plus its ghosts: the ghosts of the inner column run
// the density pass and every correction sweep, all of them sit in the staged neighbourhoods, the tiles they share with owned particles run at
// part occupancy, and each cut brings a halo's fixed costs.  Measured on config 4 at 8 slabs (one rank alone on a GPU, tools/loopback_replay.sh):
// the two end ranks, one cut each, 5.5-5.7 ms per step, the six ranks between them 6.0-6.7 ms at the same owned count -- 200 k more ghosts cost
// what 200 k owned particles cost.  So: load of slab [y, x) = its particles + the particles of the `layers` columns beyond each cut it has, and
// the cuts minimise the largest load (then keep the smallest slab as large as they can, then the sum of squares), by dynamic programming over the cut positions (slabs x columns^2 steps, host,
// identical on every rank: integers only).  lo[k] <= cut[k] <= hi[k]; every slab >= kMinSlabColumns wide.  layers = 0: plain equal counts.
void balanced_cuts(const std::vector<long long> &hist, int gx, int nslab, int layers, const std::vector<int> &lo, const std::vector<int> &hi, std::vector<int> &cut)
{
    std::vector<long long> pre((size_t)gx + 1, 0);
    for (int x = 0; x < gx; ++x) pre[(size_t)x + 1] = pre[(size_t)x] + hist[(size_t)x];
    // (in quarters of a particle: a ghost weighs 5/4 -- in-order protocol, config 4 at 8 slabs: 200 k ghosts less and 21 k owned more = -0.54 ms,
    // where 100 k owned cost 0.21 ms)
    auto load = [&](int r, int y, int x) {
        long long v = 4 * (pre[(size_t)x] - pre[(size_t)y]);
        if (r > 0) v += 5 * (pre[(size_t)y] - pre[(size_t)std::max(y - layers, 0)]);
        if (r < nslab - 1) v += 5 * (pre[(size_t)std::min(x + layers, gx)] - pre[(size_t)x]);
        return v;
    };
    struct Val { long long mx, mn; double sq; int from; };          // largest load, smallest OWNED count (no slab left empty for a tie), sum of squares
    const Val none{-1, 0, 0.0, -1};
    std::vector<std::vector<Val>> best((size_t)nslab + 1, std::vector<Val>((size_t)gx + 1, none));
    best[0][0] = Val{0, 0x7fffffffffffffffLL, 0.0, -1};
    for (int k = 1; k <= nslab; ++k)
        for (int x = lo[(size_t)k]; x <= hi[(size_t)k]; ++x) {
            Val b = none;
            for (int y = lo[(size_t)k - 1]; y <= std::min(hi[(size_t)k - 1], x - kMinSlabColumns); ++y) {
                const Val &p = best[(size_t)k - 1][(size_t)y];
                if (p.mx < 0) continue;
                const long long l = load(k - 1, y, x);
                const Val c{std::max(p.mx, l), std::min(p.mn, pre[(size_t)x] - pre[(size_t)y]), p.sq + (double)l * (double)l, y};
                if (b.mx < 0 || c.mx < b.mx || (c.mx == b.mx && (c.mn > b.mn || (c.mn == b.mn && c.sq < b.sq)))) b = c;
            }
            best[(size_t)k][(size_t)x] = b;
        }
    // no assignment within the bounds (the callers' bounds always admit one: every slab >= kMinSlabColumns wide is checked where the cuts are
    // first planned): keep what the caller had rather than walk back through an empty table
    if (best[(size_t)nslab][(size_t)gx].mx < 0) return;
    std::vector<int> found((size_t)nslab + 1, 0);
    found[(size_t)nslab] = gx;
    for (int k = nslab; k >= 1; --k) {
        const int from = best[(size_t)k][(size_t)found[(size_t)k]].from;
        if (from < 0) return;
        found[(size_t)k - 1] = from;
    }
    cut = found;
}

// Re-balancing ("re-chosen every M steps because a dam break migrates mass along x"):
// new equal-count cuts from the current global column histogram, clamped so that (a) every slab keeps >= 2
// columns and (b) a particle's new owner is its current rank or a direct neighbour -- the migration step only
// talks to the left and right neighbour.  A particle resident on rank r sits in columns
// [old[r] - 1, old[r+1]] (it may have crossed one column since the last exchange), hence
// old[k-1] + 1 <= new[k] <= old[k+1] - 1.
void replan_slab_cuts(const std::vector<long long> &hist, int gx, int nslab, const std::vector<int> &old_cut, std::vector<int> &cut, int layers = 0)
{
    if (layers > 0) {          // by cost (balanced_cuts), within the same bounds
        std::vector<int> lo((size_t)nslab + 1, 0), hi((size_t)nslab + 1, gx);
        lo[(size_t)nslab] = gx; hi[0] = 0;
        for (int k = 1; k < nslab; ++k) {
            lo[(size_t)k] = std::max(old_cut[(size_t)k - 1] + 1, kMinSlabColumns * k);
            hi[(size_t)k] = std::min(old_cut[(size_t)k + 1] - 1, gx - kMinSlabColumns * (nslab - k));
        }
        cut = old_cut;                  // (kept if no assignment fits the bounds)
        balanced_cuts(hist, gx, nslab, layers, lo, hi, cut);
        return;
    }
    long long N = 0;
    for (long long v : hist) N += v;
    cuts_from_histogram(hist, N, gx, nslab, cut);
    cut[0] = 0; cut[nslab] = gx;
    for (int k = 1; k < nslab; ++k) {
        int lo = std::max(old_cut[k - 1] + 1, cut[k - 1] + kMinSlabColumns);
        int hi = std::min(old_cut[k + 1] - 1, gx - kMinSlabColumns * (nslab - k));
        cut[k] = std::min(std::max(cut[k], lo), hi);
    }
}

// Equal-count cuts along the cell x index, computed identically on every rank from the full lattice:
// slab k owns cell columns [cut[k], cut[k+1]).
bool plan_slab_cuts(const std::vector<float> &pos, int N, float hcell, int gx, int nslab, std::vector<int> &col, std::vector<int> &cut,
                    std::string &why, int layers = 0)
{
    std::vector<long long> hist((size_t)gx, 0);
    col.resize((size_t)N);
    for (int i = 0; i < N; ++i) {
        int cx = (int)floorf(pos[3 * (size_t)i] / hcell);
        cx = cx < 0 ? 0 : (cx >= gx ? gx - 1 : cx);
        col[i] = cx;
        hist[cx]++;
    }
    if (gx < kMinSlabColumns * nslab) {
        char buf[160];
        snprintf(buf, sizeof(buf), "%d slabs need at least %d cell columns along x, the grid has %d: too many slabs for this scene", nslab, kMinSlabColumns * nslab, gx);
        why = buf;
        return false;
    }
    if (layers > 0) {                   // by cost: owned particles + the ghosts of every cut (balanced_cuts)
        std::vector<int> lo((size_t)nslab + 1, 0), hi((size_t)nslab + 1, gx);
        lo[(size_t)nslab] = gx; hi[0] = 0;
        for (int k = 1; k < nslab; ++k) { lo[(size_t)k] = kMinSlabColumns * k; hi[(size_t)k] = gx - kMinSlabColumns * (nslab - k); }
        cut.clear();
        balanced_cuts(hist, gx, nslab, layers, lo, hi, cut);
        if ((int)cut.size() != nslab + 1) { why = "no slab cuts of at least three columns each fit this grid"; return false; }
        return true;
    }
    cuts_from_histogram(hist, N, gx, nslab, cut);
    for (int k = 1; k < nslab; ++k)     // every slab at least kMinSlabColumns wide, even where the fluid is narrow
        cut[k] = std::min(std::max(cut[k], cut[k - 1] + kMinSlabColumns), gx - kMinSlabColumns * (nslab - k));
    return true;
}

// this rank's columns, its neighbours' far cuts and the ghost columns whose particles own lists, from h->cuts
void set_slab_geometry(SphHandle *h)
{
    const std::vector<int> &cut = h->cuts;
    const int r = h->slab_rank;
    h->geom.x_lo = cut[r]; h->geom.x_hi = cut[r + 1];
    h->geom.far_left = r > 0 ? cut[r - 1] : 0;
    h->geom.far_right = r + 2 <= h->nslab ? cut[r + 2] : h->c.gx;
    h->c.gw_left = (h->geom.layers == 2 && h->geom.has_left) ? h->geom.x_lo - 1 : -1;
    h->c.gw_right = (h->geom.layers == 2 && h->geom.has_right) ? h->geom.x_hi : -1;
}

struct HostScene {
    std::vector<float> fluid_pos;                 // 3N, original order
    std::vector<float> wall_pos, wall_vol;        // original order
    std::vector<float4> wall_sorted;              // cell-sorted (x,y,z,V)
    std::vector<int> wcell_start;                 // C+1
};

// The Python scalars of the viscosity / tension expressions folded in f64 and rounded once, as Taichi does with a kernel's compile-time
// constants (solver_base.py:187-188, :216), and the dfsph attributes its kernels bake in (dfsph_solver.py:113-117, :396, :404)
void fold_params(SphHandle *h)
{
    Consts &c = h->c;
    const double r = h->cfg.particle_radius, m = 1000 * (r * r * r) * 8;       // ParticleSystem.py:83
    const double kernel_h = h->cfg.particle_radius * 4;               // solver_base.py:17
    c.visc_num = (float)(2 * h->p.viscosity_alpha * kernel_h * h->p.viscosity_c_s);
    c.visc_eps_h2 = (float)(h->p.viscosity_epsilon * kernel_h * kernel_h);
    c.tens_c = (float)(-h->p.tension_k / m * m);
    c.warm_start = h->p.warm_start;
    c.adaptive_dt = h->p.adaptive_dt;
    c.max_dt = (float)h->p.max_dt;
    c.min_dt = (float)h->p.min_dt;
}
// the loop parameters live next to the loop state on the device (DevScalars.p_*); `ds` = the host mirror to fill
void loop_params(const SphHandle *h, DevScalars *ds)
{
    ds->p_dens_thr = h->p.density_threshold * 1000 * 0.01;           // dfsph_solver.py:225 (rho_0 = 1000, solver_base.py:19)
    ds->p_div_thr = h->p.density_divergence_threshold;               // :400
    ds->p_min_dens = h->p.min_iteration_density;
    ds->p_min_div = h->p.min_iteration_density_divergence;
    ds->p_max_div = h->p.max_iteration_density_divergence;
}

int build_scene(SphHandle *h, HostScene &sc)
{
    const SphConfig &cf = h->cfg;
    Consts &c = h->c;
    const double r = cf.particle_radius;
    const double d = r * 2;                        // ParticleSystem.py:81
    const double support = 4 * r;                  // :82
    const double m = 1000 * (r * r * r) * 8;       // :83
    if (!(r > 0)) return fail(h, SPH_E_INVALID, "particle_radius must be > 0");
    // :85-86, Python f64, left to right
    h->N = (int)(cf.water_size[0] / d * cf.water_size[1] / d * cf.water_size[2] / d);
    {   // compute_boundary_particles_count, :129-137 (Python f64)
        double bx = cf.box_max[0] - cf.box_min[0], by = cf.box_max[1] - cf.box_min[1], bz = cf.box_max[2] - cf.box_min[2];
        int x_cnt = (int)(bx / d + 1), z_cnt = (int)(bz / d + 1);
        int bottom = x_cnt * z_cnt;
        int ring = x_cnt * z_cnt - (x_cnt - 2) * (z_cnt - 2);
        int layer = (int)std::ceil((by - d) / d);
        h->Nb = layer * ring + bottom * 2;
    }
    h->Nr = 0;
    // Verlet lists (wcsph, relaxed arithmetic, one GPU, no body): cells of edge h + skin and lists of every pair within it, rebuilt only when a
    // particle has moved skin / 2 (sph_relaxed_kernels.h).  SPH_VERLET_SKIN sets the skin as a fraction of h (0 turns the reuse off).
    double skin = 0.0;
    h->verlet = cf.solver == SPH_SOLVER_WCSPH && h->relaxed && cf.slab_count <= 1 && !g_creating_with_rigid;
    if (h->verlet) {
        const char *e = dev_env(&h->overrides, "SPH_VERLET_SKIN");
        skin = e ? std::min(std::max(atof(e), 0.0), 0.5) : 0.05;
    }
    const double cell_edge = support * (1.0 + skin);
    int g[3];
    for (int a = 0; a < 3; ++a) g[a] = (int)std::ceil((cf.box_max[a] - cf.box_min[a]) / cell_edge) + 1;   // :100-101 (cell_edge = support but on Verlet handles)
    if (h->N <= 0) return fail(h, SPH_E_INVALID, "scene has no fluid particles");
    long long C = (long long)g[0] * g[1] * g[2];
    if (C <= 0 || C > 0x7ffffff0LL) return fail(h, SPH_E_INVALID, "grid too large");

    memset(&c, 0, sizeof(c));
    c.h = (float)support;
    c.hcell = h->verlet ? (float)cell_edge : c.h;
    c.verlet = h->verlet ? 1 : 0;
    c.verlet_thr2 = (float)((0.5 * skin * support) * (0.5 * skin * support));
    c.m = (float)m;
    c.d = (float)d;
    c.rho0 = 1000.0f;
    c.gravity = (float)cf.gravity;
    const float pi_f = (float)3.141592653589793;
    const float h3 = c.h * (c.h * c.h);            // ti.pow(h, 3) by squaring
    c.kw = 8.0f / (pi_f * h3);                     // solver_base.py:79
    c.rh = 1.0f / c.h;
    c.rh_s = c.rh * 0x1p-32f; c.h_s = c.h * 0x1p32f;     // exact
    const float kg = 48.0f / (pi_f * h3);          // :95
    c.kg6 = kg * 6.0f;
    c.neg_kg6 = -kg * 6.0f;
    {   // the relaxed sweeps' constants (sph_relaxed_kernels.h), folded in f64
        const double kg6d = 48.0 / (3.141592653589793 * support * support * support) * 6.0;
        c.rx_k1a = (float)(3.0 * m * kg6d / (support * support));
        c.rx_k1b = (float)(-2.0 * m * kg6d / (support * support));
        c.rx_k2 = (float)(-m * kg6d / support);
        c.rx_rho0_m = (float)(1000.0 / m);
    }
    {   // r2_cut: largest f32 t with sqrtf(t) <= h, so that (sqrt(r2) > h) == (r2 > r2_cut) exactly
        float t = c.h * c.h;
        while (sqrtf(t) > c.h) t = nextafterf(t, 0.0f);
        while (sqrtf(nextafterf(t, INFINITY)) <= c.h) t = nextafterf(t, INFINITY);
        c.r2_cut = t;
        if (h->verlet) c.r2_cut = c.hcell * c.hcell;      // Verlet lists: every pair within h + skin
    }
    h->p.viscosity_c_s = cf.solver == SPH_SOLVER_WCSPH ? 10 : 13;    // wcsph_solver.py:18 vs solver_base.py:24
    h->p.tension_k = cf.solver == SPH_SOLVER_WCSPH ? 0.2 : 0.5;      // wcsph_solver.py:20 vs solver_base.py:26
    fold_params(h);
    c.neg_m = (float)(-m);
    c.dt_cfl_num = (float)(0.4 * r * 2);
    const float clamp_off = cf.solver == SPH_SOLVER_WCSPH ? c.d : (float)r;   // wcsph_solver.py:57 vs dfsph_solver.py:244, pcisph_solver.py:82, iisph_solver.py:201
    for (int a = 0; a < 3; ++a) {
        c.clamp_lo[a] = (float)cf.box_min[a] + clamp_off;
        c.clamp_hi[a] = (float)cf.box_max[a] - clamp_off;
    }
    c.gx = g[0]; c.gy = g[1]; c.gz = g[2]; c.C = (int)C;
    {
        // Storage order of the cells (cell_slot() in sph_kernels.h).  The Morton curve pays once the particle state no longer sits
        // in one XCD's L2 (measured, Mparticle-steps/s linear -> Morton: dfsph 1M 167 -> 203, 10M 162 -> 195, 250k 138 -> 148;
        // iisph 1M 37 -> 48; wcsph 1M 1348 -> 1423, 250k equal or 3% slower); scenes of tens of thousands of particles are launch-bound and
        // run 5-8% faster in the reference's own order.  SPH_CELL_ORDER=linear|morton forces one, SPH_CELL_TILE=4|8|16 the tile edge.
        const char *e = dev_env(&h->overrides, "SPH_CELL_ORDER"), *t = dev_env(&h->overrides, "SPH_CELL_TILE");
        const bool morton = e && !strcmp(e, "morton") ? true : e && !strcmp(e, "linear") ? false : h->N >= (cf.solver == SPH_SOLVER_WCSPH ? 1 << 19 : 1 << 17);
        c.order = morton ? CELL_ORDER_TILED : CELL_ORDER_LINEAR;
        const int edge = t ? atoi(t) : 4;
        c.tbits = edge >= 16 ? 4 : edge >= 8 ? 3 : 2;
        const int te = 1 << c.tbits;
        c.tnx = (c.gx + te - 1) / te;
        c.tnxz = c.tnx * ((c.gz + te - 1) / te);
        const long long slots = c.order == CELL_ORDER_TILED ? ((long long)c.tnxz * ((c.gy + te - 1) / te)) << (3 * c.tbits) : C;
        if (slots + 2 > 0x7fffffffLL) return fail(h, SPH_E_INVALID, "grid of %lld cell slots is too large", slots);
        c.S = (int)slots;
    }
    c.sy = g[0] * g[2]; c.sz = g[0];               // :102
    c.boundary_handle = cf.boundary_handle ? 1 : 0;
    c.strict_cells = cf.slab_count > 1 ? 1 : 0;
    c.n = h->N;                                    // refined below for slab handles
    c.gw_left = c.gw_right = -1; c.ghost_walk = 0;
    c.stride = (h->N + 63) / 64 * 64;
    // (Verlet lists hold (1 + skin)^3 as many pairs: default capacity 80 there)
    c.kmax = ((cf.max_neighbors > 0 ? cf.max_neighbors : (h->verlet ? 80 : 64)) + 3) & ~3;        // rows come in groups of four
    c.kbmax = ((cf.max_wall_neighbors > 0 ? cf.max_wall_neighbors : (h->verlet ? 80 : 64)) + 3) & ~3;
    if (cf.boundary_handle == 0) c.kbmax = 4;      // clamp walls: no wall particles, the wall lists stay empty (one row group, never walked)
    if (c.kmax > 0xffff || c.kbmax > 0x7fff) return fail(h, SPH_E_INVALID, "neighbour capacity too large");
    {
        // A tile's rows are 1 KiB each, so with kmax = 64 every tile starts 16 KiB after the previous one and, because all waves
        // walk their lists at about the same pace, the rows in flight at any moment agree in address bits 10-13: the HBM channel
        // hash then sees a fraction of its inputs and the read latency of a sweep depends on where the allocator put the list
        // (measured: 833 vs 1090 cycles per request, sweeps 110 vs 145 us for identical handles).  An odd number of row groups
        // per tile walks the rows of consecutive tiles through all residues.
        const int pad = 4;
        // (every tile also keeps at least one spare group beyond kmax entries: the walks read one group ahead, NlWriter::flush)
        c.kpitch = c.kmax + (((c.kmax >> 2) & 1) ? 2 * pad : pad);
        c.kbpitch = c.kbmax + (((c.kbmax >> 2) & 1) ? 2 * pad : pad);
        if (c.kpitch < c.kmax + 4) c.kpitch = c.kmax + 4;
        if (c.kbpitch < c.kbmax + 4) c.kbpitch = c.kbmax + 4;
    }
    if ((long long)h->N >= (1LL << 28) || (long long)h->Nb >= (1LL << 28))
        return fail(h, SPH_E_INVALID, "%d fluid / %d wall particles: one handle addresses its particle arrays with 32-bit byte offsets (< 2^28 particles); shard the scene over slabs", h->N, h->Nb);

    // ---- fluid lattice, init_particle_pos :142-151 (f32 index arithmetic, constants f64-folded) ----
    const int N = h->N;
    sc.fluid_pos.resize(3 * (size_t)N);
    {
        const float x_num = (float)(cf.water_size[0] / d);
        const float z_num = (float)(cf.water_size[2] / d);
        const float xz_num = (float)((cf.water_size[0] / d) * (cf.water_size[2] / d));
        const float radius = (float)r;
        const float sp[3] = {(float)cf.start_pos[0], (float)cf.start_pos[1], (float)cf.start_pos[2]};
        // The reference forms the lattice coordinates from the particle index in f32, which is exact only below 2^24 particles: beyond
        // that its own initial condition degenerates (indices collide).  From 2^24 on the same expressions are evaluated in f64 -- the
        // continuation the formulas intend; below 2^24 the f32 path is kept bit for bit (SPH_LATTICE_F64=1 forces f64 everywhere: a test
        // checks that both agree there).
        const char *force64 = dev_env(&h->overrides, "SPH_LATTICE_F64");
        const int f32_limit = (force64 && force64[0] == '1') ? 0 : (1 << 24);
        for (int i = 0; i < N; ++i) {
            float x, z; int y;
            if (i < f32_limit) {
                float fi = (float)i;
                x = fmod_py(fi, x_num);
                z = fmod_py(floorf(fi / x_num), z_num);
                y = (int)(fi / xz_num);
            } else {
                const double di = (double)i, xn = (double)x_num, zn = (double)z_num;
                const double row = floor(di / xn);
                x = (float)(di - xn * floor(di / xn));
                z = (float)(row - zn * floor(row / zn));
                y = (int)(di / (double)xz_num);
            }
            sc.fluid_pos[3 * (size_t)i + 0] = x * radius * 2.0f + sp[0];
            sc.fluid_pos[3 * (size_t)i + 1] = (float)y * radius * 2.0f + sp[1];
            sc.fluid_pos[3 * (size_t)i + 2] = z * radius * 2.0f + sp[2];
        }
    }
    // ---- ownership: everything on one GPU, or the particles of this rank's x-slab ----
    h->slab = cf.slab_count > 1;
    h->init_ids.resize((size_t)N);
    for (int i = 0; i < N; ++i) h->init_ids[i] = i;
    h->n_owned = N;
    h->ncap = N;
    if (h->slab) {
        h->slab_rank = cf.slab_rank; h->nslab = cf.slab_count;
        if (h->slab_rank < 0 || h->slab_rank >= h->nslab) return fail(h, SPH_E_INVALID, "slab_rank %d out of range [0,%d)", h->slab_rank, h->nslab);
        std::vector<int> col, cut;
        std::string why;
        if (!plan_slab_cuts(sc.fluid_pos, N, c.h, c.gx, h->nslab, col, cut, why, slab_layers_of(cf))) return fail(h, SPH_E_INVALID, "%s", why.c_str());
        h->cuts = cut;
        h->rebalance_every = cf.slab_rebalance_every > 0 ? cf.slab_rebalance_every : 0;
        h->geom.has_left = h->slab_rank > 0; h->geom.has_right = h->slab_rank < h->nslab - 1;
        // two ghost columns per side by default for dfsph (one halo refresh per solver iteration, see step_dfsph_device_loops); the
        // other solvers keep the one-column protocol
        h->geom.layers = slab_layers_of(cf);
        if (h->geom.layers == 2 && cf.solver != SPH_SOLVER_DFSPH) return fail(h, SPH_E_INVALID, "slab_ghost_layers = 2 is the dfsph protocol");
        c.ghost_walk = h->geom.layers == 2 ? 1 : 0;
        set_slab_geometry(h);
        if (cf.solver == SPH_SOLVER_PCISPH) h->pci_fluid_pos = sc.fluid_pos;   // pre_compute looks at the whole lattice on every slab
        std::vector<float> own_pos; std::vector<int> own_id;
        for (int i = 0; i < N; ++i)
            if (col[i] >= h->geom.x_lo && col[i] < h->geom.x_hi) {
                own_id.push_back(i);
                own_pos.push_back(sc.fluid_pos[3 * (size_t)i]); own_pos.push_back(sc.fluid_pos[3 * (size_t)i + 1]); own_pos.push_back(sc.fluid_pos[3 * (size_t)i + 2]);
            }
        sc.fluid_pos.swap(own_pos);
        h->init_ids.swap(own_id);
        h->n_owned = (int)h->init_ids.size();
        long long cap = cf.slab_capacity > 0 ? cf.slab_capacity : (long long)((h->geom.layers == 2 ? 2.0 : 1.75) * N / h->nslab) + 262144;
        if (cap < h->n_owned) cap = h->n_owned;
        h->ncap = (int)std::min<long long>(cap, 0x7fffff00LL);
        c.n = h->n_owned;
        c.stride = (h->ncap + 63) / 64 * 64;
    }
    // ---- wall particles, init_particle_pos :155-195 (kernel-local f32) ----
    const int Nb = h->Nb;
    sc.wall_pos.assign(3 * (size_t)(Nb > 0 ? Nb : 1), 0.f);
    sc.wall_vol.assign((size_t)(Nb > 0 ? Nb : 1), 0.f);
    {
        const float dd = c.d;
        const float boxx = (float)cf.box_max[0] - (float)cf.box_min[0];
        const float boxz = (float)cf.box_max[2] - (float)cf.box_min[2];
        const int x_cnt = (int)(boxx / dd + 1.0f), z_cnt = (int)(boxz / dd + 1.0f);
        const int xr = x_cnt - 1, zr = z_cnt - 1;
        const int bottom = x_cnt * z_cnt;
        const int ring = x_cnt * z_cnt - (x_cnt - 2) * (z_cnt - 2);
        if (Nb > 0 && (xr <= 0 || zr <= 0 || ring <= 0)) return fail(h, SPH_E_INVALID, "box too small for wall particles");
        for (int i = 0; i < Nb; ++i) {
            float x = 0.f, y = 0.f, z = 0.f;
            if (i < bottom) {
                x = (float)(i % x_cnt) * dd;
                z = floorf((float)i / (float)x_cnt) * dd;
            } else if (i < Nb - bottom) {
                int index = i - bottom;
                int layer = (int)floorf((float)index / (float)ring);
                y = dd * (float)(layer + 1);
                index -= layer * ring;
                index += 1;
                if (index <= xr) { x = (float)(index % xr) * dd; z = 0.f; }
                else if (index <= xr + zr) { x = (float)xr * dd; z = (float)((index - x_cnt) % zr) * dd; }
                else if (index <= 2 * xr + zr) { x = (float)((2 * xr + zr - index) % xr + 1) * dd; z = (float)zr * dd; }
                else if (index <= 2 * (xr + zr)) { x = 0.f; z = (float)((2 * (xr + zr) - index) % zr + 1) * dd; }
            } else {
                int index = i - (Nb - bottom);
                x = (float)(index % x_cnt) * dd;
                y = (float)cf.box_max[1];
                z = (float)((int)((float)index / (float)x_cnt)) * dd;
            }
            sc.wall_pos[3 * (size_t)i] = x; sc.wall_pos[3 * (size_t)i + 1] = y; sc.wall_pos[3 * (size_t)i + 2] = z;
        }
    }
    // ---- static wall cell list (reset/update_boundary_grids :322-335), canonical order ----
    std::vector<int> wcell(Nb > 0 ? Nb : 1), wc3(3 * (size_t)(Nb > 0 ? Nb : 1));
    sc.wcell_start.assign((size_t)c.C + 1, 0);
    for (int i = 0; i < Nb; ++i) {
        int cx = (int)floorf(sc.wall_pos[3 * (size_t)i] / c.hcell);
        int cy = (int)floorf(sc.wall_pos[3 * (size_t)i + 1] / c.hcell);
        int cz = (int)floorf(sc.wall_pos[3 * (size_t)i + 2] / c.hcell);
        int id = cx + cy * c.sy + cz * c.sz;
        if (id < 0 || id >= c.C) return fail(h, SPH_E_INVALID, "wall particle %d falls outside the grid", i);
        wcell[i] = id; wc3[3 * (size_t)i] = cx; wc3[3 * (size_t)i + 1] = cy; wc3[3 * (size_t)i + 2] = cz;
        sc.wcell_start[(size_t)id + 1]++;
    }
    for (int k = 0; k < c.C; ++k) sc.wcell_start[(size_t)k + 1] += sc.wcell_start[k];
    std::vector<int> fill(sc.wcell_start.begin(), sc.wcell_start.end() - 1), order(Nb > 0 ? Nb : 1);
    for (int i = 0; i < Nb; ++i) order[fill[wcell[i]]++] = i;
    // ---- wall volumes, compute_all_boundary_volume :309-320 ----
    for (int i = 0; i < Nb; ++i) {
        float volume = 0.f;
        const float pix = sc.wall_pos[3 * (size_t)i], piy = sc.wall_pos[3 * (size_t)i + 1], piz = sc.wall_pos[3 * (size_t)i + 2];
        for (int dx = -1; dx <= 1; ++dx)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dz = -1; dz <= 1; ++dz) {
                    int x = wc3[3 * (size_t)i] + dx, y = wc3[3 * (size_t)i + 1] + dy, z = wc3[3 * (size_t)i + 2] + dz;
                    if (x >= c.gx || y >= c.gy || z >= c.gz) continue;
                    if (x < 0 || y < 0 || z < 0) continue;
                    int cid = x + y * c.sy + z * c.sz;
                    for (int e = sc.wcell_start[cid]; e < sc.wcell_start[(size_t)cid + 1]; ++e) {
                        int j = order[e];
                        if (j == i) continue;
                        float ddx = pix - sc.wall_pos[3 * (size_t)j], ddy = piy - sc.wall_pos[3 * (size_t)j + 1], ddz = piz - sc.wall_pos[3 * (size_t)j + 2];
                        float q = sqrtf((ddx * ddx + ddy * ddy) + ddz * ddz);
                        if (q > c.h) continue;
                        volume += host_cubic_w(q, c.h, c.kw);
                    }
                }
        sc.wall_vol[i] = 1.0f / volume;                                 // :314
    }
    sc.wall_sorted.resize(Nb > 0 ? Nb : 1);
    for (int e = 0; e < Nb; ++e) {
        int j = order[e];
        sc.wall_sorted[e] = make_float4(sc.wall_pos[3 * (size_t)j], sc.wall_pos[3 * (size_t)j + 1], sc.wall_pos[3 * (size_t)j + 2], sc.wall_vol[j]);
    }
    return SPH_OK;
}

// kernel<T0, RIGID, MODE> / kernel<RIGID, MODE> chosen at run time (rigid coupling active; the sweep mode of the handle):
// sweeps with a MODE parameter (SWEEP_PLAIN / SWEEP_STAGED / SWEEP_QUAD, sph_kernels.h); the grid follows the mode (quad sweeps: 64 particles per workgroup)
#define SPH_LAUNCH_RM(K, T0, rg, mode, n, lds, s, ...)                                                                                   \
    do {                                                                                                                                 \
        const dim3 g_ = (mode) == SWEEP_QUAD ? dim3((unsigned)std::max(1, ((n) + 63) / 64)) : grid_for(n), b_(kBlock);                   \
        if ((rg) && (mode) == SWEEP_STAGED) hipLaunchKernelGGL((K<T0, true, SWEEP_STAGED>), g_, b_, lds, s, __VA_ARGS__);                \
        else if ((rg) && (mode) == SWEEP_QUAD) hipLaunchKernelGGL((K<T0, true, SWEEP_QUAD>), g_, b_, 0, s, __VA_ARGS__);                 \
        else if (rg) hipLaunchKernelGGL((K<T0, true, SWEEP_PLAIN>), g_, b_, lds, s, __VA_ARGS__);                                        \
        else if ((mode) == SWEEP_STAGED) hipLaunchKernelGGL((K<T0, false, SWEEP_STAGED>), g_, b_, lds, s, __VA_ARGS__);                  \
        else if ((mode) == SWEEP_QUAD) hipLaunchKernelGGL((K<T0, false, SWEEP_QUAD>), g_, b_, 0, s, __VA_ARGS__);                        \
        else hipLaunchKernelGGL((K<T0, false, SWEEP_PLAIN>), g_, b_, lds, s, __VA_ARGS__);                                               \
    } while (0)
#define SPH_LAUNCH_RM0(K, rg, mode, n, lds, s, ...)                                                                                      \
    do {                                                                                                                                 \
        const dim3 g_ = (mode) == SWEEP_QUAD ? dim3((unsigned)std::max(1, ((n) + 63) / 64)) : grid_for(n), b_(kBlock);                   \
        if ((rg) && (mode) == SWEEP_STAGED) hipLaunchKernelGGL((K<true, SWEEP_STAGED>), g_, b_, lds, s, __VA_ARGS__);                    \
        else if ((rg) && (mode) == SWEEP_QUAD) hipLaunchKernelGGL((K<true, SWEEP_QUAD>), g_, b_, 0, s, __VA_ARGS__);                     \
        else if (rg) hipLaunchKernelGGL((K<true, SWEEP_PLAIN>), g_, b_, lds, s, __VA_ARGS__);                                            \
        else if ((mode) == SWEEP_STAGED) hipLaunchKernelGGL((K<false, SWEEP_STAGED>), g_, b_, lds, s, __VA_ARGS__);                      \
        else if ((mode) == SWEEP_QUAD) hipLaunchKernelGGL((K<false, SWEEP_QUAD>), g_, b_, 0, s, __VA_ARGS__);                            \
        else hipLaunchKernelGGL((K<false, SWEEP_PLAIN>), g_, b_, lds, s, __VA_ARGS__);                                                   \
    } while (0)
// the dfsph sweeps of UNSTAGED handles under the relaxed arithmetic (relaxed_unstaged): plain and quad sweeps with KF<true> (sph_device.h)
#define SPH_LAUNCH_RMX(K, T0, rg, mode, rx, n, lds, s, ...)                                                                              \
    do {                                                                                                                                 \
        if ((rx) && !(rg) && (mode) == SWEEP_QUAD)                                                                                       \
            hipLaunchKernelGGL((K<T0, false, SWEEP_QUAD, true>), dim3((unsigned)std::max(1, ((n) + 63) / 64)), dim3(kBlock), 0, s, __VA_ARGS__); \
        else if ((rx) && !(rg) && (mode) == SWEEP_PLAIN) hipLaunchKernelGGL((K<T0, false, SWEEP_PLAIN, true>), grid_for(n), dim3(kBlock), lds, s, __VA_ARGS__); \
        else SPH_LAUNCH_RM(K, T0, rg, mode, n, lds, s, __VA_ARGS__);                                                                      \
    } while (0)
#define SPH_LAUNCH_RMXQ0(K, rg, mode, rx, n, lds, s, ...)                                                                                \
    do {                                                                                                                                 \
        if ((rx) && !(rg) && (mode) == SWEEP_QUAD)                                                                                       \
            hipLaunchKernelGGL((K<false, SWEEP_QUAD, true>), dim3((unsigned)std::max(1, ((n) + 63) / 64)), dim3(kBlock), 0, s, __VA_ARGS__); \
        else if ((rx) && !(rg) && (mode) == SWEEP_PLAIN) hipLaunchKernelGGL((K<false, SWEEP_PLAIN, true>), grid_for(n), dim3(kBlock), lds, s, __VA_ARGS__); \
        else SPH_LAUNCH_RM0(K, rg, mode, n, lds, s, __VA_ARGS__);                                                                         \
    } while (0)
// the pcisph / iisph sweeps: the same with the kernel functions of the relaxed arithmetic (KF<true>, sph_device.h) where the handle asks for it --
// plain and staged sweeps without a coupled body
#define SPH_LAUNCH_RMX0(K, rg, mode, rx, n, lds, s, ...)                                                                                 \
    do {                                                                                                                                 \
        if ((rx) && !(rg) && (mode) == SWEEP_STAGED) hipLaunchKernelGGL((K<false, SWEEP_STAGED, true>), grid_for(n), dim3(kBlock), lds, s, __VA_ARGS__); \
        else if ((rx) && !(rg) && (mode) == SWEEP_PLAIN) hipLaunchKernelGGL((K<false, SWEEP_PLAIN, true>), grid_for(n), dim3(kBlock), lds, s, __VA_ARGS__); \
        else SPH_LAUNCH_RM0(K, rg, mode, n, lds, s, __VA_ARGS__);                                                                         \
    } while (0)
constexpr int kBnlSplit9Below = 65536, kBnlSplitBelow = 100000;   // k_build_nl_split with nine / three waves per 64 particles up to these sizes (unstaged handles)
// dynamic LDS of a staged sweep: bytes per staged particle x capacity (else the occupancy-experiment knob)
inline int sweep_mode(const SphHandle *h)
{
    if (h->staged) return SWEEP_STAGED;
    return (!h->slab && h->opt_quad && h->c.n <= h->quad_below) ? SWEEP_QUAD : SWEEP_PLAIN;
}
// partials of the block reductions: one per 256 particles, or one per 64 from quad sweeps (k_finalize_mean adds them in groups of four)
inline int partial_group(const SphHandle *h) { return sweep_mode(h) == SWEEP_QUAD ? 4 : 1; }
inline int partial_count(const SphHandle *h) { return sweep_mode(h) == SWEEP_QUAD ? (h->c.n + 63) / 64 : h->nblocks; }
inline size_t sweep_lds(const SphHandle *h, size_t bytes_per_staged) { return h->staged ? (size_t)h->c.stage_cap * bytes_per_staged : (size_t)h->sweep_lds; }
inline RigidView rigid_view_or_none(const SphHandle *h);

inline bool is_dfsph(const SphHandle *h) { return h->cfg.solver == SPH_SOLVER_DFSPH; }
// (grad W_ib, V_b) of every wall-list entry, written by D1 and read by D2-D7 of the same step (for_wall_cache); nullptr: the sweeps walk the wall lists
inline float4 *wall_cache(const SphHandle *h) { return h->wall_gc; }
inline bool is_pressure_solver(const SphHandle *h) { return h->cfg.solver == SPH_SOLVER_PCISPH || h->cfg.solver == SPH_SOLVER_IISPH; }
// solvers with a per-particle scalar that must follow the particle through the sort: dfsph warm_start_k, iisph p_past
inline bool carries_scalar(const SphHandle *h) { return h->cfg.solver == SPH_SOLVER_DFSPH || h->cfg.solver == SPH_SOLVER_IISPH; }

// Device memory of a handle comes from ONE allocation per build phase (fluid state, rigid body): dalloc() records a request,
// dcommit() sizes the arena, allocates and zeroes it and hands out the pointers.  Identical handles have identical layouts,
// arrays of 2 MiB and more start on a 2 MiB boundary, and closing a handle is one hipFree per phase.
template <class T>
int dalloc(SphHandle *h, T **p, size_t count)
{
    *p = nullptr;
    h->plan.push_back({(void **)p, sizeof(T) * (count > 0 ? count : 1)});
    return SPH_OK;
}

int dcommit(SphHandle *h)
{
    const size_t big = (size_t)2 << 20;
    std::vector<size_t> off(h->plan.size());
    size_t cur = 0;
    for (size_t k = 0; k < h->plan.size(); ++k) {
        const size_t bytes = h->plan[k].second, align = bytes >= big ? big : 256;
        cur = (cur + align - 1) / align * align;
        off[k] = cur;
        cur += bytes;
    }
    char *base = nullptr;
    HIP_TRY(h, hipMalloc((void **)&base, cur > 0 ? cur : 1));
    h->arenas.push_back(base);
    for (size_t k = 0; k < h->plan.size(); ++k) *h->plan[k].first = base + off[k];
    HIP_TRY(h, hipMemsetAsync(base, 0, cur, h->stream));
    h->plan.clear();
    return SPH_OK;
}

inline uint64_t morton_spread(uint64_t v)          // 21 bits -> every third bit
{
    v &= 0x1fffffull;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

// Consts.tile_rank: position of every tile (index tx + tz*tnx + ty*tnxz) along the Morton curve of (tx, ty, tz)
std::vector<int> morton_tile_ranks(const Consts &c)
{
    auto spread = morton_spread;
    const int te = 1 << c.tbits, tnx = c.tnx, tnz = c.tnxz / c.tnx, tny = (c.gy + te - 1) / te;
    std::vector<std::pair<uint64_t, int>> key;
    key.reserve((size_t)tnx * tnz * tny);
    for (int ty = 0; ty < tny; ++ty)
        for (int tz = 0; tz < tnz; ++tz)
            for (int tx = 0; tx < tnx; ++tx)
                key.push_back({spread((uint64_t)tx) | spread((uint64_t)ty) << 1 | spread((uint64_t)tz) << 2, tx + tz * c.tnx + ty * c.tnxz});
    std::sort(key.begin(), key.end());
    std::vector<int> rank(key.size());
    for (size_t r = 0; r < key.size(); ++r) rank[(size_t)key[r].second] = (int)r;
    return rank;
}

int alloc_device(SphHandle *h, const HostScene &sc)
{
    const Consts &c = h->c;
    const size_t n = (size_t)c.stride;
    int rc;
    std::vector<int> tile_rank;
    if (c.order == CELL_ORDER_TILED) {
        tile_rank = morton_tile_ranks(c);
        if ((rc = dalloc(h, &h->tile_rank, tile_rank.size()))) return rc;
    }
    for (int k = 0; k < 2; ++k) {
        if ((rc = dalloc(h, &h->P[k], n + 64))) return rc;      // k_build_nl reads whole groups of four candidates
        if ((rc = dalloc(h, &h->V[k], n))) return rc;
        if ((rc = dalloc(h, &h->VA[k], n))) return rc;
        if ((rc = dalloc(h, &h->warm[k], n))) return rc;
        if ((rc = dalloc(h, &h->id[k], n))) return rc;
    }
    if (is_pressure_solver(h) || h->cfg.solver == SPH_SOLVER_PBF)      // pbf: delta_pos, new position, phase-1 velocity
        for (int k = 0; k < (is_pressure_solver(h) ? 5 : 3); ++k) {
            if ((rc = dalloc(h, &h->X[k], n))) return rc;
        }
    if ((rc = dalloc(h, &h->rho, n))) return rc;
    if ((rc = dalloc(h, &h->aux, n))) return rc;
    if ((rc = dalloc(h, &h->drho, n))) return rc;
    if ((rc = dalloc(h, &h->rho_adv, n))) return rc;
    if ((rc = dalloc(h, &h->krho, n))) return rc;
    if ((rc = dalloc(h, &h->cnt, n))) return rc;
    // one spare 64-particle tile at the end: the software-pipelined walks read one row ahead
    if ((rc = dalloc(h, &h->nl, (n + 64) * (size_t)c.kpitch))) return rc;
    if ((rc = dalloc(h, &h->nlb, (n + 64) * (size_t)c.kbpitch))) return rc;
    // the wall terms of the solver loops from a per-step cache: 16 B per wall-list row (1 GiB per million particles at 64 rows, allocated like the list
    // itself; only the rows of particles next to a wall are ever touched).  Not for quad sweeps (small scenes), not where the relaxed sweeps run.
    const bool want_wall_cache = h->cfg.solver == SPH_SOLVER_DFSPH && c.boundary_handle && h->Nb > 0 && h->opt_wall_cache;
    {
        // LDS staging of the gather operands (plan in k_build_nl): DFSPH, PCISPH and IISPH on the Morton curve; SPH_STAGE=0 turns it off, SPH_STAGE_CAP sets the capacity
        const char *e = dev_env(&h->overrides, "SPH_STAGE"), *cap = dev_env(&h->overrides, "SPH_STAGE_CAP");
        h->staged = c.order == CELL_ORDER_TILED && h->cfg.solver != SPH_SOLVER_WCSPH && h->cfg.solver != SPH_SOLVER_PBF && !(e && atoi(e) == 0);
        h->c.stage_cap = h->staged ? std::min(std::max(cap ? atoi(cap) : 1664, 64), 2560) : 0;
        if (h->staged) {
            if ((rc = dalloc(h, &h->stage_src, (n + kBlock - 1) / kBlock * (size_t)kStageMaxCells))) return rc;
            if ((rc = dalloc(h, &h->stage_cnt, (n + kBlock - 1) / kBlock))) return rc;
            if (h->cfg.solver == SPH_SOLVER_DFSPH && h->opt_tile_skip) {
                if ((rc = dalloc(h, &h->dens_hot, (n + kBlock - 1) / kBlock + 1))) return rc;
                if ((rc = dalloc(h, &h->dens_order, (n + kBlock - 1) / kBlock + 2))) return rc;
            }
            if (h->cfg.solver == SPH_SOLVER_DFSPH && h->opt_tile_skip) {
                if ((rc = dalloc(h, &h->wave_dirty, (n + kBlock - 1) / kBlock * (size_t)(kBlock / 64) + 64))) return rc;
                if ((rc = dalloc(h, &h->changed8, n + 256))) return rc;
            }
            if ((h->cfg.solver == SPH_SOLVER_PCISPH || h->cfg.solver == SPH_SOLVER_IISPH) && !h->slab && h->opt_tile_skip)
                if ((rc = dalloc(h, &h->pci_zero_press, (n + kBlock - 1) / kBlock + 64))) return rc;
            if (h->relaxed && h->cfg.solver == SPH_SOLVER_DFSPH)      // the relaxed sweeps' per-step wall sums (use_relaxed)
                if ((rc = dalloc(h, &h->wall_grad, n)) || (rc = dalloc(h, &h->wall_gsq, n))) return rc;
        }
        // (up to 64 GiB of it, ~58 M particles at 64 rows: beyond that the sweeps walk the wall lists and the memory goes to the scene)
        // ... and never more than half of what is free on the device right now: the cache is an optimisation, the scene is not
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)1 << 62;
        const size_t gc_bytes = (n + 64) * (size_t)c.kbpitch * sizeof(float4);
        if (want_wall_cache && sweep_mode(h) != SWEEP_QUAD && !h->wall_grad && gc_bytes <= ((size_t)64 << 30) && gc_bytes <= free_b / 2)
            if ((rc = dalloc(h, &h->wall_gc, (n + 64) * (size_t)c.kbpitch))) return rc;
    }
    if (h->verlet) {      // the wall sums of the step (density -> force kernel) and the positions of the last list build
        if ((rc = dalloc(h, &h->wall_grad, n))) return rc;
        if ((rc = dalloc(h, &h->x0, n))) return rc;
    }
    if ((rc = dalloc(h, &h->cell_of, n))) return rc;
    if ((rc = dalloc(h, &h->rank, n))) return rc;
    if ((rc = dalloc(h, &h->slot_src, n))) return rc;
    const size_t ncell = (size_t)c.S + 2;
    h->ntiles = (int)((ncell + kScanTile - 1) / kScanTile);
    if ((rc = dalloc(h, &h->cell_count, ncell))) return rc;
    if ((rc = dalloc(h, &h->cell_start, ncell))) return rc;
    if ((rc = dalloc(h, &h->tile_sums, (size_t)h->ntiles))) return rc;
    if ((rc = dalloc(h, &h->WP, (size_t)h->Nb + 64))) return rc;
    if ((rc = dalloc(h, &h->wcell_start, (size_t)c.C + 1))) return rc;
    h->nblocks = (c.n + kBlock - 1) / kBlock;
    const size_t nblocks_cap = (n + 63) / 64;    // quad sweeps: one partial per 64 particles; others one per 256 (a few KB either way, and no second predicate to keep in step with sweep_mode)
    if ((rc = dalloc(h, &h->psum, nblocks_cap))) return rc;
    if ((rc = dalloc(h, &h->pcnt, nblocks_cap))) return rc;
    if ((rc = dalloc(h, &h->pmax, nblocks_cap))) return rc;
    if (h->slab) {
        if ((rc = dalloc(h, &h->dead, n))) return rc;
        for (int k = 0; k < 8; ++k)
            if ((rc = dalloc(h, &h->edge_off[k], (size_t)c.gy * c.gz + 1))) return rc;
        for (int k = 0; k < 4; ++k)
            if ((rc = dalloc(h, &h->edge_list[k], n))) return rc;
        if ((rc = dalloc(h, &h->counters, kSlabCounters))) return rc;
        if ((rc = dalloc(h, &h->class_cnt, (size_t)kSlabCounted * (n / kBlock + 2)))) return rc;
        HIP_TRY(h, hipHostMalloc((void **)&h->counters_host, sizeof(int) * kSlabCounters, hipHostMallocDefault));
        // edge / interior split of the residual sweeps (dfsph, two ghost columns): tile flags and the edge-first tile order
        h->overlap = h->geom.layers == 2 && h->cfg.slab_overlap != 1;
        if (h->overlap) {
            if ((rc = dalloc(h, &h->tile_flag, (n + kBlock - 1) / kBlock + 1))) return rc;
            if ((rc = dalloc(h, &h->tile_order, (n + kBlock - 1) / kBlock + 2))) return rc;
            HIP_TRY(h, hipStreamCreateWithFlags(&h->xstream, hipStreamNonBlocking));
            HIP_TRY(h, hipEventCreateWithFlags(&h->ev_edge, hipEventDisableTiming));
            HIP_TRY(h, hipEventCreateWithFlags(&h->ev_halo, hipEventDisableTiming));
            if ((rc = dalloc(h, &h->spec_v, n))) return rc;
            if ((rc = dalloc(h, &h->spec_w, n))) return rc;
            HIP_TRY(h, hipStreamCreateWithFlags(&h->rstream, hipStreamNonBlocking));
            HIP_TRY(h, hipEventCreateWithFlags(&h->ev_red, hipEventDisableTiming));
            HIP_TRY(h, hipEventCreateWithFlags(&h->ev_dec, hipEventDisableTiming));
        }
        if ((rc = dalloc(h, &h->col_hist, (size_t)c.gx))) return rc;
        HIP_TRY(h, hipHostMalloc((void **)&h->col_hist_host, sizeof(int) * (size_t)c.gx, hipHostMallocDefault));
    }
    // one GPU, dfsph: the divergence correction runs ahead of its loop decision, which rides in the same launch (fin_ride_block): what it overwrites
    if (!h->slab && is_dfsph(h)) {
        if ((rc = dalloc(h, &h->spec_v, n))) return rc;
        if ((rc = dalloc(h, &h->spec_w, n))) return rc;
    }
    if ((rc = dalloc(h, &h->ds, 1))) return rc;
    HIP_TRY(h, hipHostMalloc((void **)&h->ds_host, sizeof(DevScalars), hipHostMallocDefault));
    if (hipHostMalloc((void **)&h->pub_host, sizeof(DevScalarsPub), hipHostMallocMapped) == hipSuccess) {
        memset(h->pub_host, 0, sizeof(DevScalarsPub));
        if (hipHostGetDevicePointer((void **)&h->pub_dev, h->pub_host, 0) != hipSuccess) { (void)hipHostFree(h->pub_host); h->pub_host = nullptr; h->pub_dev = nullptr; }
    } else {
        (void)hipGetLastError();
        h->pub_host = nullptr;
    }
    size_t stg = 3 * std::max(n, (size_t)h->Nb);
    if ((rc = dalloc(h, &h->staging, stg))) return rc;

    if ((rc = dcommit(h))) return rc;
    if (c.order == CELL_ORDER_TILED) {
        HIP_TRY(h, hipMemcpyAsync(h->tile_rank, tile_rank.data(), sizeof(int) * tile_rank.size(), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        h->c.tile_rank = h->tile_rank;
    }

    // upload the scene
    std::vector<float4> p4((size_t)h->n_owned);
    for (int i = 0; i < h->n_owned; ++i)
        p4[i] = make_float4(sc.fluid_pos[3 * (size_t)i], sc.fluid_pos[3 * (size_t)i + 1], sc.fluid_pos[3 * (size_t)i + 2], 0.f);
    if (h->n_owned > 0) {
        HIP_TRY(h, hipMemcpyAsync(h->P[0], p4.data(), sizeof(float4) * p4.size(), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->id[0], h->init_ids.data(), sizeof(int) * h->init_ids.size(), hipMemcpyHostToDevice, h->stream));
    }
    if (h->Nb > 0)
        HIP_TRY(h, hipMemcpyAsync(h->WP, sc.wall_sorted.data(), sizeof(float4) * (size_t)h->Nb, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->wcell_start, sc.wcell_start.data(), sizeof(int) * ((size_t)c.C + 1), hipMemcpyHostToDevice, h->stream));
    memset(h->ds_host, 0, sizeof(DevScalars));
    h->ds_host->dt = (float)h->cfg.delta_time;                       // solver_base.py:16
    h->ds_host->dt2 = h->ds_host->dt * h->ds_host->dt;               // dfsph_solver.py:20
    h->ds_host->ps_dt = 0.f;                                         // ParticleSystem.py:37
    h->ds_host->moved = 1;                                           // Verlet handles: the first step builds the lists
    loop_params(h, h->ds_host);
    HIP_TRY(h, hipMemcpyAsync(h->ds, h->ds_host, sizeof(DevScalars), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->dt_wcsph = (float)h->cfg.delta_time;
    return SPH_OK;
}

// the sharded per-build maxima of the list lengths (note_list_lengths), folded into ds_host after a read-back
inline void fold_list_maxima(SphHandle *h)
{
    for (int k = 0; k < kNoteShards; ++k) {
        h->ds_host->max_nbrs = std::max(h->ds_host->max_nbrs, h->ds_host->nbr_shard[k]);
        h->ds_host->max_wall_nbrs = std::max(h->ds_host->max_wall_nbrs, h->ds_host->wall_shard[k]);
    }
}
int read_scalars(SphHandle *h);
// read_scalars for the read-back a solver loop waits on: the device writes the block to mapped host memory itself and the host spins on its
// sequence number (k_publish_scalars) -- no copy command, no interrupt.  Falls back to the copy if the block has not arrived after 2 ms.
int read_scalars_fast(SphHandle *h)
{
    if (!h->pub_dev) return read_scalars(h);
    const unsigned long long seq = ++h->pub_seq;
    hipLaunchKernelGGL(k_publish_scalars, dim3(1), dim3(kBlock), 0, h->stream, h->ds, h->pub_dev, seq);
    volatile unsigned long long *flag = &h->pub_host->seq;
    const auto t0 = std::chrono::steady_clock::now();
    long spins = 0;
    while (*flag != seq) {
        if ((++spins & 0x3ff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
            HIP_TRY(h, hipStreamSynchronize(h->stream));          // (a long chunk, or a launch that failed: the stream's status tells)
            if (*flag != seq) return read_scalars(h);
            break;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    memcpy(h->ds_host, &h->pub_host->ds, sizeof(DevScalars));
    fold_list_maxima(h);
    return SPH_OK;
}
int read_scalars(SphHandle *h)
{
    HIP_TRY(h, hipMemcpyAsync(h->ds_host, h->ds, sizeof(DevScalars), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    fold_list_maxima(h);
    return SPH_OK;
}

// ---------------------------------------------------------------------------------------------
// multi-GPU slab transport.  The library packs/unpacks on the device; the caller's
// callbacks move the bytes (RCCL send/recv over xGMI in production, gloo in the tests).
// ---------------------------------------------------------------------------------------------
int comm_fail(SphHandle *h, const char *what, int rc) { return fail(h, SPH_E_STATE, "comm callback %s failed (%d)", what, rc); }
inline bool slab_stream_ordered(const SphHandle *h) { return h->slab && (h->native || (!h->comm.on_host && h->comm.stream_ordered)); }
// sharded DFSPH with the device-side loop control of the single-GPU path (needs the transport's in-place all-reduce of reduce_buf)
inline bool slab_async(const SphHandle *h) { return h->slab && (h->native || h->comm.allreduce_stream) && h->red_dev; }

int native_allreduce_stream(SphHandle *h, int n, int op, hipStream_t stream = nullptr);

// all-reduce red_dev[0..n) over the slabs, ordered on `stream` (default: the handle's stream; a stream-ordered CALLBACK transport always uses the handle's)
int slab_allreduce_stream(SphHandle *h, int n, int op, hipStream_t stream = nullptr)
{
    if (!stream) stream = h->stream;
    h->comm_stat[4] += 1;
    if (h->native) return native_allreduce_stream(h, n, op, stream);
    const SphComm &cm = h->comm;
    if (cm.on_host) {                                  // host transport: stage through the caller's host buffer
        HIP_TRY(h, hipMemcpyAsync(cm.reduce_buf, h->red_dev, sizeof(double) * n, hipMemcpyDeviceToHost, stream));
        HIP_TRY(h, hipStreamSynchronize(stream));
    }
    // synchronous discipline on device buffers: the transport works on its own stream, so the pair must be complete before it reads
    // (it returns only when the reduced values are in place)
    if (!cm.on_host && !cm.stream_ordered) HIP_TRY(h, hipStreamSynchronize(stream));
    int rc = cm.allreduce_stream(cm.user, n, op);
    if (rc) return comm_fail(h, "allreduce_stream", rc);
    if (cm.on_host) HIP_TRY(h, hipMemcpyAsync(h->red_dev, cm.reduce_buf, sizeof(double) * n, hipMemcpyHostToDevice, stream));
    return SPH_OK;
}

// ---------------------------------------------------------------------------------------------
// native RCCL transport: ncclSend / ncclRecv to the left and right slab neighbour (one direct xGMI link per pair) and
// ncclAllReduce of the residual pair, issued by the library on its own stream -- no Python, no host waits.  librccl is
// dlopen'ed so that the library itself has no link-time dependency on it.
// ---------------------------------------------------------------------------------------------
struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string why;
    bool ok = false;
};

RcclApi &rccl()
{
    static RcclApi api = [] {
        RcclApi a;
        // development override (SPH_DEV=1): another library with librccl's entry points -- tests/loopback_rccl.hip drives this transport
        // with several handles of ONE process on one GPU.  sph_rccl_attach records it in the handle's overrides.
        if (const char *dev = dev_env(nullptr, "SPH_RCCL_LIB")) {
            a.lib = dlopen(dev, RTLD_NOW | RTLD_LOCAL);
            if (!a.lib) { a.why = std::string("SPH_RCCL_LIB: ") + dlerror(); return a; }
        }
        for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            if (a.lib) break;
            a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!a.lib) { a.why = "librccl.so not found"; return a; }
#define SPH_RCCL_SYM(field, sym) a.field = (decltype(a.field))dlsym(a.lib, sym); if (!a.field) { a.why = std::string("missing symbol ") + sym; return a; }
        SPH_RCCL_SYM(GetUniqueId, "ncclGetUniqueId") SPH_RCCL_SYM(CommInitRank, "ncclCommInitRank") SPH_RCCL_SYM(CommDestroy, "ncclCommDestroy")
        SPH_RCCL_SYM(GroupStart, "ncclGroupStart") SPH_RCCL_SYM(GroupEnd, "ncclGroupEnd") SPH_RCCL_SYM(Send, "ncclSend") SPH_RCCL_SYM(Recv, "ncclRecv")
        SPH_RCCL_SYM(AllReduce, "ncclAllReduce") SPH_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef SPH_RCCL_SYM
        a.ok = true;
        return a;
    }();
    return api;
}

#define NCCL_TRY(h, expr)                                                                                        \
    do {                                                                                                         \
        ncclResult_t r_ = (expr);                                                                                \
        if (r_ != ncclSuccess) return fail(h, SPH_E_HIP, "%s failed: %s", #expr, rccl().GetErrorString(r_));     \
    } while (0)

// exchange_buffers of the native transport: one group of up to four point-to-point transfers, ordered on the handle's stream
// gather_doubles > 0: the same group also carries this slab's gath_dev slot (that many doubles) to EVERY other slab and theirs back -- the residual's
// (sum, count, flags) travel with the halo, one start-up latency per solver iteration instead of the halo's plus an all-reduce's
int native_exchange(SphHandle *h, size_t sl, size_t sr, size_t rl, size_t rr, hipStream_t stream = nullptr, int gather_doubles = 0)
{
    RcclApi &n = rccl();
    if (!stream) stream = h->stream;
    const int left = h->slab_rank > 0 ? h->slab_rank - 1 : -1, right = h->slab_rank < h->nslab - 1 ? h->slab_rank + 1 : -1;
    if (!gather_doubles && !((left >= 0 && (sl || rl)) || (right >= 0 && (sr || rr)))) return SPH_OK;
    NCCL_TRY(h, n.GroupStart());
    for (int p = 0; gather_doubles && p < h->nslab; ++p) {
        if (p == h->slab_rank) continue;
        NCCL_TRY(h, n.Send(h->gath_dev + 4 * h->slab_rank, (size_t)gather_doubles, ncclDouble, p, h->nccl, stream));
        NCCL_TRY(h, n.Recv(h->gath_dev + 4 * p, (size_t)gather_doubles, ncclDouble, p, h->nccl, stream));
    }
    if (left >= 0) {
        if (sl) NCCL_TRY(h, n.Send(h->dsend[0], sl, ncclChar, left, h->nccl, stream));
        if (rl) NCCL_TRY(h, n.Recv(h->drecv[0], rl, ncclChar, left, h->nccl, stream));
    }
    if (right >= 0) {
        if (sr) NCCL_TRY(h, n.Send(h->dsend[1], sr, ncclChar, right, h->nccl, stream));
        if (rr) NCCL_TRY(h, n.Recv(h->drecv[1], rr, ncclChar, right, h->nccl, stream));
    }
    NCCL_TRY(h, n.GroupEnd());
    return SPH_OK;
}

// exchange_counts of the native transport: n ints each way with each neighbour, then the host reads what it received
constexpr int kCountInts = 8;
int native_exchange_counts_n(SphHandle *h, int n, const int32_t *sl, const int32_t *sr, int32_t *rl, int32_t *rr)
{
    RcclApi &api = rccl();
    const int left = h->slab_rank > 0 ? h->slab_rank - 1 : -1, right = h->slab_rank < h->nslab - 1 ? h->slab_rank + 1 : -1;
    memset(h->cnt_host, 0, sizeof(int) * 4 * kCountInts);
    for (int k = 0; k < n; ++k) { h->cnt_host[k] = sl[k]; h->cnt_host[kCountInts + k] = sr[k]; }
    HIP_TRY(h, hipMemcpyAsync(h->cnt_dev, h->cnt_host, sizeof(int) * 4 * kCountInts, hipMemcpyHostToDevice, h->stream));
    if (left >= 0 || right >= 0) {
        NCCL_TRY(h, api.GroupStart());
        if (left >= 0) {
            NCCL_TRY(h, api.Send(h->cnt_dev + 0, (size_t)n, ncclInt32, left, h->nccl, h->stream));
            NCCL_TRY(h, api.Recv(h->cnt_dev + 2 * kCountInts, (size_t)n, ncclInt32, left, h->nccl, h->stream));
        }
        if (right >= 0) {
            NCCL_TRY(h, api.Send(h->cnt_dev + kCountInts, (size_t)n, ncclInt32, right, h->nccl, h->stream));
            NCCL_TRY(h, api.Recv(h->cnt_dev + 3 * kCountInts, (size_t)n, ncclInt32, right, h->nccl, h->stream));
        }
        NCCL_TRY(h, api.GroupEnd());
    }
    HIP_TRY(h, hipMemcpyAsync(h->cnt_host, h->cnt_dev, sizeof(int) * 4 * kCountInts, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int k = 0; k < n; ++k) { rl[k] = h->cnt_host[2 * kCountInts + k]; rr[k] = h->cnt_host[3 * kCountInts + k]; }
    return SPH_OK;
}
int native_exchange_counts(SphHandle *h, int32_t sl, int32_t sr, int32_t *rl, int32_t *rr) { return native_exchange_counts_n(h, 1, &sl, &sr, rl, rr); }
// the same with the n ints per side already in cnt_dev[0..n) / cnt_dev[kCountInts..] (k_classify_scan): no upload; the classification's
// counters come back in the same read-back (counters_host)
int native_exchange_counts_dev(SphHandle *h, int n, int32_t *rl, int32_t *rr)
{
    RcclApi &api = rccl();
    const int left = h->slab_rank > 0 ? h->slab_rank - 1 : -1, right = h->slab_rank < h->nslab - 1 ? h->slab_rank + 1 : -1;
    h->comm_stat[3] += 1;
    HIP_TRY(h, hipMemsetAsync(h->cnt_dev + 2 * kCountInts, 0, sizeof(int) * 2 * kCountInts, h->stream));
    if (left >= 0 || right >= 0) {
        NCCL_TRY(h, api.GroupStart());
        if (left >= 0) {
            NCCL_TRY(h, api.Send(h->cnt_dev + 0, (size_t)n, ncclInt32, left, h->nccl, h->stream));
            NCCL_TRY(h, api.Recv(h->cnt_dev + 2 * kCountInts, (size_t)n, ncclInt32, left, h->nccl, h->stream));
        }
        if (right >= 0) {
            NCCL_TRY(h, api.Send(h->cnt_dev + kCountInts, (size_t)n, ncclInt32, right, h->nccl, h->stream));
            NCCL_TRY(h, api.Recv(h->cnt_dev + 3 * kCountInts, (size_t)n, ncclInt32, right, h->nccl, h->stream));
        }
        NCCL_TRY(h, api.GroupEnd());
    }
    HIP_TRY(h, hipMemcpyAsync(h->cnt_host, h->cnt_dev, sizeof(int) * 4 * kCountInts, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->counters_host, h->counters, sizeof(int) * kSlabCounters, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int k = 0; k < n; ++k) { rl[k] = h->cnt_host[2 * kCountInts + k]; rr[k] = h->cnt_host[3 * kCountInts + k]; }
    return SPH_OK;
}

int native_allreduce_stream(SphHandle *h, int n, int op, hipStream_t stream)
{
    NCCL_TRY(h, rccl().AllReduce(h->red_dev, h->red_dev, (size_t)n, ncclDouble, op == 0 ? ncclSum : ncclMax, h->nccl, stream ? stream : h->stream));
    return SPH_OK;
}

// neighbour counts / host-side all-reduce through whichever transport the handle has
// n ints to each neighbour, n from each (absent neighbour: zeros): one host round trip where the transport can (native RCCL, a SphComm with
// exchange_counts_n), n of them through a plain exchange_counts
// doubles the transport's reduce buffer must hold on this handle (a rigid body's by-id sums: 4 per sample)
inline size_t slab_reduce_need(const SphHandle *h) { return h->rigid ? 4 * (size_t)h->Nr + 8 : 4; }

int slab_exchange_counts_n(SphHandle *h, int n, const int32_t *sl, const int32_t *sr, int32_t *rl, int32_t *rr)
{
    if (n > kCountInts) return fail(h, SPH_E_INVALID, "count exchange of %d ints", n);
    for (int k = 0; k < n; ++k) rl[k] = rr[k] = 0;
    if (h->native) { h->comm_stat[3] += 1; return native_exchange_counts_n(h, n, sl, sr, rl, rr); }
    if (h->comm.exchange_counts_n) {
        h->comm_stat[3] += 1;
        int rc = h->comm.exchange_counts_n(h->comm.user, n, sl, sr, rl, rr);
        return rc ? comm_fail(h, "exchange_counts_n", rc) : SPH_OK;
    }
    for (int k = 0; k < n; ++k) {
        h->comm_stat[3] += 1;
        int rc = h->comm.exchange_counts(h->comm.user, sl[k], sr[k], &rl[k], &rr[k]);
        if (rc) return comm_fail(h, "exchange_counts", rc);
    }
    return SPH_OK;
}

int slab_allreduce_host(SphHandle *h, double *v, int n, int op)
{
    h->comm_stat[5] += 1;
    if (!h->native) {
        int rc = h->comm.allreduce(h->comm.user, v, n, op);
        return rc ? comm_fail(h, "allreduce", rc) : SPH_OK;
    }
    if (n > h->red_cap) return fail(h, SPH_E_INVALID, "all-reduce of %d doubles exceeds the reduce buffer (%d)", n, h->red_cap);
    memcpy(h->red_host, v, sizeof(double) * n);
    HIP_TRY(h, hipMemcpyAsync(h->red_dev, h->red_host, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
    int rc = native_allreduce_stream(h, n, op);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->red_host, h->red_dev, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    memcpy(v, h->red_host, sizeof(double) * n);
    return SPH_OK;
}

// `stream`: where the packed data was produced and the unpack will run (the handle's stream, or the halo stream of an overlapped refresh).
// A stream-ordered CALLBACK transport enqueues on the handle's own stream whatever we say, so overlapped refreshes are only taken with the
// native transport or a synchronous one (slab_can_overlap).
int slab_xfer(SphHandle *h, size_t sl, size_t sr, size_t rl, size_t rr, hipStream_t stream = nullptr, int gather_doubles = 0)
{
    if (!stream) stream = h->stream;
    const SphComm &cm = h->comm;
    if (sl > cm.capacity || sr > cm.capacity || rl > cm.capacity || rr > cm.capacity)
        return fail(h, SPH_E_OVERFLOW, "halo message of %zu bytes exceeds the comm buffer capacity %zu", std::max(std::max(sl, sr), std::max(rl, rr)), cm.capacity);
    if (cm.on_host) {
        if (sl) HIP_TRY(h, hipMemcpyAsync(cm.send_left, h->dsend[0], sl, hipMemcpyDeviceToHost, stream));
        if (sr) HIP_TRY(h, hipMemcpyAsync(cm.send_right, h->dsend[1], sr, hipMemcpyDeviceToHost, stream));
    }
    if (!slab_stream_ordered(h)) HIP_TRY(h, hipStreamSynchronize(stream));     // packed data complete before the transport reads it
    h->comm_stat[0] += 1; h->comm_stat[1] += (long long)(sl + sr); h->comm_stat[2] += (long long)(rl + rr);       // (counted even when this rank's share of the exchange is empty)
    int rc;
    if (h->native) {
        if ((rc = native_exchange(h, sl, sr, rl, rr, stream, gather_doubles))) return rc;
    } else {
        rc = cm.exchange_buffers(cm.user, sl, sr, rl, rr);      // stream-ordered transports enqueue behind the pack kernels instead
        if (rc) return comm_fail(h, "exchange_buffers", rc);
    }
    if (cm.on_host) {
        if (rl) HIP_TRY(h, hipMemcpyAsync(h->drecv[0], cm.recv_left, rl, hipMemcpyHostToDevice, stream));
        if (rr) HIP_TRY(h, hipMemcpyAsync(h->drecv[1], cm.recv_right, rr, hipMemcpyHostToDevice, stream));
    }
    return SPH_OK;
}
inline bool slab_can_overlap(const SphHandle *h) { return h->overlap && h->overlap_on && (h->native || !slab_stream_ordered(h)); }

int read_counters(SphHandle *h)
{
    HIP_TRY(h, hipMemcpyAsync(h->counters_host, h->counters, sizeof(int) * kSlabCounters, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SPH_OK;
}

// Every `slab_rebalance_every` steps: global per-column particle histogram (one all-reduce of gx counts), new
// equal-count cuts on every rank alike.  Only the cuts change here; the migration that follows moves the
// particles of the shifted columns to the neighbour that now owns them.  Results do not depend on the cuts
// (every sum runs in (cell, id) order), so re-balancing is invisible in the output.
int slab_rebalance(SphHandle *h)
{
    if (!h->comm_set) return fail(h, SPH_E_STATE, "slab handle needs sph_set_comm before stepping");
    Consts &c = h->c;
    hipStream_t s = h->stream;
    HIP_TRY(h, hipMemsetAsync(h->col_hist, 0, sizeof(int) * (size_t)c.gx, s));
    {
        ProfScope ps(h, K_SLAB);
        hipLaunchKernelGGL(k_column_histogram, grid_for(c.n), dim3(kBlock), 0, s, c, h->P[h->pcur], h->id[h->icur], h->col_hist);
    }
    HIP_TRY(h, hipMemcpyAsync(h->col_hist_host, h->col_hist, sizeof(int) * (size_t)c.gx, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    std::vector<double> v((size_t)c.gx);
    for (int x = 0; x < c.gx; ++x) v[x] = (double)h->col_hist_host[x];
    int rc = slab_allreduce_host(h, v.data(), c.gx, 0);
    if (rc) return rc;
    std::vector<long long> hist((size_t)c.gx);
    for (int x = 0; x < c.gx; ++x) hist[x] = (long long)v[x];
    std::vector<int> cut;
    replan_slab_cuts(hist, c.gx, h->nslab, h->cuts, cut, h->geom.layers);
    h->cuts_moved = cut != h->cuts;
    if (h->cuts_moved) {
        h->cuts = cut;
        set_slab_geometry(h);
        ++h->n_recuts;
    }
    return SPH_OK;
}

// Start of a step on a slab handle: particles that left [x_lo, x_hi) move to their new owner, last step's ghosts go, and the `layers`
// columns next to each cut are copied to the neighbour as this step's ghosts.  Old ghosts and leavers are only MARKED dead; the counting
// sort drops them.
//   ordinary step   ONE message per neighbour carries migrants and ghost copies together (k_classify_slab, all three modes), after ONE
//                   count exchange of five ints per side: records, ghost copies per column, and -- because a leaver that lands in one of
//                   my ghost columns simply stays here as a ghost, the new owner does not send it back -- how many I kept per column, which
//                   is how many of the receiver's arrivals belong to the columns it copies to me.  Two host round trips per step (the
//                   counters read-back and the count exchange) where the two-round form takes four.
//   re-cut step     two rounds (migrate, then ghost copies over what arrived): moved cuts can carry whole columns across a slab, so what
//                   arrives from one side may belong to the columns copied to the other.
// Afterwards edge_n[k][l] = particles of column l of ordered edge list k (0 ghost-left, 1 send-left, 2 send-right, 3 ghost-right), known on
// both sides of a cut alike without looking at the sorted arrays.
int slab_exchange_particles(SphHandle *h)
{
    if (!h->comm_set) return fail(h, SPH_E_STATE, "slab handle needs sph_set_comm before stepping");
    Consts &c = h->c;
    hipStream_t s = h->stream;
    const dim3 b(kBlock);
    float *warm = carries_scalar(h) ? h->warm[h->wcur] : nullptr;   // dfsph warm_start_k / iisph p_past travel with the particle
    const int cap_rec = (int)std::min<size_t>(h->comm.capacity / 32, 0x7fffffff);
    int rc;
    int n_res = c.n;                                   // resident slots, dead ones included
    int ndead = 0;
    int own_ghost[2][2] = {{0, 0}, {0, 0}}, own_kept[2][2] = {{0, 0}, {0, 0}};      // [side][column]: ghost copies I sent, leavers I kept as ghosts
    int got_ghost[2][2] = {{0, 0}, {0, 0}}, got_kept[2][2] = {{0, 0}, {0, 0}};      // ... and what the neighbour on that side reported
    auto round = [&](int mode) -> int {
        {
            ProfScope ps(h, K_SLAB);
            const int nblk = (int)grid_for(n_res).x;
            hipLaunchKernelGGL(k_classify_count, dim3(nblk), b, 0, s, c, h->geom, mode, h->P[h->pcur], h->id[h->icur], h->dead, nblk, h->class_cnt);
            hipLaunchKernelGGL(k_classify_scan, dim3(kSlabCounted), dim3(kScanBlock), 0, s, nblk, h->class_cnt, h->counters, h->native ? h->cnt_dev : (int *)nullptr);
            hipLaunchKernelGGL(k_classify_write, dim3(nblk), b, 0, s, c, h->geom, mode, h->P[h->pcur], h->V[h->vcur], warm, h->id[h->icur], h->dead,
                               (float4 *)h->dsend[0], (float4 *)h->dsend[1], cap_rec, nblk, h->class_cnt, h->ds);
        }
        int r;
        const int *ct = h->counters_host;
        int32_t sl[5], sr[5], rl[5], rr[5];
        if (h->native) {
            // the counts go from device to device (k_classify_scan left them in wire order) and come back to the host together with what the
            // neighbours sent: ONE host round trip per exchange round
            if ((r = native_exchange_counts_dev(h, 5, rl, rr))) return r;
        } else if ((r = read_counters(h))) return r;
        if (ct[0] > cap_rec || ct[1] > cap_rec) return fail(h, SPH_E_OVERFLOW, "%d/%d particle records exceed the comm buffer (%d records)", ct[0], ct[1], cap_rec);
        if (mode & kSlabMigrate) ndead = ct[2];
        { const int32_t a[5] = {ct[0], ct[3], ct[4], ct[5], ct[6]}, b2[5] = {ct[1], ct[7], ct[8], ct[9], ct[10]}; for (int q = 0; q < 5; ++q) { sl[q] = a[q]; sr[q] = b2[q]; } }
        if (!h->native && (r = slab_exchange_counts_n(h, 5, sl, sr, rl, rr))) return r;
        for (int l = 0; l < 2; ++l) {
            own_ghost[0][l] += sl[1 + l]; own_kept[0][l] += sl[3 + l]; own_ghost[1][l] += sr[1 + l]; own_kept[1][l] += sr[3 + l];
            got_ghost[0][l] += rl[1 + l]; got_kept[0][l] += rl[3 + l]; got_ghost[1][l] += rr[1 + l]; got_kept[1][l] += rr[3 + l];
        }
        if ((long long)n_res + rl[0] + rr[0] > h->ncap) return fail(h, SPH_E_OVERFLOW, "slab capacity %d exceeded by the particle exchange", h->ncap);
        if ((r = slab_xfer(h, 32 * (size_t)sl[0], 32 * (size_t)sr[0], 32 * (size_t)rl[0], 32 * (size_t)rr[0]))) return r;
        {
            ProfScope ps(h, K_SLAB);
            if (rl[0]) hipLaunchKernelGGL(k_append_records, grid_for(rl[0]), b, 0, s, (const float4 *)h->drecv[0], rl[0], n_res, h->P[h->pcur], h->V[h->vcur], warm, h->id[h->icur], h->dead);
            if (rr[0]) hipLaunchKernelGGL(k_append_records, grid_for(rr[0]), b, 0, s, (const float4 *)h->drecv[1], rr[0], n_res + rl[0], h->P[h->pcur], h->V[h->vcur], warm, h->id[h->icur], h->dead);
        }
        // owned particles: migrants out, migrants in (a record is a migrant unless it is a ghost copy)
        h->n_owned += -(sl[0] - sl[1] - sl[2]) - (sr[0] - sr[1] - sr[2]) + (rl[0] - rl[1] - rl[2]) + (rr[0] - rr[1] - rr[2]);
        n_res += rl[0] + rr[0];
        c.n = n_res;
        return SPH_OK;
    };
    if (h->cuts_moved) {
        if ((rc = round(kSlabMigrate))) return rc;
        if ((rc = round(kSlabGhosts))) return rc;
        h->cuts_moved = false;
    } else {
        if ((rc = round(kSlabMigrate | kSlabGhosts | kSlabKeep))) return rc;
    }
    HIP_TRY(h, hipGetLastError());
    h->n_dead = ndead;                        // the sort runs over everything resident, dead slots included
    for (int l = 0; l < 2; ++l) {
        h->edge_n[0][l] = got_ghost[0][l] + own_kept[0][l];      // ghost-left column l: the left neighbour's copies + my leavers that stayed as ghosts
        h->edge_n[1][l] = own_ghost[0][l] + got_kept[0][l];      // send-left column l: my copies + arrivals the left neighbour kept as ghosts
        h->edge_n[2][l] = own_ghost[1][l] + got_kept[1][l];
        h->edge_n[3][l] = got_ghost[1][l] + own_kept[1][l];
    }
    h->n_ghost = h->edge_n[0][0] + h->edge_n[0][1] + h->edge_n[3][0] + h->edge_n[3][1];
    return SPH_OK;
}

// refresh one field of the ghosts after the sweep that produced it (mode: see k_pack_field).  cols: how many of the ghost columns per side
// (1 = the column next to the cut only; the lists hold it first).
int slab_exchange_field(SphHandle *h, int mode, float4 *P, float4 *V, float *rho, int cols = 2, int gather_doubles = 0)
{
    hipStream_t s = h->stream;
    const dim3 b(kBlock);
    const size_t fl = mode == 0 ? 1 : (mode == 1 ? 3 : 2);      // modes 2 and 3: two floats
    auto cnt = [&](int k) { return h->edge_n[k][0] + (cols >= 2 && h->geom.layers >= 2 ? h->edge_n[k][1] : 0); };
    const int nsl = cnt(1), nsr = cnt(2), nrl = cnt(0), nrr = cnt(3);
    float *S = h->c.kr_split ? h->krho : nullptr;               // where the per-sweep scalar k / rho lives (else P.w)
    {
        ProfScope ps(h, K_SLAB);
        if (nsl + nsr)
            hipLaunchKernelGGL(k_pack_field, grid_for(nsl + nsr), b, 0, s, h->edge_list[1], nsl, (float *)h->dsend[0], h->edge_list[2], nsr,
                               (float *)h->dsend[1], mode, P, V, S);
    }
    int rc = slab_xfer(h, 4 * fl * nsl, 4 * fl * nsr, 4 * fl * nrl, 4 * fl * nrr, nullptr, gather_doubles);
    if (rc) return rc;
    {
        ProfScope ps(h, K_SLAB);
        if (nrl + nrr)
            hipLaunchKernelGGL(k_unpack_field, grid_for(nrl + nrr), b, 0, s, h->edge_list[0], nrl, (const float *)h->drecv[0], h->edge_list[3], nrr,
                               (const float *)h->drecv[1], mode, P, V, rho, S);
    }
    HIP_TRY(h, hipGetLastError());
    return SPH_OK;
}

// The one halo refresh of a dfsph solver iteration on a two-column slab handle: the residual sweep's value (rho_derivative / rho_adv) for
// the inner ghost column, the owner's k / rho for the outer one -- 4 bytes per ghost (k_pack_resid / k_unpack_resid).  With `overlap` the
// caller has run the EDGE tiles of the sweep only: the pack waits for them (ev_edge) on the halo's own stream, and whoever reads the ghosts
// next waits for ev_halo -- the interior tiles of the sweep run under the transfer.
int slab_exchange_resid(SphHandle *h, bool dens, float *val, bool overlap, bool wait_edge = true)
{
    hipStream_t s = overlap ? h->xstream : h->stream;
    const dim3 b(kBlock);
    const int nsl = h->edge_n[1][0] + h->edge_n[1][1], nsr = h->edge_n[2][0] + h->edge_n[2][1];
    const int nrl = h->edge_n[0][0] + h->edge_n[0][1], nrr = h->edge_n[3][0] + h->edge_n[3][1];
    float *S = h->c.kr_split ? h->krho : nullptr;
    float4 *P = h->P[1 - h->pcur];
    if (overlap && wait_edge) HIP_TRY(h, hipStreamWaitEvent(s, h->ev_edge, 0));
    {
        ProfScope ps(h, K_SLAB, s);
        if (nsl + nsr)
            hipLaunchKernelGGL(k_pack_resid, grid_for(nsl + nsr), b, 0, s, h->edge_list[1], nsl, h->edge_n[1][0], (float *)h->dsend[0], h->edge_list[2], nsr,
                               h->edge_n[2][0], (float *)h->dsend[1], val, P, S);
    }
    int rc = slab_xfer(h, 4 * (size_t)nsl, 4 * (size_t)nsr, 4 * (size_t)nrl, 4 * (size_t)nrr, s);
    if (rc) return rc;
    {
        ProfScope ps(h, K_SLAB, s);
        if (nrl + nrr)
            hipLaunchKernelGGL(k_unpack_resid, grid_for(nrl + nrr), b, 0, s, h->c, h->edge_list[0], nrl, h->edge_n[0][0], (const float *)h->drecv[0], h->edge_list[3], nrr,
                               h->edge_n[3][0], (const float *)h->drecv[1], dens ? 1 : 0, h->aux, h->rho, h->ds, val, P, S);
    }
    HIP_TRY(h, hipGetLastError());
    if (overlap) HIP_TRY(h, hipEventRecord(h->ev_halo, s));
    return SPH_OK;
}

// ---------------------------------------------------------------------------------------------
// rigid body of config 5: host-side construction and rigid_solver.step orchestration
// ---------------------------------------------------------------------------------------------
void cross3h(const float a[3], const float b[3], float out[3])
{
    out[0] = a[1] * b[2] - a[2] * b[1];
    out[1] = a[2] * b[0] - a[0] * b[2];
    out[2] = a[0] * b[1] - a[1] * b[0];
}
void matvec3h(const float m[9], const float v[3], float out[3])
{
    for (int r = 0; r < 3; ++r) out[r] = (m[3 * r] * v[0] + m[3 * r + 1] * v[1]) + m[3 * r + 2] * v[2];
}
void matmul3h(const float a[9], const float b[9], float out[9])
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) out[3 * r + c] = (a[3 * r] * b[c] + a[3 * r + 1] * b[3 + c]) + a[3 * r + 2] * b[6 + c];
}
// ti.math.inverse for a 3x3 matrix (cofactor form, [taichi-semantics, unverifiable here])
void inverse3h(const float m[9], float out[9])
{
    auto E = [&](int x, int y) { return m[3 * (x % 3) + (y % 3)]; };
    float det = (m[0] * (m[4] * m[8] - m[7] * m[5]) - m[3] * (m[1] * m[8] - m[7] * m[2])) + m[6] * (m[1] * m[5] - m[4] * m[2]);
    float inv_det = 1.0f / det;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            out[3 * j + i] = inv_det * (E(i + 1, j + 1) * E(i + 2, j + 2) - E(i + 2, j + 1) * E(i + 1, j + 2));
}
// ti.math.rotation3d(ang_x, ang_y, ang_z), 3x3 block ([taichi-semantics]: the body's orientation is 'parity unpinned')
void rotation3dh(float ang_x, float ang_y, float ang_z, float m[9])
{
    float ca = cosf(ang_x), sa = sinf(ang_x), cb = cosf(ang_z), sb = sinf(ang_z), cy = cosf(ang_y), sy = sinf(ang_y);
    m[0] = cb * cy + sb * sa * sy; m[1] = sb * ca; m[2] = -cb * sy + sb * sa * cy;
    m[3] = -sb * cy + cb * sa * sy; m[4] = cb * ca; m[5] = sb * sy + cb * sa * cy;
    m[6] = ca * sy; m[7] = -sa; m[8] = ca * cy;
}

RigidView rigid_view(const SphHandle *h)
{
    RigidView rv;
    memset(&rv, 0, sizeof(rv));
    rv.RP = h->RPs; rv.rid = h->rid; rv.rcell_start = h->rcell_start; rv.pos_orig = h->pos_orig; rv.rho_orig = h->rho_orig;
    for (int a = 0; a < 3; ++a) {
        rv.c[a] = h->centroid[a]; rv.vel[a] = h->r_vel[a]; rv.acc[a] = h->r_acc[a]; rv.omega[a] = h->r_omega[a]; rv.alpha[a] = h->r_alpha[a];
    }
    rv.n_fluid = h->N;
    return rv;
}

inline bool rigid_coupled(const SphHandle *h) { return h->rigid && h->rigid_active && h->cfg.fs_couple; }
inline RigidView rigid_view_or_none(const SphHandle *h) { return rigid_coupled(h) ? rigid_view(h) : RigidView(); }

// the tolerance-grade sweeps (sph_relaxed_kernels.h) run on this handle
// (with a coupled body -- rx_split -- they cover the workgroups with 16-bit lists, i.e. without a rigid sample in reach, and the exact RIGID sweeps the thin
// shell around the body: two launches per sweep over the two halves of tile_order)
inline bool use_relaxed(const SphHandle *h) { return h->relaxed && h->staged && h->c.kr_split && h->wall_grad && (!rigid_coupled(h) || (h->tile_order && !h->slab)); }
inline bool rx_split(const SphHandle *h) { return use_relaxed(h) && rigid_coupled(h); }
// dfsph handles the tolerance-grade kernels of sph_relaxed_kernels.h do not cover because their sweeps are not staged (scenes below 131 k particles in
// the reference's cell order: plain and quad sweeps): the exact sweeps with the kernel functions KF<true> -- same lists, same order of the sums
inline bool relaxed_unstaged(const SphHandle *h) { return h->relaxed && h->cfg.solver == SPH_SOLVER_DFSPH && !h->staged && !rigid_coupled(h); }
// pcisph / iisph under the relaxed arithmetic: the sweeps take KF<true> (sph_device.h); plain and staged sweeps, no coupled body (the quad sweeps of
// small scenes and the RIGID instantiations stay exact)
inline bool relaxed_pressure(const SphHandle *h)
{
    return h->relaxed && (h->cfg.solver == SPH_SOLVER_PCISPH || h->cfg.solver == SPH_SOLVER_IISPH) && !rigid_coupled(h) &&
           (h->staged || !(!h->slab && h->opt_quad && h->c.n <= h->quad_below));
}

// init_rigid_particles_pos + init_rigid_particles_data (ParticleSystem.py:198-223, 249-295), once, on the host
int build_rigid(SphHandle *h, const SphRigid *rg)
{
    const Consts &c = h->c;
    h->Nr = rg->n_particles; h->Nv = rg->n_vertices;
    h->rigid_active = rg->active ? 1 : 0;
    h->rigid_rho = (float)rg->rho_0;
    const int Nr = h->Nr, Nv = h->Nv;
    const double pi = 3.141592653589793;
    float att[3], m[9], off[3];
    for (int a = 0; a < 3; ++a) { att[a] = (float)(rg->attitude_offset[a] / 180.0 * pi); off[a] = (float)rg->pos_offset[a]; }   // :52
    rotation3dh(att[0], att[2], att[1], m);                                                                                     // :200
    std::vector<float> rpos(3 * (size_t)Nr), rvert(3 * (size_t)(Nv > 0 ? Nv : 1));
    for (int pass = 0; pass < 2; ++pass) {
        const int n = pass == 0 ? Nr : Nv;
        const float *src = pass == 0 ? rg->points : rg->vertices;
        float *dst = pass == 0 ? rpos.data() : rvert.data();
        for (int i = 0; i < n; ++i) {
            const float p[3] = {src[3 * i], src[3 * i + 1], src[3 * i + 2]};
            for (int r = 0; r < 3; ++r) {
                float v = ((m[3 * r] * p[0] + m[3 * r + 1] * p[1]) + m[3 * r + 2] * p[2]) + 0.0f * 1.0f;   // mat4 @ (p, 1), :205-207
                dst[3 * i + r] = v + off[r];                                                                // :218, :223
            }
        }
    }
    // rigid cell list (canonical: ascending index inside a cell) for the one-time volume sums
    std::vector<int> rc3(3 * (size_t)Nr), rcell(Nr), rstart((size_t)c.C + 1, 0);
    for (int i = 0; i < Nr; ++i) {
        int cx = (int)floorf(rpos[3 * (size_t)i] / c.h), cy = (int)floorf(rpos[3 * (size_t)i + 1] / c.h), cz = (int)floorf(rpos[3 * (size_t)i + 2] / c.h);
        int id = cx + cy * c.sy + cz * c.sz;
        if (id < 0 || id >= c.C) return fail(h, SPH_E_INVALID, "rigid particle %d starts outside the grid", i);
        rc3[3 * (size_t)i] = cx; rc3[3 * (size_t)i + 1] = cy; rc3[3 * (size_t)i + 2] = cz;
        rcell[i] = id;
        rstart[(size_t)id + 1]++;
    }
    for (int k = 0; k < c.C; ++k) rstart[(size_t)k + 1] += rstart[k];
    std::vector<int> fill(rstart.begin(), rstart.end() - 1), order(Nr);
    for (int i = 0; i < Nr; ++i) order[fill[rcell[i]]++] = i;
    h->rvol_host.assign(Nr, 0.f);
    h->rmass_host.assign(Nr, 0.f);
    for (int i = 0; i < Nr; ++i) {                                                     // :252-259
        float volume = 0.f;
        if (h->rigid_active) {
            const float *pi_ = &rpos[3 * (size_t)i];
            for (int dx = -1; dx <= 1; ++dx)
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dz = -1; dz <= 1; ++dz) {
                        int x = rc3[3 * (size_t)i] + dx, y = rc3[3 * (size_t)i + 1] + dy, z = rc3[3 * (size_t)i + 2] + dz;
                        if (x >= c.gx || y >= c.gy || z >= c.gz) continue;
                        if (x < 0 || y < 0 || z < 0) continue;
                        int cid = x + y * c.sy + z * c.sz;
                        for (int e = rstart[cid]; e < rstart[(size_t)cid + 1]; ++e) {
                            int j = order[e];
                            if (j == i) continue;
                            float ddx = pi_[0] - rpos[3 * (size_t)j], ddy = pi_[1] - rpos[3 * (size_t)j + 1], ddz = pi_[2] - rpos[3 * (size_t)j + 2];
                            float q = sqrtf((ddx * ddx + ddy * ddy) + ddz * ddz);
                            if (q > c.h) continue;
                            volume += host_cubic_w(q, c.h, c.kw);
                        }
                    }
        }
        h->rvol_host[i] = volume < 1e-6f ? 0.0f : 1.0f / volume;
    }
    for (int i = 0; i < Nr; ++i) h->rmass_host[i] = h->rigid_rho * h->rvol_host[i];     // :262-263
    float cs[3] = {0, 0, 0}, sum_mass = 0.f;                                            // :266-271
    for (int i = 0; i < Nr; ++i) {
        for (int a = 0; a < 3; ++a) cs[a] += rpos[3 * (size_t)i + a] * h->rmass_host[i];
        sum_mass += h->rmass_host[i];
    }
    for (int a = 0; a < 3; ++a) h->centroid[a] = cs[a] / sum_mass;
    float Ixx = 0, Iyy = 0, Izz = 0, Ixy = 0, Ixz = 0, Iyz = 0;                         // :275-288
    for (int i = 0; i < Nr; ++i) {
        float x = rpos[3 * (size_t)i] - h->centroid[0], y = rpos[3 * (size_t)i + 1] - h->centroid[1], z = rpos[3 * (size_t)i + 2] - h->centroid[2];
        float mi = h->rmass_host[i];
        Ixx += mi * (y * y + z * z);
        Iyy += mi * (x * x + z * z);
        Izz += mi * (x * x + y * y);
        Ixy += -mi * (x * y);
        Ixz += -mi * (x * z);
        Iyz += -mi * (z * y);
    }
    const float I[9] = {Ixx, Ixy, Ixz, Ixy, Iyy, Iyz, Ixz, Iyz, Izz};
    inverse3h(I, h->inertia_inv);                                                       // :291
    h->rs_dt = (float)h->cfg.delta_time;                                                // rigid_solver.py:13
    h->rigid_pos_host = rpos;
    // device buffers
    int rc;
    const size_t nr = (size_t)Nr;
    if ((rc = dalloc(h, &h->RPos, nr))) return rc;
    if ((rc = dalloc(h, &h->RPs, nr))) return rc;
    if ((rc = dalloc(h, &h->rid, nr))) return rc;
    if ((rc = dalloc(h, &h->rcell_of, nr))) return rc;
    if ((rc = dalloc(h, &h->rrank, nr))) return rc;
    if ((rc = dalloc(h, &h->rslot, nr))) return rc;
    if ((rc = dalloc(h, &h->rcell_count, (size_t)c.S + 2))) return rc;
    if ((rc = dalloc(h, &h->rcell_start, (size_t)c.S + 2))) return rc;
    if ((rc = dalloc(h, &h->rforce, 3 * nr))) return rc;
    if ((rc = dalloc(h, &h->rvert, 3 * (size_t)(Nv > 0 ? Nv : 1)))) return rc;
    // (indexed by ORIGINAL particle id: on a slab handle that is the whole scene's id range, whatever this rank holds)
    const size_t by_id = std::max((size_t)h->c.stride, (size_t)h->N);
    if ((rc = dalloc(h, &h->pos_orig, by_id))) return rc;
    if ((rc = dalloc(h, &h->rho_orig, by_id))) return rc;
    if ((rc = dalloc(h, &h->ncount, (size_t)h->c.stride))) return rc;
    if ((rc = dalloc(h, &h->rred, kRigidParts))) return rc;
    if ((rc = dalloc(h, &h->rvmax_part, kRigidParts))) return rc;
    if ((rc = dalloc(h, &h->rnl, (nr + 64) * (size_t)c.kpitch))) return rc;
    if ((rc = dalloc(h, &h->rcnt, nr))) return rc;
    if (h->relaxed && h->staged && !h->tile_order) {       // relaxed arithmetic next to a body: the tile order of the exact / relaxed split (rx_split)
        if ((rc = dalloc(h, &h->tile_flag, (size_t)(h->c.stride + kBlock - 1) / kBlock + 1))) return rc;
        if ((rc = dalloc(h, &h->tile_order, (size_t)(h->c.stride + kBlock - 1) / kBlock + 2))) return rc;
    }
    const size_t stg_need = 3 * std::max(nr, (size_t)Nv);
    if (stg_need > 3 * std::max((size_t)h->c.stride, (size_t)h->Nb))
        if ((rc = dalloc(h, &h->staging, stg_need))) return rc;      // the fluid arena's staging buffer is too small for this body
    if ((rc = dcommit(h))) return rc;
    HIP_TRY(h, hipHostMalloc((void **)&h->rred_host, sizeof(RigidReduce) * (kRigidParts + 1), hipHostMallocDefault));
    std::vector<float4> rp4(nr);
    for (int i = 0; i < Nr; ++i) rp4[i] = make_float4(rpos[3 * (size_t)i], rpos[3 * (size_t)i + 1], rpos[3 * (size_t)i + 2], h->rvol_host[i]);
    HIP_TRY(h, hipMemcpyAsync(h->RPos, rp4.data(), sizeof(float4) * nr, hipMemcpyHostToDevice, h->stream));
    if (Nv > 0) HIP_TRY(h, hipMemcpyAsync(h->rvert, rvert.data(), sizeof(float) * 3 * (size_t)Nv, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemsetAsync(h->rforce, 0, sizeof(float) * 3 * nr, h->stream));
    HIP_TRY(h, hipMemsetAsync(h->rcell_start, 0, sizeof(int) * ((size_t)c.S + 2), h->stream));
    HIP_TRY(h, hipMemsetAsync(h->rho_orig, 0, sizeof(float) * by_id, h->stream));
    HIP_TRY(h, hipMemsetAsync(h->ncount, 0, sizeof(int) * (size_t)h->c.stride, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->rigid = true;
    return SPH_OK;
}

// per step: cell-sort the rigid sample particles (update_grid_rigid_particles, ParticleSystem.py:399-407)
int stage_sort_rigid(SphHandle *h)
{
    Consts cr = h->c;
    cr.n = h->Nr;
    hipStream_t s = h->stream;
    const dim3 g = grid_for(h->Nr), b(kBlock);
    const size_t ncell = (size_t)cr.S + 2;
    ProfScope ps(h, K_RIGID);
    HIP_TRY(h, hipMemsetAsync(h->rcell_count, 0, sizeof(int) * ncell, s));
    hipLaunchKernelGGL(k_hash_count, g, b, 0, s, cr, h->RPos, (const int *)nullptr, h->rcell_of, h->rrank, h->rcell_count, (DevScalars *)nullptr);
    hipLaunchKernelGGL(k_scan_tiles, dim3(h->ntiles), b, 0, s, h->rcell_count, h->rcell_start, h->tile_sums, (int)ncell);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), b, 0, s, h->tile_sums, h->ntiles);
    hipLaunchKernelGGL(k_scan_add, grid_for((int)ncell), b, 0, s, h->rcell_start, h->tile_sums, (int)ncell);
    hipLaunchKernelGGL(k_scatter, g, b, 0, s, cr, h->rcell_of, h->rrank, h->rcell_start, h->rslot);
    hipLaunchKernelGGL(k_rigid_order, g, b, 0, s, h->Nr, h->rcell_of, h->rcell_start, h->rslot, h->RPos, h->RPs, h->rid);
    HIP_TRY(h, hipGetLastError());
    return SPH_OK;
}

RigidBodyState rigid_state(const SphHandle *h, const float vel[3], const float ori[3])
{
    RigidBodyState st;
    memset(&st, 0, sizeof(st));
    for (int a = 0; a < 3; ++a) {
        st.c[a] = h->centroid[a]; st.omega[a] = h->rs_omega[a];
        st.vel[a] = vel ? vel[a] : 0.f; st.ori[a] = ori ? ori[a] : 0.f;
        st.lo[a] = (float)h->cfg.box_min[a] + h->c.d;                   // rigid_solver.py:56
        st.hi[a] = (float)h->cfg.box_max[a] - h->c.d;                   // :65
    }
    return st;
}

inline dim3 rigid_parts_grid(const SphHandle *h) { return dim3((unsigned)std::max(1, std::min(kRigidParts, (h->Nr + kBlock - 1) / kBlock))); }

// the partials of k_rigid_torque_force / k_rigid_collide (one per workgroup) combined in index order into rred_host[0]
int read_rigid_reduce(SphHandle *h)
{
    const int np = (int)rigid_parts_grid(h).x;
    RigidReduce *part = h->rred_host + 1;
    HIP_TRY(h, hipMemcpyAsync(part, h->rred, sizeof(RigidReduce) * (size_t)np, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    RigidReduce r = part[0];
    int lo[3], hi[3];
    for (int a = 0; a < 3; ++a) { lo[a] = r.cnorm[a] & 1; hi[a] = (r.cnorm[a] >> 1) & 1; }
    for (int k = 1; k < np; ++k) {
        const RigidReduce &q = part[k];
        for (int a = 0; a < 3; ++a) {
            r.torque[a] += q.torque[a]; r.force[a] += q.force[a]; r.cp[a] += q.cp[a];
            r.dmax[a] = fmaxf(r.dmax[a], q.dmax[a]); r.dmin[a] = fminf(r.dmin[a], q.dmin[a]);
            lo[a] |= q.cnorm[a] & 1; hi[a] |= (q.cnorm[a] >> 1) & 1;
        }
        r.ccount += q.ccount;
    }
    // collision_norm[j]: -1 from the lower wall, +1 from the upper wall; if both fire in one step the later write wins in the reference
    // (a race); here the upper wall wins, as in the oracle's particle loop order per axis
    for (int a = 0; a < 3; ++a) r.cnorm[a] = hi[a] ? 1 : (lo[a] ? -1 : 0);
    h->rred_host[0] = r;
    return SPH_OK;
}

// rigid_solver.step                                                      rigid_solver.py:216-232
int rigid_step(SphHandle *h)
{
    hipStream_t s = h->stream;
    const dim3 b(kBlock), gr = grid_for(h->Nr), gv = grid_for(h->Nv > 0 ? h->Nv : 1);
    int rc;
    if (!h->rs_run_once) {                                              // compute_sum_mass :156-162
        float sm = 0.f;
        for (int i = 0; i < h->Nr; ++i) sm += h->rmass_host[i];
        h->rs_mass = sm;
        h->rs_run_once = true;
    }
    h->rs_cnt += 1;
    if (h->cfg.solver == SPH_SOLVER_DFSPH) {
        if ((rc = read_scalars(h))) return rc;
        if (h->ds_host->ps_dt > 0.0f) h->rs_dt = h->ds_host->ps_dt;     // :223-224
    }
    const float dt = h->rs_dt;
    ProfScope ps(h, K_RIGID);
    // compute_attitude :118-128 (+ the force sum of kinematic :35-38: the forces do not change in between)
    hipLaunchKernelGGL(k_rigid_torque_force, rigid_parts_grid(h), b, 0, s, h->Nr, h->RPos, h->rforce, rigid_state(h, nullptr, nullptr), h->rred);
    if ((rc = read_rigid_reduce(h))) return rc;
    {
        const float torque[3] = {(float)h->rred_host->torque[0], (float)h->rred_host->torque[1], (float)h->rred_host->torque[2]};
        float alpha[3];
        matvec3h(h->inertia_inv, torque, alpha);
        for (int a = 0; a < 3; ++a) {
            h->rs_omega[a] += alpha[a] * dt;
            h->rs_attitude[a] = h->rs_omega[a] * dt;
            h->r_alpha[a] = alpha[a];
        }
    }
    // rotation :130-141
    {
        Mat3 R;
        float mt[9], tmp[9], out[9];
        rotation3dh(-h->rs_attitude[0], -h->rs_attitude[2], -h->rs_attitude[1], R.m);
        const RigidBodyState st = rigid_state(h, nullptr, nullptr);
        hipLaunchKernelGGL(k_rigid_rotate, gr, b, 0, s, h->Nr, h->RPos, (float *)nullptr, R, st);
        if (h->Nv > 0) hipLaunchKernelGGL(k_rigid_rotate, gv, b, 0, s, h->Nv, (float4 *)nullptr, h->rvert, R, st);
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) mt[3 * r + c] = R.m[3 * c + r];
        matmul3h(R.m, h->inertia_inv, tmp);
        matmul3h(tmp, mt, out);
        memcpy(h->inertia_inv, out, sizeof(out));
    }
    // kinematic :33-104
    float vel[3], disp[3], ori[3];
    {
        const float force[3] = {(float)h->rred_host->force[0], (float)h->rred_host->force[1], (float)h->rred_host->force[2]};
        const float g[3] = {h->c.gravity * 0.0f, h->c.gravity * -1.0f, h->c.gravity * 0.0f};
        for (int a = 0; a < 3; ++a) {
            h->r_acc[a] = force[a] / h->rs_mass + g[a];                 // :40-41
            vel[a] = h->r_acc[a] * dt + h->r_vel[a];                    // :43
            disp[a] = vel[a] * dt;                                      // :45
            ori[a] = disp[a];
        }
    }
    hipLaunchKernelGGL(k_rigid_collide, rigid_parts_grid(h), b, 0, s, h->Nr, h->RPos, rigid_state(h, vel, ori), h->rred);
    if ((rc = read_rigid_reduce(h))) return rc;
    const RigidReduce &rr = *h->rred_host;
    for (int j = 0; j < 3; ++j) {
        disp[j] = disp[j] > rr.dmax[j] ? disp[j] : rr.dmax[j];          // :58 (all lower-wall maxima, then the upper-wall minima)
        disp[j] = rr.dmin[j] < disp[j] ? rr.dmin[j] : disp[j];          // :67
    }
    if (rr.ccount > 0) {                                                // :80-94
        const float cnorm[3] = {(float)rr.cnorm[0], (float)rr.cnorm[1], (float)rr.cnorm[2]};
        float cpt[3], cv[3], wr[3];
        for (int a = 0; a < 3; ++a) cpt[a] = ((float)rr.cp[a] + ori[a]) / (float)rr.ccount - h->centroid[a];
        cross3h(h->rs_omega, cpt, wr);
        for (int a = 0; a < 3; ++a) cv[a] = vel[a] + wr[a];
        const float mu_n = 0.1f, mu_c = (float)(0.8 * (1 + 0.1));        // compute_new_vel :106-116
        float vdn = (cv[0] * cnorm[0] + cv[1] * cnorm[1]) + cv[2] * cnorm[2];
        float vn[3], vt[3], vnew[3];
        for (int a = 0; a < 3; ++a) { vn[a] = vdn * cnorm[a]; vt[a] = cv[a] - vn[a]; }
        float nvn = sqrtf((vn[0] * vn[0] + vn[1] * vn[1]) + vn[2] * vn[2]);
        float nvt = sqrtf((vt[0] * vt[0] + vt[1] * vt[1]) + vt[2] * vt[2]);
        float a_ = 1.0f - mu_c * nvn / nvt;
        a_ = a_ > 0.0f ? a_ : 0.0f;
        for (int a = 0; a < 3; ++a) vnew[a] = a_ * vt[a] + (-mu_n * vn[a]);
        const float rx[9] = {0, -cpt[2], cpt[1], cpt[2], 0, -cpt[0], -cpt[1], cpt[0], 0};
        float t1[9], t2[9], K[9], Kinv[9], dv[3], jimp[3], cj[3], dw[3];
        matmul3h(rx, h->inertia_inv, t1);
        matmul3h(t1, rx, t2);
        for (int q = 0; q < 9; ++q) K[q] = ((q % 4 == 0) ? 1.0f / h->rs_mass : 0.0f / h->rs_mass) - t2[q];
        inverse3h(K, Kinv);
        for (int a = 0; a < 3; ++a) dv[a] = vnew[a] - cv[a];
        matvec3h(Kinv, dv, jimp);
        for (int a = 0; a < 3; ++a) vel[a] += jimp[a] / h->rs_mass;
        cross3h(cpt, jimp, cj);
        matvec3h(h->inertia_inv, cj, dw);
        for (int a = 0; a < 3; ++a) h->rs_omega[a] += dw[a];
    }
    for (int a = 0; a < 3; ++a) { h->r_omega[a] = h->rs_omega[a]; h->r_vel[a] = vel[a]; }   // :96-97
    hipLaunchKernelGGL(k_rigid_translate, gr, b, 0, s, h->Nr, h->RPos, (float *)nullptr, disp[0], disp[1], disp[2], h->rforce);   // :98-99, :38
    if (h->Nv > 0) hipLaunchKernelGGL(k_rigid_translate, gv, b, 0, s, h->Nv, (float4 *)nullptr, h->rvert, disp[0], disp[1], disp[2], (float *)nullptr);
    for (int a = 0; a < 3; ++a) h->centroid[a] += disp[a];                                  // :104
    HIP_TRY(h, hipGetLastError());
    h->nl_valid = false;
    return SPH_OK;
}

// ---------------------------------------------------------------------------------------------
// step stages
// ---------------------------------------------------------------------------------------------
// solver_base.step() prologue: reset_grid + update_grid (solver_base.py:136-143) as a counting sort,
// then the neighbour lists.
int stage_sort_and_lists(SphHandle *h)
{
    int rc;
    if (h->slab) {
        if (h->rebalance_every > 0 && ++h->steps_since_rebalance >= h->rebalance_every) {
            h->steps_since_rebalance = 0;
            if ((rc = slab_rebalance(h))) return rc;
        }
        if ((rc = slab_exchange_particles(h))) return rc;
    }
    Consts &c = h->c;
    // 16-bit local indices in the fluid lists of staged workgroups of the dfsph sweeps (the pcisph / iisph sweeps keep the 32-bit walks).  With a
    // coupled body the list build decides per workgroup: tagged rigid entries need 32 bits, so the workgroups with a rigid sample in one of their
    // neighbourhood cells keep 32-bit local indices (kStageLists16 in stage_cnt).  SPH_NL16=0 at sph_create turns it off (A/B, tests/test_cell_order_gpu.py)
    c.nl16 = (h->staged && is_dfsph(h) && h->opt_nl16) ? 1 : 0;
    // k / rho in its own array: dfsph handles with staged sweeps (on slab handles the ghost refreshes write it)
    c.kr_split = (c.nl16 && h->opt_kr_split) ? 1 : 0;
    hipStream_t s = h->stream;
    dim3 g = grid_for(c.n);
    const dim3 b(kBlock);
    const size_t ncell = (size_t)c.S + 2;       // cell slots, "outside the grid" bucket S, end
    const bool dfsph = h->cfg.solver == SPH_SOLVER_DFSPH;
    const bool carry = carries_scalar(h);
    (void)dfsph;
    // Verlet handles: every kernel of the sort and the list build is enqueued every step and leaves at once unless the integrator of the
    // step before found a particle skin / 2 away from where the lists were built (k_verlet_decide: DevScalars.moved -> rebuild)
    const int *gate = h->verlet ? &h->ds->moved : nullptr;
    {
        ProfScope ps(h, K_HASH);
        // cell_count is clean: the arena starts zeroed and k_scan_tiles zeroes the histogram as it consumes it
        hipLaunchKernelGGL(k_hash_count, g, b, 0, s, c, h->P[h->pcur], h->slab ? h->dead : (const int *)nullptr, h->cell_of, h->rank,
                           h->cell_count, h->ds, gate);
    }
    {
        ProfScope ps(h, K_SCAN);
        hipLaunchKernelGGL(k_scan_tiles, dim3(h->ntiles), b, 0, s, h->cell_count, h->cell_start, h->tile_sums, (int)ncell, gate);
        const int fold = h->ntiles <= kScanFoldTiles ? 1 : 0;
        if (!fold) hipLaunchKernelGGL(k_scan_sums, dim3(1), b, 0, s, h->tile_sums, h->ntiles, gate);
        hipLaunchKernelGGL(k_scan_add, grid_for((int)ncell), b, 0, s, h->cell_start, h->tile_sums, (int)ncell, gate, fold);
    }
    {
        ProfScope ps(h, K_SCATTER);
        hipLaunchKernelGGL(k_scatter, g, b, 0, s, c, h->cell_of, h->rank, h->cell_start, h->slot_src, gate);
    }
    if (h->slab) {
        // dead slots took no part in the sort: the sorted arrays end after the live particles
        c.n -= h->n_dead;
        h->n_dead = 0;
        h->nblocks = (c.n + kBlock - 1) / kBlock;
        g = grid_for(c.n);
    }
    {
        ProfScope ps(h, K_ORDER_GATHER);
        hipLaunchKernelGGL(k_order_gather, g, b, 0, s, c, h->cell_of, h->cell_start, h->slot_src, h->P[h->pcur], h->V[h->vcur],
                           carry ? h->warm[h->wcur] : (const float *)nullptr, h->id[h->icur], h->P[1 - h->pcur], h->V[1 - h->vcur],
                           h->warm[1 - h->wcur], h->id[1 - h->icur], rigid_coupled(h) ? h->pos_orig : (float4 *)nullptr, gate, h->x0);
        h->pcur ^= 1; h->vcur ^= 1; h->icur ^= 1;
        if (carry) h->wcur ^= 1;
    }
    if (h->slab) {
        HIP_TRY(h, hipMemsetAsync(h->dead, 0, sizeof(int) * (size_t)c.n, s));
        ProfScope ps(h, K_SLAB);
        // ordered edge lists: list k, column l (0 = next to the cut): ghost-left x_lo - 1 - l, send-left x_lo + l, send-right x_hi - 1 - l, ghost-right x_hi + l
        const SlabGeom &sg = h->geom;
        LayerJobs jobs;
        jobs.n = 0;
        for (int k = 0; k < 4; ++k) {
            if (!(k < 2 ? sg.has_left : sg.has_right)) continue;
            for (int l = 0; l < sg.layers; ++l) {
                jobs.col[jobs.n] = k == 0 ? sg.x_lo - 1 - l : k == 1 ? sg.x_lo + l : k == 2 ? sg.x_hi - 1 - l : sg.x_hi + l;
                jobs.off[jobs.n] = h->edge_off[2 * k + l];
                jobs.list[jobs.n] = h->edge_list[k] + (l ? h->edge_n[k][0] : 0);
                jobs.n += 1;
            }
        }
        if (jobs.n) {
            hipLaunchKernelGGL(k_layer_offsets, dim3(jobs.n), dim3(kScanBlock), 0, s, c, h->cell_start, jobs);
            hipLaunchKernelGGL(k_layer_list, dim3(grid_for(c.gy * c.gz).x, jobs.n), b, 0, s, c, h->cell_start, jobs);
        }
        if (dev_env(&h->overrides, "SPH_SLAB_CHECK")) {       // the host's bookkeeping of the column populations against the sorted arrays
            for (int k = 0; k < 4; ++k)
                for (int l = 0; l < sg.layers; ++l) {
                    if (!(k < 2 ? sg.has_left : sg.has_right)) continue;
                    int tot = -1;
                    HIP_TRY(h, hipMemcpyAsync(&tot, h->edge_off[2 * k + l] + (size_t)c.gy * c.gz, sizeof(int), hipMemcpyDeviceToHost, s));
                    HIP_TRY(h, hipStreamSynchronize(s));
                    if (tot != h->edge_n[k][l])
                        return fail(h, SPH_E_STATE, "slab %d step %d: edge list %d column %d holds %d particles, the exchange counted %d", h->slab_rank, h->simulate_cnt, k, l, tot, h->edge_n[k][l]);
                }
        }
        if (h->overlap && h->overlap_on) {       // edge tiles first, then the interior (k_tile_order); tile_order[ntiles] = number of edge tiles
            hipLaunchKernelGGL(k_tile_flags, g, b, 0, s, c, h->geom, h->P[h->pcur], h->tile_flag);
            hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, s, h->tile_flag, h->nblocks, h->tile_order);
        }
    }
    if (h->slab && rigid_coupled(h)) {      // fluid positions by original id < Nr, from whichever rank owns them (the get_neighbour_count quirk)
        ProfScope ps(h, K_RIGID);
        HIP_TRY(h, hipMemsetAsync(h->red_dev, 0, sizeof(double) * 4 * (size_t)h->Nr, s));
        hipLaunchKernelGGL(k_collect_by_id, g, b, 0, s, c.n, h->id[h->icur], h->P[h->pcur], (const float *)nullptr, h->Nr, h->red_dev);
        if ((rc = slab_allreduce_stream(h, 4 * h->Nr, 0))) return rc;
        hipLaunchKernelGGL(k_spread_by_id, grid_for(h->Nr), b, 0, s, h->Nr, h->red_dev, h->pos_orig, (float *)nullptr);
    }
    if (rigid_coupled(h) && (rc = stage_sort_rigid(h))) return rc;
    {
        ProfScope ps(h, K_BUILD_NL);
        // (the per-build maxima were zeroed by k_hash_count; `overflow` stays sticky until check_overflow reports it)
#define SPH_BNL(R, S) hipLaunchKernelGGL((k_build_nl<R, S>), g, b, 0, s, c, h->P[h->pcur], h->cell_start, h->WP, h->wcell_start, h->id[h->icur], \
                                             h->nl, h->nlb, h->cnt, h->ds, rigid_view_or_none(h), h->ncount, h->stage_src, h->stage_cnt, gate)
#define SPH_BNL_SPLIT(R, NW) hipLaunchKernelGGL((k_build_nl_split<R, NW>), dim3((unsigned)std::max(1, (c.n + 63) / 64)), dim3(NW * 64), 0, s, c, h->P[h->pcur], \
                                                h->cell_start, h->WP, h->wcell_start, h->id[h->icur], h->nl, h->nlb, h->cnt, h->ds, rigid_view_or_none(h), h->ncount, gate)
        // small unstaged scenes: one wave per dx-plane (3) or per (dx, dy) column (9) of the same 64 particles.  Measured (tools/split_sweep.sh):
        // 22 k particles 77 -> 55 -> 34 us, 29 k 54 -> 32 -> 27 us, 55 k 146 -> 81 -> 64 us (rigid) / 56 -> 44 -> 48 us; 250 k 69 -> 87 -> 122 us.
        const bool rg = rigid_coupled(h);
        const int split = h->staged ? 0 : h->opt_bnl_split >= 0 ? h->opt_bnl_split : c.n <= kBnlSplit9Below ? 9 : c.n <= kBnlSplitBelow ? 3 : 0;
        if (rg && h->staged) SPH_BNL(true, true);
        else if (h->staged) SPH_BNL(false, true);
        else if (rg && split) { if (split == 9) SPH_BNL_SPLIT(true, 9); else SPH_BNL_SPLIT(true, 3); }
        else if (rg) SPH_BNL(true, false);
        else if (split) { if (split == 9) SPH_BNL_SPLIT(false, 9); else SPH_BNL_SPLIT(false, 3); }
        else SPH_BNL(false, false);
#undef SPH_BNL
#undef SPH_BNL_SPLIT
    }
    if (rx_split(h)) {       // tiles with a rigid sample in reach (32-bit lists) first: the exact RIGID sweeps take them, the relaxed sweeps the rest
        ProfScope ps(h, K_BUILD_NL);
        hipLaunchKernelGGL(k_tile_flags_exact, grid_for(h->nblocks), b, 0, s, h->stage_cnt, h->nblocks, h->tile_flag);
        hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, s, h->tile_flag, h->nblocks, h->tile_order);
    }
    if (rigid_coupled(h)) {      // the body's view of the fluid, for the force kernels of this step
        ProfScope ps(h, K_RIGID);
        hipLaunchKernelGGL(k_build_rnl, grid_for(h->Nr), b, 0, s, c, h->Nr, h->RPs, h->P[h->pcur], h->cell_start, h->rnl, h->rcnt, h->ds);
    }
    if (h->wall_grad && h->c.kr_split && h->c.boundary_handle && use_relaxed(h)) {     // the wall sums of this step's positions
        ProfScope ps(h, K_BUILD_NL);
        hipLaunchKernelGGL(k_rx_wall_grad, g, b, 0, s, c, h->P[h->pcur], h->WP, h->nlb, h->cnt, h->wall_grad, h->wall_gsq);
    }
    HIP_TRY(h, hipGetLastError());
    h->nl_valid = true;
    h->density_valid = false;
    return SPH_OK;
}

int check_overflow(SphHandle *h)
{
    // ds_host must be fresh
    if (h->ds_host->overflow) {
        (void)hipMemsetAsync(&h->ds->overflow, 0, sizeof(int), h->stream);
        if (h->ds_host->overflow & 2)
            return fail(h, SPH_E_OVERFLOW, "internal: a cell was missing from a workgroup's staging plan (run with SPH_STAGE=0 and report)");
        if (h->ds_host->overflow & 4)
            return fail(h, SPH_E_OVERFLOW, "a particle crossed a whole slab in one step (it left its slab and landed beyond the neighbour's): the one-message particle "
                                           "exchange assumes a fraction of a cell per step -- lower delta_time or use fewer, wider slabs");
        return fail(h, SPH_E_OVERFLOW, "neighbour list overflow: %d fluid / %d wall neighbours, capacity %d / %d (raise max_neighbors)",
                    h->ds_host->max_nbrs, h->ds_host->max_wall_nbrs, h->c.kmax, h->c.kbmax);
    }
    return SPH_OK;
}

PbfConsts pbf_consts(const SphHandle *h);


int stage_density(SphHandle *h)
{
    const Consts &c = h->c;
    hipStream_t s = h->stream;
    (void)kBlock;
    if (h->cfg.solver == SPH_SOLVER_PBF) {
        // compute_all_rho on a pbf solver: pbf_solver.py:166-174 overrides the two rho callbacks with the poly6 kernel.  The rho part of
        // the lambda sweep alone: pbf_lambda (aux), the (pos, lambda) scratch and the P / V roles stay as they are.
        ProfScope ps(h, K_B_LAMBDA);
        const PbfConsts k = pbf_consts(h);
        if (sweep_mode(h) == SWEEP_QUAD)
            hipLaunchKernelGGL(k_pbf_lambda<true>, dim3((unsigned)std::max(1, (c.n + 63) / 64)), dim3(kBlock), 0, s, c, k, h->P[h->pcur], h->WP, h->nl, h->nlb,
                               h->cnt, h->rho, h->aux, h->P[1 - h->pcur], 1);
        else
            hipLaunchKernelGGL(k_pbf_lambda<false>, grid_for(c.n), dim3(kBlock), 0, s, c, k, h->P[h->pcur], h->WP, h->nl, h->nlb, h->cnt, h->rho, h->aux,
                               h->P[1 - h->pcur], 1);
        HIP_TRY(h, hipGetLastError());
        h->density_valid = true;
        return SPH_OK;
    }
    const bool dfsph = h->cfg.solver == SPH_SOLVER_DFSPH;
    if (h->verlet) {      // wcsph under the relaxed arithmetic: Verlet lists hold pairs beyond h, only the clamped kernel functions may walk them
        ProfScope ps(h, K_W_DENSITY);
        hipLaunchKernelGGL(k_wcsph_density_rx, grid_for(c.n), dim3(kBlock), 0, s, c, h->P[h->pcur], h->V[h->vcur], h->WP, h->nl, h->nlb, h->cnt,
                           h->rho, h->aux, h->P[1 - h->pcur], h->V[1 - h->vcur], h->wall_grad, h->ds, 1);
        h->pcur ^= 1; h->vcur ^= 1;                         // P = (pos, rho), V = (vel, p / rho^2)
        HIP_TRY(h, hipGetLastError());
        h->density_valid = true;
        return SPH_OK;
    }
    if (dfsph) {
        // DFSPH buffer roles for the whole step: P[pcur] = sorted positions (never written until the integrator),
        // P[1-pcur] = (pos, k/rho) scratch rewritten by D1/D3/D6, V[vcur] and VA[0] updated in place (a thread only ever
        // writes its own element and no sweep reads the array it writes from its neighbours)
        ProfScope ps(h, K_D_DENSITY_ALPHA);
        const bool split = rx_split(h);
        if (use_relaxed(h))
            hipLaunchKernelGGL(k_density_rx, grid_for(c.n), dim3(kBlock), sweep_lds(h, sizeof(float4)), s, c, h->P[h->pcur], h->V[h->vcur], h->wall_grad, h->wall_gsq,
                               h->nl, h->cnt, h->warm[h->wcur], h->ds, h->rho, h->aux, h->V[h->vcur], h->stage_src, h->stage_cnt, h->krho,
                               split ? TilePhase{h->tile_order, h->nblocks, 2} : TilePhase{nullptr, 0, 0}, h->id[h->icur], split ? h->rho_orig : (float *)nullptr);
        if (!use_relaxed(h) || split)
        SPH_LAUNCH_RMX(k_density, true, rigid_coupled(h), sweep_mode(h), relaxed_unstaged(h), c.n, sweep_lds(h, sizeof(float4)), s, c, h->P[h->pcur], h->V[h->vcur], h->WP, h->nl, h->nlb,
                      h->cnt, h->warm[h->wcur], h->ds, h->rho, h->aux, h->P[1 - h->pcur], h->V[h->vcur], rigid_view_or_none(h), h->id[h->icur],
                      h->rho_orig, h->stage_src, h->stage_cnt, h->krho, wall_cache(h), split ? TilePhase{h->tile_order, h->nblocks, 1} : TilePhase{nullptr, 0, 0});
    } else {
        ProfScope ps(h, K_W_DENSITY);
        SPH_LAUNCH_RM(k_density, false, rigid_coupled(h), sweep_mode(h), c.n, sweep_lds(h, sizeof(float4)), s, c, h->P[h->pcur], h->V[h->vcur], h->WP, h->nl, h->nlb,
                      h->cnt, (const float *)nullptr, h->ds, h->rho, h->aux, h->P[1 - h->pcur], h->V[1 - h->vcur], rigid_view_or_none(h), h->id[h->icur],
                      h->rho_orig, h->stage_src, h->stage_cnt, h->krho, (float4 *)nullptr);
        h->pcur ^= 1; h->vcur ^= 1;   // P = (pos, rho), V = (vel, p/rho^2)
    }
    HIP_TRY(h, hipGetLastError());
    if (h->slab && dfsph && rigid_coupled(h)) {      // fluid densities by original id < Nr (the viscosity quirk), summed over the owners
        ProfScope ps(h, K_RIGID);
        HIP_TRY(h, hipMemsetAsync(h->red_dev, 0, sizeof(double) * (size_t)h->Nr, s));
        hipLaunchKernelGGL(k_collect_by_id, grid_for(c.n), dim3(kBlock), 0, s, c.n, h->id[h->icur], (const float4 *)nullptr, h->rho, h->Nr, h->red_dev);
        int rc = slab_allreduce_stream(h, h->Nr, 0);
        if (rc) return rc;
        hipLaunchKernelGGL(k_spread_by_id, grid_for(h->Nr), dim3(kBlock), 0, s, h->Nr, h->red_dev, (float4 *)nullptr, h->rho_orig);
    }
    if (h->slab && dfsph && h->geom.layers == 2) {
        // two ghost columns: the inner one computed rho, alpha and its warm-start k / rho itself (same inputs, same order as on its owner); the
        // outer one is only ever read as a neighbour of the warm start: k / rho
        int rc = slab_exchange_field(h, 0, h->P[1 - h->pcur], nullptr, nullptr);
        if (rc) return rc;
    } else if (h->slab) {   // ghosts need (k/rho, rho) resp. (rho, p/rho^2) from their owners
        const bool ps = is_pressure_solver(h);            // their sweeps read rho[] of the neighbours: mode 3 fills it from P.w
        int rc = slab_exchange_field(h, ps ? 3 : 2, dfsph ? h->P[1 - h->pcur] : h->P[h->pcur], h->V[h->vcur], (dfsph || ps) ? h->rho : nullptr);
        if (rc) return rc;
    }
    h->density_valid = true;
    return SPH_OK;
}

// force of the fluid on the body for wcsph (S = pressure) / pcisph / iisph (PB.w = press_iter / p_iter); see k_rigid_force_p
template <int MODE>
void launch_rigid_force_p(SphHandle *h, const float4 *P, const float4 *PB, int gate)
{
    ProfScope ps(h, K_RIGID);
    hipLaunchKernelGGL(k_rigid_force_p<MODE>, grid_for(h->Nr), dim3(kBlock), 0, h->stream, h->c, h->Nr, h->RPs, h->rid, P, h->rnl, h->rcnt, h->rho,
                       h->aux, PB, h->ds, h->rforce, gate);
}

int step_wcsph_once(SphHandle *h)
{
    int rc;
    h->simulate_cnt += 1;                                   // solver_base.py:137
    h->comm_stat[6] += 1;
    if ((rc = stage_sort_and_lists(h))) return rc;          // :139-141
    if (h->verlet) {                                        // the relaxed arithmetic: two kernels over the Verlet lists (sph_relaxed_kernels.h)
        const Consts &cv = h->c;
        if ((rc = stage_density(h))) return rc;             // pressure_phase, wcsph_solver.py:32-38
        {
            ProfScope ps(h, K_W_FORCE);                     // + kinematic_phase :40-63
            hipLaunchKernelGGL(k_wcsph_force_rx, grid_for(cv.n), dim3(kBlock), 0, h->stream, cv, h->dt_wcsph, h->P[h->pcur], h->V[h->vcur], h->nl, h->cnt,
                               h->wall_grad, h->x0, h->P[1 - h->pcur], h->V[1 - h->vcur], h->VA[0], h->ds);
            h->pcur ^= 1; h->vcur ^= 1;
        }
        HIP_TRY(h, hipGetLastError());
        h->nl_valid = false;
        h->density_valid = false;
        return SPH_OK;
    }
    if ((rc = stage_density(h))) return rc;                 // wcsph_solver.py:34-35
    const Consts &c = h->c;
    if (rigid_coupled(h)) launch_rigid_force_p<RF_WCSPH>(h, h->P[h->pcur], nullptr, GATE_NONE);   // wcsph_solver.py:127, positions of this step
    {
        ProfScope ps(h, K_W_FORCE);                          // wcsph_solver.py:36-38 + kinematic_phase :40-63
        const bool quad = sweep_mode(h) == SWEEP_QUAD;
        const dim3 gf = quad ? dim3((unsigned)std::max(1, (c.n + 63) / 64)) : grid_for(c.n);
#define SPH_WFORCE(R, Q, RV) hipLaunchKernelGGL((k_wcsph_force<R, Q>), gf, dim3(kBlock), 0, h->stream, c, h->dt_wcsph, h->P[h->pcur], h->V[h->vcur], h->WP, \
                                                h->nl, h->nlb, h->cnt, h->aux, h->P[1 - h->pcur], h->V[1 - h->vcur], h->VA[0], RV)
        if (rigid_coupled(h)) { if (quad) SPH_WFORCE(true, true, rigid_view(h)); else SPH_WFORCE(true, false, rigid_view(h)); }
        else { if (quad) SPH_WFORCE(false, true, RigidView()); else SPH_WFORCE(false, false, RigidView()); }
#undef SPH_WFORCE
        h->pcur ^= 1; h->vcur ^= 1;
    }
    HIP_TRY(h, hipGetLastError());
    h->nl_valid = false;
    h->density_valid = false;
    return SPH_OK;
}

// every slab must see a list overflow at the same point, or the others would wait in a collective forever
int check_overflow_all(SphHandle *h, bool reduced_on_device = false)
{
    int ovf = h->ds_host->overflow;
    if (h->slab && reduced_on_device) {          // dfsph device loops: the flags of all slabs came with the density loop's first reduction
        if (h->ds_host->overflow_any && !ovf) return fail(h, SPH_E_OVERFLOW, "list overflow or exchange failure on another slab");
    } else if (h->slab) {
        double v[1] = {(double)ovf};
        int rc = slab_allreduce_host(h, v, 1, 1);
        if (rc) return rc;
        if (v[0] > 0.0 && !ovf) return fail(h, SPH_E_OVERFLOW, "list overflow or exchange failure on another slab (flags %d)", (int)v[0]);
    }
    return check_overflow(h);
}

// ---- DFSPH launch helpers (buffer roles: see stage_density) --------------------------------------------------
// tiles of the density loop whose inputs did not change are not recomputed (staged dfsph handles)
inline bool tile_skip(const SphHandle *h) { return h->wave_dirty && h->staged; }
// the tolerance-grade sweeps cover kr_split handles (single GPU, staged, 16-bit lists, no rigid entries); all others stay exact
inline TilePhase tile_phase(const SphHandle *h, int phase)
{
    TilePhase tp{h->tile_order, h->nblocks, phase};
    // (the un-split launches of the density loop; the overlapped slab protocol's split launches keep their edge-first order)
    if (phase == 0 && h->dens_order && tile_skip(h)) { tp.hot = h->dens_hot; tp.sparse = h->dens_sparse ? h->dens_order : nullptr; }
    return tp;
}
void launch_div_residual(SphHandle *h, int gate, int phase = 0, SpecUndo un = SpecUndo{nullptr, nullptr, nullptr, nullptr, 0}, hipStream_t st = nullptr)          // derivative_iter_all_rho sweep, dfsph_solver.py:252-277
{
    const Consts &c = h->c;
    if (!st) st = h->stream;
    ProfScope ps(h, K_D_DIV_RESIDUAL, st);
    const bool split = rx_split(h);
    const TilePhase tp = split ? TilePhase{h->tile_order, h->nblocks, 1} : tile_phase(h, phase);
    if (use_relaxed(h)) {
        const TilePhase tpr = split ? TilePhase{h->tile_order, h->nblocks, 2} : tp;
        hipLaunchKernelGGL(k_residual_rx<false>, grid_for(c.n), dim3(kBlock), sweep_lds(h, sizeof(float4) + sizeof(float2)), st, c, h->P[h->pcur], h->V[h->vcur],
                           h->wall_grad, h->nl, h->cnt, h->rho, h->aux, h->ds, h->drho, h->psum, h->pcnt, gate, h->stage_src, h->stage_cnt, h->krho, (const int *)nullptr, (const unsigned char *)nullptr, 1, tpr, un);
        if (!split) return;
    }
    SPH_LAUNCH_RMX(k_residual, false, rigid_coupled(h), sweep_mode(h), relaxed_unstaged(h), c.n, sweep_lds(h, sizeof(float4) + sizeof(float2)), st, c,
                  h->P[h->pcur], h->V[h->vcur], h->WP, h->nl, h->nlb, h->cnt, h->rho, h->aux, h->ds, h->drho, h->P[1 - h->pcur], h->psum, h->pcnt,
                  rigid_view_or_none(h), h->ncount, gate, h->stage_src, h->stage_cnt, h->krho, (const int *)nullptr, (const unsigned char *)nullptr, 1,
                  (const float4 *)wall_cache(h), tp, un);
}

// ride_mode >= 0 (one GPU, fin_rides): workgroup 0 of the launch takes the loop decision of evaluation `ride_eval` -- the residual sweep enqueued
// before this one -- and the grid is one workgroup larger (fin_ride_block in sph_kernels.h)
inline bool fin_rides(const SphHandle *h) { return !h->slab && h->spec_v != nullptr && !rx_split(h); }
template <int MODE>
void launch_correct(SphHandle *h, int kid, const float *src, float4 *V, int gate, SpecSave sv = SpecSave{nullptr, nullptr}, int ride_mode = -1, int ride_eval = -1)
{
    const Consts &c = h->c;
    ProfScope ps(h, kid);
    int *wdirty = (MODE == CORR_DENS && tile_skip(h) && !h->tune_all) ? h->wave_dirty : nullptr;      // change propagation in the density loop
    const bool split = rx_split(h);
    const bool ride = ride_mode >= 0;
    const FinRide fr = ride ? FinRide{h->psum, h->pcnt, h->ds, h->nblocks, ride_mode, partial_group(h), partial_count(h), ride_eval} : kNoRide;
    TilePhase tp0 = tile_phase(h, 0);
    tp0.shift = ride ? 1 : 0;
    const int n_grid = c.n + (ride ? (sweep_mode(h) == SWEEP_QUAD ? 64 : kBlock) : 0);           // one more workgroup
    if (use_relaxed(h)) {
        hipLaunchKernelGGL(k_correct_rx<MODE>, grid_for(split ? c.n : n_grid), dim3(kBlock), sweep_lds(h, sizeof(float4)), h->stream, c, h->P[h->pcur], h->wall_grad, h->nl, h->cnt,
                           h->rho, h->aux, src, h->warm[h->wcur], h->ds, V, V, gate, h->stage_src, h->stage_cnt, h->krho, wdirty, h->changed8,
                           split ? TilePhase{h->tile_order, h->nblocks, 2} : tp0, sv, split ? kNoRide : fr);
        if (!split) return;
    }
    SPH_LAUNCH_RMX(k_correct, MODE, rigid_coupled(h), sweep_mode(h), relaxed_unstaged(h), split ? c.n : n_grid, sweep_lds(h, sizeof(float4)), h->stream, c,
                  c.kr_split ? h->P[h->pcur] : h->P[1 - h->pcur], h->WP,
                  h->nl, h->nlb, h->cnt, h->rho, h->aux, src, h->warm[h->wcur], h->ds, V, V, rigid_view_or_none(h), gate, h->stage_src, h->stage_cnt, h->krho, wdirty, h->changed8,
                  (const float4 *)wall_cache(h), split ? TilePhase{h->tile_order, h->nblocks, 1} : tp0, sv, split ? kNoRide : fr);
}

void launch_dens_residual(SphHandle *h, int gate, int phase = 0, hipStream_t st = nullptr)          // compute_all_rho_adv sweep, dfsph_solver.py:124-141
{
    const Consts &c = h->c;
    if (!st) st = h->stream;
    ProfScope ps(h, K_D_DENS_RESIDUAL, st);
    const bool split = rx_split(h);
    const TilePhase tp = split ? TilePhase{h->tile_order, h->nblocks, 1} : tile_phase(h, phase);
    const int *wdirty = tile_skip(h) ? h->wave_dirty : nullptr;
    const int force_all = (h->dens_first || h->tune_all) ? 1 : 0;      // the first compute_all_rho_adv of a step computes every tile
    if (phase != 1) h->dens_first = false;                              // (an edge launch is followed by the interior launch of the same sweep)
    if (use_relaxed(h)) {
        const TilePhase tpr = split ? TilePhase{h->tile_order, h->nblocks, 2} : tp;
        hipLaunchKernelGGL(k_residual_rx<true>, grid_for(c.n), dim3(kBlock), sweep_lds(h, sizeof(float4) + sizeof(float2)), st, c, h->P[h->pcur], h->VA[0],
                           h->wall_grad, h->nl, h->cnt, h->rho, h->aux, h->ds, h->rho_adv, h->psum, h->pcnt, gate, h->stage_src, h->stage_cnt, h->krho, wdirty, h->changed8, force_all, tpr);
        if (!split) return;
    }
    SPH_LAUNCH_RMX(k_residual, true, rigid_coupled(h), sweep_mode(h), relaxed_unstaged(h), c.n, sweep_lds(h, sizeof(float4) + sizeof(float2)), st, c,
                  h->P[h->pcur], h->VA[0], h->WP, h->nl, h->nlb, h->cnt, h->rho, h->aux, h->ds, h->rho_adv, h->P[1 - h->pcur], h->psum, h->pcnt,
                  rigid_view_or_none(h), h->ncount, gate, h->stage_src, h->stage_cnt, h->krho, wdirty, h->changed8, force_all, (const float4 *)wall_cache(h), tp);
}

// The same in two halves, for the handles that hide the all-reduce (step_dfsph_device_loops): this slab's (sum, count) on the handle's stream ...
int launch_finalize_reduce(SphHandle *h, int mode)
{
    ProfScope ps(h, K_FINALIZE);
    hipLaunchKernelGGL(k_finalize_mean, dim3(1), dim3(kFinBlock), 0, h->stream, h->psum, h->pcnt, h->nblocks, h->ds, mode, FINP_REDUCE, h->red_dev, partial_group(h), partial_count(h));
    HIP_TRY(h, hipEventRecord(h->ev_red, h->stream));
    return SPH_OK;
}
// ... and the all-reduce + the decision of evaluation `eval` on the third stream; whoever needs the decision waits for ev_dec
int launch_finalize_decide(SphHandle *h, int mode, int eval)
{
    hipStream_t r = h->rstream;
    HIP_TRY(h, hipStreamWaitEvent(r, h->ev_red, 0));
    int rc = slab_allreduce_stream(h, mode == FIN_DENS ? 3 : 2, 0, r);       // (the density loop's carries the overflow flags, k_finalize_mean)
    if (rc) return rc;
    {
        ProfScope ps(h, K_FINALIZE, r);
        hipLaunchKernelGGL(k_finalize_mean, dim3(1), dim3(kFinBlock), 0, r, h->psum, h->pcnt, h->nblocks, h->ds, mode, FINP_DECIDE, h->red_dev, partial_group(h), partial_count(h), eval);
    }
    HIP_TRY(h, hipEventRecord(h->ev_dec, r));
    return SPH_OK;
}
int launch_finalize(SphHandle *h, int mode)
{
    if (slab_async(h)) {       // this slab's (sum, count) -> all-reduce over the slabs -> the loop decision, all on the stream
        {
            ProfScope ps(h, K_FINALIZE);
            hipLaunchKernelGGL(k_finalize_mean, dim3(1), dim3(kFinBlock), 0, h->stream, h->psum, h->pcnt, h->nblocks, h->ds, mode, FINP_REDUCE, h->red_dev, partial_group(h), partial_count(h));
        }
        int rc = slab_allreduce_stream(h, mode == FIN_DENS ? 3 : 2, 0);
        if (rc) return rc;
        ProfScope ps(h, K_FINALIZE);
        hipLaunchKernelGGL(k_finalize_mean, dim3(1), dim3(kFinBlock), 0, h->stream, h->psum, h->pcnt, h->nblocks, h->ds, mode, FINP_DECIDE, h->red_dev, partial_group(h), partial_count(h));
        return SPH_OK;
    }
    ProfScope ps(h, K_FINALIZE);
    hipLaunchKernelGGL(k_finalize_mean, dim3(1), dim3(kFinBlock), 0, h->stream, h->psum, h->pcnt, h->nblocks, h->ds, mode, FINP_ALL, (double *)nullptr, partial_group(h), partial_count(h));
    return SPH_OK;
}

void launch_rigid_force(SphHandle *h, int gate)            // dfsph_solver.py:212
{
    const Consts &c = h->c;
    ProfScope ps(h, K_RIGID);
    hipLaunchKernelGGL(k_rigid_force, grid_for(h->Nr), dim3(kBlock), 0, h->stream, c, h->Nr, h->RPs, h->rid, h->P[h->pcur], h->rnl, h->rcnt, h->rho,
                       h->rho_adv, h->aux, h->ds, h->rforce, gate, h->slab ? h->geom.x_lo : -0x7fffffff, h->slab ? h->geom.x_hi : 0x7fffffff);
}

// The in-order protocol of a two-column slab handle: the residual's refresh AND its mean in four enqueues instead of six -- [pack + this slab's
// (sum, count)] -> the halo transfer -> the all-reduce -> [unpack + the loop decision] (k_pack_resid_reduce / k_unpack_resid_decide).
int slab_exchange_resid_and_finalize(SphHandle *h, bool dens, float *val, int mode)
{
    hipStream_t s = h->stream;
    const int nsl = h->edge_n[1][0] + h->edge_n[1][1], nsr = h->edge_n[2][0] + h->edge_n[2][1];
    const int nrl = h->edge_n[0][0] + h->edge_n[0][1], nrr = h->edge_n[3][0] + h->edge_n[3][1];
    float *S = h->c.kr_split ? h->krho : nullptr;
    float4 *P = h->P[1 - h->pcur];
    // native transport: this slab's (sum, count, flags) go to every slab in the halo's own group of transfers and the decision sums the gathered
    // triples in slab order -- ONE start-up latency per solver iteration where the halo and an all-reduce paid two (what a step costs on a link
    // that is not free: profiles/r04/loopback/link_latency_sweep.txt)
    const bool gather = h->native && h->gath_dev && h->opt_gather;
    if (gather) h->comm_stat[4] += 1;           // (counted with the all-reduces it replaces)
    {
        ProfScope ps(h, K_SLAB);
        const ResidLists L{h->edge_list[1], nsl, h->edge_n[1][0], (float *)h->dsend[0], h->edge_list[2], nsr, h->edge_n[2][0], (float *)h->dsend[1]};
        hipLaunchKernelGGL(k_pack_resid_reduce, dim3((unsigned)((nsl + nsr + kFinBlock - 1) / kFinBlock + 1)), dim3(kFinBlock), 0, s, L, val, P, S,
                           h->psum, h->pcnt, h->nblocks, h->ds, mode, gather ? h->gath_dev + 4 * h->slab_rank : h->red_dev, partial_group(h), partial_count(h));
    }
    int rc = slab_xfer(h, 4 * (size_t)nsl, 4 * (size_t)nsr, 4 * (size_t)nrl, 4 * (size_t)nrr, s, gather ? 3 : 0);
    if (rc) return rc;
    if (!gather && (rc = slab_allreduce_stream(h, mode == FIN_DENS ? 3 : 2, 0))) return rc;
    {
        ProfScope ps(h, K_SLAB);
        const ResidLists L{h->edge_list[0], nrl, h->edge_n[0][0], (float *)h->drecv[0], h->edge_list[3], nrr, h->edge_n[3][0], (float *)h->drecv[1]};
        hipLaunchKernelGGL(k_unpack_resid_decide, dim3((unsigned)((nrl + nrr + kFinBlock - 1) / kFinBlock + 1)), dim3(kFinBlock), 0, s, h->c, L, dens ? 1 : 0, h->aux, h->rho,
                           val, P, S, h->psum, h->pcnt, h->nblocks, h->ds, mode, gather ? h->gath_dev : h->red_dev, partial_group(h), partial_count(h), gather ? h->nslab : 0);
    }
    HIP_TRY(h, hipGetLastError());
    return SPH_OK;
}

// host-driven evaluation of a mean (sharded runs: the (sum, count) pair is all-reduced over the slabs)
int reduce_mean_host(SphHandle *h, float dflt, float *mean)
{
    hipLaunchKernelGGL(k_finalize_mean, dim3(1), dim3(kFinBlock), 0, h->stream, h->psum, h->pcnt, h->nblocks, h->ds, (int)FIN_PLAIN, (int)FINP_ALL, (double *)nullptr, partial_group(h), partial_count(h));
    int rc = read_scalars(h);
    if (rc) return rc;
    double v[2] = {h->ds_host->sum, (double)h->ds_host->cnt};
    if (h->slab && (rc = slab_allreduce_host(h, v, 2, 0))) return rc;
    *mean = v[1] > 0.0 ? (float)(v[0] / v[1]) : dflt;       // dfsph_solver.py:148-149, 278-279
    return SPH_OK;
}

// ext forces, v*, CFL dt                                    dfsph_solver.py:91-122
int dfsph_ext_and_dt(SphHandle *h)
{
    const Consts &c = h->c;
    hipStream_t s = h->stream;
    const dim3 b(kBlock);
    int rc;
    {
        ProfScope ps(h, K_D_EXT);
        const bool split = rx_split(h);
        if (use_relaxed(h))
            hipLaunchKernelGGL(k_dfsph_ext_rx, grid_for(c.n), b, sweep_lds(h, sizeof(float4) + sizeof(uint32_t)), s, c, h->P[h->pcur], h->V[h->vcur], h->nl, h->cnt, h->ds,
                               h->VA[0], h->pmax, h->stage_src, h->stage_cnt, split ? TilePhase{h->tile_order, h->nblocks, 2} : TilePhase{nullptr, 0, 0});
        if (!use_relaxed(h) || split)
        SPH_LAUNCH_RMXQ0(k_dfsph_ext, rigid_coupled(h), sweep_mode(h), relaxed_unstaged(h), c.n, sweep_lds(h, sizeof(float4) + sizeof(uint32_t)), s, c, h->P[h->pcur], h->V[h->vcur], h->nl,
                       h->cnt, h->ds, h->VA[0], h->pmax, rigid_view_or_none(h), h->stage_src, h->stage_cnt, split ? TilePhase{h->tile_order, h->nblocks, 1} : TilePhase{nullptr, 0, 0});
        if (h->rigid) {   // max_rigid_vel, :104-110 (loops over the rigid particles whether or not the body is active)
            RigidBodyState st = rigid_state(h, nullptr, nullptr);
            for (int a = 0; a < 3; ++a) st.omega[a] = h->r_omega[a];
            const float vn = sqrtf((h->r_vel[0] * h->r_vel[0] + h->r_vel[1] * h->r_vel[1]) + h->r_vel[2] * h->r_vel[2]);
            hipLaunchKernelGGL(k_rigid_vmax, rigid_parts_grid(h), b, 0, s, h->Nr, h->RPos, st, vn, h->ds, h->rvmax_part, 0);
            hipLaunchKernelGGL(k_rigid_vmax, dim3(1), b, 0, s, h->Nr, h->RPos, st, vn, h->ds, h->rvmax_part, (int)rigid_parts_grid(h).x);
        }
    }
    const bool async = slab_async(h);
    // native transport: this slab's max |v*| goes to every slab in the group of transfers that refreshes v* on the ghosts (one group instead of a
    // group and an all-reduce, as in the solver loops)
    const bool gather = async && h->native && h->gath_dev && h->opt_gather;
    {
        ProfScope ps(h, K_FINALIZE);
        hipLaunchKernelGGL(k_finalize_max, dim3(1), b, 0, s, h->pmax, partial_count(h), h->ds, gather ? h->gath_dev + 4 * h->slab_rank : async ? h->red_dev : (double *)nullptr,
                           c, h->slab ? 0 : 1, h->pending_div);
        h->pending_div = kNoRide;
    }
    if (!h->slab) return SPH_OK;          // (the maximum's thread applied the CFL rule: :112-119)
    if (h->slab) {
        if ((rc = slab_exchange_field(h, 1, nullptr, h->VA[0], nullptr, 1, gather ? 1 : 0))) return rc;     // v* of the column next to the cut (all the density residual reads)
        if (gather) {
            h->comm_stat[4] += 1;
        } else if (async) {
            if ((rc = slab_allreduce_stream(h, 1, 1))) return rc;          // max |v*| over all slabs, stays on the device
        } else {
            if ((rc = read_scalars(h))) return rc;
            double v[1] = {(double)h->ds_host->vmax};
            if ((rc = slab_allreduce_host(h, v, 1, 1))) return rc;
            h->ds_host->vmax = (float)v[0];
            HIP_TRY(h, hipMemcpyAsync(&h->ds->vmax, &h->ds_host->vmax, sizeof(float), hipMemcpyHostToDevice, s));
        }
    }
    {
        ProfScope ps(h, K_FINALIZE);
        hipLaunchKernelGGL(k_apply_dt, dim3(1), dim3(1), 0, s, c, h->ds, gather ? h->gath_dev : async ? h->red_dev : (const double *)nullptr, gather ? h->nslab : 0);   // :112-119
    }
    return SPH_OK;
}

int dfsph_integrate(SphHandle *h)
{
    const Consts &c = h->c;
    ProfScope ps(h, K_D_INTEGRATE);                          // compute_all_position :235-250
    // new positions go to the scratch buffer (nobody reads it any more), new velocities in place
    hipLaunchKernelGGL(k_dfsph_integrate, grid_for(c.n), dim3(kBlock), 0, h->stream, c, h->P[h->pcur], h->VA[0], h->ds, h->P[1 - h->pcur],
                       h->V[h->vcur]);
    h->pcur ^= 1;
    HIP_TRY(h, hipGetLastError());
    h->nl_valid = false;
    h->density_valid = false;
    return SPH_OK;
}

// One DFSPH step on a single GPU: the reference's two host loops run on the device (k_finalize_mean applies their
// conditions; kernels of iterations that would not run exit at once), the host only reads the control block back
// once per chunk of iterations.
int step_dfsph_device_loops(SphHandle *h, SphStepStats *st)
{
    int rc;
    hipStream_t s = h->stream;
    const int cap = h->cfg.max_density_iters > 0 ? h->cfg.max_density_iters : 100;
    // (one GPU with the warm start on: workgroup 0 of the warm-start launch resets the loop state instead -- FIN_BEGIN below -- one launch less)
    const bool begin_rides = fin_rides(h) && h->p.warm_start;
    if (!begin_rides) hipLaunchKernelGGL(k_ctrl_begin, dim3(1), dim3(1), 0, s, h->ds, cap);
    h->dens_first = true;
    // (one GPU: dens_sparse stays -- the region of the scene that keeps the density loop busy moves slowly, last step's order serves the loop's first
    // launches; a slab's tiles change with every particle exchange)
    if (h->slab) h->dens_sparse = false;
    // ---- correct_divergence_error, dfsph_solver.py:393-416 ----
    // On a slab handle every sweep whose output the neighbours read is followed by the refresh of that field on the ghosts (enqueued,
    // not waited for, with a stream-ordered transport); gated sweeps still take part in the exchanges so that all slabs issue the same
    // sequence of transfers (they re-send unchanged values).
    // Two ghost columns (slab_ghost_layers = 2, the dfsph default): the inner ghost column runs the correction sweeps itself -- its neighbours
    // are all resident, its inputs are the owner's, so are its results -- and a solver iteration needs ONE refresh, the residual's
    // (slab_exchange_resid); with slab_can_overlap the residual sweep runs its edge tiles first and its interior tiles under that transfer.
    const bool two = h->slab && h->geom.layers == 2;
    const bool ovl = two && slab_can_overlap(h);
    auto ghosts_v = [&](float4 *V) -> int { return (h->slab && !two) ? slab_exchange_field(h, 1, nullptr, V, nullptr) : SPH_OK; };
    // a residual sweep and the refresh of what it produced on the ghosts
    // reduce_mode >= 0 (the handles that hide the all-reduce): this slab's (sum, count) is reduced right behind the sweep's last tile -- in front of
    // the halo's enqueue and of the wait for it, which only the NEXT sweep needs
    auto residual_sweep = [&](bool dens, int gate, SpecUndo un = SpecUndo{nullptr, nullptr, nullptr, nullptr, 0}, int reduce_mode = -1) -> int {
        int r = SPH_OK;
        if (ovl) {
            if (dens) launch_dens_residual(h, gate, 1); else launch_div_residual(h, gate, 1, un);
            HIP_TRY(h, hipEventRecord(h->ev_edge, s));
            if (dens) launch_dens_residual(h, gate, 2); else launch_div_residual(h, gate, 2, un);       // enqueued before the host turns to the transfer
            if (reduce_mode >= 0 && (r = launch_finalize_reduce(h, reduce_mode))) return r;
            if ((r = slab_exchange_resid(h, dens, dens ? h->rho_adv : h->drho, true))) return r;
            HIP_TRY(h, hipStreamWaitEvent(s, h->ev_halo, 0));                                       // the next sweep reads the ghosts
            return SPH_OK;
        }
        if (dens) launch_dens_residual(h, gate); else launch_div_residual(h, gate, 0, un);
        if (two) return slab_exchange_resid(h, dens, dens ? h->rho_adv : h->drho, false);
        return h->slab ? slab_exchange_field(h, 0, h->P[1 - h->pcur], nullptr, nullptr) : SPH_OK;
    };
    // ... followed by the loop decision in a launch of its own (k_finalize_mean; around the all-reduce on slabs)
    auto residual = [&](bool dens, int gate, int fin_mode) -> int {
        int r = SPH_OK;
        if (two && !ovl && slab_async(h)) {          // in order: the small launches of the refresh and of the mean ride together
            if (dens) launch_dens_residual(h, gate); else launch_div_residual(h, gate);
            return slab_exchange_resid_and_finalize(h, dens, dens ? h->rho_adv : h->drho, fin_mode);
        }
        if ((r = residual_sweep(dens, gate))) return r;
        return launch_finalize(h, fin_mode);
    };
    // Hiding the all-reduce (two-column handles whose halo may run on its own stream, `ovl`).  What a solver iteration still waited for was the
    // two-double all-reduce of its residual, because the decision it feeds gates the next sweep.  The reduction and the decision kernel now run on a
    // third stream while the NEXT sweep runs on the handle's:
    //   density loop     that sweep is the correction D7 of the SAME iteration, which the reference runs whatever the new mean says
    //                    (dfsph_solver.py:227-231: the condition is tested at the loop's head): no speculation at all;
    //   divergence loop  that sweep is the correction D4 of the NEXT iteration (:402-408), which the decision may cancel: it runs ahead, keeps what
    //                    it overwrote (SpecSave), and if the decision closed the loop the following residual launch -- gated off -- puts it back
    //                    (SpecUndo).  Wrong at most once per step; never in a loop that runs into its cap of 15.
    // A sweep that is enqueued behind evaluation e's reduction must not read the gate evaluation e is about to write: it reads the decision of
    // e - 1 from DevScalars.gate_hist[(e - 1) & 1].  Bit-identical to the plain order by construction (tests/test_slab_gpu.py).
    const bool spec = ovl && slab_async(h) && h->rstream;
    const int max_div = h->p.max_iteration_density_divergence;                       // :24 (15)
    if (h->p.warm_start) {
        launch_correct<CORR_WARM>(h, K_D_WARM, nullptr, h->V[h->vcur], GATE_NONE, SpecSave{nullptr, nullptr}, begin_rides ? FIN_BEGIN : -1, cap);   // :396-397
        if ((rc = ghosts_v(h->V[h->vcur]))) return rc;
    }
    // One GPU: the same reordering without a second stream -- the decision of evaluation e is taken by workgroup 0 of the correction launch that
    // runs ahead of it (launch_correct's ride_mode / fin_ride_block): no single-workgroup launch between two sweeps any more.
    const bool ride = fin_rides(h);
    if (ride) {
        launch_div_residual(h, GATE_NONE);                                                                                   // :398, evaluation 1
        for (int e = 1; e <= max_div; ++e) {
            launch_correct<CORR_DIV>(h, K_D_DIV_CORRECT, h->drho, h->V[h->vcur], GATE_HIST0 + ((e - 1) & 1), SpecSave{h->spec_v, h->spec_w},
                                     e == 1 ? FIN_DIV_FIRST : FIN_DIV_LOOP, e);                                              // :402-405 + decision e
            launch_div_residual(h, GATE_DIV, 0, SpecUndo{h->V[h->vcur], h->spec_v, h->warm[h->wcur], h->spec_w, e});     // :408, evaluation e + 1
        }
        // the decision of the last evaluation has no correction launch to ride in: it is taken by the launch that reduces max |v*| (dfsph_ext_and_dt;
        // the sweep in between, D5, writes other partials and reads no loop state)
        h->pending_div = FinRide{h->psum, h->pcnt, h->ds, h->nblocks, max_div == 0 ? (int)FIN_DIV_FIRST : (int)FIN_DIV_LOOP, partial_group(h), partial_count(h), max_div + 1};
    } else if (spec) {
        if ((rc = residual_sweep(false, GATE_NONE, SpecUndo{nullptr, nullptr, nullptr, nullptr, 0}, FIN_DIV_FIRST))) return rc;     // :398, evaluation 1
        for (int e = 1; e <= max_div; ++e) {
            // the correction of evaluation e first (the GPU works on it while the host may block in a synchronous all-reduce) ...
            launch_correct<CORR_DIV>(h, K_D_DIV_CORRECT, h->drho, h->V[h->vcur], GATE_HIST0 + ((e - 1) & 1), SpecSave{h->spec_v, h->spec_w});   // :402-405
            // ... then evaluation e's reduction and decision on the third stream
            if ((rc = launch_finalize_decide(h, e == 1 ? FIN_DIV_FIRST : FIN_DIV_LOOP, e))) return rc;
            HIP_TRY(h, hipStreamWaitEvent(s, h->ev_dec, 0));
            if ((rc = residual_sweep(false, GATE_DIV, SpecUndo{h->V[h->vcur], h->spec_v, h->warm[h->wcur], h->spec_w, e}, FIN_DIV_LOOP))) return rc;   // :408, evaluation e + 1
        }
        if ((rc = launch_finalize_decide(h, max_div == 0 ? FIN_DIV_FIRST : FIN_DIV_LOOP, max_div + 1))) return rc;
        HIP_TRY(h, hipStreamWaitEvent(s, h->ev_dec, 0));
    } else {
    if ((rc = residual(false, GATE_NONE, FIN_DIV_FIRST))) return rc;                 // :398
    // all max_iteration_density_divergence (15) possible iterations are enqueued at once: the ones the reference's loop would not run exit at
    // their first instruction, and the host does not need the outcome before the density loop's first read-back
    for (int done = 0; done < max_div; ++done) {
        launch_correct<CORR_DIV>(h, K_D_DIV_CORRECT, h->drho, h->V[h->vcur], GATE_DIV);   // :402-405
        if ((rc = ghosts_v(h->V[h->vcur]))) return rc;
        if ((rc = residual(false, GATE_DIV, FIN_DIV_LOOP))) return rc;                    // :408
    }
    }
    if ((rc = dfsph_ext_and_dt(h))) return rc;
    // ---- correct_density_error, :221-233: first chunk = last step's iteration count (it changes slowly), then two at a time ----
    bool first = true;
    int d = 0;                                                                       // evaluations of the density loop so far
    // behind the loop's second residual launch -- the first that skips unchanged tiles and notes which did not: the tiles that had work first, for the
    // rest of the loop's launches (TilePhase.sparse)
    auto order_working_tiles_first = [&]() {
        if (d != 2 || !h->dens_order || !tile_skip(h) || h->tune_all) return;
        ProfScope ps(h, K_BUILD_NL);
        hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, s, h->dens_hot, h->nblocks, h->dens_order);
        h->dens_sparse = true;
    };
    for (int chunk = std::max(2, h->last_iters);; chunk = 2) {
        for (int k = 0; k < chunk; ++k) {
            ++d;
            if (ride) {
                launch_dens_residual(h, GATE_DENS);                                                                          // :227, evaluation d
                order_working_tiles_first();
                // D7 of iteration d runs iff iteration d runs (the decision of evaluation d - 1; gate_hist starts open) and carries decision d
                launch_correct<CORR_DENS>(h, K_D_DENS_CORRECT, h->rho_adv, h->VA[0], GATE_HIST0 + ((d - 1) & 1), SpecSave{nullptr, nullptr}, FIN_DENS, d);   // :229
                if (rigid_coupled(h)) launch_rigid_force(h, GATE_HIST0 + ((d - 1) & 1));
                continue;
            }
            if (spec) {
                if ((rc = residual_sweep(true, GATE_DENS, SpecUndo{nullptr, nullptr, nullptr, nullptr, 0}, FIN_DENS))) return rc;      // :227, evaluation d
                order_working_tiles_first();
                // D7 of iteration d runs iff iteration d runs: the decision of evaluation d - 1 (gate_hist starts open)
                launch_correct<CORR_DENS>(h, K_D_DENS_CORRECT, h->rho_adv, h->VA[0], GATE_HIST0 + ((d - 1) & 1));   // :229
                if (rigid_coupled(h)) launch_rigid_force(h, GATE_HIST0 + ((d - 1) & 1));
                if ((rc = launch_finalize_decide(h, FIN_DENS, d))) return rc;
                HIP_TRY(h, hipStreamWaitEvent(s, h->ev_dec, 0));
                continue;
            }
            if ((rc = residual(true, GATE_DENS, FIN_DENS))) return rc;               // :227
            order_working_tiles_first();
            launch_correct<CORR_DENS>(h, K_D_DENS_CORRECT, h->rho_adv, h->VA[0], GATE_DENS_D7);   // :229
            if (rigid_coupled(h)) launch_rigid_force(h, GATE_DENS_D7);
            if ((rc = ghosts_v(h->VA[0]))) return rc;
        }
        if ((rc = read_scalars_fast(h))) return rc;
        if (first) {
            if ((rc = check_overflow_all(h, slab_async(h)))) return rc;     // first read-back of the step: list overflow?
            first = false;
        }
        if (!h->ds_host->dens_active) break;
    }
    h->last_iters = h->ds_host->dens_it;
    st->max_nbrs = h->ds_host->max_nbrs;
    st->max_wall_nbrs = h->ds_host->max_wall_nbrs;
    st->lost = h->ds_host->lost;
    st->n_div = h->ds_host->div_it;
    st->n_div_evals = h->ds_host->div_evals;
    st->div_first_err = h->ds_host->div_first;
    st->div_err = h->ds_host->div_err;
    st->n_dens = h->ds_host->dens_it;
    st->capped = h->ds_host->dens_capped;
    st->dens_err = (float)((double)h->ds_host->dens_avg - 1000.0);
    st->dt = h->ds_host->dt;
    return dfsph_integrate(h);
}

// The same step with the loops on the host (sharded runs: every residual needs an all-reduce and every sweep a ghost refresh)
int step_dfsph_host_loops(SphHandle *h, SphStepStats *st)
{
    int rc;
    h->dens_first = true;
    const bool two = h->slab && h->geom.layers == 2;       // (see step_dfsph_device_loops)
    auto ghosts_v = [&](float4 *V) -> int { return (h->slab && !two) ? slab_exchange_field(h, 1, nullptr, V, nullptr) : SPH_OK; };
    auto ghosts_k = [&](bool dens) -> int {
        if (two) return slab_exchange_resid(h, dens, dens ? h->rho_adv : h->drho, false);
        return h->slab ? slab_exchange_field(h, 0, h->P[1 - h->pcur], nullptr, nullptr) : SPH_OK;
    };
    if (h->p.warm_start) {
        launch_correct<CORR_WARM>(h, K_D_WARM, nullptr, h->V[h->vcur], GATE_NONE);   // :396-397
        if ((rc = ghosts_v(h->V[h->vcur]))) return rc;
    }
    float err = 0.f, past = 0.f;
    auto residual = [&](float *out) -> int {
        launch_div_residual(h, GATE_NONE);
        int r;
        if ((r = ghosts_k(false))) return r;
        return reduce_mean_host(h, 0.0f, out);
    };
    if ((rc = residual(&err))) return rc;                                            // :398
    if ((rc = check_overflow_all(h))) return rc;
    st->max_nbrs = h->ds_host->max_nbrs;
    st->max_wall_nbrs = h->ds_host->max_wall_nbrs;
    st->lost = h->ds_host->lost;
    st->n_div_evals = 1;
    st->div_first_err = err;
    int iter_cnt = 0;
    while ((iter_cnt < h->p.min_iteration_density_divergence || (double)err > h->p.density_divergence_threshold) && iter_cnt < h->p.max_iteration_density_divergence) {   // :400
        launch_correct<CORR_DIV>(h, K_D_DIV_CORRECT, h->drho, h->V[h->vcur], GATE_NONE);
        if ((rc = ghosts_v(h->V[h->vcur]))) return rc;
        past = err;
        if ((rc = residual(&err))) return rc;                                        // :408
        st->n_div_evals += 1;
        if (std::fabs((double)err - (double)past) < 1e-5) break;                     // :410-412
        iter_cnt += 1;
    }
    st->n_div = iter_cnt;
    st->div_err = err;
    if ((rc = dfsph_ext_and_dt(h))) return rc;
    const int cap = h->cfg.max_density_iters > 0 ? h->cfg.max_density_iters : 100;
    double rho_avg = INFINITY;
    int it = 0;
    while (it < h->p.min_iteration_density || rho_avg - 1000.0 > h->p.density_threshold * 1000 * 0.01) {     // :225
        if (it >= cap) { st->capped = 1; break; }
        launch_dens_residual(h, GATE_NONE);
        if ((rc = ghosts_k(true))) return rc;
        float avg;
        if ((rc = reduce_mean_host(h, 1000.0f, &avg))) return rc;
        launch_correct<CORR_DENS>(h, K_D_DENS_CORRECT, h->rho_adv, h->VA[0], GATE_NONE);
        if (rigid_coupled(h)) launch_rigid_force(h, GATE_NONE);
        if ((rc = ghosts_v(h->VA[0]))) return rc;
        rho_avg = (double)avg;
        it += 1;
    }
    st->n_dens = it;
    st->dens_err = (float)(rho_avg - 1000.0);
    if ((rc = read_scalars(h))) return rc;
    st->dt = h->ds_host->dt;
    return dfsph_integrate(h);
}

int step_dfsph_once(SphHandle *h, SphStepStats *st)
{
    int rc;
    memset(st, 0, sizeof(*st));
    h->simulate_cnt += 1;                                   // solver_base.py:137
    h->comm_stat[6] += 1;
    if ((rc = stage_sort_and_lists(h))) return rc;          // :139-141 (reset() is the no-op override, dfsph_solver.py:418-421)
    if ((rc = stage_density(h))) return rc;                 // initialize(): dfsph_solver.py:423-426
    const bool host_loops = h->slab && !slab_async(h);      // a transport without allreduce_stream
    return host_loops ? step_dfsph_host_loops(h, st) : step_dfsph_device_loops(h, st);
}

// ---------------------------------------------------------------------------------------------
// PBF (csrc/sph_pbf_kernels.h)                                pbf_solver.py:176-187
// ---------------------------------------------------------------------------------------------
PbfConsts pbf_consts(const SphHandle *h)
{
    const Consts &c = h->c;
    PbfConsts k;
    const double pi = 3.141592653589793, r = h->cfg.particle_radius;
    k.kpoly = 315.0f / ((float)(64 * pi) * (c.h * (c.h * c.h)));                         // solver_base.py:128 (64 * pi folds in f64)
    k.pih4 = (float)pi * ((c.h * c.h) * (c.h * c.h));                                    // :120
    k.neg_k = -(float)1e-7; k.c_visc = (float)9e-6; k.eps = (float)1.0e-6;               // pbf_solver.py:17-21
    {   // poly_kernel(s_corr_factor * kernel_h, kernel_h), the argument a Python float (:148)
        const float rc = (float)(0.3 * (r * 4)), q = rc / c.h, q2 = q * q, t = 1.0f - q2;
        k.w_corr = q <= 1.0f ? k.kpoly * (t * (t * t)) : 0.0f;
    }
    for (int a = 0; a < 3; ++a) {                                                        // :74-81
        k.lo[a] = (float)h->cfg.box_min[a] + (float)r;
        k.hi[a] = (float)h->cfg.box_max[a] - (float)r;
    }
    return k;
}

int step_pbf_once(SphHandle *h)
{
    int rc;
    h->simulate_cnt += 1;                                   // solver_base.py:137
    h->comm_stat[6] += 1;
    if ((rc = stage_sort_and_lists(h))) return rc;          // :139-141
    const Consts &c = h->c;
    const PbfConsts k = pbf_consts(h);
    const bool quad = sweep_mode(h) == SWEEP_QUAD;          // four lanes per particle in all three sweeps (small scenes)
    const dim3 g = grid_for(c.n), b(kBlock), gq((unsigned)std::max(1, (c.n + 63) / 64));
    hipStream_t s = h->stream;
    {
        ProfScope ps(h, K_B_LAMBDA);                          // compute_all_lambda :32-52
        if (quad) hipLaunchKernelGGL(k_pbf_lambda<true>, gq, b, 0, s, c, k, h->P[h->pcur], h->WP, h->nl, h->nlb, h->cnt, h->rho, h->aux, h->P[1 - h->pcur], 0);
        else hipLaunchKernelGGL(k_pbf_lambda<false>, g, b, 0, s, c, k, h->P[h->pcur], h->WP, h->nl, h->nlb, h->cnt, h->rho, h->aux, h->P[1 - h->pcur], 0);
    }
    {
        ProfScope ps(h, K_B_DELTA);                           // compute_all_delta_pos :55-64, the prediction :26-29, update_all_pos phase 1 :66-84
        if (quad) hipLaunchKernelGGL(k_pbf_delta<true>, gq, b, 0, s, c, k, h->dt_wcsph, h->P[1 - h->pcur], h->V[h->vcur], h->WP, h->nl, h->nlb, h->cnt, h->X[0], h->X[1], h->X[2]);
        else hipLaunchKernelGGL(k_pbf_delta<false>, g, b, 0, s, c, k, h->dt_wcsph, h->P[1 - h->pcur], h->V[h->vcur], h->WP, h->nl, h->nlb, h->cnt, h->X[0], h->X[1], h->X[2]);
    }
    {
        ProfScope ps(h, K_B_XSPH);                            // update_all_pos phases 2-3 :86-98
        if (quad) hipLaunchKernelGGL(k_pbf_xsph<true>, gq, b, 0, s, c, k, h->P[h->pcur], h->X[1], h->X[2], h->cell_start, h->P[1 - h->pcur], h->V[1 - h->vcur]);
        else hipLaunchKernelGGL(k_pbf_xsph<false>, g, b, 0, s, c, k, h->P[h->pcur], h->X[1], h->X[2], h->cell_start, h->P[1 - h->pcur], h->V[1 - h->vcur]);
    }
    h->pcur ^= 1; h->vcur ^= 1;
    HIP_TRY(h, hipGetLastError());
    h->nl_valid = false;
    h->density_valid = false;
    return SPH_OK;
}

// ---------------------------------------------------------------------------------------------
// PCISPH / IISPH (SURVEY.md section 8f "next": the solvers coupling_demo.json and breaking_dam_30k.json name)
// ---------------------------------------------------------------------------------------------
// The pressure refresh of the ghosts and the residual's mean in ONE group of transfers on the native transport (as the dfsph loops do,
// slab_exchange_resid_and_finalize): this slab's (sum, count) goes to every slab with the ghosts' pressures, the decision sums the gathered pairs.
int launch_pressure_finalize(SphHandle *h, int mode);
int slab_refresh_w_and_pressure_finalize(SphHandle *h, float4 *A, int mode)
{
    int rc;
    if (!(h->slab && h->native && h->gath_dev && h->opt_gather)) {
        if (h->slab && (rc = slab_exchange_field(h, 0, A, nullptr, nullptr))) return rc;
        return launch_pressure_finalize(h, mode);
    }
    {
        ProfScope ps(h, K_FINALIZE);
        hipLaunchKernelGGL(k_finalize_pressure, dim3(1), dim3(kFinBlock), 0, h->stream, h->psum, h->pcnt, h->nblocks, h->ds, mode, (int)FINP_REDUCE, h->gath_dev + 4 * h->slab_rank,
                           partial_group(h), partial_count(h), 0);
    }
    h->comm_stat[4] += 1;
    if ((rc = slab_exchange_field(h, 0, A, nullptr, nullptr, 2, 3))) return rc;      // (sum, count, overflow flags): check_overflow_all trusts the gathered flags
    ProfScope ps(h, K_FINALIZE);
    hipLaunchKernelGGL(k_finalize_pressure, dim3(1), dim3(kFinBlock), 0, h->stream, h->psum, h->pcnt, h->nblocks, h->ds, mode, (int)FINP_DECIDE, h->gath_dev, partial_group(h), partial_count(h), h->nslab);
    return SPH_OK;
}
int launch_pressure_finalize(SphHandle *h, int mode)
{
    if (h->slab) {
        {
            ProfScope ps(h, K_FINALIZE);
            hipLaunchKernelGGL(k_finalize_pressure, dim3(1), dim3(kFinBlock), 0, h->stream, h->psum, h->pcnt, h->nblocks, h->ds, mode, (int)FINP_REDUCE, h->red_dev, partial_group(h), partial_count(h));
        }
        int rc = slab_allreduce_stream(h, 3, 0);
        if (rc) return rc;
        ProfScope ps(h, K_FINALIZE);
        hipLaunchKernelGGL(k_finalize_pressure, dim3(1), dim3(kFinBlock), 0, h->stream, h->psum, h->pcnt, h->nblocks, h->ds, mode, (int)FINP_DECIDE, h->red_dev, partial_group(h), partial_count(h));
        return SPH_OK;
    }
    ProfScope ps(h, K_FINALIZE);
    hipLaunchKernelGGL(k_finalize_pressure, dim3(1), dim3(kFinBlock), 0, h->stream, h->psum, h->pcnt, h->nblocks, h->ds, mode, (int)FINP_ALL, (double *)nullptr, partial_group(h), partial_count(h));
    return SPH_OK;
}

// sharded pcisph / iisph need the device-side loop control (an in-place all-reduce on the stream)
int require_async_slab(SphHandle *h)
{
    if (h->slab && !slab_async(h))
        return fail(h, SPH_E_STATE, "pcisph / iisph on slabs need a transport with allreduce_stream (TorchComm) or the native RCCL transport");
    return SPH_OK;
}

// pcisph_solver.step :252-259
int step_pcisph_once(SphHandle *h, SphStepStats *st)
{
    int rc;
    memset(st, 0, sizeof(*st));
    h->simulate_cnt += 1;                                   // solver_base.py:137
    h->comm_stat[6] += 1;
    if ((rc = require_async_slab(h))) return rc;
    if ((rc = stage_sort_and_lists(h))) return rc;          // :139-141
    if ((rc = stage_density(h))) return rc;                 // compute_all_rho :239; P = (pos, rho)
    const Consts &c = h->c;
    hipStream_t s = h->stream;
    const dim3 g = grid_for(c.n), b(kBlock);
    const float dt = h->dt_wcsph;                           // delta_time never changes in pcisph
    const bool rg = rigid_coupled(h);
    const RigidView rv = rg ? rigid_view(h) : RigidView();
    float4 *EF = h->X[0], *PF = h->X[1], *PP = h->X[2], *PB[2] = {h->X[3], h->X[4]};
    const int cap = 80;                                     // max_iteration :21
    hipLaunchKernelGGL(k_pressure_ctrl_begin, dim3(1), dim3(1), 0, s, h->ds, cap);
    {
        ProfScope ps(h, K_P_EXT);                           // compute_ext_force, reset(), first predict_vel_pos
        SPH_LAUNCH_RMX0(k_pci_ext, rg, sweep_mode(h), relaxed_pressure(h), c.n, sweep_lds(h, sizeof(float4) + sizeof(uint32_t)), s, c, dt, h->P[h->pcur], h->V[h->vcur], h->nl, h->cnt, EF, PF,
                       PB[0], PP, rv, h->stage_src, h->stage_cnt);
    }
    // sharded: the ghosts' predicted positions / pressures come from their owners after the sweep that produced them
    auto ghosts_xyz = [&](float4 *A) -> int { return h->slab ? slab_exchange_field(h, 1, nullptr, A, nullptr) : SPH_OK; };
    if ((rc = ghosts_xyz(PP))) return rc;
    // tiles without pressure skip update_press_force (k_pci_press): single-GPU staged handles without rigid entries
    int *zero_press = (h->pci_zero_press && h->staged && !h->slab && !rg) ? h->pci_zero_press : nullptr;
    if (zero_press)       // after k_pci_ext: press_force = 0 and pos_predict = the zero-pressure prediction everywhere
        HIP_TRY(h, hipMemsetAsync(zero_press, 1, sizeof(int) * (size_t)h->nblocks, s));
    auto predict_rho = [&](int k, int gate) {               // the k-th predict_rho + residual: reads press from PB[k&1]
        ProfScope ps(h, K_P_PREDICT_RHO);
        SPH_LAUNCH_RMX0(k_pci_predict_rho, rg, sweep_mode(h), relaxed_pressure(h), c.n, sweep_lds(h, sizeof(float4)), s, c, h->pci_delta, PP, h->WP, h->nl, h->nlb, h->cnt, h->ds, PB[k & 1],
                       PB[(k + 1) & 1], h->rho_adv, h->psum, h->pcnt, gate, rv, h->stage_src, h->stage_cnt);
    };
    predict_rho(0, GATE_NONE);                              // :53-56
    if ((rc = slab_refresh_w_and_pressure_finalize(h, PB[1], PFIN_PCI_FIRST))) return rc;
    bool first = true;
    for (int k = 1, chunk = std::max(2, h->last_iters); k <= cap; chunk = 2) {
        for (int q = 0; q < chunk && k <= cap; ++q, ++k) {
            {
                ProfScope ps(h, K_P_PRESS);                 // iter_press (already in PB[k&1]), update_press_force, predict_vel_pos
                SPH_LAUNCH_RMX0(k_pci_press, rg, sweep_mode(h), relaxed_pressure(h), c.n, sweep_lds(h, sizeof(float4)), s, c, dt, PB[k & 1], h->WP, h->nl, h->nlb, h->cnt, h->rho, h->V[h->vcur],
                               EF, h->ds, PF, PP, GATE_DENS, rv, h->stage_src, h->stage_cnt, zero_press);
            }
            if (rg) launch_rigid_force_p<RF_PCISPH>(h, h->P[h->pcur], PB[k & 1], GATE_DENS);   // :209, every iteration
            if ((rc = ghosts_xyz(PP))) return rc;
            predict_rho(k, GATE_DENS);
            if ((rc = slab_refresh_w_and_pressure_finalize(h, PB[(k + 1) & 1], PFIN_PCI_LOOP))) return rc;
        }
        if ((rc = read_scalars(h))) return rc;
        if (first) {
            if ((rc = check_overflow_all(h, slab_async(h)))) return rc;      // (the slabs' flags came with the loop's first reduction)
            first = false;
        }
        if (!h->ds_host->dens_active) break;
    }
    st->max_nbrs = h->ds_host->max_nbrs;
    st->max_wall_nbrs = h->ds_host->max_wall_nbrs;
    st->lost = h->ds_host->lost;
    st->n_dens = h->ds_host->dens_it;
    st->capped = h->ds_host->dens_capped;
    st->dens_err = h->ds_host->dens_avg;
    st->dt = dt;
    h->pb_final = h->ds_host->dens_it & 1;
    h->last_iters = h->ds_host->dens_it;
    {
        ProfScope ps(h, K_P_INTEGRATE);
        hipLaunchKernelGGL(k_pci_integrate, g, b, 0, s, c, dt, h->P[h->pcur], h->V[h->vcur], EF, PF, h->P[1 - h->pcur], h->V[1 - h->vcur]);
        h->pcur ^= 1; h->vcur ^= 1;
    }
    HIP_TRY(h, hipGetLastError());
    h->nl_valid = false;
    h->density_valid = false;
    return SPH_OK;
}

// iisph_solver.step :340-347
int step_iisph_once(SphHandle *h, SphStepStats *st)
{
    int rc;
    memset(st, 0, sizeof(*st));
    h->simulate_cnt += 1;
    h->comm_stat[6] += 1;
    if ((rc = require_async_slab(h))) return rc;
    if ((rc = stage_sort_and_lists(h))) return rc;
    if ((rc = stage_density(h))) return rc;                 // predict_advection :38; P = (pos, rho)
    const Consts &c = h->c;
    hipStream_t s = h->stream;
    const dim3 g = grid_for(c.n), b(kBlock);
    const float dt = h->dt_wcsph;
    const bool rg = rigid_coupled(h);
    const RigidView rv = rg ? rigid_view(h) : RigidView();
    float4 *DII = h->X[0], *DIJ = h->X[1], *FP = h->X[2], *PB[2] = {h->X[3], h->X[4]}, *VA = h->VA[0];
    const int cap = 180;                                    // max_iter_cnt :27
    hipLaunchKernelGGL(k_pressure_ctrl_begin, dim3(1), dim3(1), 0, s, h->ds, cap);
    {
        ProfScope ps(h, K_I_ADVECT);                        // :43-56
        SPH_LAUNCH_RMX0(k_ii_advect, rg, sweep_mode(h), relaxed_pressure(h), c.n, sweep_lds(h, sizeof(float4) + sizeof(uint32_t)), s, c, dt, h->P[h->pcur], h->V[h->vcur], h->WP, h->nl, h->nlb,
                       h->cnt, VA, DII, rv, h->stage_src, h->stage_cnt);
    }
    auto ghosts_xyz = [&](float4 *A) -> int { return h->slab ? slab_exchange_field(h, 1, nullptr, A, nullptr) : SPH_OK; };
    if ((rc = ghosts_xyz(VA))) return rc;                   // v_adv and d_ii of the ghosts (their 0.5 p_past travels with the particle)
    if ((rc = ghosts_xyz(DII))) return rc;
    {
        ProfScope ps(h, K_I_RHO_ADV);                       // :58-82; a_ii lives in aux, p_past in the carried scalar
        SPH_LAUNCH_RMX0(k_ii_rho_adv, rg, sweep_mode(h), relaxed_pressure(h), c.n, sweep_lds(h, sizeof(float4) + sizeof(float2)), s, c, dt, h->P[h->pcur], VA, h->WP, h->nl, h->nlb, h->cnt,
                       DII, h->warm[h->wcur], h->rho_adv, h->aux, PB[0], rv, h->stage_src, h->stage_cnt);
    }
    int *zero_dij = (h->pci_zero_press && h->staged && !h->slab) ? h->pci_zero_press : nullptr;      // tiles without pressure skip compute_all_d_ij (k_ii_dij)
    if (zero_dij) HIP_TRY(h, hipMemsetAsync(zero_dij, 0, sizeof(int) * (size_t)h->nblocks, s));      // DIJ still holds last step's sums
    bool first = true;
    for (int k = 1, chunk = std::max(2, h->last_iters); k <= cap; chunk = 2) {
        for (int q = 0; q < chunk && k <= cap; ++q, ++k) {
            {
                ProfScope ps(h, K_I_DIJ);                   // compute_all_d_ij :91
                SPH_LAUNCH_RMX0(k_ii_dij, rg, sweep_mode(h), relaxed_pressure(h), c.n, sweep_lds(h, sizeof(float4) + sizeof(uint32_t)), s, c, dt, PB[(k - 1) & 1], h->rho, h->nl, h->cnt, h->ds,
                               DIJ, GATE_DENS, rv, h->stage_src, h->stage_cnt, zero_dij);
            }
            if ((rc = ghosts_xyz(DIJ))) return rc;
            {
                ProfScope ps(h, K_I_UPDATE_P);              // update_p :93 + compute_residual :97
                SPH_LAUNCH_RMX0(k_ii_update_p, rg, sweep_mode(h), relaxed_pressure(h), c.n, sweep_lds(h, sizeof(float4) + sizeof(uint32_t) + 3 * sizeof(float)), s, c, dt, PB[(k - 1) & 1], DII, DIJ, h->WP, h->nl,
                               h->nlb, h->cnt, h->rho, h->rho_adv, h->aux, h->ds, PB[k & 1], h->psum, h->pcnt, GATE_DENS, rv, h->stage_src, h->stage_cnt);
            }
            if ((rc = slab_refresh_w_and_pressure_finalize(h, PB[k & 1], PFIN_II_LOOP))) return rc;
        }
        if ((rc = read_scalars(h))) return rc;
        if (first) {
            if ((rc = check_overflow_all(h, slab_async(h)))) return rc;      // (the slabs' flags came with the loop's first reduction)
            first = false;
        }
        if (!h->ds_host->dens_active) break;
    }
    st->max_nbrs = h->ds_host->max_nbrs;
    st->max_wall_nbrs = h->ds_host->max_wall_nbrs;
    st->lost = h->ds_host->lost;
    st->n_dens = h->ds_host->dens_it;
    st->capped = h->ds_host->dens_capped;
    st->n_div = h->ds_host->res_diverged;                   // 1: the loop left on "Iteration trend to divergence" (:97-99)
    st->dens_err = h->ds_host->dens_avg;
    st->dt = dt;
    h->pb_final = h->ds_host->dens_it & 1;
    h->last_iters = h->ds_host->dens_it;
    if (rg) launch_rigid_force_p<RF_IISPH>(h, h->P[h->pcur], PB[h->pb_final], GATE_NONE);   // compute_all_press_force :172-179
    {
        ProfScope ps(h, K_I_INTEGRATE);
        hipLaunchKernelGGL(k_ii_integrate, g, b, 0, s, c, dt, h->P[h->pcur], VA, DII, DIJ, PB[h->pb_final], h->P[1 - h->pcur], h->V[1 - h->vcur],
                           FP, h->warm[h->wcur]);
        h->pcur ^= 1; h->vcur ^= 1;
    }
    HIP_TRY(h, hipGetLastError());
    h->nl_valid = false;
    h->density_valid = false;
    return SPH_OK;
}

// host copy of solver_base.cubic_kernel_derivative (:90-103), same f32 operations as the device's grad_w
void grad_w_host(const Consts &c, float rx, float ry, float rz, float out[3])
{
    const float r_norm = sqrtf((rx * rx + ry * ry) + rz * rz);
    const float q = r_norm / c.h;
    out[0] = out[1] = out[2] = 0.f;
    float sc;
    if (1e-5f < q && q <= 0.5f) sc = c.kg6 * (3.0f * (q * q) - 2.0f * q);
    else if (0.5f < q && q <= 1.0f) { const float t = 1.0f - q; sc = c.neg_kg6 * (t * t); }
    else return;
    const float den = c.h * r_norm;
    out[0] = sc * rx / den; out[1] = sc * ry / den; out[2] = sc * rz / den;
}

// pcisph_solver.__init__ :23-26 + pre_compute :28-47: beta, the fullest neighbourhood of the initial lattice, delta
int pcisph_precompute(SphHandle *h)
{
    int rc;
    const Consts &c = h->c;
    const int N = h->N;
    const double r = h->cfg.particle_radius;
    const double m = 1000 * (r * r * r) * 8;
    const double dtf = (double)h->dt_wcsph;                            // self.delta_time[None] read back as a Python float
    const double beta = dtf * dtf * m * m * 2 / (double)(1000 * 1000); // :23 (Python f64, left to right)
    h->pci_beta = (float)beta;
    // get_max_neighbor_particle_index (ParticleSystem.py:410-422): counts from the device lists, then the single-thread
    // reading of the atomic_max idiom -- the last particle whose count ties the running maximum
    std::vector<float> counts((size_t)N);
    if (h->slab) {
        // every slab needs the same delta: neighbour counts of the WHOLE initial lattice, on the host (same r2 > r2_cut criterion as
        // k_build_nl; one-time, O(216 N))
        const float *pos = h->pci_fluid_pos.data();
        std::vector<int> cid((size_t)N), start((size_t)c.C + 1, 0), order((size_t)N);
        for (int i = 0; i < N; ++i) {
            const int x = (int)floorf(pos[3 * (size_t)i] / c.h), y = (int)floorf(pos[3 * (size_t)i + 1] / c.h), z = (int)floorf(pos[3 * (size_t)i + 2] / c.h);
            int id = x + y * c.sy + z * c.sz;
            if (x < 0 || y < 0 || z < 0 || x >= c.gx || y >= c.gy || z >= c.gz) id = -1;
            cid[i] = id;
            if (id >= 0) start[(size_t)id + 1]++;
        }
        for (int k = 0; k < c.C; ++k) start[(size_t)k + 1] += start[k];
        std::vector<int> fill(start.begin(), start.end() - 1);
        for (int i = 0; i < N; ++i) if (cid[i] >= 0) order[fill[cid[i]]++] = i;
        for (int i = 0; i < N; ++i) {
            int cnt = 0;
            if (cid[i] >= 0) {
                const int x = cid[i] % c.gx, z = (cid[i] / c.gx) % c.gz, y = cid[i] / (c.gx * c.gz);
                for (int dx = -1; dx <= 1; ++dx)
                    for (int dy = -1; dy <= 1; ++dy)
                        for (int dz = -1; dz <= 1; ++dz) {
                            const int xx = x + dx, yy = y + dy, zz = z + dz;
                            if (xx < 0 || yy < 0 || zz < 0 || xx >= c.gx || yy >= c.gy || zz >= c.gz) continue;
                            const int nb = xx + yy * c.sy + zz * c.sz;
                            for (int e = start[nb]; e < start[(size_t)nb + 1]; ++e) {
                                const int j = order[e];
                                if (j == i) continue;
                                const float ax = pos[3 * (size_t)i] - pos[3 * (size_t)j], ay = pos[3 * (size_t)i + 1] - pos[3 * (size_t)j + 1],
                                            az = pos[3 * (size_t)i + 2] - pos[3 * (size_t)j + 2];
                                if (!((ax * ax + ay * ay) + az * az > c.r2_cut)) ++cnt;
                            }
                        }
            }
            counts[i] = (float)cnt;
        }
    } else {
    if ((rc = stage_sort_and_lists(h))) return rc;
    if ((rc = read_scalars(h))) return rc;
    if ((rc = check_overflow(h))) return rc;
    if (rigid_coupled(h))   // get_neighbour_count with its rigid-entry quirk (ParticleSystem.py:436-444)
        hipLaunchKernelGGL(k_unsort_scalar_int, grid_for(N), dim3(kBlock), 0, h->stream, N, h->ncount, h->id[h->icur], h->staging);
    else
        hipLaunchKernelGGL(k_unsort_count, grid_for(N), dim3(kBlock), 0, h->stream, N, h->cnt, h->id[h->icur], h->staging);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(counts.data(), h->staging, sizeof(float) * (size_t)N, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    int max_count = -1, max_index = -1;
    for (int i = 0; i < N; ++i) {
        const int cnt = (int)counts[i];
        const int old = max_count;
        if (cnt > max_count) max_count = cnt;
        if (old == cnt) max_index = i;
    }
    h->pci_max_index = max_index; h->pci_max_count = max_count;
    float sx = 0.f, sy = 0.f, sz = 0.f, sq = 0.f;
    if (max_index >= 0) {
        // for_all_neighbor(max_index) on the host: 27 cells, dx outermost; inside a cell ascending fluid ids, then the rigid entries
        const float *pos = h->pci_fluid_pos.data();
        auto cell = [&](const float *p, int cc[3]) { for (int a = 0; a < 3; ++a) cc[a] = (int)floorf(p[a] / c.h); };
        int ci[3];
        cell(pos + 3 * (size_t)max_index, ci);
        std::vector<int> bucket[27];
        for (int j = 0; j < N; ++j) {
            int cj[3];
            cell(pos + 3 * (size_t)j, cj);
            const int dx = cj[0] - ci[0], dy = cj[1] - ci[1], dz = cj[2] - ci[2];
            if (dx < -1 || dx > 1 || dy < -1 || dy > 1 || dz < -1 || dz > 1) continue;
            if (cj[0] < 0 || cj[0] >= c.gx || cj[1] < 0 || cj[1] >= c.gy || cj[2] < 0 || cj[2] >= c.gz) continue;
            bucket[(dx + 1) * 9 + (dy + 1) * 3 + (dz + 1)].push_back(j);
        }
        std::vector<int> rbucket[27];
        if (rigid_coupled(h))
            for (int j = 0; j < h->Nr; ++j) {
                int cj[3];
                cell(h->rigid_pos_host.data() + 3 * (size_t)j, cj);
                const int dx = cj[0] - ci[0], dy = cj[1] - ci[1], dz = cj[2] - ci[2];
                if (dx < -1 || dx > 1 || dy < -1 || dy > 1 || dz < -1 || dz > 1) continue;
                rbucket[(dx + 1) * 9 + (dy + 1) * 3 + (dz + 1)].push_back(j);
            }
        const float *pi = pos + 3 * (size_t)max_index;
        auto add = [&](const float *pj) {
            const float x = pi[0] - pj[0], y = pi[1] - pj[1], z = pi[2] - pj[2];
            if (sqrtf((x * x + y * y) + z * z) > c.h) return;
            float gw[3];
            grad_w_host(c, x, y, z, gw);
            sx += gw[0]; sy += gw[1]; sz += gw[2];                 // compute_sum :179-183 (any material)
            sq += (gw[0] * gw[0] + gw[1] * gw[1]) + gw[2] * gw[2]; // compute_square_sum :185-190
        };
        for (int bk = 0; bk < 27; ++bk) {
            for (int j : bucket[bk])
                if (j != max_index) add(pos + 3 * (size_t)j);
            for (int j : rbucket[bk]) add(h->rigid_pos_host.data() + 3 * (size_t)j);
        }
    }
    h->pci_delta = 1.0f / ((((sx * sx + sy * sy) + sz * sz) + sq) * h->pci_beta);   // :47
    return SPH_OK;
}

int field_floats(SphHandle *h, int species, int field, size_t *count, bool *vec)
{
    *vec = false;
    if (species == SPH_SPECIES_FLUID) {
        switch (field) {
        case SPH_F_POS: case SPH_F_VEL: case SPH_F_ACC: case SPH_F_VEL_ADV: case SPH_F_PRESS_FORCE: case SPH_F_POS_PREDICT: case SPH_F_D_II:
        case SPH_F_D_IJ: case SPH_F_PBF_DELTA_POS:
            *vec = true; *count = 3 * (size_t)h->N; return SPH_OK;
        case SPH_F_RHO: case SPH_F_PRESSURE: case SPH_F_ALPHA: case SPH_F_WARM_K: case SPH_F_RHO_ADV: case SPH_F_RHO_DER:
        case SPH_F_NBR_COUNT: case SPH_F_PRESS_ITER: case SPH_F_A_II: case SPH_F_PBF_LAMBDA:
            *count = (size_t)h->N; return SPH_OK;
        default: break;
        }
    } else if (species == SPH_SPECIES_WALL) {
        if (field == SPH_F_WALL_POS) { *vec = true; *count = 3 * (size_t)h->Nb; return SPH_OK; }
        if (field == SPH_F_WALL_VOL) { *count = (size_t)h->Nb; return SPH_OK; }
    } else if (species == SPH_SPECIES_RIGID && h->rigid) {
        if (field == SPH_F_RIGID_POS || field == SPH_F_RIGID_FORCE) { *vec = true; *count = 3 * (size_t)h->Nr; return SPH_OK; }
        if (field == SPH_F_RIGID_VOL || field == SPH_F_RIGID_MASS) { *count = (size_t)h->Nr; return SPH_OK; }
        if (field == SPH_F_RIGID_VERT) { *vec = true; *count = 3 * (size_t)h->Nv; return SPH_OK; }
    }
    return fail(h, SPH_E_INVALID, "unknown species/field %d/%d", species, field);
}

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
    { const char *e = dev_env(&h->overrides, "SPH_NL16"); h->opt_nl16 = !(e && atoi(e) == 0); }
    { const char *e = dev_env(&h->overrides, "SPH_KR_SPLIT"); h->opt_kr_split = !(e && atoi(e) == 0); }
    { const char *e = dev_env(&h->overrides, "SPH_TILE_SKIP"); h->opt_tile_skip = !(e && atoi(e) == 0); }
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
    SphHandle::Params &p = h->p;
    const bool dfsph = h->cfg.solver == SPH_SOLVER_DFSPH;
    auto count = [&](int *dst, int lo) -> int {
        if (!(value >= lo && value <= 100000.0) || value != std::floor(value)) return fail(h, SPH_E_INVALID, "sph_set_scalar(%d): an integer >= %d expected, got %g", which, lo, value);
        *dst = (int)value; return SPH_OK;
    };
    auto positive = [&](double *dst, bool zero_ok) -> int {
        if (!(value > 0.0 || (zero_ok && value == 0.0)) || !std::isfinite(value)) return fail(h, SPH_E_INVALID, "sph_set_scalar(%d): a finite value %s 0 expected, got %g", which, zero_ok ? ">=" : ">", value);
        *dst = value; return SPH_OK;
    };
    int rc = SPH_OK;
    if (which >= SPH_P_DENSITY_THRESHOLD && which <= SPH_P_MIN_DT && !dfsph) return fail(h, SPH_E_STATE, "sph_set_scalar(%d): a dfsph_solver attribute on a handle of another solver", which);
    switch (which) {
    case SPH_P_DENSITY_THRESHOLD: rc = positive(&p.density_threshold, true); break;
    case SPH_P_MIN_ITERATION_DENSITY: rc = count(&p.min_iteration_density, 0); break;
    case SPH_P_MIN_ITERATION_DENSITY_DIVERGENCE: rc = count(&p.min_iteration_density_divergence, 0); break;
    case SPH_P_MAX_ITERATION_DENSITY_DIVERGENCE: rc = count(&p.max_iteration_density_divergence, 0); break;
    case SPH_P_DENSITY_DIVERGENCE_THRESHOLD: rc = positive(&p.density_divergence_threshold, true); break;
    case SPH_P_WARM_START: p.warm_start = value != 0.0; break;
    case SPH_P_ADAPTIVE_DT: p.adaptive_dt = value != 0.0; break;
    case SPH_P_MAX_DT: rc = positive(&p.max_dt, false); break;
    case SPH_P_MIN_DT: rc = positive(&p.min_dt, false); break;
    case SPH_P_VISCOSITY_C_S: rc = positive(&p.viscosity_c_s, true); break;
    case SPH_P_VISCOSITY_ALPHA: rc = positive(&p.viscosity_alpha, true); break;
    case SPH_P_VISCOSITY_EPSILON: rc = positive(&p.viscosity_epsilon, false); break;
    case SPH_P_TENSION_K: rc = positive(&p.tension_k, true); break;
    default: return fail(h, SPH_E_INVALID, "unknown scalar %d", which);
    }
    if (rc) return rc;
    if (h->cfg.solver == SPH_SOLVER_PBF && which >= SPH_P_VISCOSITY_C_S) return fail(h, SPH_E_STATE, "sph_set_scalar(%d): pbf_solver has no such attribute", which);
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
