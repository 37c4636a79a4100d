// sph_host_scene.h -- a SECTION of csrc/sph_mi355x.hip's one translation unit (included there once, inside its anonymous namespace, in file order):
// the cuts planner, the one-time scene construction (mirrors ParticleSystem.__init__), the device arena and the control block's read-back.  Not a stand-alone header: it uses SphHandle and the helpers defined above its include.

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

// Slab handles on the Morton curve: the cell slots of THIS slab only.  The whole grid's slots were replicated on every rank -- at config 4 on 8 slabs
// 6.2 M cells scanned, zeroed and walked per step by a rank that holds 17 of 401 columns (k_scan_tiles 26 us against 5 at 1 M on one GPU, k_scan_sums,
// k_scan_add, k_layer_offsets: ~0.1 ms of a 5 ms step).  The tiles (4 x 4 x 4 cells) that hold one of the slab's columns [x_lo - layers, x_hi + layers)
// keep their order along the curve and are ranked among themselves; every other tile ranks -1: cell_slot() bins a particle there nowhere and the
// 27-cell walks skip it (`slot < 0`).  Re-applied when a re-balancing moves the cuts (the cell arrays keep the whole grid's size, S_full).
int slab_local_grid(SphHandle *h)
{
    Consts &c = h->c;
    if (!h->slab || c.order != CELL_ORDER_TILED || h->tile_rank_full.empty()) return SPH_OK;
    const int lo = std::max(h->geom.x_lo - h->geom.layers, 0) >> c.tbits, hi = std::min(h->geom.x_hi + h->geom.layers - 1, c.gx - 1) >> c.tbits;
    std::vector<std::pair<int, int>> held;                   // (rank along the whole grid's curve, tile)
    for (size_t t = 0; t < h->tile_rank_full.size(); ++t) {
        const int tx = (int)(t % (size_t)c.tnx);
        if (tx >= lo && tx <= hi) held.push_back({h->tile_rank_full[t], (int)t});
    }
    std::sort(held.begin(), held.end());
    h->tile_rank_local.assign(h->tile_rank_full.size(), -1);
    for (size_t r = 0; r < held.size(); ++r) h->tile_rank_local[(size_t)held[r].second] = (int)r;
    HIP_TRY(h, hipMemcpyAsync(h->tile_rank, h->tile_rank_local.data(), sizeof(int) * h->tile_rank_local.size(), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    c.S = (int)(held.size() << (3 * c.tbits));
    h->ntiles = (int)(((size_t)c.S + 2 + kScanTile - 1) / kScanTile);
    return SPH_OK;
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
            // DensFlow: one GPU, and slab handles of the two-column protocol (the inner ghosts' velocities are corrected HERE like everybody's, the ghosts'
            // k / rho arrives with the residual's refresh, whose unpack kernel pushes for them; a one-column handle's ghosts change behind the rank's back)
            if (h->cfg.solver == SPH_SOLVER_DFSPH && h->opt_tile_skip && h->opt_dens_push && (!h->slab || h->geom.layers == 2)) {
                const size_t nt = (n + kBlock - 1) / kBlock + 1;
                if ((rc = dalloc(h, &h->tile_nbr, nt * (size_t)kNbrStride))) return rc;
                if ((rc = dalloc(h, &h->need6, nt)) || (rc = dalloc(h, &h->need7, nt)) || (rc = dalloc(h, &h->tile_nz, nt))) return rc;
                if ((rc = dalloc(h, &h->worked6, nt)) || (rc = dalloc(h, &h->worked7, nt))) return rc;
                if ((rc = dalloc(h, &h->dens_bcast, 64))) return rc;
                // (SPH_NBR_CAP: fewer tiles per row than fit -- most rows then say "unknown" and their tiles take the broadcast / always-run fallback: tests)
                const char *cap = dev_env(&h->overrides, "SPH_NBR_CAP");
                h->c.nbr_cap = cap ? std::min(std::max(atoi(cap), 0), kNbrStride - 1) : kNbrStride - 1;
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
    h->S_full = c.S;
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
    // (coherent / fine-grained asked for by name: read_scalars_fast spins on a word the device writes while the stream is still busy, and
    // HIP_HOST_COHERENT=0 in the environment would otherwise make that word visible only after the stream has drained -- ADVICE r5)
    if (hipHostMalloc((void **)&h->pub_host, sizeof(DevScalarsPub), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
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
        const char *grid_full = dev_env(&h->overrides, "SPH_SLAB_GRID_FULL");
        if (h->slab && !(grid_full && atoi(grid_full) != 0)) {         // (SPH_SLAB_GRID_FULL=1: the whole grid's slots on every rank, as before round 6 -- A/B, tests)
            h->tile_rank_full = tile_rank;
            if ((rc = slab_local_grid(h))) return rc;
        }
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
            if (*flag != seq) { h->pub_dev = nullptr; return read_scalars(h); }      // the word never arrives on this system: no second 2 ms wait
            // it arrived only with the drain: either a chunk that really took longer than 2 ms, or writes that are not visible while the stream runs.
            // Three such waits in a row and the handle goes back to the copy (a step of a large scene does take > 2 ms per chunk: keep trying there)
            if (++h->pub_late >= 3 && h->c.n < (1 << 21)) h->pub_dev = nullptr;
            break;
        }
    }
    if (spins > 0 && std::chrono::steady_clock::now() - t0 <= std::chrono::milliseconds(2)) h->pub_late = 0;
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
