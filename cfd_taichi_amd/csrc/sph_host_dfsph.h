// sph_host_dfsph.h -- a SECTION of csrc/sph_mi355x.hip's one translation unit (included there once, inside its anonymous namespace, in file order):
// the step stages (sort + lists, density), the wcsph step and the dfsph step with its device-side loop control.  Not a stand-alone header: it uses SphHandle and the helpers defined above its include.

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
                           h->warm[1 - h->wcur], h->id[1 - h->icur], rigid_coupled(h) ? h->pos_orig : (float4 *)nullptr, gate, h->x0,
                           h->slab ? h->dead : (int *)nullptr);
        h->pcur ^= 1; h->vcur ^= 1; h->icur ^= 1;
        if (carry) h->wcur ^= 1;
    }
    if (h->slab) {
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
            hipLaunchKernelGGL(k_layer_offsets, dim3(jobs.n), dim3(kScanBlock), 0, s, c, h->cell_start, jobs, h->opt_layer_generic ? 1 : 0);
            hipLaunchKernelGGL(k_layer_list, dim3(grid_for(c.gy * c.gz).x, jobs.n), b, 0, s, c, h->cell_start, jobs);
        }
        if (h->opt_slab_check) {       // the host's bookkeeping of the column populations against the sorted arrays
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
                                             h->nl, h->nlb, h->cnt, h->ds, rigid_view_or_none(h), h->ncount, h->stage_src, h->stage_cnt, gate, h->tile_nbr)
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
// the density loop's sweeps: who must run is said by the sweep before (DensFlow in sph_kernels.h).  Every sweep takes the next stamp and consumes the
// stamp of the sweep enqueued before it; d6 = the residual sweep (consumes need6 / pushes need7), else the correction sweep
inline DensFlow dens_flow(SphHandle *h, bool d6)
{
    if (!h->tile_nbr || !tile_skip(h) || h->tune_all) return kNoFlow;
    DensFlow df{h->tile_nbr, d6 ? h->need6 : h->need7, d6 ? h->need7 : h->need6, h->tile_nz, h->dens_bcast, h->flow_last, ++h->flow_stamp, d6 ? 0 : 1, d6 ? 1 : 0,
                d6 ? h->worked6 : h->worked7};
    h->flow_last = df.stamp_out;
    return df;
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
    const DensFlow df = (MODE == CORR_DENS && wdirty) ? dens_flow(h, false) : kNoFlow;
    const int n_grid = c.n + (ride ? (sweep_mode(h) == SWEEP_QUAD ? 64 : kBlock) : 0);           // one more workgroup
    if (use_relaxed(h)) {
        hipLaunchKernelGGL(k_correct_rx<MODE>, grid_for(split ? c.n : n_grid), dim3(kBlock), sweep_lds(h, sizeof(float4)), h->stream, c, h->P[h->pcur], h->wall_grad, h->nl, h->cnt,
                           h->rho, h->aux, src, h->warm[h->wcur], h->ds, V, V, gate, h->stage_src, h->stage_cnt, h->krho, wdirty, h->changed8,
                           split ? TilePhase{h->tile_order, h->nblocks, 2} : tp0, sv, split ? kNoRide : fr, df);
        if (!split) return;
    }
    SPH_LAUNCH_RMX(k_correct, MODE, rigid_coupled(h), sweep_mode(h), relaxed_unstaged(h), split ? c.n : n_grid, sweep_lds(h, sizeof(float4)), h->stream, c,
                  c.kr_split ? h->P[h->pcur] : h->P[1 - h->pcur], h->WP,
                  h->nl, h->nlb, h->cnt, h->rho, h->aux, src, h->warm[h->wcur], h->ds, V, V, rigid_view_or_none(h), gate, h->stage_src, h->stage_cnt, h->krho, wdirty, h->changed8,
                  (const float4 *)wall_cache(h), split ? TilePhase{h->tile_order, h->nblocks, 1} : tp0, sv, split ? kNoRide : fr, df);
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
    // (the interior launch of a split sweep is the same sweep as its edge launch: the same stamps)
    const DensFlow df = wdirty ? (phase == 2 ? h->flow_d6 : dens_flow(h, true)) : kNoFlow;
    h->flow_d6 = df;
    const SpecUndo no_undo{nullptr, nullptr, nullptr, nullptr, 0};
    if (use_relaxed(h)) {
        const TilePhase tpr = split ? TilePhase{h->tile_order, h->nblocks, 2} : tp;
        hipLaunchKernelGGL(k_residual_rx<true>, grid_for(c.n), dim3(kBlock), sweep_lds(h, sizeof(float4) + sizeof(float2)), st, c, h->P[h->pcur], h->VA[0],
                           h->wall_grad, h->nl, h->cnt, h->rho, h->aux, h->ds, h->rho_adv, h->psum, h->pcnt, gate, h->stage_src, h->stage_cnt, h->krho, wdirty, h->changed8, force_all, tpr, no_undo, df);
        if (!split) return;
    }
    SPH_LAUNCH_RMX(k_residual, true, rigid_coupled(h), sweep_mode(h), relaxed_unstaged(h), c.n, sweep_lds(h, sizeof(float4) + sizeof(float2)), st, c,
                  h->P[h->pcur], h->VA[0], h->WP, h->nl, h->nlb, h->cnt, h->rho, h->aux, h->ds, h->rho_adv, h->P[1 - h->pcur], h->psum, h->pcnt,
                  rigid_view_or_none(h), h->ncount, gate, h->stage_src, h->stage_cnt, h->krho, wdirty, h->changed8, force_all, (const float4 *)wall_cache(h), tp, no_undo, df);
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
                           val, P, S, h->psum, h->pcnt, h->nblocks, h->ds, mode, gather ? h->gath_dev : h->red_dev, partial_group(h), partial_count(h), gather ? h->nslab : 0,
                           dens ? h->flow_d6 : kNoFlow);
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
    // All max_iteration_density_divergence (15) possible iterations are enqueued at once -- the ones the reference's loop would not run leave at their first
    // instruction.  A caller who writes the "no cap" idiom (solver.max_iteration_density_divergence = 1000) must not pay a thousand gated full-grid launches
    // per step for that: beyond kDivChunk iterations the host looks at the loop state every kDivChunk iterations and stops enqueuing once the loop has
    // ended.  1 = ended, 0 = go on, < 0 = error
    constexpr int kDivChunk = 32;
    auto div_ended = [&](int e) -> int {
        if (max_div <= kDivChunk || e % kDivChunk != 0 || e >= max_div) return 0;
        const int r = read_scalars_fast(h);
        if (r) return r;
        return h->ds_host->div_active ? 0 : 1;
    };
    if (ride) {
        launch_div_residual(h, GATE_NONE);                                                                                   // :398, evaluation 1
        int last_e = max_div;
        for (int e = 1; e <= max_div; ++e) {
            launch_correct<CORR_DIV>(h, K_D_DIV_CORRECT, h->drho, h->V[h->vcur], GATE_HIST0 + ((e - 1) & 1), SpecSave{h->spec_v, h->spec_w},
                                     e == 1 ? FIN_DIV_FIRST : FIN_DIV_LOOP, e);                                              // :402-405 + decision e
            launch_div_residual(h, GATE_DIV, 0, SpecUndo{h->V[h->vcur], h->spec_v, h->warm[h->wcur], h->spec_w, e});     // :408, evaluation e + 1
            if ((rc = div_ended(e)) < 0) return rc;
            if (rc) { last_e = e; break; }
        }
        // the decision of the last evaluation has no correction launch to ride in: it is taken by the launch that reduces max |v*| (dfsph_ext_and_dt;
        // the sweep in between, D5, writes other partials and reads no loop state)
        h->pending_div = FinRide{h->psum, h->pcnt, h->ds, h->nblocks, max_div == 0 ? (int)FIN_DIV_FIRST : (int)FIN_DIV_LOOP, partial_group(h), partial_count(h), last_e + 1};
    } else if (spec) {
        if ((rc = residual_sweep(false, GATE_NONE, SpecUndo{nullptr, nullptr, nullptr, nullptr, 0}, FIN_DIV_FIRST))) return rc;     // :398, evaluation 1
        int last_spec = max_div;
        for (int e = 1; e <= max_div; ++e) {
            // the correction of evaluation e first (the GPU works on it while the host may block in a synchronous all-reduce) ...
            launch_correct<CORR_DIV>(h, K_D_DIV_CORRECT, h->drho, h->V[h->vcur], GATE_HIST0 + ((e - 1) & 1), SpecSave{h->spec_v, h->spec_w});   // :402-405
            // ... then evaluation e's reduction and decision on the third stream
            if ((rc = launch_finalize_decide(h, e == 1 ? FIN_DIV_FIRST : FIN_DIV_LOOP, e))) return rc;
            HIP_TRY(h, hipStreamWaitEvent(s, h->ev_dec, 0));
            if ((rc = residual_sweep(false, GATE_DIV, SpecUndo{h->V[h->vcur], h->spec_v, h->warm[h->wcur], h->spec_w, e}, FIN_DIV_LOOP))) return rc;   // :408, evaluation e + 1
            if ((rc = div_ended(e)) < 0) return rc;
            if (rc) { last_spec = e; break; }
        }
        if ((rc = launch_finalize_decide(h, max_div == 0 ? FIN_DIV_FIRST : FIN_DIV_LOOP, last_spec + 1))) return rc;
        HIP_TRY(h, hipStreamWaitEvent(s, h->ev_dec, 0));
    } else {
    if ((rc = residual(false, GATE_NONE, FIN_DIV_FIRST))) return rc;                 // :398
    // (enqueued at once, see div_ended above: the host does not need the outcome before the density loop's first read-back)
    for (int done = 0; done < max_div; ++done) {
        launch_correct<CORR_DIV>(h, K_D_DIV_CORRECT, h->drho, h->V[h->vcur], GATE_DIV);   // :402-405
        if ((rc = ghosts_v(h->V[h->vcur]))) return rc;
        if ((rc = residual(false, GATE_DIV, FIN_DIV_LOOP))) return rc;                    // :408
        if ((rc = div_ended(done + 1)) < 0) return rc;
        if (rc) break;
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
