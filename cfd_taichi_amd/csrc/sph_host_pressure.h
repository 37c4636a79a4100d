// sph_host_pressure.h -- a SECTION of csrc/sph_mi355x.hip's one translation unit (included there once, inside its anonymous namespace, in file order):
// the pbf, pcisph and iisph steps and the field table of upload / download.  Not a stand-alone header: it uses SphHandle and the helpers defined above its include.

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
