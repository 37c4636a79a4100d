// sph_host_rigid.h -- a SECTION of csrc/sph_mi355x.hip's one translation unit (included there once, inside its anonymous namespace, in file order):
// the rigid body of config 5: host-side construction and rigid_solver.step.  Not a stand-alone header: it uses SphHandle and the helpers defined above its include.

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
    if ((rc = dalloc(h, &h->rcell_count, (size_t)h->S_full + 2))) return rc;      // (S_full: a slab handle's own slots grow and shrink with its cuts)
    if ((rc = dalloc(h, &h->rcell_start, (size_t)h->S_full + 2))) return rc;
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
    HIP_TRY(h, hipMemsetAsync(h->rcell_start, 0, sizeof(int) * ((size_t)h->S_full + 2), h->stream));
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
