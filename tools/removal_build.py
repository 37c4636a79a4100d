#!/usr/bin/env python3
"""Builds of the library for REMOVAL experiments and soak variants -- from a patched COPY of csrc/, never from switches inside the
product kernels (`-DSPH_X_*` compiled wrong-on-purpose variants of the hot kernels).

    python tools/removal_build.py nofluid nowall nogather rx_nofluid ...      ->  ab/libsph_<name>.so each

Timing only: except `bnl_notable` the results of these builds are wrong on purpose.  Compare with
    TUNE_COPY_STATE=1 python tools/tune_libs.py dfsph_1m 60 ref=cfd_taichi_amd/libsph_mi355x.so x=ab/libsph_nofluid.so
(only the first build advances the scene; the others receive its state).  A patch is a list of (old, new) text replacements that must
each match exactly once in the named file: a kernel that changed under a patch fails loudly instead of measuring something else."""
import os
os.environ.setdefault("SPH_DEV", "1")     # tools run with development overrides enabled (sph_overrides reports them)
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cfd_taichi_amd import build as hip_build  # noqa: E402

K, R = "sph_kernels.h", "sph_relaxed_kernels.h"
PATCHES = {
    # exact k_residual (sph_kernels.h): the fluid pair loop / the wall loop removed
    "nofluid": [(K, "    else if (staged && (RIGID ? stage_lists16(stage_cnt, blk) : c.nl16 != 0)) for_staged16_nbrs_pv2(nlp, skip ? 0 : kf, s_operand, s_v2, pair_scaled);",
                 "    else if (staged && (RIGID ? stage_lists16(stage_cnt, blk) : c.nl16 != 0)) for_staged16_nbrs_pv2(nlp, 0, s_operand, s_v2, pair_scaled);")],
    "nowall": [(K, "    else for_nbrs_p(nlbp, skip ? 0 : kb, WP, wall);\n    float val = 0.f, kr = 0.f;", "    else for_nbrs_p(nlbp, 0, WP, wall);\n    float val = 0.f, kr = 0.f;")],
    "nowall_correct": [(K, "    else for_nbrs_p(nlbp, kb, WP, wall);\n    if (track) {       // did any lane", "    else for_nbrs_p(nlbp, 0, WP, wall);\n    if (track) {       // did any lane")],
    # CORRECT variants (speed only): the solver-loop sweeps take their tiles in chunks of C consecutive tiles dealt round-robin over the XCDs
    # instead of one contiguous eighth per XCD (1.5-2 % faster at 16-64, 19 % more HBM traffic: not in the product)
    "xcd_chunk16": [(K, "    return xcd_block(orig, nwg);\n}\n\n// Edge / interior split of a sweep",
                    "    const int C = 16, xcd = orig & 7, s = orig >> 3, per = nwg / (8 * C);\n    if (per < 4) return xcd_block(orig, nwg);\n    if (s >= per * C) return orig;\n    return ((s / C) * 8 + xcd) * C + s % C;\n}\n\n// Edge / interior split of a sweep")],
    "xcd_chunk32": [(K, "    return xcd_block(orig, nwg);\n}\n\n// Edge / interior split of a sweep",
                    "    const int C = 32, xcd = orig & 7, s = orig >> 3, per = nwg / (8 * C);\n    if (per < 4) return xcd_block(orig, nwg);\n    if (s >= per * C) return orig;\n    return ((s / C) * 8 + xcd) * C + s % C;\n}\n\n// Edge / interior split of a sweep")],
    "xcd_chunk64": [(K, "    return xcd_block(orig, nwg);\n}\n\n// Edge / interior split of a sweep",
                    "    const int C = 64, xcd = orig & 7, s = orig >> 3, per = nwg / (8 * C);\n    if (per < 4) return xcd_block(orig, nwg);\n    if (s >= per * C) return orig;\n    return ((s / C) * 8 + xcd) * C + s % C;\n}\n\n// Edge / interior split of a sweep")],
    "xcd_chunk128": [(K, "    return xcd_block(orig, nwg);\n}\n\n// Edge / interior split of a sweep",
                    "    const int C = 128, xcd = orig & 7, s = orig >> 3, per = nwg / (8 * C);\n    if (per < 4) return xcd_block(orig, nwg);\n    if (s >= per * C) return orig;\n    return ((s / C) * 8 + xcd) * C + s % C;\n}\n\n// Edge / interior split of a sweep")],
    # exact k_residual: workgroups whose set did not fit the LDS capacity (1-2 % of them at 1 M) return at once: what do they cost the launch?
    "nounstaged": [(K, "(spread && direct) ? changed8 : nullptr, &would);   // positions * 2^32\n",
                    "(spread && direct) ? changed8 : nullptr, &would);   // positions * 2^32\n        if (STAGED && !staged) return;\n")],
    # a CORRECT variant with a side effect: every workgroup of the exact divergence-residual sweep leaves (begin, end, XCC id) of its life in a
    # device array that sph_debug_timeline() copies out (tools/wg_timeline.py): how full is the chip over a launch, where is the tail?
    "wg_timeline": [(K, "    else block_partial_mean(blk, (double)val, flag, psum, pcnt);\n    if (flow) {                                                      // does this tile hold",
                     "    else block_partial_mean(blk, (double)val, flag, psum, pcnt);\n    if (!DENS && threadIdx.x == 0 && blockIdx.x < 16384) {\n        unsigned xcc; asm volatile(\"s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)\" : \"=s\"(xcc));\n        g_timeline[blockIdx.x * 4 + 0] = t_begin; g_timeline[blockIdx.x * 4 + 1] = wall_clock64(); g_timeline[blockIdx.x * 4 + 2] = (xcc & 15) | ((t_staged - t_begin) << 8) | ((t_pairs - t_begin) << 24) | ((t_walls - t_begin) << 40); g_timeline[blockIdx.x * 4 + 3] = (unsigned long long)blk;\n    }\n    if (flow) {                                                      // does this tile hold"),
                    (K, "    extern __shared__ float4 s_operand[];\n    // (see k_correct: round-robin tiles when most of them return at once; the body",
                     "    extern __shared__ float4 s_operand[];\n    const unsigned long long t_begin = wall_clock64();\n    // (see k_correct: round-robin tiles when most of them return at once; the body"),
                    (K, "// D3 / D6: divergence residual and predicted density.", "__device__ unsigned long long g_timeline[16384 * 4];\n// D3 / D6: divergence residual and predicted density."),
                    (K, "    const float4 vi = V[ii];\n    float fa[1] = {0.f};\n    float &acc = fa[0];\n    const int nq = RIGID", "    const unsigned long long t_staged = wall_clock64();\n    const float4 vi = V[ii];\n    float fa[1] = {0.f};\n    float &acc = fa[0];\n    const int nq = RIGID"),
                    (K, "    float wa[1] = {0.f};\n    float &accb = wa[0];\n    auto wall = [&](const float4 pj) {\n        float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;\n        float r = K::norm3(dx, dy, dz);\n        F3 g = K::grad_in(c, dx, dy, dz, r);\n        accb += pj.w * dot3(vi.x, vi.y, vi.z, g.x, g.y, g.z);",
                     "    __builtin_amdgcn_s_barrier();\n    const unsigned long long t_pairs = wall_clock64();\n    float wa[1] = {0.f};\n    float &accb = wa[0];\n    auto wall = [&](const float4 pj) {\n        float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;\n        float r = K::norm3(dx, dy, dz);\n        F3 g = K::grad_in(c, dx, dy, dz, r);\n        accb += pj.w * dot3(vi.x, vi.y, vi.z, g.x, g.y, g.z);"),
                    (K, "    else for_nbrs_p(nlbp, skip ? 0 : kb, WP, wall);\n    float val = 0.f, kr = 0.f;", "    else for_nbrs_p(nlbp, skip ? 0 : kb, WP, wall);\n    __builtin_amdgcn_s_barrier();\n    const unsigned long long t_walls = wall_clock64();\n    float val = 0.f, kr = 0.f;"),
                    ("sph_mi355x.hip", "int sph_tune_time(SphHandle *h, int which, unsigned lds_bytes, int reps, double *avg_us)\n{",
                     "int sph_debug_timeline(unsigned long long *out, int n)\n{\n    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sph::g_timeline), sizeof(unsigned long long) * 4 * (size_t)n);\n}\nint sph_tune_time(SphHandle *h, int which, unsigned lds_bytes, int reps, double *avg_us)\n{")],
    # the same stamps in the relaxed divergence-residual sweep (tools/wg_timeline.py with SPH_ARITH=relaxed reads the same array)
    "wg_timeline_rx": [(K, "constexpr int kStageBatch = 7;          // 7 x 256 = 1792 >= the default capacity of 1664: one trip", "__device__ unsigned long long g_sub2[16384 * 4];\nconstexpr int kStageBatch = 7;          // 7 x 256 = 1792 >= the default capacity of 1664: one trip"),
                       (K, "    if (w < 0) return -1;                                   // uniform per workgroup\n    const int nst = w & 0xffff, nruns = (w >> 16) & 0x3fff;",
                        "    if (w < 0) return -1;                                   // uniform per workgroup\n    if (threadIdx.x == 0 && blockIdx.x < 16384) g_sub2[blockIdx.x * 4 + 0] = wall_clock64();\n    const int nst = w & 0xffff, nruns = (w >> 16) & 0x3fff;"),
                       (K, "        for (int k = 0; k < n; ++k) s_idx[base + k] = rn.x + (uint32_t)k;\n    }\n    __syncthreads();\n    return nst;",
                        "        for (int k = 0; k < n; ++k) s_idx[base + k] = rn.x + (uint32_t)k;\n    }\n    if (threadIdx.x == 0 && blockIdx.x < 16384) g_sub2[blockIdx.x * 4 + 1] = wall_clock64();\n    __syncthreads();\n    return nst;"),
                       (R, "    const uint32_t *nlb = nullptr;\n    SPH_SWEEP_PROLOGUE_B(false, tile)\n    (void)nlbp;\n    float2 *s_v2",
                        "    const uint32_t *nlb = nullptr;\n    if (threadIdx.x == 0 && blockIdx.x < 16384) g_sub2[blockIdx.x * 4 + 2] = wall_clock64();\n    SPH_SWEEP_PROLOGUE_B(false, tile)\n    (void)nlbp;\n    float2 *s_v2"),
                       ("sph_mi355x.hip", "int sph_set_scalar(", "int sph_debug_sub2(unsigned long long *out, int n)\n{\n    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sph::g_sub2), sizeof(unsigned long long) * 4 * (size_t)n);\n}\nint sph_set_scalar("),
                       (K, "// both operands of the residual sweeps staged: (x, y, z, vx) and (vy, vz) -- 24 B per staged particle",
                        "__device__ unsigned long long g_sub[16384 * 4];\n// both operands of the residual sweeps staged: (x, y, z, vx) and (vy, vz) -- 24 B per staged particle"),
                       (K, "    const int nst = stage_expand(stage_runs, stage_cnt, blk, reinterpret_cast<uint32_t *>(s_A), pre);\n    if (nst < 0) return false;\n    if (nst == 0) return true;                              // a workgroup of ghosts only (slab handles): nothing to stage, uniform\n    const StageIdx x = stage_take(reinterpret_cast<const uint32_t *>(s_A), nst);\n    int any = 0;\n",
                        "    const int nst = stage_expand(stage_runs, stage_cnt, blk, reinterpret_cast<uint32_t *>(s_A), pre);\n    const unsigned long long ts1 = wall_clock64();\n    if (nst < 0) return false;\n    if (nst == 0) return true;\n    const StageIdx x = stage_take(reinterpret_cast<const uint32_t *>(s_A), nst);\n    const unsigned long long ts2 = wall_clock64();\n    if (threadIdx.x == 0 && blockIdx.x < 16384) { g_sub[blockIdx.x * 4 + 0] = ts1; g_sub[blockIdx.x * 4 + 1] = ts2; }\n    int any = 0;\n"),
                       (K, "    if (changed) *any_changed = __syncthreads_or(any);\n    else __syncthreads();\n    return true;\n}\n// The same with a look at a per-particle byte first",
                        "    if (threadIdx.x == 0 && blockIdx.x < 16384) g_sub[blockIdx.x * 4 + 2] = wall_clock64();\n    if (changed) *any_changed = __syncthreads_or(any);\n    else __syncthreads();\n    return true;\n}\n// The same with a look at a per-particle byte first"),
                       ("sph_mi355x.hip", "int sph_tune_time(SphHandle *h, int which, unsigned lds_bytes, int reps, double *avg_us)\n{",
                        "int sph_debug_sub(unsigned long long *out, int n)\n{\n    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sph::g_sub), sizeof(unsigned long long) * 4 * (size_t)n);\n}\nint sph_tune_time(SphHandle *h, int which, unsigned lds_bytes, int reps, double *avg_us)\n{"),
                       (K, "// D3 / D6: divergence residual and predicted density.", "__device__ unsigned long long g_timeline[16384 * 4];\n// D3 / D6: divergence residual and predicted density."),
                       ("sph_mi355x.hip", "int sph_tune_time(SphHandle *h, int which, unsigned lds_bytes, int reps, double *avg_us)\n{",
                        "int sph_debug_timeline(unsigned long long *out, int n)\n{\n    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sph::g_timeline), sizeof(unsigned long long) * 4 * (size_t)n);\n}\nint sph_tune_time(SphHandle *h, int which, unsigned lds_bytes, int reps, double *avg_us)\n{"),
                       (R, "    extern __shared__ float4 s_operand[];\n    const bool spread = DENS && wave_dirty && !force_all;           // (round-robin",
                        "    extern __shared__ float4 s_operand[];\n    const unsigned long long t_begin = wall_clock64();\n    const bool spread = DENS && wave_dirty && !force_all;           // (round-robin"),
                       (R, "    const float4 vi = V[ii];\n    float acc = 0.f;\n    const bool skip = !DENS && kf < 20;", "    const unsigned long long t_staged = wall_clock64();\n    const float4 vi = V[ii];\n    float acc = 0.f;\n    const bool skip = !DENS && kf < 20;"),
                       (R, "    float val = 0.f, kr = 0.f;\n    int flag = 0;\n    if (live) {\n        float sum = acc;", "    __builtin_amdgcn_s_barrier();\n    const unsigned long long t_pairs = wall_clock64();\n    const unsigned long long t_walls = t_pairs;\n    float val = 0.f, kr = 0.f;\n    int flag = 0;\n    if (live) {\n        float sum = acc;"),
                       (R, "    block_partial_mean(blk, (double)val, flag, psum, pcnt);\n    if (flow) {\n        const int nzf",
                        "    block_partial_mean(blk, (double)val, flag, psum, pcnt);\n    if (!DENS && threadIdx.x == 0 && blockIdx.x < 16384) {\n        unsigned xcc; asm volatile(\"s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)\" : \"=s\"(xcc));\n        g_timeline[blockIdx.x * 4 + 0] = t_begin; g_timeline[blockIdx.x * 4 + 1] = wall_clock64(); g_timeline[blockIdx.x * 4 + 2] = (xcc & 15) | ((t_staged - t_begin) << 8) | ((t_pairs - t_begin) << 24) | ((t_walls - t_begin) << 40); g_timeline[blockIdx.x * 4 + 3] = (unsigned long long)blk;\n    }\n    if (flow) {\n        const int nzf")],
    # a CORRECT variant with a side effect (round 6): every workgroup of ONE density-loop launch pair -- the residual sweep D6 and the correction sweep D7 whose
    # DensFlow stamps the host names (sph_debug_dens_capture) -- leaves (begin, end, outcome | XCC id, tile): outcome 0 = left after the need word, 1 = left
    # after the per-particle check, 2 = worked (tools/dens_timeline.py: where does a sparse launch's time go?)
    "dens_timeline": [(K, "constexpr DensFlow kNoFlow{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, nullptr};",
                       "constexpr DensFlow kNoFlow{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, nullptr};\n__device__ unsigned long long g_tl6[16384 * 4], g_tl7[16384 * 4];\n__device__ int g_tl_capture[2];\n"
                       "__device__ __forceinline__ void tl_note(unsigned long long *buf, int want, const DensFlow &df, unsigned long long t_begin, int outcome, int tile, int extra = 0, unsigned long long ph = 0ull)\n{\n"
                       "    if (df.nbr == nullptr || df.stamp_out != want || threadIdx.x != 0 || blockIdx.x >= 16384) return;\n"
                       "    unsigned xcc; asm volatile(\"s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)\" : \"=s\"(xcc));\n"
                       "    buf[blockIdx.x * 4 + 0] = t_begin; buf[blockIdx.x * 4 + 1] = wall_clock64(); buf[blockIdx.x * 4 + 2] = (unsigned long long)outcome | ((unsigned long long)(xcc & 15) << 8) | ((unsigned long long)(unsigned)extra << 32); buf[blockIdx.x * 4 + 3] = (unsigned long long)tile | (ph << 16);\n}\n"),
                      (K, "    if (fr.mode >= 0 && blockIdx.x == 0) { fin_ride_block(fr); return; }          // the loop decision of the residual sweep before (whatever the gate says)\n",
                       "    const unsigned long long t_begin = wall_clock64();\n    if (fr.mode >= 0 && blockIdx.x == 0) { fin_ride_block(fr); return; }\n"),
                      (K, "            if (direct && threadIdx.x == 0) df.worked[tile] = 0;\n            return;\n        }\n    }\n    const int my_nbr = (flow && threadIdx.x < 64) ? df.nbr[(size_t)tile * kNbrStride + threadIdx.x] : 0;       // requested now, used by the push at the end",
                       "            if (direct && threadIdx.x == 0) df.worked[tile] = 0;\n            tl_note(g_tl7, g_tl_capture[1], df, t_begin, 0, tile);\n            return;\n        }\n    }\n    const int my_nbr = (flow && threadIdx.x < 64) ? df.nbr[(size_t)tile * kNbrStride + threadIdx.x] : 0;"),
                      (K, "            if (live) changed8[i] = foreign ? 1 : 0;\n            return;\n        }\n        staged = verdict == 1;",
                       "            if (live) changed8[i] = foreign ? 1 : 0;\n            if (MODE == CORR_DENS) tl_note(g_tl7, g_tl_capture[1], df, t_begin, 1, tile);\n            return;\n        }\n        staged = verdict == 1;"),
                      (K, "            if (threadIdx.x == 0) df.worked[tile] = moved ? 1 : 0;\n        }\n    }\n    if (!owner) return;",
                       "            if (threadIdx.x == 0) df.worked[tile] = moved ? 1 : 0;\n        }\n    }\n    if (MODE == CORR_DENS) { __builtin_amdgcn_s_barrier(); const unsigned long long t_w = wall_clock64(); tl_note(g_tl7, g_tl_capture[1], df, t_begin, 2, tile, STAGED ? stage_cnt[tile] : 0, ((t_st - t_begin) & 0xffff) | (((t_pr - t_begin) & 0xffff) << 16) | (((t_w - t_begin) & 0xffff) << 32)); }\n    if (!owner) return;"),
                      (K, "    extern __shared__ float4 s_operand[];\n    // (see k_correct: round-robin tiles when most of them return at once; the body",
                       "    extern __shared__ float4 s_operand[];\n    const unsigned long long t_begin = wall_clock64();\n    // (see k_correct: round-robin tiles when most of them return at once; the body"),
                      (K, "        if (idle) return;                                            // rho*, k / rho and the block partial of the last iteration stand",
                       "        if (idle) { tl_note(g_tl6, g_tl_capture[0], df, t_begin, 0, tile); return; }"),
                      (K, "            if (tp.hot && threadIdx.x == 0) tp.hot[blk] = 1;\n            return;\n        }\n        staged = verdict == 1;",
                       "            if (tp.hot && threadIdx.x == 0) tp.hot[blk] = 1;\n            tl_note(g_tl6, g_tl_capture[0], df, t_begin, 1, tile);\n            return;\n        }\n        staged = verdict == 1;"),
                      (K, "    if (flow) {                                                      // does this tile hold a k / rho != 0?  (a NaN counts)",
                       "    if (DENS) tl_note(g_tl6, g_tl_capture[0], df, t_begin, 2, tile, STAGED ? stage_cnt[tile] : 0, ((t_st - t_begin) & 0xffff) | (((t_pr - t_begin) & 0xffff) << 16) | (((t_w - t_begin) & 0xffff) << 32));\n    if (flow) {                                                      // does this tile hold a k / rho != 0?  (a NaN counts)"),
                      (K, "    const float dt = ds->dt;\n    const float rho_i = rho[ii];\n    float k_i;\n    if (MODE == CORR_WARM) k_i = warm[ii] / dt;                                   // :333",
                       "    __builtin_amdgcn_s_barrier();\n    const unsigned long long t_st = wall_clock64();\n    const float dt = ds->dt;\n    const float rho_i = rho[ii];\n    float k_i;\n    if (MODE == CORR_WARM) k_i = warm[ii] / dt;                                   // :333"),
                      (K, "    float wa[3] = {0.f, 0.f, 0.f};\n    float &bx = wa[0], &by = wa[1], &bz = wa[2];\n    auto wall = [&](const float4 pj) {\n        float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;\n        float r = K::norm3(dx, dy, dz);\n        F3 g = K::grad_in(c, dx, dy, dz, r);\n        float s = pj.w * k_i / rho_i;                                             // :354 / :390 / :219",
                       "    __builtin_amdgcn_s_barrier();\n    const unsigned long long t_pr = wall_clock64();\n    float wa[3] = {0.f, 0.f, 0.f};\n    float &bx = wa[0], &by = wa[1], &bz = wa[2];\n    auto wall = [&](const float4 pj) {\n        float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;\n        float r = K::norm3(dx, dy, dz);\n        F3 g = K::grad_in(c, dx, dy, dz, r);\n        float s = pj.w * k_i / rho_i;                                             // :354 / :390 / :219"),
                      (K, "    const float4 vi = V[ii];\n    float fa[1] = {0.f};\n    float &acc = fa[0];\n    const int nq = RIGID", "    __builtin_amdgcn_s_barrier();\n    const unsigned long long t_st = wall_clock64();\n    const float4 vi = V[ii];\n    float fa[1] = {0.f};\n    float &acc = fa[0];\n    const int nq = RIGID"),
                      (K, "    float wa[1] = {0.f};\n    float &accb = wa[0];\n    auto wall = [&](const float4 pj) {\n        float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;\n        float r = K::norm3(dx, dy, dz);\n        F3 g = K::grad_in(c, dx, dy, dz, r);\n        accb += pj.w * dot3(vi.x, vi.y, vi.z, g.x, g.y, g.z);",
                       "    __builtin_amdgcn_s_barrier();\n    const unsigned long long t_pr = wall_clock64();\n    float wa[1] = {0.f};\n    float &accb = wa[0];\n    auto wall = [&](const float4 pj) {\n        float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;\n        float r = K::norm3(dx, dy, dz);\n        F3 g = K::grad_in(c, dx, dy, dz, r);\n        accb += pj.w * dot3(vi.x, vi.y, vi.z, g.x, g.y, g.z);"),
                      (K, "    else for_nbrs_p(nlbp, skip ? 0 : kb, WP, wall);\n    float val = 0.f, kr = 0.f;", "    else for_nbrs_p(nlbp, skip ? 0 : kb, WP, wall);\n    __builtin_amdgcn_s_barrier();\n    const unsigned long long t_w = wall_clock64();\n    float val = 0.f, kr = 0.f;"),
                      ("sph_mi355x.hip", "int sph_tune_time(SphHandle *h, int which, unsigned lds_bytes, int reps, double *avg_us)\n{",
                       "int sph_debug_flow_stamp(SphHandle *h) { return h->flow_stamp; }\nint sph_debug_dens_capture(int stamp6, int stamp7)\n{\n    int v[2] = {stamp6, stamp7};\n    return (int)hipMemcpyToSymbol(HIP_SYMBOL(sph::g_tl_capture), v, sizeof(v));\n}\n"
                       "int sph_debug_dens_timeline(int which, unsigned long long *out, int n)\n{\n    return which == 6 ? (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sph::g_tl6), sizeof(unsigned long long) * 4 * (size_t)n)\n                      : (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sph::g_tl7), sizeof(unsigned long long) * 4 * (size_t)n);\n}\n"
                       "int sph_tune_time(SphHandle *h, int which, unsigned lds_bytes, int reps, double *avg_us)\n{")],
    # a CORRECT variant with a side effect (round 6): every workgroup of the staged list build leaves the clock of its thread 0 at the
    # boundaries of its life -- begin, cell set built, plan written (runs, bases, tile row), after each of the three dx planes of the walk, end
    # (tools/bnl_timeline.py: which phase of k_build_nl is the time?)
    "bnl_timeline": [(K, "constexpr int kRunCap = 13;", "__device__ unsigned long long g_bnl[16384 * 8];\nconstexpr int kRunCap = 13;"),
                     (K, "    if (gate && *gate == 0) return;       // Verlet handles: the lists still hold\n    __shared__ uint32_t s_stage[4 * kBlock];",
                      "    if (gate && *gate == 0) return;       // Verlet handles: the lists still hold\n    const unsigned long long tl0 = wall_clock64();\n    unsigned long long tl1 = tl0, tl2 = tl0, tlp[3] = {tl0, tl0, tl0};\n    __shared__ uint32_t s_stage[4 * kBlock];"),
                     (K, "        __syncthreads();\n        // (2) local base of every cell of the set (table order), the ordered source list, the verdict",
                      "        __syncthreads();\n        tl1 = wall_clock64();\n        // (2) local base of every cell of the set (table order), the ordered source list, the verdict"),
                     (K, "    constexpr int CHUNK = 4;\n    const bool staged = STAGED && s_ok != 0;", "    tl2 = wall_clock64();\n    constexpr int CHUNK = 4;\n    const bool staged = STAGED && s_ok != 0;"),
                     (K, "        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\");\n        __builtin_amdgcn_wave_barrier();\n    }\n    if (walker) {\n        wf.flush(self_local);",
                      "        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\");\n        __builtin_amdgcn_wave_barrier();\n        tlp[dx + 1] = wall_clock64();\n    }\n    if (walker) {\n        wf.flush(self_local);"),
                     (K, "    note_list_lengths(c, kf, kb, ds);\n}\n\n// ---- the list build for small scenes: one wave per dx-plane",
                      "    note_list_lengths(c, kf, kb, ds);\n    if (STAGED && threadIdx.x == 0 && blockIdx.x < 16384) {\n        unsigned long long *o = g_bnl + (size_t)blockIdx.x * 8;\n        o[0] = tl0; o[1] = tl1; o[2] = tl2; o[3] = tlp[0]; o[4] = tlp[1]; o[5] = tlp[2]; o[6] = wall_clock64(); o[7] = (unsigned long long)blk | ((unsigned long long)(unsigned)nruns << 32);\n    }\n}\n\n// ---- the list build for small scenes: one wave per dx-plane"),
                     ("sph_mi355x.hip", "int sph_tune_time(SphHandle *h, int which, unsigned lds_bytes, int reps, double *avg_us)\n{",
                      "int sph_debug_bnl(unsigned long long *out, int n)\n{\n    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sph::g_bnl), sizeof(unsigned long long) * 8 * (size_t)n);\n}\nint sph_tune_time(SphHandle *h, int which, unsigned lds_bytes, int reps, double *avg_us)\n{")],
    # k_build_nl: eight candidates of a cell per trip instead of four (a CORRECT variant: same lists; round 6 A/B)
    "bnl_chunk8": [(K, "    constexpr int CHUNK = 4;\n    const bool staged = STAGED && s_ok != 0;", "    constexpr int CHUNK = 8;\n    const bool staged = STAGED && s_ok != 0;")],
    # k_build_nl at eight waves per SIMD (a CORRECT variant; the compiler must fit 96 SGPRs): does residency move the list build?  (round 6 A/B)
    "bnl_waves8": [(K, "template <bool RIGID, bool STAGED>\n__global__ __launch_bounds__(kBlock) void k_build_nl(", "template <bool RIGID, bool STAGED>\n__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_build_nl(")],
    # the staging gathers of the residual sweeps removed (plan expansion and barriers kept)
    "nogather": [(K, "        float4 a[kStageBatch], b[kStageBatch];\n#pragma unroll\n        for (int u = 0; u < kStageBatch; ++u) { a[u] = A[x.j[t][u]]; b[u] = B[x.j[t][u]]; }\n#pragma unroll\n        for (int u = 0; u < kStageBatch; ++u)\n            if (base + u * kBlock < nst) {\n                const int e = base + u * kBlock;",
                  "        float4 a[kStageBatch], b[kStageBatch];\n#pragma unroll\n        for (int u = 0; u < kStageBatch; ++u) { a[u] = make_float4((float)x.j[t][u], 0.f, 0.f, 0.f); b[u] = a[u]; }\n#pragma unroll\n        for (int u = 0; u < kStageBatch; ++u)\n            if (base + u * kBlock < nst) {\n                const int e = base + u * kBlock;"),
                 (K, "        for (int u = 0; u < kStageBatch; ++u) { a[u] = A[x.j[t][u]]; b[u] = B[x.j[t][u]]; f[u] = changed ? changed[x.j[t][u]] : (unsigned char)0; }",
                  "        for (int u = 0; u < kStageBatch; ++u) { a[u] = make_float4((float)x.j[t][u], 0.f, 0.f, 0.f); b[u] = a[u]; f[u] = 0; }")],
    # k_build_nl: every wave works its 27 cell entries out per lane (a CORRECT variant: tools/soak_libs.py holds it against the default)
    "bnl_notable": [(K, "    const bool table = nruns <= kRunCap;                                       // wave-uniform",
                     "    const bool table = false;")],
    # relaxed k_residual_rx: pair loop removed / staging gathers removed / whole staging removed (the decomposition in DESIGN.md section 4b)
    "rx_nofluid": [(R, "        rx_walk8(nlp, kfx, [&](const Nl16Group &g) {\n            float4 a[8]; float2 b[8];",
                    "        rx_walk8(nlp, 0, [&](const Nl16Group &g) {\n            float4 a[8]; float2 b[8];")],
    # relaxed sweeps with HALF the index stream (every second 16-byte group of a list is not loaded, the walk reuses the group before it: same
    # LDS gathers, same arithmetic, half the list bytes from HBM): what a 2x more compact list encoding could buy at most
    "rx_halflist": [(R, "        if (kk + 8 < cnt) jn = nl_load(base + (size_t)((kk >> 3) + 1) * 256);\n        pair8(g);",
                     "        if (kk + 8 < cnt && ((kk >> 3) & 1)) jn = nl_load(base + (size_t)((kk >> 3) + 1) * 256);\n        pair8(g);")],
    "rx_nostage": [(R, "        staged = stage_operand_pv<false>(c, s_operand, s_v2, P, V, stage_src, stage_cnt, blk, pre, (spread && direct) ? changed8 : nullptr, &would);\n",
                    "        staged = true;\n"),
                   (R, "        rx_walk8(nlp, kfx, [&](const Nl16Group &g) {\n            float4 a[8]; float2 b[8];",
                    "        rx_walk8(nlp, 0, [&](const Nl16Group &g) {\n            float4 a[8]; float2 b[8];")],
}


def main():
    names = sys.argv[1:]
    if not names or any(n not in PATCHES for n in names):
        raise SystemExit("usage: removal_build.py %s" % " | ".join(sorted(PATCHES)))
    os.makedirs(os.path.join(ROOT, "ab"), exist_ok=True)
    for name in names:
        with tempfile.TemporaryDirectory() as tmp:
            csrc = os.path.join(tmp, "cfd_taichi_amd", "csrc")
            shutil.copytree(hip_build.CSRC, csrc)
            shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
            for fname, old, new, *want in PATCHES[name]:          # (file, old, new[, matches expected: default 1])
                path = os.path.join(csrc, fname)
                text = open(path).read()
                if text.count(old) != (want[0] if want else 1):
                    raise SystemExit("patch %s: %r matches %d times in %s" % (name, old[:60], text.count(old), fname))
                open(path, "w").write(text.replace(old, new))
            out = os.path.join(ROOT, "ab", "libsph_%s.so" % name)
            cmd = [hip_build.hipcc()] + hip_build.FLAGS + [os.path.join(csrc, s) for s in hip_build.SOURCES] + ["-o", out]
            subprocess.run(cmd, check=True)
            print(out)


if __name__ == "__main__":
    main()
