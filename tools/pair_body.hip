// pair_body.hip -- how fast can gfx950 run the DFSPH pair body itself?  The residual sweep's per-neighbour arithmetic (difference,
// correctly rounded norm, grad_w_scaled with its Newton divisions, velocity dot product) in a loop with its LDS operand reads and
// nothing else: no index stream, no staging, no tails.  Compares, at the sweep's occupancy (4 workgroups of 256 per CU):
//   seq    four bodies per group, each in its own EXEC-masked region (the shape of for_staged_nbrs_pv2)
//   ilp    four bodies per group in one basic block (walk_staged_pv)
//   x, c   operands gathered from LDS at scattered (x) or wave-uniform, conflict-free (c) addresses
// prints pairs per second and the equivalent microseconds for `PAIRS` pairs (default 38e6 = one sweep over dfsph_1m late in the collapse).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I cfd_taichi_amd/csrc tools/pair_body.hip -o gpurun_out/pair_body
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "sph_device.h"
using namespace sph;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int kCap = 1664;

// ---- candidate forms of the body (KIND >= 2); each must reproduce KIND 0 bit for bit --------------------------------------
struct PairK { float rh_s, h_s, kg6, neg_kg6, h, m; };
template <bool PIN>
__device__ __forceinline__ PairK pair_consts(const Consts &c)
{
    PairK k = {c.rh_s, c.h_s, c.kg6, c.neg_kg6, c.h, c.m};
    if (PIN) asm volatile("" : "+v"(k.rh_s), "+v"(k.h_s), "+v"(k.kg6), "+v"(k.neg_kg6), "+v"(k.h), "+v"(k.m));   // VGPR residents: an SGPR operand halves the issue rate
    return k;
}
__device__ __forceinline__ float sqrt_fma(float x)      // LLVM's correctly rounded f32 sqrt for the no-denormal case: rsq + 2 mul + 5 fma, no selects
{
    const float r0 = __builtin_amdgcn_rsqf(x);
    float s = x * r0, h = 0.5f * r0;
    const float e = __builtin_fmaf(-h, s, 0.5f);
    h = __builtin_fmaf(h, e, h);
    s = __builtin_fmaf(s, e, s);
    const float d = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(d, h, s);
}
template <bool FMASQRT>
__device__ __forceinline__ float norm3_scaled2(float xs, float ys, float zs)
{
    const float x = (xs * xs + ys * ys) + zs * zs;
    if (FMASQRT) return sqrt_fma(x);
    return norm3_scaled(xs, ys, zs);
}
__device__ __forceinline__ F3 grad_w_scaled2(const PairK &k, float dxs, float dys, float dzs, float rs)
{
    float q0 = rs * k.rh_s;
    float e = __builtin_fmaf(-q0, k.h_s, rs);
    float q = __builtin_fmaf(e, k.rh_s, q0);
    float q2 = q * q;
    float s1 = k.kg6 * (3.0f * q2 - 2.0f * q);
    float t = 1.0f - q;
    float s2 = k.neg_kg6 * (t * t);
    float s = q <= 0.5f ? s1 : s2;
    s = 1e-5f < q ? s : 0.0f;                                      // ONE select on the scalar instead of three on the components (a v_cndmask whose
                                                                   // mask does not come straight from the preceding v_cmp costs ~23 cycles)
    const Recip den = recip_prepare(__builtin_fmaxf(k.h * rs, 1e-30f));   // r = 0 (coincident particles): finite divisor, numerators are 0
    F3 o;
    o.x = div_shared(s * dxs, den); o.y = div_shared(s * dys, den); o.z = div_shared(s * dzs, den);
    return o;
}

template <int KIND, bool SCATTER>
__global__ __launch_bounds__(256) void k_body(Consts c, float *out, int groups, const uint32_t *idx)
{
    extern __shared__ float4 s_A[];
    float2 *s_B = reinterpret_cast<float2 *>(s_A + kCap);
    for (int e = threadIdx.x; e < kCap; e += 256) {
        const float f = (float)(e % 97) * 0.0011f;
        s_A[e] = make_float4((1.0f + f) * 0x1p32f, (2.0f - f) * 0x1p32f, (0.5f + 0.5f * f) * 0x1p32f, 0.1f * f);
        s_B[e] = make_float2(0.3f - f, 0.2f + f);
    }
    __syncthreads();
    const float sx_i = 1.03f * 0x1p32f, sy_i = 1.98f * 0x1p32f, sz_i = 0.52f * 0x1p32f;
    const float4 vi = make_float4(0.1f, -0.2f, 0.3f, 0.f);
    float acc = 0.f;
    const int cnt = groups * 4 - (threadIdx.x & 3);          // ragged like a real list
    uint32_t j0 = SCATTER ? idx[threadIdx.x] : 0u;
    const PairK pk = pair_consts<(KIND == 3 || KIND == 4)>(c);
    auto eval = [&](const float4 pa, const float2 pb) -> float {
        const float dx = sx_i - pa.x, dy = sy_i - pa.y, dz = sz_i - pa.z;
        if (KIND >= 2) {
            const float r = norm3_scaled2<(KIND == 4)>(dx, dy, dz);
            const F3 g = grad_w_scaled2(pk, dx, dy, dz, r);
            return pk.m * dot3(vi.x - pa.w, vi.y - pb.x, vi.z - pb.y, g.x, g.y, g.z);
        }
        const float r = norm3_scaled(dx, dy, dz);
        const F3 g = grad_w_scaled(c, dx, dy, dz, r);
        return c.m * dot3(vi.x - pa.w, vi.y - pb.x, vi.z - pb.y, g.x, g.y, g.z);
    };
    for (int kk = 0; kk < cnt; kk += 4) {
        uint32_t j[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { j0 = (j0 * 5u + 7u + (SCATTER ? 0u : 0u)) % kCap; j[u] = SCATTER ? j0 : (uint32_t)((kk + u) % kCap); }
        const float4 a0 = s_A[j[0]], a1 = s_A[j[1]], a2 = s_A[j[2]], a3 = s_A[j[3]];
        const float2 b0 = s_B[j[0]], b1 = s_B[j[1]], b2 = s_B[j[2]], b3 = s_B[j[3]];
        if (KIND != 1 && KIND != 5) {
            acc += eval(a0, b0);
            if (kk + 1 < cnt) acc += eval(a1, b1);
            if (kk + 2 < cnt) acc += eval(a2, b2);
            if (kk + 3 < cnt) acc += eval(a3, b3);
        } else if (KIND == 1) {
            const float t0 = eval(a0, b0), t1 = eval(a1, b1), t2 = eval(a2, b2), t3 = eval(a3, b3);
            acc += t0;
            { const float n = acc + t1; acc = kk + 1 < cnt ? n : acc; }
            { const float n = acc + t2; acc = kk + 2 < cnt ? n : acc; }
            { const float n = acc + t3; acc = kk + 3 < cnt ? n : acc; }
        } else {
            // one basic block, no selects: a term past the lane's count is ANDed to +0 with an arithmetic-shift mask (plain integer VALU,
            // no VCC), and acc + (+0) == acc bit for bit (acc is never -0)
            const float t0 = eval(a0, b0), t1 = eval(a1, b1), t2 = eval(a2, b2), t3 = eval(a3, b3);
            acc += t0;
            acc += __int_as_float(__float_as_int(t1) & ((kk + 1 - cnt) >> 31));
            acc += __int_as_float(__float_as_int(t2) & ((kk + 2 - cnt) >> 31));
            acc += __int_as_float(__float_as_int(t3) & ((kk + 3 - cnt) >> 31));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int KIND, bool SCATTER>
double run(const Consts &c, int cus, float *dout, const uint32_t *didx, int groups)
{
    const size_t lds = (size_t)kCap * 24;       // 39 KiB: four workgroups per CU, as in the sweep
    const int grid = cus * 4 * 4;               // four rounds of a full chip
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_body<KIND, SCATTER>), dim3(grid), dim3(256), lds, 0, c, dout, groups, didx);
    CHECK(hipDeviceSynchronize());
    double best = 1e30;
    for (int rep = 0; rep < 7; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_body<KIND, SCATTER>), dim3(grid), dim3(256), lds, 0, c, dout, groups, didx);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double pairs = (double)grid * 256.0 * (groups * 4 - 1.5);
    return pairs / (best * 1e-3);
}

int main(int argc, char **argv)
{
    const double target = argc > 1 ? atof(argv[1]) : 31.1e6;    // wave-level body executions x 64 of one sweep over dfsph_1m at step 60 (tools/nbr_stats.py)
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    Consts c = {};
    c.h = 0.1f; c.m = 0.125f; c.rho0 = 1000.f;
    const float pi_f = (float)3.141592653589793, h3 = c.h * (c.h * c.h);
    c.kw = 8.0f / (pi_f * h3); c.rh = 1.0f / c.h; c.rh_s = c.rh * 0x1p-32f; c.h_s = c.h * 0x1p32f;
    const float kg = 48.0f / (pi_f * h3);
    c.kg6 = kg * 6.0f; c.neg_kg6 = -kg * 6.0f;
    float *dout; uint32_t *didx;
    CHECK(hipMalloc((void **)&dout, (size_t)cus * 16 * 256 * 4));
    CHECK(hipMalloc((void **)&didx, 256 * 4));
    uint32_t hidx[256];
    for (int t = 0; t < 256; ++t) hidx[t] = (uint32_t)((t * 2654435761u) % kCap);
    CHECK(hipMemcpy(didx, hidx, sizeof(hidx), hipMemcpyHostToDevice));
    const int groups = 250;
    const int nout = cus * 16 * 256;
    std::vector<float> ref(nout), got(nout);
    const char *names[6] = {"seq (current body)", "one basic block", "seq, gate on the scalar + max(den)", "+ constants in VGPRs", "+ fma square root",
                            "one basic block, masked terms (no selects), gate on the scalar"};
    printf("{\"pairs_per_sweep\": %.3g, \"results\": {\n", target);
    for (int kind = 0; kind < 6; ++kind) {
        double rx, rc;
        switch (kind) {
        case 0: rx = run<0, true>(c, cus, dout, didx, groups); break;
        case 1: rx = run<1, true>(c, cus, dout, didx, groups); break;
        case 2: rx = run<2, true>(c, cus, dout, didx, groups); break;
        case 3: rx = run<3, true>(c, cus, dout, didx, groups); break;
        case 4: rx = run<4, true>(c, cus, dout, didx, groups); break;
        default: rx = run<5, true>(c, cus, dout, didx, groups); break;
        }
        CHECK(hipMemcpy(got.data(), dout, (size_t)nout * 4, hipMemcpyDeviceToHost));
        if (kind == 0) ref = got;
        size_t bad = 0;
        for (int i = 0; i < nout; ++i) bad += memcmp(&ref[i], &got[i], 4) != 0;
        switch (kind) {
        case 0: rc = run<0, false>(c, cus, dout, didx, groups); break;
        case 1: rc = run<1, false>(c, cus, dout, didx, groups); break;
        case 2: rc = run<2, false>(c, cus, dout, didx, groups); break;
        case 3: rc = run<3, false>(c, cus, dout, didx, groups); break;
        case 4: rc = run<4, false>(c, cus, dout, didx, groups); break;
        default: rc = run<5, false>(c, cus, dout, didx, groups); break;
        }
        printf("  \"%s\": {\"scattered_us\": %.1f, \"conflict_free_us\": %.1f, \"outputs_differing_from_current\": %zu}%s\n", names[kind], target / rx * 1e6,
               target / rc * 1e6, bad, kind == 5 ? "" : ",");
    }
    printf("}}\n");
    return 0;
}
