// valu_issue.hip -- sustained wave64 VALU issue rate on gfx950: v_fma_f32 vs v_pk_fma_f32 (and v_pk_mul_f32 / v_pk_add_f32), at 1..8
// waves per SIMD.  The number bench.py's roofline.valu is priced against is MEASURED with this, not assumed.
//
//   hipcc --offload-arch=gfx950 -O3 tools/valu_issue.hip -o gpurun_out/valu_issue && gpurun_out/valu_issue > profiles/r02a/valu_issue.json
//
// Each lane runs ITER iterations of 16 independent accumulator chains (no dependency stalls: 16 > the 4..8 cycle VALU latency at
// any occupancy), written in inline asm so the compiler can neither fuse, reorder nor eliminate them.  One workgroup = 256 threads
// = 1 wave per SIMD of a CU; `waves per SIMD` w is set by launching CUs * w workgroups with LDS sized so that exactly w fit a CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int kChains = 16;
constexpr int kUnroll = 4;     // the 16 chains, 4 times per loop trip: 64 VALU instructions per trip + 2 SALU

template <int KIND>   // 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_pk_mul_f32, 3 v_pk_add_f32, 4 v_mul_f32, 5 v_sqrt_f32, 6 v_rcp_f32, 7 v_cndmask_b32 (vcc),
                      // 8 v_cmp_lt_f32 + s_nop 1 + v_cndmask_b32 (counted as two VALU), 9 v_add_u32, 10 v_mul_f32 with an SGPR operand,
                      // 11-14 v_fma_f32 / v_mul_f32 with DISTINCT register operands per chain (kinds 0-4 share their b, c operands over all chains)
__global__ __launch_bounds__(256) void k_issue(float *out, int iters, float seed)
{
    extern __shared__ float pad[];
    float a[kChains];
    f32x2 p[kChains];
    const float b = seed, c = 1.0f - seed;
    const f32x2 b2 = {seed, seed}, c2 = {c, c};
#pragma unroll
    for (int k = 0; k < kChains; ++k) { a[k] = threadIdx.x * 1e-3f + k; p[k] = f32x2{a[k], a[k] + 0.5f}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
#pragma unroll
            for (int k = 0; k < kChains; ++k) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
                else if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(b2), "v"(c2));
                else if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(b2));
                else if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(c2));
                else if (KIND == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
                else if (KIND == 5) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[k]));
                else if (KIND == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
                else if (KIND == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b));
                else if (KIND == 8) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[k]) : "v"(b), "v"(c) : "vcc");
                else if (KIND == 9) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
                else if (KIND == 10) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[k]) : "s"(seed));
                else if (KIND == 11) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(p[k].x), "v"(p[(k + 5) % kChains].y));   // three distinct VGPRs per instruction
                else if (KIND == 12) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(p[k].x));                                 // two distinct VGPRs
                else if (KIND == 13) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(p[k].x), "v"(p[k].y));                 // accumulate form
                else if (KIND == 14) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(p[k].x), "s"(seed));                   // two VGPRs + one SGPR
                else if (KIND == 15) asm volatile("v_sub_f32 %0, 1.0, %0" : "+v"(a[k]));                                              // inline constant
                else if (KIND == 16) asm volatile("v_mul_f32 %0, 0x40400000, %0" : "+v"(a[k]));                                       // 32-bit literal
                else if (KIND == 17) asm volatile("v_cmp_lt_f32 vcc, %0, %3\n s_nop 1\n v_cndmask_b32 %0, %0, %3, vcc\n v_cndmask_b32 %1, %1, %3, vcc\n v_cndmask_b32 %2, %2, %3, vcc"
                                                  : "+v"(a[k]), "+v"(p[k].x), "+v"(p[k].y) : "v"(b) : "vcc");                          // one compare, three selects (4 VALU)
                else if (KIND == 18) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %2, s[20:21]" : "+v"(a[k]) : "v"(b), "v"(c) : "s20", "s21");   // SGPR-pair mask (2 VALU)
                else if (KIND == 19) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[k]));
                else if (KIND == 20) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(p[k].x));
                else asm volatile("v_fma_f32 %0, -%0, %1, 0.5" : "+v"(a[k]) : "v"(p[k].x));                                            // inline constant addend + neg modifier
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kChains; ++k) s += a[k] + p[k].x + p[k].y;
    if (s == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = s + pad[threadIdx.x];
}

template <int KIND>
double run(int cus, int waves_per_simd, int iters, float *dout, double *ms_out)
{
    // LDS per workgroup so that exactly `waves_per_simd` workgroups are resident per CU (160 KiB per CU; 64 KiB max per workgroup)
    size_t lds = waves_per_simd >= 3 ? (size_t)(160 * 1024) / waves_per_simd - 512 : 0;
    if (lds > 64 * 1024) lds = 64 * 1024;
    const int grid = cus * waves_per_simd;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_issue<KIND>, dim3(grid), dim3(256), lds, 0, dout, iters / 8, 0.999f);   // warm-up (clocks)
    CHECK(hipDeviceSynchronize());
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_issue<KIND>, dim3(grid), dim3(256), lds, 0, dout, iters, 0.999f);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    *ms_out = best;
    const double wave_insts = (double)grid * 4.0 * (double)iters * kUnroll * kChains * (KIND == 8 || KIND == 18 ? 2.0 : KIND == 17 ? 4.0 : 1.0);    // 4 waves per workgroup
    return wave_insts / (best * 1e-3) / 1e9;                                          // G wave64-instructions per second
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *dout = nullptr;
    CHECK(hipMalloc((void **)&dout, (size_t)cus * 8 * 256 * 4));
    const int iters = 20000;
    const int kKinds = 22;
    const char *names[kKinds] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_mul_f32", "v_sqrt_f32", "v_rcp_f32", "v_cndmask_b32_vcc",
                                 "v_cmp_lt_f32+s_nop1+v_cndmask_b32", "v_add_u32", "v_mul_f32_sgpr",
                                 "v_fma_f32_3_distinct_vgpr", "v_mul_f32_2_distinct_vgpr", "v_fma_f32_accumulate_distinct", "v_fma_f32_2vgpr_1sgpr",
                                 "v_sub_f32_inline_const", "v_mul_f32_literal", "v_cmp+s_nop1+3x_v_cndmask", "v_cmp_e64_sgpr_pair+s_nop1+v_cndmask_e64", "v_rsq_f32", "v_max_f32",
                                 "v_fma_f32_neg_inline_addend"};
    printf("{\"device\": \"%s\", \"compute_units\": %d, \"clock_mhz_reported\": %d, \"unit\": \"G wave64-instructions/s\",\n", prop.gcnArchName, cus, prop.clockRate / 1000);
    printf(" \"nominal_issue_peak\": %.1f, \"results\": {\n", cus * 2.4);
    double best_fma = 0, best_pk = 0;
    for (int kind = 0; kind < kKinds; ++kind) {
        printf("  \"%s\": {", names[kind]);
        for (int w = 1; w <= 8; ++w) {
            if (w == 3 || w == 5 || w == 6 || w == 7) continue;
            double ms, g;
            switch (kind) {
            case 0: g = run<0>(cus, w, iters, dout, &ms); break;
            case 1: g = run<1>(cus, w, iters, dout, &ms); break;
            case 2: g = run<2>(cus, w, iters, dout, &ms); break;
            case 3: g = run<3>(cus, w, iters, dout, &ms); break;
            case 4: g = run<4>(cus, w, iters, dout, &ms); break;
            case 5: g = run<5>(cus, w, iters / 4, dout, &ms); break;
            case 6: g = run<6>(cus, w, iters / 4, dout, &ms); break;
            case 7: g = run<7>(cus, w, iters, dout, &ms); break;
            case 8: g = run<8>(cus, w, iters, dout, &ms); break;
            case 9: g = run<9>(cus, w, iters, dout, &ms); break;
            case 10: g = run<10>(cus, w, iters, dout, &ms); break;
            case 11: g = run<11>(cus, w, iters, dout, &ms); break;
            case 12: g = run<12>(cus, w, iters, dout, &ms); break;
            case 13: g = run<13>(cus, w, iters, dout, &ms); break;
            case 14: g = run<14>(cus, w, iters, dout, &ms); break;
            case 15: g = run<15>(cus, w, iters, dout, &ms); break;
            case 16: g = run<16>(cus, w, iters, dout, &ms); break;
            case 17: g = run<17>(cus, w, iters / 2, dout, &ms); break;
            case 18: g = run<18>(cus, w, iters, dout, &ms); break;
            case 19: g = run<19>(cus, w, iters / 4, dout, &ms); break;
            case 20: g = run<20>(cus, w, iters, dout, &ms); break;
            default: g = run<21>(cus, w, iters, dout, &ms); break;
            }
            if (kind == 0 && g > best_fma) best_fma = g;
            if (kind == 1 && g > best_pk) best_pk = g;
            printf("\"%d_waves_per_simd\": %.1f%s", w, g, w == 8 ? "" : ", ");
        }
        printf("}%s\n", kind == kKinds - 1 ? "" : ",");
    }
    printf(" },\n \"valu_issue_ginst_measured\": %.1f, \"fp32_tflops_unpacked_fma\": %.1f, \"fp32_tflops_packed_fma\": %.1f}\n", best_fma > best_pk ? best_fma : best_pk,
           best_fma * 128e-3, best_pk * 256e-3);
    CHECK(hipFree(dout));
    return 0;
}
