// grid_barrier.hip -- what a grid-wide hand-off costs INSIDE a kernel on gfx950, against the kernel boundary it would replace.
// The go / no-go measurement for a persistent-tile solver loop: such a loop replaces the three dependent
// launches of a dfsph solver iteration (correction sweep, residual sweep, single-workgroup decision) by in-kernel synchronisation -- per
// iteration at least: everybody's k / rho visible to the neighbours (1), everybody's v visible (2), the reduced residual and the decision (3).
//
//   hipcc --offload-arch=gfx950 -O3 tools/grid_barrier.hip -o gpurun_out/grid_barrier && gpurun_out/grid_barrier > profiles/r04/grid_barrier.json
//
// Measured per grid size G (workgroups of 256 threads, all co-resident: G <= 4 per CU):
//   launch        K dependent launches of an (almost) empty kernel of G workgroups -- the kernel boundary
//   barrier_1     K grid barriers inside ONE launch: thread 0 of each workgroup adds to ONE agent-scope counter, then polls it with sc1 loads
//   barrier_s     the same with the adds on 8 sharded counters (own cache lines) and the poll over all shards
//   handoff       K rounds of: every workgroup stores a 16-B payload write-through (sc1), waits for it, barrier_s, then reads the payloads of
//                 26 other workgroups (its "halo") with sc1 loads -- the visibility a tile-to-tile dependency needs on top of the barrier
// Every spin is BOUNDED (kSpinCap polls, then the workgroup raises an error flag and leaves): a wrong protocol ends the run, not the box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kSpinCap = 1 << 18;
constexpr int kShards = 8, kStride = 32;      // 32 ints = 128 B per counter

__global__ __launch_bounds__(256) void k_empty(int *sink, int round)
{
    if (threadIdx.x == 0 && blockIdx.x == 0x7fffffff) sink[0] = round;     // never true: the launch does nothing but exist
}

// one grid barrier; returns false on timeout.  `gen` = barriers completed so far (the counters only grow)
template <bool SHARDED>
__device__ __forceinline__ bool grid_barrier(int *cnt, int gen, int *err)
{
    __shared__ int s_ok;
    __syncthreads();
    if (threadIdx.x == 0) {
        int ok = 1;
        if (SHARDED) {
            __hip_atomic_fetch_add(&cnt[(blockIdx.x % kShards) * kStride], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int target = gen + 1;
            for (int sh = 0; sh < kShards && ok; ++sh) {
                const int want = target * (((int)gridDim.x + kShards - 1 - sh) / kShards);
                int spins = 0;
                while (__hip_atomic_load(&cnt[sh * kStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    if (++spins > kSpinCap) { ok = 0; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
        } else {
            __hip_atomic_fetch_add(&cnt[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int want = (gen + 1) * (int)gridDim.x;
            int spins = 0;
            while (__hip_atomic_load(&cnt[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (++spins > kSpinCap) { ok = 0; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (!ok) *err = 1;
        s_ok = ok;
    }
    __syncthreads();
    return s_ok != 0;
}

template <bool SHARDED>
__global__ __launch_bounds__(256) void k_barriers(int *cnt, int rounds, int *err)
{
    for (int r = 0; r < rounds; ++r)
        if (!grid_barrier<SHARDED>(cnt, r, err)) return;
}

__global__ __launch_bounds__(256) void k_handoff(int *cnt, int rounds, int *err, float4 *payload, float *sink)
{
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        if (threadIdx.x == 0) {
            const float v = (float)(r + 1);
            __hip_atomic_store(&payload[blockIdx.x].x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&payload[blockIdx.x].y, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (!grid_barrier<true>(cnt, r, err)) return;
        if (threadIdx.x < 26) {             // the "halo": 26 other workgroups' payloads, sc1 loads
            const int other = (int)((blockIdx.x + 1 + threadIdx.x * 7) % gridDim.x);
            const float got = __hip_atomic_load(&payload[other].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (got != (float)(r + 1)) *err = 2;          // stale payload: the protocol is wrong
            acc += got;
        }
        // (a second barrier would be needed before the payloads are overwritten; the next round's barrier is counted as that one: r + 1 values differ)
        if (!grid_barrier<true>(cnt + (kShards + 1) * kStride, r, err)) return;
    }
    if (acc == -1.f) sink[0] = acc;
}

int main()
{
    int *cnt, *err, *sink_i; float4 *payload; float *sink;
    CHECK(hipMalloc(&cnt, sizeof(int) * 4 * (kShards + 1) * kStride));
    CHECK(hipMalloc(&err, sizeof(int)));
    CHECK(hipMalloc(&sink_i, sizeof(int)));
    CHECK(hipMalloc(&sink, sizeof(float)));
    CHECK(hipMalloc(&payload, sizeof(float4) * 2048));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int rounds = 200;
    printf("{\"what\": \"dependent kernel boundary vs in-kernel grid barrier / hand-off on gfx950, us per round (median of 7 runs of %d rounds)\", \"spin_cap\": %d, \"grids\": {", rounds, kSpinCap);
    const int grids[] = {114, 455, 613, 1024};
    for (int gi = 0; gi < 4; ++gi) {
        const int G = grids[gi];
        double res[4] = {0, 0, 0, 0};
        int errs = 0;
        for (int kind = 0; kind < 4; ++kind) {
            std::vector<float> t;
            for (int rep = 0; rep < 8; ++rep) {
                CHECK(hipMemset(cnt, 0, sizeof(int) * 4 * (kShards + 1) * kStride));
                CHECK(hipMemset(err, 0, sizeof(int)));
                CHECK(hipMemset(payload, 0, sizeof(float4) * 2048));
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(a, 0));
                if (kind == 0) { for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(k_empty, dim3(G), dim3(256), 0, 0, sink_i, r); }
                else if (kind == 1) hipLaunchKernelGGL(k_barriers<false>, dim3(G), dim3(256), 0, 0, cnt, rounds, err);
                else if (kind == 2) hipLaunchKernelGGL(k_barriers<true>, dim3(G), dim3(256), 0, 0, cnt, rounds, err);
                else hipLaunchKernelGGL(k_handoff, dim3(G), dim3(256), 0, 0, cnt, rounds, err, payload, sink);
                CHECK(hipEventRecord(b, 0));
                CHECK(hipEventSynchronize(b));
                float ms = 0.f;
                CHECK(hipEventElapsedTime(&ms, a, b));
                int e = 0;
                CHECK(hipMemcpy(&e, err, sizeof(int), hipMemcpyDeviceToHost));
                errs |= e;
                if (rep > 0) t.push_back(ms * 1e3f / rounds);
            }
            std::sort(t.begin(), t.end());
            res[kind] = t[t.size() / 2];
        }
        printf("%s\"%d\": {\"launch_us\": %.2f, \"barrier_one_counter_us\": %.2f, \"barrier_8_shards_us\": %.2f, \"store_barrier_halo_loads_barrier_us\": %.2f, \"error_flag\": %d}",
               gi ? ", " : "", G, res[0], res[1], res[2], res[3], errs);
    }
    printf("}}\n");
    return 0;
}
