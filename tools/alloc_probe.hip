// Does the speed of a latency-bound gather depend on WHICH allocation it reads?  hipcc --offload-arch=gfx950 -O3 tools/alloc_probe.hip -o ab/alloc_probe
// For a number of buffers (separately allocated, or carved from one large allocation) time (a) a gather of float4 through a
// pseudo-random but local index pattern (like the neighbour sweeps: 16 MiB table, indices within +-4096 of the reader),
// (b) a streaming read.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

__global__ void k_gather(const float4 *__restrict__ tab, const uint32_t *__restrict__ idx, float *__restrict__ out, int n, int k)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float acc = 0.f;
    for (int q = 0; q < k; ++q) {
        uint32_t j = idx[(size_t)q * n + i];
        float4 v = tab[j];
        acc += v.x + v.w;
    }
    out[i] = acc;
}

__global__ void k_fill_idx(uint32_t *idx, int n, int k)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s = i * 2654435761u + 12345u;
    for (int q = 0; q < k; ++q) {
        s = s * 1664525u + 1013904223u;
        int off = (int)((s >> 8) % 8192u) - 4096;
        int j = i + off;
        if (j < 0) j = 0;
        if (j >= n) j = n - 1;
        idx[(size_t)q * n + i] = (uint32_t)j;
    }
}

int main(int argc, char **argv)
{
    const int nbuf = argc > 1 ? atoi(argv[1]) : 8;
    const int carve = argc > 2 ? atoi(argv[2]) : 0;      // 1: carve all buffers from one allocation
    const int n = 1 << 20, k = 40;
    const size_t tab_bytes = (size_t)n * 16, idx_bytes = (size_t)n * k * 4, out_bytes = (size_t)n * 4;
    const size_t per = ((tab_bytes + idx_bytes + out_bytes) + (2u << 20) - 1) / (2u << 20) * (2u << 20) + (4u << 20);
    char *big = nullptr;
    if (carve) hipMalloc((void **)&big, per * nbuf);
    std::vector<float4 *> tab(nbuf); std::vector<uint32_t *> idx(nbuf); std::vector<float *> out(nbuf);
    for (int b = 0; b < nbuf; ++b) {
        if (carve) {
            char *base = big + per * b;
            tab[b] = (float4 *)base; idx[b] = (uint32_t *)(base + (tab_bytes + (2u << 20) - 1) / (2u << 20) * (2u << 20)); out[b] = (float *)((char *)idx[b] + idx_bytes);
        } else {
            hipMalloc((void **)&tab[b], tab_bytes); hipMalloc((void **)&idx[b], idx_bytes); hipMalloc((void **)&out[b], out_bytes);
        }
        hipMemset(tab[b], 0, tab_bytes);
        k_fill_idx<<<n / 256, 256>>>(idx[b], n, k);
    }
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<std::vector<float>> t(nbuf);
    for (int round = 0; round < 12; ++round)
        for (int bb = 0; bb < nbuf; ++bb) {
            int b = (bb * 5 + round * 3) % nbuf;
            hipEventRecord(e0);
            for (int r = 0; r < 5; ++r) k_gather<<<n / 256, 256>>>(tab[b], idx[b], out[b], n, k);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            t[b].push_back(ms * 1000.f / 5);
        }
    for (int b = 0; b < nbuf; ++b) {
        std::sort(t[b].begin(), t[b].end());
        printf("buf %d  tab %p idx %p  gather median %.1f us\n", b, (void *)tab[b], (void *)idx[b], t[b][t[b].size() / 2]);
    }
    return 0;
}
