// Latency probes for the primitives of the beam-search wave (one wavefront, idle chip): cycles per dependent step.
// build: hipcc --offload-arch=gfx950 -O3 scratch/probe_lat.hip -o scratch/bin/probe_lat     run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define N 512
__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memtime(); }
__global__ void k_lds(unsigned long long* out, uint32_t seed) {
    __shared__ uint32_t a[4096];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 4096; i += 64) a[i] = (i * 2654435761u + seed) & 4095u;
    __syncthreads();
    uint32_t x = lane;
    unsigned long long t0 = now();
    for (int i = 0; i < N; ++i) x = a[x];
    unsigned long long t1 = now();
    uint32_t y = x & 4095u;
    for (int i = 0; i < N; ++i) y = atomicCAS(&a[y], 0xFFFFFFFFu, y) & 4095u;  // never matches: pure returning atomic
    unsigned long long t2 = now();
    uint32_t z = y;
    for (int i = 0; i < N; ++i) { a[(z + lane) & 4095u] = z; z = a[(z * 7u + lane) & 4095u] & 4095u; }
    unsigned long long t3 = now();
    uint32_t w = z | 1u;
    for (int i = 0; i < N; ++i) w = __umulhi(w * 2654435761u, 40961u) + 1u;
    unsigned long long t4 = now();
    // merge-loop shaped chain: ctz of a mask, readlane, compare, ballot, popcount
    float f = __uint_as_float(0x3f800000u + (w & 0xffffu) + lane);
    uint32_t acc = 0;
    unsigned long long mm = 0xFFFFFFFFFFFFFFFFull;
    for (int i = 0; i < 64; ++i) {
        const int j = __builtin_ctzll(mm);
        mm &= mm - 1;
        const float dj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, f), j));
        const unsigned long long b = __ballot(f < dj);
        acc += (uint32_t)__popcll(b);
        acc = ((int)lane == j) ? acc + 1 : acc;
    }
    unsigned long long t5 = now();
    uint32_t v = acc;
    for (int i = 0; i < N; ++i) { __syncthreads(); v += i; }
    unsigned long long t6 = now();
    uint32_t u = v;
    for (int i = 0; i < N; ++i) u = __shfl_up(u, 1) + 1u;
    unsigned long long t7 = now();
    uint32_t q = u;
    for (int i = 0; i < N; ++i) q = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)q, 0x138, 0xf, 0xf, false) + 1u;  // wave_shr:1
    unsigned long long t8 = now();
    if (lane == 0) {
        out[0] = (t1 - t0); out[1] = (t2 - t1); out[2] = (t3 - t2); out[3] = (t4 - t3); out[4] = (t5 - t4);
        out[5] = (t6 - t5); out[6] = (t7 - t6); out[7] = (t8 - t7);
        out[8] = x + y + z + w + acc + v + u + q;
    }
    if (lane == 5) out[9] = q;  // wave_shr check: lane 5 should hold (lane-chain) value
}
__global__ void k_chase(const uint32_t* __restrict__ p, uint32_t start, int n, unsigned long long* out) {
    uint32_t x = start + threadIdx.x * 0;  // whole wave follows one chain (one line per step)
    unsigned long long t0 = now();
    for (int i = 0; i < n; ++i) x = p[x];
    unsigned long long t1 = now();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = x; }
}
// 32 independent random rows of 512 B per step (the gather's shape): 16 x 16-byte loads per lane, dependent steps
__global__ void k_gather(const uint4* __restrict__ rows, const uint32_t* __restrict__ ids, int steps, uint64_t nrows,
                         unsigned long long* out) {
    const uint32_t lane = threadIdx.x, g = lane >> 3, v = lane & 7;
    uint32_t next = ids[lane];
    float acc = 0.f;
    unsigned long long t0 = now();
    for (int s = 0; s < steps; ++s) {
        uint4 r[16];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint64_t row = (uint64_t)(__shfl(next, (g + 8 * u) & 63) % nrows);
#pragma unroll
            for (int t = 0; t < 4; ++t) r[u * 4 + t] = rows[row * 32 + t * 8 + v];
        }
        uint32_t h = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) h ^= r[i].x + r[i].y * 3u + r[i].z * 5u + r[i].w * 7u;
        acc += (float)h;
        next = h * 2654435761u + lane;  // dependent: next step's rows come from this step's data
    }
    unsigned long long t1 = now();
    if (lane == 0) { out[0] = t1 - t0; }
    if (acc == 12345.f) out[1] = 1;
}
int main() {
    unsigned long long *d, h[16];
    hipMalloc(&d, 128);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_lds, dim3(1), dim3(64), 0, 0, d, 12345u + rep);
        hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
    }
    printf("per step (cycles): ds_read chain %.1f | LDS atomicCAS(rtn) chain %.1f | ds_write+ds_read %.1f | mul_lo+mul_hi chain %.1f\n",
           h[0] / (double)N, h[1] / (double)N, h[2] / (double)N, h[3] / (double)N);
    printf("merge-shaped iteration (ctz, readlane, cmp, ballot, popc, select) %.1f | __syncthreads (1 wave) %.1f | __shfl_up %.1f | dpp wave_shr %.1f (lane5=%llu)\n",
           h[4] / 64.0, h[5] / (double)N, h[6] / (double)N, h[7] / (double)N, h[9]);
    // pointer chase: L2-resident (1 MB) and HBM (4 GB)
    for (size_t bytes : {(size_t)1 << 20, (size_t)64 << 20, (size_t)4 << 30}) {
        const size_t n = bytes / 4;
        std::vector<uint32_t> hp(n);
        // one random cycle through 128-byte lines
        const size_t lines = n / 32;
        std::vector<uint32_t> perm(lines);
        for (size_t i = 0; i < lines; ++i) perm[i] = (uint32_t)i;
        uint64_t s = 88172645463325252ull;
        for (size_t i = lines - 1; i > 0; --i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; size_t j = s % (i + 1); std::swap(perm[i], perm[j]); }
        for (size_t i = 0; i < lines; ++i) hp[(size_t)perm[i] * 32] = perm[(i + 1) % lines] * 32;
        uint32_t* dp;
        hipMalloc(&dp, bytes);
        hipMemcpy(dp, hp.data(), bytes, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, 0, dp, perm[0] * 32, 2000, d);
            hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        }
        printf("pointer chase over %zu MB: %.0f cycles per dependent load\n", bytes >> 20, h[0] / 2000.0);
        if (bytes == ((size_t)4 << 30)) {
            uint32_t* ids;
            hipMalloc(&ids, 256);
            std::vector<uint32_t> hi(64);
            for (int i = 0; i < 64; ++i) hi[i] = (uint32_t)(i * 2654435761u);
            hipMemcpy(ids, hi.data(), 256, hipMemcpyHostToDevice);
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(k_gather, dim3(1), dim3(64), 0, 0, (const uint4*)dp, ids, 500, (uint64_t)(bytes / 512), d);
                hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            }
            printf("gather-shaped step (32 random 512 B rows, dependent steps): %.0f cycles per step\n", h[0] / 500.0);
            hipFree(ids);
        }
        hipFree(dp);
    }
    // wall clock vs s_memtime: how long is a cycle
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    uint32_t* dp; hipMalloc(&dp, 1 << 20);
    std::vector<uint32_t> hp(1 << 18);
    for (size_t i = 0; i < hp.size(); ++i) hp[i] = (uint32_t)((i + 32) % hp.size());
    hipMemcpy(dp, hp.data(), 1 << 20, hipMemcpyHostToDevice);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, 0, dp, 0u, 200000, d);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("s_memtime ticks per microsecond: %.1f (kernel %.3f ms, %llu ticks)\n", h[0] / (ms * 1e3), ms, h[0]);
    return 0;
}
