// Is the f32 matrix core an IEEE FMA chain in k order?  For v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32:
// D = C + sum_k A[i][k] B[k][j] compared bit for bit with  fma(a_{K-1}, b_{K-1}, ... fma(a_0, b_0, c))  (ascending k),
// the descending chain, and the unfused forms -- on random data with wide exponent spread, denormals included.
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off scratch/probe_mfma_exact.hip -o scratch/bin/probe_mfma_exact
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k32(const float* A, const float* B, const float* C, float* D, int chain) {
    // A: 32 x K row-major (K = 2 * chain), B: K x 32, C/D: 32 x 32
    const int lane = threadIdx.x, K = 2 * chain;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) {
        const int i = 8 * (r / 4) + 4 * (lane / 32) + r % 4, j = lane % 32;
        acc[r] = C[i * 32 + j];
    }
    for (int s = 0; s < chain; ++s) {
        const float a = A[(lane % 32) * K + 2 * s + lane / 32];
        const float b = B[(2 * s + lane / 32) * 32 + lane % 32];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) {
        const int i = 8 * (r / 4) + 4 * (lane / 32) + r % 4, j = lane % 32;
        D[i * 32 + j] = acc[r];
    }
}
__global__ void k16(const float* A, const float* B, const float* C, float* D, int chain) {
    // A: 16 x K (K = 4 * chain), B: K x 16, C/D: 16 x 16; lane l: a = A[l % 16][4 s + l / 16], D rows 4 (l / 16) + r, col l % 16
    const int lane = threadIdx.x, K = 4 * chain;
    f32x4 acc;
    for (int r = 0; r < 4; ++r) acc[r] = C[(4 * (lane / 16) + r) * 16 + lane % 16];
    for (int s = 0; s < chain; ++s) {
        const float a = A[(lane % 16) * K + 4 * s + lane / 16];
        const float b = B[(4 * s + lane / 16) * 16 + lane % 16];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[(4 * (lane / 16) + r) * 16 + lane % 16] = acc[r];
}
static uint32_t rs = 12345;
static uint32_t rnd() { rs = rs * 1664525u + 1013904223u; return rs; }
static float rfloat(int mode) {
    if (mode == 0) return (float)((int)(rnd() % 17) - 8);  // small integers: exact in any order (layout check)
    uint32_t sign = rnd() & 0x80000000u, mant = rnd() & 0x7FFFFFu;
    int spread = mode == 1 ? 6 : mode == 2 ? 40 : 120;
    uint32_t e = 127 - spread / 2 + rnd() % spread;
    if (mode == 4) e = rnd() % 3;  // denormals and the smallest normals
    uint32_t b = sign | (e << 23) | mant;
    float f; memcpy(&f, &b, 4); return f;
}
int main() {
    for (int shape = 0; shape < 2; ++shape) {
        const int M = shape == 0 ? 32 : 16, kstep = shape == 0 ? 2 : 4;
        for (int mode = 0; mode < 5; ++mode) {
            for (int chain : {1, 4, 64}) {
                const int K = kstep * chain;
                long bad_asc = 0, bad_desc = 0, bad_unfused = 0, bad_pair = 0, total = 0;
                for (int rep = 0; rep < 20; ++rep) {
                    std::vector<float> A(M * K), B(K * M), Cm(M * M), Dm(M * M);
                    for (auto& x : A) x = rfloat(mode);
                    for (auto& x : B) x = rfloat(mode == 4 ? 1 : mode);
                    for (auto& x : Cm) x = rfloat(mode);
                    float *dA, *dB, *dC, *dD;
                    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, Cm.size() * 4); hipMalloc(&dD, Dm.size() * 4);
                    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
                    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
                    hipMemcpy(dC, Cm.data(), Cm.size() * 4, hipMemcpyHostToDevice);
                    if (shape == 0) hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, chain);
                    else hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, chain);
                    hipMemcpy(Dm.data(), dD, Dm.size() * 4, hipMemcpyDeviceToHost);
                    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dD);
                    for (int i = 0; i < M; ++i)
                        for (int j = 0; j < M; ++j) {
                            float asc = Cm[i * M + j], unf = asc, pr = asc;
                            for (int k = 0; k < K; ++k) {
                                asc = __builtin_fmaf(A[i * K + k], B[k * M + j], asc);
                                volatile float p = A[i * K + k] * B[k * M + j];
                                unf = unf + p;
                            }
                            // per instruction: the kstep products summed first (in f32, ascending), then added to c
                            for (int s = 0; s < chain; ++s) {
                                float t = 0.f;
                                for (int k = 0; k < kstep; ++k) t = __builtin_fmaf(A[i * K + s * kstep + k], B[(s * kstep + k) * M + j], t);
                                pr = pr + t;
                            }
                            float desc = Cm[i * M + j];
                            for (int s = 0; s < chain; ++s)
                                for (int k = kstep - 1; k >= 0; --k)
                                    desc = __builtin_fmaf(A[i * K + s * kstep + k], B[(s * kstep + k) * M + j], desc);
                            uint32_t g, a, d, u, p2;
                            memcpy(&g, &Dm[i * M + j], 4); memcpy(&a, &asc, 4); memcpy(&d, &desc, 4); memcpy(&u, &unf, 4); memcpy(&p2, &pr, 4);
                            const bool nan_both = (Dm[i * M + j] != Dm[i * M + j]) && (asc != asc);
                            bad_asc += (g != a) && !nan_both; bad_desc += g != d; bad_unfused += g != u; bad_pair += g != p2; ++total;
                        }
                }
                printf("mfma_%dx%dx%d mode %d chain %3d: of %ld entries differ from ascending-fma %ld, descending-within-instruction %ld, unfused %ld, pair-sum-first %ld\n",
                       M, M, kstep, mode, chain, total, bad_asc, bad_desc, bad_unfused, bad_pair);
            }
        }
    }
    return 0;
}
