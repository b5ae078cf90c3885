// v_mfma_f32_32x32x16_f16 on gfx950: operand layout (lane l holds row l % 32, the 8 halfs k = 8 (l / 32) .. + 7 of a 16-wide
// K step) and its arithmetic against (a) the exact dot product in f64, (b) the ascending f32 FMA chain the f32 matrix core
// is.  G = A A^T for A = 32 x K f16.  Prints the worst |G - exact| in units of u * sum |a_ik a_jk| (u = 2^-24).
// Build: hipcc --offload-arch=gfx950 -O2 scratch/probe_mfma_f16.hip -o /tmp/bin/probe_mfma_f16
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(const _Float16* A, float* D, int K) {
    const int lane = threadIdx.x, row = lane % 32, h = lane / 32;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int s = 0; s < K / 16; ++s) {
        f16x8 a;
        for (int e = 0; e < 8; ++e) a[e] = A[row * K + 16 * s + 8 * h + e];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) {
        const int i = 8 * (r / 4) + 4 * h + r % 4, j = lane % 32;
        D[i * 32 + j] = acc[r];
    }
}
static uint32_t rs = 777;
static uint32_t rnd() { rs = rs * 1664525u + 1013904223u; return rs; }
int main() {
    for (int K : {16, 64, 768, 1536}) {
        for (int mode = 0; mode < 3; ++mode) {
            std::vector<_Float16> A(32 * K);
            for (auto& x : A) {
                float v = ((int)(rnd() % 2001) - 1000) / 1000.0f;
                if (mode == 1) v *= std::ldexp(1.0f, (int)(rnd() % 12) - 6);
                if (mode == 2) v = (float)((int)(rnd() % 9) - 4);
                x = (_Float16)v;
            }
            _Float16* dA; float* dD;
            hipMalloc(&dA, A.size() * 2); hipMalloc(&dD, 32 * 32 * 4);
            hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dD, K);
            std::vector<float> D(32 * 32);
            hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
            double worst = 0, worst_chain = 0; long same_as_chain = 0;
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) {
                    double ex = 0, ab = 0; float ch = 0.f;
                    for (int kk = 0; kk < K; ++kk) {
                        const float x = (float)A[i * K + kk], y = (float)A[j * K + kk];
                        ex += (double)x * y; ab += std::fabs((double)x * y);
                        ch = __builtin_fmaf(x, y, ch);
                    }
                    const double u = std::ldexp(1.0, -24);
                    if (ab > 0) {
                        worst = std::fmax(worst, std::fabs(D[i * 32 + j] - ex) / (u * ab));
                        worst_chain = std::fmax(worst_chain, std::fabs(ch - ex) / (u * ab));
                    }
                    same_as_chain += D[i * 32 + j] == ch;
                }
            printf("K %4d mode %d: worst |mfma - exact| = %.3f u sum|xy|   (f32 chain: %.3f)   entries equal to the chain: %ld / 1024\n",
                   K, mode, worst, worst_chain, same_as_chain);
            hipFree(dA); hipFree(dD);
        }
    }
    return 0;
}
