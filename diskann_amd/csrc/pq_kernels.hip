// pq_kernels.hip -- product-quantisation lookup table build + scan.
//
//   pq_lut_kernel   == FixedChunkPQTable::populate_chunk_distances_impl
//                      diskann-providers/src/model/pq/fixed_chunk_pq_table.rs:152-192
//   pq_scan_kernel  == pq_dist_lookup_single (:82-100), batched over (query, candidate) pairs
//
// LUT entries are SquaredL2 / InnerProduct over a *chunk slice* evaluated by the reference's
// f32 SIMD kernel (PureDistanceFunction for &[f32]); chunks are short (dim / nchunks elements),
// so one thread evaluates one (chunk, centroid) entry by walking the 4-accumulator x 8-lane
// schedule sequentially -- the same emulation the CPU oracle uses, hence bit-identical.  The
// scan adds LUT entries in chunk order in f32, one lane per candidate, LUT staged in LDS.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <vector>

#include "dann_device.h"
#include "dann_internal.h"

namespace dann {
namespace {

template <bool IS_L2>
__global__ __launch_bounds__(256) void pq_lut_kernel(const float* pivots, const uint32_t* offsets, uint32_t nchunks,
                                                     uint32_t dim, const float* queries, float* lut) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float* q = reinterpret_cast<float*>(smem);
    const uint32_t qi = blockIdx.x;
    for (uint32_t i = threadIdx.x; i < dim; i += blockDim.x) q[i] = queries[(uint64_t)qi * dim + i];
    __syncthreads();
    const uint32_t total = nchunks * 256u;
    for (uint32_t t = threadIdx.x; t < total; t += blockDim.x) {
        const uint32_t chunk = t >> 8, centroid = t & 255u;
        const uint32_t s = offsets[chunk], e = offsets[chunk + 1];
        const float raw = simd_op_seq<IS_L2>(q + s, pivots + (uint64_t)centroid * dim + s, e - s);
        // PostOp<f32, f32>: SquaredL2 -> x, InnerProduct -> -x (implementations.rs:215-314)
        lut[((uint64_t)qi * nchunks + chunk) * 256u + centroid] = IS_L2 ? raw : -raw;
    }
}

__global__ __launch_bounds__(256) void pq_scan_kernel(const float* lut, uint32_t nchunks, const uint8_t* codes,
                                                      const uint32_t* ids, const uint64_t* offsets, float* out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float* l = reinterpret_cast<float*>(smem);
    const uint32_t qi = blockIdx.x;
    const uint64_t lo = offsets[qi] + (uint64_t)blockIdx.y * 4096u, hi_all = offsets[qi + 1];
    if (lo >= hi_all) return;
    const uint64_t hi = lo + 4096u < hi_all ? lo + 4096u : hi_all;
    const uint32_t n = nchunks * 256u;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) l[i] = lut[(uint64_t)qi * n + i];
    __syncthreads();
    for (uint64_t c = lo + threadIdx.x; c < hi; c += blockDim.x) {
        const uint8_t* code = codes + (uint64_t)ids[c] * nchunks;
        float accum = 0.0f;
        for (uint32_t ch = 0; ch < nchunks; ++ch) accum += l[ch * 256u + code[ch]];
        out[c] = accum;
    }
}

// ScalarQuantizer::compress_into, 8 bits (quantizer.rs:189-236, 395-430); one thread per vector:
// the compensation is a sequential FMA chain over the dimensions.
__global__ void sq8_compress_kernel(const float* x, uint32_t n, uint32_t dim, const float* shift, float scale,
                                    uint8_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float inverse_scale = 255.0f / scale;
    const float* v = x + (uint64_t)i * dim;
    uint8_t* o = out + (uint64_t)i * (dim + 4);
    float dot = 0.0f;
    for (uint32_t d = 0; d < dim; ++d) {
        float c = (v[d] - shift[d]) * inverse_scale;
        c = c < 0.0f ? 0.0f : (c > 255.0f ? 255.0f : c);  // NaN stays NaN -> code 0 (`as u8`)
        c = roundf(c);
        dot = __builtin_fmaf(c, shift[d], dot);
        o[d] = (c != c) ? 0 : (uint8_t)c;
    }
    const float comp = scale * (1.0f / 255.0f) * dot;
    const uint32_t u = __builtin_bit_cast(uint32_t, comp);
    o[dim] = (uint8_t)u;
    o[dim + 1] = (uint8_t)(u >> 8);
    o[dim + 2] = (uint8_t)(u >> 16);
    o[dim + 3] = (uint8_t)(u >> 24);
}

// kmeans::square_norm (diskann-quantization/src/algorithms/kmeans/common.rs:8-62): four 8-lane accumulators over
// 32-element trips, combined (s0+s1)+(s2+s3), remaining 8-blocks and the zero-padded tail into the combined vector,
// then sum_tree.  One thread emulates the 8 lanes.
__device__ float pq_square_norm(const float* x, uint32_t len) {
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t i = 0;
    if (i + 32 <= len) {
        float a[4][8];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int l = 0; l < 8; ++l) a[b][l] = 0.0f;
        while (i + 32 <= len) {
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int l = 0; l < 8; ++l) {
                    const float v = x[i + 8 * b + l];
                    a[b][l] = __builtin_fmaf(v, v, a[b][l]);
                }
            i += 32;
        }
#pragma unroll
        for (int l = 0; l < 8; ++l) s[l] = (a[0][l] + a[1][l]) + (a[2][l] + a[3][l]);
    }
    while (i + 8 <= len) {
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            const float v = x[i + l];
            s[l] = __builtin_fmaf(v, v, s[l]);
        }
        i += 8;
    }
    if (len - i) {
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            const float v = (i + l < len) ? x[i + l] : 0.0f;
            s[l] = __builtin_fmaf(v, v, s[l]);
        }
    }
    return ((s[0] + s[4]) + (s[2] + s[6])) + ((s[1] + s[5]) + (s[3] + s[7]));
}

// TransposedTable::compress_into -> Chunk::find_closest (product/tables/transposed/table.rs:382-403,
// pivots.rs:253-345): block = 256 rows x one chunk; the chunk's pivot slab and norms live in LDS (every thread
// reads the same pivot element: broadcast), each thread keeps the reference's 8 lane-wise running minima.
__global__ __launch_bounds__(256) void pq_compress_kernel(const float* pivots, uint32_t ncenters, const uint32_t* offsets,
                                                          uint32_t nchunks, uint32_t dim, const float* rows, uint64_t n,
                                                          uint8_t* codes, unsigned long long* first_bad) {
    extern __shared__ __attribute__((aligned(16))) float pq_smem[];
    const uint32_t c = blockIdx.y;
    const uint32_t s0 = offsets[c], len = offsets[c + 1] - s0;
    float* slab = pq_smem;                  // ncenters x len
    float* norms = pq_smem + ncenters * len;  // ncenters
    for (uint32_t t = threadIdx.x; t < ncenters * len; t += blockDim.x)
        slab[t] = pivots[(uint64_t)(t / len) * dim + s0 + (t % len)];
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < ncenters; j += blockDim.x) norms[j] = pq_square_norm(slab + j * len, len);
    __syncthreads();
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const float* x = rows + r * dim + s0;
    float best_d[8];
    uint32_t best_i[8];
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        best_d[l] = __builtin_inff();
        best_i[l] = 0xFFFFFFFFu;
    }
    for (uint32_t j0 = 0; j0 < ncenters; j0 += 8) {
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            const uint32_t j = j0 + l;
            if (j < ncenters) {
                const float* pj = slab + j * len;
                float ip = 0.0f;
                for (uint32_t d = 0; d < len; ++d) ip = __builtin_fmaf(x[d], pj[d], ip);
                const float score = norms[j] - (ip + ip);
                if (score < best_d[l]) {
                    best_d[l] = score;
                    best_i[l] = j;
                }
            }
        }
    }
    float md = 3.402823466e+38f;
    uint32_t mi = 0xFFFFFFFFu;
#pragma unroll
    for (int l = 0; l < 8; ++l)
        if (best_d[l] < md) {
            md = best_d[l];
            mi = best_i[l];
        }
    const bool finite = (md - md) == 0.0f;
    if (!finite || mi == 0xFFFFFFFFu) {
        atomicMin(first_bad, r * nchunks + c);
        mi = 0;
    }
    codes[r * nchunks + c] = (uint8_t)mi;
}

// ---- k-means++ seeding of the PQ trainer (kmeans::plusplus::kmeans_plusplus_into_inner, plusplus.rs:366-497) --------
// update_distances (:239-311) with the BlockTransposed<f32, 16> micro-kernel (:87-237): one thread per row evaluates
// the reference's per-row chain (fma over the chunk's columns in order, * -2, (norm + |centre|^2) + that, strict-<
// minimum); the 16 rows of a block fold their minima in the reference's order (lane k with lane k + 8, the 8 pair sums
// left to right, in f64).  A second kernel adds the block sums sequentially (the reference's rolling f64 sum).
__global__ __launch_bounds__(256) void kpp_update_kernel(const float* data, uint64_t n, uint32_t dim, const uint32_t* offsets,
                                                         const float* norms, const float* last, const float* last_norm,
                                                         float* mins, double* block_sums, uint64_t nblocks) {
    __shared__ float cur[256];
    const uint32_t c = blockIdx.y;
    const uint64_t row = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t s0 = offsets[c], len = offsets[c + 1] - s0;
    float v = 0.0f;  // lanes past the end hold 0 and stay 0 (finish_last)
    if (row < n) {
        const float* x = data + row * dim + s0;
        const float* l = last + (uint64_t)c * dim + s0;
        float acc = 0.0f;
        for (uint32_t k = 0; k < len; ++k) acc = __builtin_fmaf(x[k], l[k], acc);
        acc = acc * -2.0f;
        const float d = (norms[(uint64_t)c * n + row] + last_norm[c]) + acc;
        float m = mins[(uint64_t)c * n + row];
        if (d < m) {
            m = d;
            mins[(uint64_t)c * n + row] = m;
        }
        v = m;
    }
    cur[threadIdx.x] = v;
    __syncthreads();
    if ((threadIdx.x & 15u) == 0) {
        const uint64_t b = row >> 4;
        if (b < nblocks) {
            double blk = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) blk += (double)cur[threadIdx.x + k] + (double)cur[threadIdx.x + k + 8];
            block_sums[(uint64_t)c * nblocks + b] = blk;
        }
    }
}
// copy the chosen rows' chunk columns into centre `cur` and publish them as the next `last` (+ their norms)
__global__ void kpp_commit_kernel(const float* data, uint64_t n, uint32_t dim, const uint32_t* offsets, const float* norms,
                                  const int64_t* chosen, uint32_t cur, float* centers, float* last, float* last_norm) {
    const uint32_t c = blockIdx.x;
    const int64_t i = chosen[c];
    if (i < 0) return;
    const uint32_t s0 = offsets[c], len = offsets[c + 1] - s0;
    for (uint32_t k = threadIdx.x; k < len; k += blockDim.x) {
        const float v = data[(uint64_t)i * dim + s0 + k];
        centers[(uint64_t)cur * dim + s0 + k] = v;
        last[(uint64_t)c * dim + s0 + k] = v;
    }
    if (threadIdx.x == 0) last_norm[c] = norms[(uint64_t)c * n + (uint64_t)i];
}
__global__ void kpp_fill_kernel(float* v, uint64_t n, float x) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = x;
}

// ---- Lloyd iterations of the PQ trainer (product/train.rs:96-226, kmeans/lloyds.rs:23-438) ----------------
// |x|^2 of every (row, chunk): kmeans::square_norm
__global__ __launch_bounds__(256) void pq_data_norms_kernel(const float* data, uint64_t n, uint32_t dim,
                                                            const uint32_t* offsets, float* norms) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t c = blockIdx.y;
    if (r >= n) return;
    norms[(uint64_t)c * n + r] = pq_square_norm(data + r * dim + offsets[c], offsets[c + 1] - offsets[c]);
}
__global__ void pq_center_norms_kernel(const float* centers, uint32_t ncenters, uint32_t dim, const uint32_t* offsets,
                                       float* norms) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
    if (j >= ncenters) return;
    norms[c * ncenters + j] = pq_square_norm(centers + (uint64_t)j * dim + offsets[c], offsets[c + 1] - offsets[c]);
}
// distances_in_place (lloyds.rs:23-260): score = ((n_c - ip) - ip) + |x|^2, first strictly smaller centre wins
__global__ __launch_bounds__(256) void pq_assign_kernel(const float* data, uint64_t n, uint32_t dim,
                                                        const uint32_t* offsets, const float* centers, uint32_t ncenters,
                                                        const float* cnorms, const float* dnorms, uint32_t* assign,
                                                        float* best_out) {
    extern __shared__ __attribute__((aligned(16))) float pq_smem[];
    const uint32_t c = blockIdx.y;
    const uint32_t s0 = offsets[c], len = offsets[c + 1] - s0;
    float* slab = pq_smem;
    float* cn = pq_smem + ncenters * len;
    for (uint32_t t = threadIdx.x; t < ncenters * len; t += blockDim.x)
        slab[t] = centers[(uint64_t)(t / len) * dim + s0 + (t % len)];
    for (uint32_t j = threadIdx.x; j < ncenters; j += blockDim.x) cn[j] = cnorms[c * ncenters + j];
    __syncthreads();
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const float* x = data + r * dim + s0;
    const float dn = dnorms[(uint64_t)c * n + r];
    float best = __builtin_inff();
    uint32_t bi = 0xFFFFFFFFu;
    for (uint32_t j = 0; j < ncenters; ++j) {
        const float* pj = slab + j * len;
        float ip = 0.0f;
        for (uint32_t d = 0; d < len; ++d) ip = __builtin_fmaf(pj[d], x[d], ip);
        const float sc = ((cn[j] - ip) - ip) + dn;
        if (sc < best) {
            best = sc;
            bi = j;
        }
    }
    assign[(uint64_t)c * n + r] = bi;
    best_out[(uint64_t)c * n + r] = best;
}
// The same assignment on the matrix cores ("k-means / PQ training as GEMM -> MFMA").  The reference's
// inner product of a (row, centre) pair is one FMA chain over the chunk's columns in ascending order, starting from +0.0
// (process_block_unroll_2, lloyds.rs:70-110: s = c.mul_add(d, s) per dimension), and v_mfma_f32_32x32x2_f32 IS that chain:
// D = fma(a_1, b_1, fma(a_0, b_0, C)), exact and in ascending k, denormals included (scratch/probe_mfma_exact.hip on this
// part: 0 of 20 480 entries differ from the ascending chain for every exponent spread; the descending and unfused forms
// differ in thousands).  So a 32 x 32 tile of inner products (32 centres x 32 rows) is ceil(len / 2) MFMAs on one
// accumulator set -- bit for bit the 1 024 scalar chains.  A = centres (tile row i), B = rows (tile column j): a lane
// then holds 16 centres of ONE row (C layout: register r of lane l = tile row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column
// l & 31), the score ((|c|^2 - ip) - ip) + |x|^2 and the running minimum stay in the lane, and "the first strictly
// smaller centre wins" is the lexicographic minimum of (score, centre) -- taken in the lane in ascending centre order,
// then between the lane and its partner (l ^ 32).  An odd chunk length pads the last k-pair with zeros: fma(0, 0, s) = s
// (s is never -0.0: the chain starts at +0.0).  NaN scores never win, as in the scalar kernel.
// LDS: the chunk's centres transposed ([k][centre], row stride = 32 mod 64 words: the two k of an MFMA step hit disjoint
// banks) + their norms.  One wavefront = 32 rows x all centres; four wavefronts per workgroup.
typedef float pq_f32x16 __attribute__((ext_vector_type(16)));
__host__ __device__ inline uint32_t pq_mfma_ncp(uint32_t ncenters) {
    const uint32_t t = (ncenters + 31u) & ~31u;
    return (t & 63u) == 32u ? t : t + 32u;
}
__global__ __launch_bounds__(256) void pq_assign_mfma_kernel(const float* data, uint64_t n, uint32_t dim,
                                                             const uint32_t* offsets, const float* centers,
                                                             uint32_t ncenters, const float* cnorms, const float* dnorms,
                                                             uint32_t* assign, float* best_out) {
    extern __shared__ __attribute__((aligned(16))) float pq_smem[];
    const uint32_t c = blockIdx.y;
    const uint32_t s0 = offsets[c], len = offsets[c + 1] - s0;
    const uint32_t ksteps = (len + 1u) >> 1, ncp = pq_mfma_ncp(ncenters), ntiles = (ncenters + 31u) >> 5;
    float* slab = pq_smem;                  // [2 ksteps][ncp]
    float* cn = pq_smem + 2u * ksteps * ncp;  // [32 ntiles]
    for (uint32_t t = threadIdx.x; t < 2u * ksteps * ncp; t += blockDim.x) {
        const uint32_t k = t / ncp, j = t % ncp;
        slab[t] = (k < len && j < ncenters) ? centers[(uint64_t)j * dim + s0 + k] : 0.0f;
    }
    for (uint32_t j = threadIdx.x; j < 32u * ntiles; j += blockDim.x) cn[j] = j < ncenters ? cnorms[c * ncenters + j] : 0.0f;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, l31 = lane & 31u, hi = lane >> 5;
    const uint64_t row0 = ((uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6)) * 32u;
    if (row0 >= n) return;
    const uint64_t row = row0 + l31 < n ? row0 + l31 : n - 1u;
    const float* x = data + row * dim + s0;
    const float dn = dnorms[(uint64_t)c * n + row];
    float best = __builtin_inff();
    uint32_t bi = 0xFFFFFFFFu;
    for (uint32_t t = 0; t < ntiles; ++t) {
        pq_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        const float* ap = slab + hi * ncp + (t << 5) + l31;
        for (uint32_t m = 0; m < ksteps; ++m) {
            const uint32_t k = 2u * m + hi;
            const float a = ap[2u * m * ncp];
            const float b = k < len ? x[k] : 0.0f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t j = (t << 5) + (uint32_t)((r & 3) + 8 * (r >> 2)) + 4u * hi;
            const float ip = acc[r];
            const float sc = ((cn[j] - ip) - ip) + dn;
            const bool take = (j < ncenters) & (sc < best);
            best = take ? sc : best;
            bi = take ? j : bi;
        }
    }
    // the partner lane holds the other 16 centres of each tile for the same row
    const float ob = __shfl_xor(best, 32);
    const uint32_t oi = (uint32_t)__shfl_xor((int)bi, 32);
    if ((ob < best) | ((ob == best) & (oi < bi))) {
        best = ob;
        bi = oi;
    }
    if (hi == 0u && row0 + l31 < n) {
        assign[(uint64_t)c * n + row] = bi;
        best_out[(uint64_t)c * n + row] = best;
    }
}

// ---- a sequential f64 sum, evaluated in parallel wherever that cannot change a bit ---------------------------------
// The trainer's D^2 draw and its totals are rolling f64 sums over 131 072 f32 values / 8 192 block sums in row order
// (plusplus.rs:446-462, the micro-kernel's RollingSum): one thread walking them was 2.5 of the trainer's 3.1 seconds.
// A floating-point sum is order-independent exactly when no addition rounds.  For a range of values v_i with
// g = the exponent of the lowest set bit over the v_i and the running sum acc at the range's start, and
// A = sum |v_i|: every partial sum the sequential walk forms is a multiple of 2^min(g, lsb(acc)) and at most
// |acc| + A in magnitude; if |acc| + A < 2^(min(g, lsb(acc)) + 53) they are all representable, no addition rounds,
// and the walk's result is acc + (sum of the range in ANY association).  `seq_exact` is that test (A carries a 2^-30
// margin for its own rounding; a non-finite value fails it).  Ranges that fail are walked one element at a time by one
// thread, exactly as before: a low bit pushed out when acc crosses a power of two, or a value far below the running
// sum's last place -- local events on continuous data.
struct SeqSummary {
    double S, A;
    int g;
};
constexpr int kSeqNone = 0x7FFFFFFF, kSeqBad = (int)0x80000000;
// [0] wavefront-level ranges taken in parallel, [1] thread-level ranges taken in parallel, [2] ranges walked element by
// element, [3] elements walked (dann_debug_pq_rolling_sum_stats: the tests assert that both paths ran)
__device__ unsigned long long g_seq_stats[4];
__device__ __forceinline__ int lsb_exp_f64(double v) {  // finite, non-zero
    const uint64_t b = (uint64_t)__builtin_bit_cast(long long, v);
    const int e = (int)((b >> 52) & 0x7FFu);
    uint64_t m = b & ((1ull << 52) - 1u);
    if (e == 0) return -1074 + (int)__builtin_ctzll(m);
    m |= 1ull << 52;
    return e - 1075 + (int)__builtin_ctzll(m);
}
__device__ __forceinline__ void seq_take(SeqSummary& s, double v) {
    s.S += v;
    s.A += __builtin_fabs(v);
    if (!(__builtin_fabs(v) < __builtin_inf())) s.g = kSeqBad;
    else if (v != 0.0 && s.g != kSeqBad) {
        const int g = lsb_exp_f64(v);
        s.g = g < s.g ? g : s.g;
    }
}
__device__ __forceinline__ SeqSummary seq_join(const SeqSummary& a, const SeqSummary& b) {
    SeqSummary r;
    r.S = a.S + b.S;
    r.A = a.A + b.A;
    r.g = (a.g == kSeqBad || b.g == kSeqBad) ? kSeqBad : (a.g < b.g ? a.g : b.g);
    return r;
}
__device__ __forceinline__ bool seq_exact(double acc, const SeqSummary& s) {
    if (s.g == kSeqBad || !(__builtin_fabs(acc) < __builtin_inf())) return false;
    int g = s.g;
    if (acc != 0.0) {
        const int ga = lsb_exp_f64(acc);
        g = ga < g ? ga : g;
    }
    if (g == kSeqNone) return true;  // nothing but zeros on a zero sum
    const double bound = (__builtin_fabs(acc) + s.A) * (1.0 + 0x1p-30);
    const int lim = g + 53;
    if (lim > 1023) return bound < __builtin_inf();
    if (lim < -1021) return false;
    return bound < __builtin_ldexp(1.0, lim);
}
// One workgroup of 1024 threads walks `count` values in order (LOAD(i) -> double).  Thread t owns the contiguous range
// [t seg, (t + 1) seg); wavefront w the 64 ranges of its lanes.  Returns (thread 0) the sequential sum; with SELECT also
// finds the first index whose prefix sum reaches `thr` and passes PRED(i) (the D^2 draw), UINT64_MAX if none.
// LDS: 1024 range summaries + 16 wavefront summaries + carries.
struct SeqLds {
    double S[1024], A[1024], carry[1024], wS[16], wA[16], wcarry[16];
    int g[1024], wg[16];
    uint8_t ok[1024], wok[16];
    unsigned long long hit;
    unsigned long long stop;  // the walk ended at a hit inside a range thread 0 walked itself: later ranges do not matter
};
template <bool SELECT, typename Load, typename Pred>
__device__ __forceinline__ double seq_sum_workgroup(SeqLds& L, uint64_t count, Load load, double thr, Pred pred,
                                                    unsigned long long* hit_out) {
    const uint32_t t = threadIdx.x, lane = t & 63u, w = t >> 6;
    const uint64_t seg = (count + 1023u) / 1024u;
    const uint64_t lo = (uint64_t)t * seg < count ? (uint64_t)t * seg : count;
    const uint64_t hi = lo + seg < count ? lo + seg : count;
    SeqSummary s{0.0, 0.0, kSeqNone};
    for (uint64_t i = lo; i < hi; ++i) seq_take(s, load(i));
    L.S[t] = s.S;
    L.A[t] = s.A;
    L.g[t] = s.g;
    // the wavefront's summary and the exclusive prefix of its lanes' sums (both exact where the wavefront's test passes)
    SeqSummary ws = s;
    double incl = s.S;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        SeqSummary o;
        o.S = __shfl_xor(ws.S, d);
        o.A = __shfl_xor(ws.A, d);
        o.g = __shfl_xor(ws.g, d);
        ws = seq_join(ws, o);
        const double up = __shfl_up(incl, d);
        if ((int)lane >= d) incl += up;
    }
    const double excl = incl - s.S;  // (exact under the wavefront's test: a difference of two representable partial sums)
    if (lane == 0) {
        L.wS[w] = ws.S;
        L.wA[w] = ws.A;
        L.wg[w] = ws.g;
    }
    if (t == 0) {
        L.hit = ~0ull;
        L.stop = 0;
    }
    __syncthreads();
    double total = 0.0;
    if (t == 0) {
        double acc = 0.0;
        bool done = false;
        uint32_t st_w = 0, st_r = 0, st_walk = 0;
        uint64_t st_el = 0;
        for (uint32_t ww = 0; ww < 16u && !done; ++ww) {
            const SeqSummary wsum{L.wS[ww], L.wA[ww], L.wg[ww]};
            L.wcarry[ww] = acc;
            if (seq_exact(acc, wsum)) {
                L.wok[ww] = 1;
                acc += wsum.S;
                ++st_w;
                continue;
            }
            L.wok[ww] = 0;
            for (uint32_t r = ww * 64u; r < ww * 64u + 64u && !done; ++r) {
                const SeqSummary rs{L.S[r], L.A[r], L.g[r]};
                L.carry[r] = acc;
                if (seq_exact(acc, rs)) {
                    L.ok[r] = 1;
                    acc += rs.S;
                    ++st_r;
                    continue;
                }
                L.ok[r] = 0;
                ++st_walk;
                const uint64_t rlo = (uint64_t)r * seg < count ? (uint64_t)r * seg : count;
                const uint64_t rhi = rlo + seg < count ? rlo + seg : count;
                st_el += rhi - rlo;
                for (uint64_t i = rlo; i < rhi; ++i) {
                    acc += load(i);
                    if (SELECT && acc >= thr && pred(i)) {
                        L.hit = i;
                        L.stop = 1;
                        done = true;
                        break;
                    }
                }
            }
        }
        total = acc;
        atomicAdd(&g_seq_stats[0], (unsigned long long)st_w);
        atomicAdd(&g_seq_stats[1], (unsigned long long)st_r);
        atomicAdd(&g_seq_stats[2], (unsigned long long)st_walk);
        atomicAdd(&g_seq_stats[3], (unsigned long long)st_el);
    }
    if (!SELECT) return total;
    __syncthreads();
    // ranges whose walk is exact: every thread repeats its range's walk from the carry and looks for the first hit.
    // (After a stop the ranges behind the stopping one were never classified: they lie behind the hit and are skipped.)
    {
        const unsigned long long stop_at = L.stop ? L.hit : ~0ull;
        if (lo < hi && lo < stop_at) {
            const bool wok = L.wok[w] != 0;
            if (wok || L.ok[t] != 0) {
                double acc = wok ? L.wcarry[w] + excl : L.carry[t];
                for (uint64_t i = lo; i < hi; ++i) {
                    acc += load(i);
                    if (acc >= thr && pred(i)) {
                        atomicMin(&L.hit, (unsigned long long)i);
                        break;
                    }
                }
            }
        }
    }
    __syncthreads();
    if (t == 0) *hit_out = L.hit;
    return total;
}
// rolling f64 sum of the 16-row block sums, one workgroup per chunk
__global__ __launch_bounds__(1024) void kpp_total_par_kernel(const double* block_sums, uint64_t nblocks, double* totals) {
    __shared__ SeqLds L;
    const double* b = block_sums + (uint64_t)blockIdx.x * nblocks;
    unsigned long long dummy;
    const double s = seq_sum_workgroup<false>(L, nblocks, [&](uint64_t i) { return b[i]; }, 0.0,
                                              [](uint64_t) { return false; }, &dummy);
    if (threadIdx.x == 0) totals[blockIdx.x] = s;
}
// the D^2 draw (plusplus.rs:446-462), one workgroup per chunk
__global__ __launch_bounds__(1024) void kpp_select_par_kernel(const float* mins, uint64_t n, const double* thresholds,
                                                              const uint8_t* active, uint8_t* picked, int64_t* chosen) {
    __shared__ SeqLds L;
    const uint32_t c = blockIdx.x;
    if (!active[c]) {
        if (threadIdx.x == 0) chosen[c] = -1;
        return;
    }
    const float* m = mins + (uint64_t)c * n;
    uint8_t* pk = picked + (uint64_t)c * n;
    __shared__ unsigned long long hit;
    seq_sum_workgroup<true>(L, n, [&](uint64_t i) { return (double)m[i]; }, thresholds[c],
                            [&](uint64_t i) { return m[i] > 0.0f && !pk[i]; }, &hit);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (hit == ~0ull) chosen[c] = -1;
        else {
            pk[hit] = 1;
            chosen[c] = (int64_t)hit;
        }
    }
}

// residual: SIMD lane l sums the points with index = l (mod 8) in order, then sum_tree (lloyds.rs:201, 254-259).
// One block per chunk; tiles are staged through LDS so the eight sequential chains read at LDS speed.
__global__ __launch_bounds__(256) void pq_residual_kernel(const float* best, uint64_t n, float* residuals,
                                                          uint32_t* bad_assign, const uint32_t* assign) {
    __shared__ float tile[2048];
    __shared__ float lanes[8];
    const uint32_t c = blockIdx.x;
    const float* b = best + (uint64_t)c * n;
    float acc = 0.0f;
    bool bad = false;
    for (uint64_t t0 = 0; t0 < n; t0 += 2048) {
        for (uint32_t i = threadIdx.x; i < 2048; i += blockDim.x) {
            tile[i] = (t0 + i < n) ? b[t0 + i] : 0.0f;
            if (t0 + i < n && assign[(uint64_t)c * n + t0 + i] == 0xFFFFFFFFu) bad = true;
        }
        __syncthreads();
        if (threadIdx.x < 8) {
            const uint64_t lim = (n - t0) < 2048 ? (n - t0) : 2048;
            for (uint32_t i = threadIdx.x; i < lim; i += 8) acc = acc + tile[i];
        }
        __syncthreads();
    }
    if (bad) atomicExch(bad_assign, 1u);
    if (threadIdx.x < 8) lanes[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
        residuals[c] = ((lanes[0] + lanes[4]) + (lanes[2] + lanes[6])) + ((lanes[1] + lanes[5]) + (lanes[3] + lanes[7]));
}
__global__ void pq_iota_kernel(uint32_t* v, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (uint32_t)i;
}
__global__ void pq_hist_kernel(const uint32_t* assign, uint64_t n, uint32_t ncenters, uint32_t* counts) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t c = blockIdx.y;
    if (i >= n) return;
    const uint32_t a = assign[(uint64_t)c * n + i];
    if (a < ncenters) atomicAdd(&counts[c * ncenters + a], 1u);
}
// update_centroids (lloyds.rs:273-296): f64 sums in row order; block = one centre of one chunk, thread = dimension.
// `order` lists the rows of each centre in increasing row order (stable radix sort by centre), per chunk.
__global__ void pq_update_kernel(const float* data, uint64_t n, uint32_t dim, const uint32_t* offsets, uint32_t ncenters,
                                 const uint32_t* order, const uint32_t* counts, const uint32_t* starts, float* centers) {
    const uint32_t j = blockIdx.x, c = blockIdx.y;
    const uint32_t s0 = offsets[c], len = offsets[c + 1] - s0;
    const uint32_t cnt = counts[c * ncenters + j];
    const uint32_t* list = order + (uint64_t)c * n + starts[c * ncenters + j];
    for (uint32_t d = threadIdx.x; d < len; d += blockDim.x) {
        double sum = 0.0;
        uint32_t i = 0;
        for (; i + 4 <= cnt; i += 4) {  // loads issued together, adds in order
            const float v0 = data[(uint64_t)list[i] * dim + s0 + d], v1 = data[(uint64_t)list[i + 1] * dim + s0 + d];
            const float v2 = data[(uint64_t)list[i + 2] * dim + s0 + d], v3 = data[(uint64_t)list[i + 3] * dim + s0 + d];
            sum += (double)v0;
            sum += (double)v1;
            sum += (double)v2;
            sum += (double)v3;
        }
        for (; i < cnt; ++i) sum += (double)data[(uint64_t)list[i] * dim + s0 + d];
        centers[(uint64_t)j * dim + s0 + d] = (float)(sum / (double)(cnt > 1u ? cnt : 1u));
    }
}
__global__ void pq_scan_counts_kernel(const uint32_t* counts, uint32_t ncenters, uint32_t* starts) {
    if (threadIdx.x == 0) {
        const uint32_t c = blockIdx.x;
        uint32_t acc = 0;
        for (uint32_t j = 0; j < ncenters; ++j) {
            starts[c * ncenters + j] = acc;
            acc += counts[c * ncenters + j];
        }
    }
}

// ---- ScalarQuantizationParameters::train (scalar/train.rs:33-52; utils.rs:109-140, 180-199) ------------------
// Every statistic is an f64 sum in row order, so each sum is one sequential chain: one thread per dimension walks
// the rows (a wave reads 64 consecutive columns of a row per step -- coalesced), rows' norms are computed one thread
// per row and then summed by a single chain staged through LDS.
__global__ void sq_col_sum_kernel(const float* data, uint64_t n, uint32_t dim, const double* means, double* out) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= dim) return;
    double acc = 0.0;
    if (!means) {
        for (uint64_t r = 0; r < n; ++r) acc += (double)data[r * dim + d];
    } else {
        const double m = means[d];
        for (uint64_t r = 0; r < n; ++r) {
            const double df = (double)data[r * dim + d] - m;
            acc += df * df;
        }
    }
    out[d] = acc / (double)n;
}
__global__ void sq_row_norm_kernel(const float* data, uint64_t n, uint32_t dim, double* norms) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    double sq = 0.0;
    for (uint32_t d = 0; d < dim; ++d) {
        const double x = (double)data[r * dim + d];
        sq += x * x;
    }
    norms[r] = __builtin_sqrt(sq);  // correctly rounded (checked against the oracle in tests)
}
__global__ __launch_bounds__(256) void sq_chain_sum_kernel(const double* v, uint64_t n, double* out) {
    __shared__ double tile[1024];
    double acc = 0.0;
    for (uint64_t t0 = 0; t0 < n; t0 += 1024) {
        for (uint32_t i = threadIdx.x; i < 1024; i += blockDim.x) tile[i] = (t0 + i < n) ? v[t0 + i] : 0.0;
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint64_t lim = (n - t0) < 1024 ? (n - t0) : 1024;
            for (uint32_t i = 0; i < lim; ++i) acc = acc + tile[i];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = acc / (double)n;
}

struct Buf {
    void* p = nullptr;
    ~Buf() {
        if (p) (void)hipFree(p);
    }
};

// dann_pq_pack_neighbors: one wavefront per node.  Row = [len][ids x R][pad to 16][code row (16 bytes per 16 chunks) of neighbour 0 .. R-1]
// (neighbours beyond the length, and ids beyond the index, get zero code rows: they are never candidates).
__global__ __launch_bounds__(256) void pq_pack_kernel(IndexView ix, uint8_t* pack, uint32_t stride, uint32_t codes_off) {
    const uint32_t node = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (node >= ix.nslots) return;
    const uint32_t R = ix.max_degree;
    const uint32_t cw = (ix.pq_chunks + 15u) / 16u;  // 16-byte words of a code row (1 .. 4)
    const uint32_t* arow = ix.adj + (uint64_t)node * ix.adj_stride;
    uint8_t* prow = pack + (uint64_t)node * stride;
    const uint32_t len = arow[0] < R ? arow[0] : R;
    for (uint32_t w = lane; w < codes_off / 4u; w += 64u)
        reinterpret_cast<uint32_t*>(prow)[w] = w <= R ? arow[w] : 0u;  // (the stored length word, unclamped, as the adjacency row has it)
    for (uint32_t t = lane; t < R * cw; t += 64u) {
        const uint32_t j = t / cw, g = t % cw;
        uint4 code = make_uint4(0u, 0u, 0u, 0u);
        if (j < len) {
            const uint32_t id = arow[1u + j];
            if (id < ix.nslots) code = *reinterpret_cast<const uint4*>(ix.rows + (uint64_t)id * ix.row_stride + 16u * g);
        }
        *reinterpret_cast<uint4*>(prow + codes_off + 16u * t) = code;
    }
    for (uint32_t o = codes_off + 16u * cw * R + 4u * lane; o < stride; o += 256u) *reinterpret_cast<uint32_t*>(prow + o) = 0u;
}

}  // namespace

int32_t launch_pq_pack(const IndexView& ix, uint8_t* d_pack, uint32_t stride, uint32_t codes_off, hipStream_t stream) {
    hipLaunchKernelGGL(pq_pack_kernel, dim3((ix.nslots + 3u) / 4u), dim3(256), 0, stream, ix, d_pack, stride, codes_off);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "pq_pack_kernel launch");
    return DANN_OK;
}
}  // namespace dann

using namespace dann;

extern "C" {

int32_t dann_pq_build_lut(int32_t device, int32_t metric, const float* pivots, const uint32_t* chunk_offsets,
                          uint32_t nchunks, uint32_t dim, const float* queries, uint32_t nq, float* lut) try {
    if (!pivots || !chunk_offsets || !queries || !lut || nchunks == 0 || dim == 0) return DANN_EINVAL;
    if (metric != M_L2 && metric != M_IP) {
        set_error("PQ lookup tables exist for L2 and inner product only");
        return DANN_EUNSUPPORTED;
    }
    if (chunk_offsets[0] != 0 || chunk_offsets[nchunks] != dim) {
        set_error("chunk offsets must start at 0 and end at dim");
        return DANN_EINVAL;
    }
    for (uint32_t c = 0; c < nchunks; ++c)
        if (chunk_offsets[c + 1] <= chunk_offsets[c]) return DANN_EINVAL;
    if (nq == 0) return DANN_OK;
    if (device >= 0) DANN_HIP(hipSetDevice(device));
    if ((size_t)dim * 4 > 64 * 1024) return DANN_EUNSUPPORTED;
    Buf dp, doff, dq, dl;
    const size_t lut_bytes = (size_t)nq * nchunks * 256 * 4;
    DANN_HIP(hipMalloc(&dp.p, (size_t)256 * dim * 4));
    DANN_HIP(hipMalloc(&doff.p, (size_t)(nchunks + 1) * 4));
    DANN_HIP(hipMalloc(&dq.p, (size_t)nq * dim * 4));
    DANN_HIP(hipMalloc(&dl.p, lut_bytes));
    DANN_HIP(hipMemcpy(dp.p, pivots, (size_t)256 * dim * 4, hipMemcpyHostToDevice));
    DANN_HIP(hipMemcpy(doff.p, chunk_offsets, (size_t)(nchunks + 1) * 4, hipMemcpyHostToDevice));
    DANN_HIP(hipMemcpy(dq.p, queries, (size_t)nq * dim * 4, hipMemcpyHostToDevice));
    if (metric == M_L2)
        hipLaunchKernelGGL(pq_lut_kernel<true>, dim3(nq), dim3(256), (size_t)dim * 4, 0, (const float*)dp.p,
                           (const uint32_t*)doff.p, nchunks, dim, (const float*)dq.p, (float*)dl.p);
    else
        hipLaunchKernelGGL(pq_lut_kernel<false>, dim3(nq), dim3(256), (size_t)dim * 4, 0, (const float*)dp.p,
                           (const uint32_t*)doff.p, nchunks, dim, (const float*)dq.p, (float*)dl.p);
    DANN_HIP(hipGetLastError());
    DANN_HIP(hipMemcpy(lut, dl.p, lut_bytes, hipMemcpyDeviceToHost));
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_pq_scan(int32_t device, const float* lut, uint32_t nq, uint32_t nchunks, const uint8_t* codes,
                     uint64_t npoints, const uint32_t* ids, const uint64_t* offsets, float* out) try {
    if (!lut || !codes || !ids || !offsets || !out || nchunks == 0) return DANN_EINVAL;
    if (nq == 0) return DANN_OK;
    const uint64_t total = offsets[nq];
    uint64_t maxlen = 0;
    for (uint32_t i = 0; i < nq; ++i) {
        if (offsets[i + 1] < offsets[i]) return DANN_EINVAL;
        maxlen = std::max<uint64_t>(maxlen, offsets[i + 1] - offsets[i]);
    }
    for (uint64_t i = 0; i < total; ++i)
        if (ids[i] >= npoints) return DANN_EBOUNDS;
    if (total == 0) return DANN_OK;
    const size_t lds = (size_t)nchunks * 1024;
    if (lds > 160 * 1024) {
        set_error("LUT of %u chunks does not fit in LDS", nchunks);
        return DANN_EUNSUPPORTED;
    }
    if (device >= 0) DANN_HIP(hipSetDevice(device));
    Buf dl, dc, di, doff, dout;
    DANN_HIP(hipMalloc(&dl.p, (size_t)nq * lds));
    DANN_HIP(hipMalloc(&dc.p, npoints * nchunks));
    DANN_HIP(hipMalloc(&di.p, total * 4));
    DANN_HIP(hipMalloc(&doff.p, (size_t)(nq + 1) * 8));
    DANN_HIP(hipMalloc(&dout.p, total * 4));
    DANN_HIP(hipMemcpy(dl.p, lut, (size_t)nq * lds, hipMemcpyHostToDevice));
    DANN_HIP(hipMemcpy(dc.p, codes, npoints * nchunks, hipMemcpyHostToDevice));
    DANN_HIP(hipMemcpy(di.p, ids, total * 4, hipMemcpyHostToDevice));
    DANN_HIP(hipMemcpy(doff.p, offsets, (size_t)(nq + 1) * 8, hipMemcpyHostToDevice));
    auto kern = pq_scan_kernel;
    if (lds > 64 * 1024)
        DANN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds));
    const uint32_t chunks = (uint32_t)((maxlen + 4095) / 4096);
    hipLaunchKernelGGL(kern, dim3(nq, chunks), dim3(256), lds, 0, (const float*)dl.p, nchunks, (const uint8_t*)dc.p,
                       (const uint32_t*)di.p, (const uint64_t*)doff.p, (float*)dout.p);
    DANN_HIP(hipGetLastError());
    DANN_HIP(hipMemcpy(out, dout.p, total * 4, hipMemcpyDeviceToHost));
    return DANN_OK;
} DANN_CATCH_ALL

}  // extern "C"

extern "C" int32_t dann_pq_compress(int32_t device, const float* pivots, uint32_t ncenters, const uint32_t* chunk_offsets,
                                    uint32_t nchunks, uint32_t dim, const float* rows, uint64_t n, uint8_t* codes) try {
    using namespace dann;
    if (!pivots || !chunk_offsets || !rows || !codes || nchunks == 0 || dim == 0) return DANN_EINVAL;
    if (ncenters == 0 || ncenters > 256) {  // TableCompressionError::CannotCompressToByte
        set_error("num centers (%u) must be at most 256 to compress into a byte vector", ncenters);
        return DANN_EINVAL;
    }
    if (chunk_offsets[0] != 0 || chunk_offsets[nchunks] != dim) {
        set_error("chunk offsets must start at 0 and end at dim");
        return DANN_EINVAL;
    }
    uint32_t maxlen = 0;
    for (uint32_t c = 0; c < nchunks; ++c) {
        if (chunk_offsets[c + 1] <= chunk_offsets[c]) return DANN_EINVAL;
        maxlen = std::max(maxlen, chunk_offsets[c + 1] - chunk_offsets[c]);
    }
    if (n == 0) return DANN_OK;
    const size_t lds = ((size_t)ncenters * maxlen + ncenters) * 4;
    if (lds > 160 * 1024) {
        set_error("PQ chunk of %u dimensions x %u centres does not fit the 160 KiB LDS slab", maxlen, ncenters);
        return DANN_EUNSUPPORTED;
    }
    if (device >= 0) DANN_HIP(hipSetDevice(device));
    if (lds > 64 * 1024)
        DANN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pq_compress_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    Buf dp, doff, dx, dc, dbad;
    DANN_HIP(hipMalloc(&dp.p, (size_t)ncenters * dim * 4));
    DANN_HIP(hipMalloc(&doff.p, (size_t)(nchunks + 1) * 4));
    DANN_HIP(hipMalloc(&dbad.p, 8));
    DANN_HIP(hipMemcpy(dp.p, pivots, (size_t)ncenters * dim * 4, hipMemcpyHostToDevice));
    DANN_HIP(hipMemcpy(doff.p, chunk_offsets, (size_t)(nchunks + 1) * 4, hipMemcpyHostToDevice));
    DANN_HIP(hipMemset(dbad.p, 0xFF, 8));
    // rows go through in slabs of <= 1 GiB
    const uint64_t slab_rows = std::max<uint64_t>(1, (1ull << 30) / ((uint64_t)dim * 4));
    const uint64_t cap = std::min<uint64_t>(n, slab_rows);
    DANN_HIP(hipMalloc(&dx.p, cap * dim * 4));
    DANN_HIP(hipMalloc(&dc.p, cap * nchunks));
    for (uint64_t off = 0; off < n; off += cap) {
        const uint64_t m = std::min<uint64_t>(cap, n - off);
        DANN_HIP(hipMemcpy(dx.p, rows + off * dim, m * dim * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(pq_compress_kernel, dim3((uint32_t)((m + 255) / 256), nchunks), dim3(256), lds, 0,
                           (const float*)dp.p, ncenters, (const uint32_t*)doff.p, nchunks, dim, (const float*)dx.p, m,
                           (uint8_t*)dc.p, (unsigned long long*)dbad.p);
        DANN_HIP(hipGetLastError());
        DANN_HIP(hipMemcpy(codes + off * nchunks, dc.p, m * nchunks, hipMemcpyDeviceToHost));
        unsigned long long bad = ~0ull;
        DANN_HIP(hipMemcpy(&bad, dbad.p, 8, hipMemcpyDeviceToHost));
        if (bad != ~0ull) {  // TableBatchCompressionError::InfinityOrNaN(chunk, row)
            set_error("a value of infinity or NaN was observed while compressing chunk %llu of batch input %llu",
                      bad % nchunks, off + bad / nchunks);
            return DANN_EINVAL;
        }
    }
    return DANN_OK;
} DANN_CATCH_ALL

namespace dann {
namespace {
int32_t pq_check_offsets(const uint32_t* chunk_offsets, uint32_t nchunks, uint32_t dim, uint32_t* maxlen) {
    if (chunk_offsets[0] != 0 || chunk_offsets[nchunks] != dim) {
        set_error("chunk offsets must start at 0 and end at dim");
        return DANN_EINVAL;
    }
    uint32_t m = 0;
    for (uint32_t c = 0; c < nchunks; ++c) {
        if (chunk_offsets[c + 1] <= chunk_offsets[c]) return DANN_EINVAL;
        m = std::max(m, chunk_offsets[c + 1] - chunk_offsets[c]);
    }
    if (maxlen) *maxlen = m;
    return DANN_OK;
}
// the Lloyd iterations on device-resident rows, chunk offsets and centres (dcen: updated in place)
int32_t pq_lloyds_device(const float* dx, uint64_t n, uint32_t dim, const uint32_t* chunk_offsets, const uint32_t* doff,
                         uint32_t nchunks, uint32_t ncenters, float* dcen, uint32_t max_reps, uint32_t* assignments,
                         float* residuals) {
    uint32_t maxlen = 0;
    int32_t rc = pq_check_offsets(chunk_offsets, nchunks, dim, &maxlen);
    if (rc != DANN_OK) return rc;
    // closest centres on the matrix cores while the transposed slab of one chunk fits 64 KiB of LDS (every PQ shape:
    // chunks of a few to a few dozen columns); beyond that the row kernel with the chunk's centres in up to 160 KiB
    const uint32_t ncp = pq_mfma_ncp(ncenters);
    const size_t lds_mfma = ((size_t)2 * ((maxlen + 1) / 2) * ncp + (size_t)((ncenters + 31) / 32) * 32) * 4;
    const size_t lds_row = ((size_t)ncenters * maxlen + ncenters) * 4;
    const bool mfma = lds_mfma <= 64 * 1024;
    if (!mfma && (lds_row > 160 * 1024 || maxlen > 1024)) {
        set_error("PQ chunk of %u dimensions x %u centres does not fit the 160 KiB LDS slab", maxlen, ncenters);
        return DANN_EUNSUPPORTED;
    }
    if (!mfma && lds_row > 64 * 1024)
        DANN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pq_assign_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_row));
    Buf dcn, ddn, dasg, dbest, dres, dord, dord2, dkeys2, dcounts, dstarts, dtmp, dbad;
    DANN_HIP(hipMalloc(&dcn.p, (size_t)nchunks * ncenters * 4));
    DANN_HIP(hipMalloc(&ddn.p, (size_t)nchunks * n * 4));
    DANN_HIP(hipMalloc(&dasg.p, (size_t)nchunks * n * 4));
    DANN_HIP(hipMalloc(&dbest.p, (size_t)nchunks * n * 4));
    DANN_HIP(hipMalloc(&dres.p, (size_t)nchunks * 4));
    DANN_HIP(hipMalloc(&dord.p, n * 4));
    DANN_HIP(hipMalloc(&dord2.p, (size_t)nchunks * n * 4));
    DANN_HIP(hipMalloc(&dkeys2.p, n * 4));
    DANN_HIP(hipMalloc(&dcounts.p, (size_t)nchunks * ncenters * 4));
    DANN_HIP(hipMalloc(&dstarts.p, (size_t)nchunks * ncenters * 4));
    DANN_HIP(hipMalloc(&dbad.p, 4));
    DANN_HIP(hipMemsetAsync(dbad.p, 0, 4, 0));
    int key_bits = 1;
    while ((1u << key_bits) < ncenters && key_bits < 32) ++key_bits;
    size_t tmp_bytes = 0;
    DANN_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, (const uint32_t*)dasg.p, (uint32_t*)dkeys2.p,
                                                (const uint32_t*)dord.p, (uint32_t*)dord2.p, (int)n, 0, key_bits));
    DANN_HIP(hipMalloc(&dtmp.p, tmp_bytes + 16));
    const dim3 rows_grid((uint32_t)((n + 255) / 256), nchunks);
    hipLaunchKernelGGL(pq_data_norms_kernel, rows_grid, dim3(256), 0, 0, dx, n, dim, doff, (float*)ddn.p);
    hipLaunchKernelGGL(pq_center_norms_kernel, dim3((ncenters + 255) / 256, nchunks), dim3(256), 0, 0,
                       (const float*)dcen, ncenters, dim, doff, (float*)dcn.p);
    hipLaunchKernelGGL(pq_iota_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, (uint32_t*)dord.p, n);
    for (uint32_t rep = 0; rep < max_reps; ++rep) {
        if (mfma)
            hipLaunchKernelGGL(pq_assign_mfma_kernel, dim3((uint32_t)((n + 127) / 128), nchunks), dim3(256), lds_mfma, 0, dx, n,
                               dim, doff, (const float*)dcen, ncenters, (const float*)dcn.p, (const float*)ddn.p,
                               (uint32_t*)dasg.p, (float*)dbest.p);
        else
            hipLaunchKernelGGL(pq_assign_kernel, rows_grid, dim3(256), lds_row, 0, dx, n, dim, doff, (const float*)dcen,
                               ncenters, (const float*)dcn.p, (const float*)ddn.p, (uint32_t*)dasg.p, (float*)dbest.p);
        hipLaunchKernelGGL(pq_residual_kernel, dim3(nchunks), dim3(256), 0, 0, (const float*)dbest.p, n, (float*)dres.p,
                           (uint32_t*)dbad.p, (const uint32_t*)dasg.p);
        // rows of every centre in row order (a stable sort by centre), all chunks; then one update launch
        DANN_HIP(hipMemsetAsync(dcounts.p, 0, (size_t)nchunks * ncenters * 4, 0));
        hipLaunchKernelGGL(pq_hist_kernel, rows_grid, dim3(256), 0, 0, (const uint32_t*)dasg.p, n, ncenters,
                           (uint32_t*)dcounts.p);
        hipLaunchKernelGGL(pq_scan_counts_kernel, dim3(nchunks), dim3(64), 0, 0, (const uint32_t*)dcounts.p, ncenters,
                           (uint32_t*)dstarts.p);
        for (uint32_t c = 0; c < nchunks; ++c) {
            size_t tb = tmp_bytes;
            DANN_HIP(hipcub::DeviceRadixSort::SortPairs(dtmp.p, tb, (const uint32_t*)dasg.p + (size_t)c * n,
                                                        (uint32_t*)dkeys2.p, (const uint32_t*)dord.p,
                                                        (uint32_t*)dord2.p + (size_t)c * n, (int)n, 0, key_bits));
        }
        hipLaunchKernelGGL(pq_update_kernel, dim3(ncenters, nchunks), dim3(64), 0, 0, dx, n, dim, doff, ncenters,
                           (const uint32_t*)dord2.p, (const uint32_t*)dcounts.p, (const uint32_t*)dstarts.p, dcen);
        if (rep != max_reps - 1)
            hipLaunchKernelGGL(pq_center_norms_kernel, dim3((ncenters + 255) / 256, nchunks), dim3(256), 0, 0,
                               (const float*)dcen, ncenters, dim, doff, (float*)dcn.p);
        DANN_HIP(hipGetLastError());
    }
    // a row whose scores were all NaN keeps the "no centre" mark; the later iterations skip such rows (nothing is read
    // out of bounds), and the call fails here, as the per-iteration check did
    uint32_t bad = 0;
    DANN_HIP(hipMemcpy(&bad, dbad.p, 4, hipMemcpyDeviceToHost));
    if (bad) {
        set_error("k-means assignment saw only NaN scores for some row (non-finite data or centres)");
        return DANN_EINVAL;
    }
    if (assignments) DANN_HIP(hipMemcpy(assignments, dasg.p, (size_t)nchunks * n * 4, hipMemcpyDeviceToHost));
    if (residuals) DANN_HIP(hipMemcpy(residuals, dres.p, (size_t)nchunks * 4, hipMemcpyDeviceToHost));
    return DANN_OK;
}
}  // namespace
}  // namespace dann

extern "C" int32_t dann_pq_lloyds(int32_t device, const float* data, uint64_t n, uint32_t dim,
                                  const uint32_t* chunk_offsets, uint32_t nchunks, uint32_t ncenters, float* centers,
                                  uint32_t max_reps, uint32_t* assignments, float* residuals) try {
    using namespace dann;
    if (!data || !chunk_offsets || !centers || nchunks == 0 || dim == 0 || ncenters == 0) return DANN_EINVAL;
    if (n == 0 || n > 0xFFFFFFFFull) return DANN_EINVAL;
    int32_t rc = pq_check_offsets(chunk_offsets, nchunks, dim, nullptr);
    if (rc != DANN_OK) return rc;
    if (device >= 0) DANN_HIP(hipSetDevice(device));
    Buf dx, doff, dcen;
    DANN_HIP(hipMalloc(&dx.p, n * dim * 4));
    DANN_HIP(hipMalloc(&doff.p, (size_t)(nchunks + 1) * 4));
    DANN_HIP(hipMalloc(&dcen.p, (size_t)ncenters * dim * 4));
    DANN_HIP(hipMemcpy(dx.p, data, n * dim * 4, hipMemcpyHostToDevice));
    DANN_HIP(hipMemcpy(doff.p, chunk_offsets, (size_t)(nchunks + 1) * 4, hipMemcpyHostToDevice));
    DANN_HIP(hipMemcpy(dcen.p, centers, (size_t)ncenters * dim * 4, hipMemcpyHostToDevice));
    rc = pq_lloyds_device((const float*)dx.p, n, dim, chunk_offsets, (const uint32_t*)doff.p, nchunks, ncenters,
                          (float*)dcen.p, max_reps, assignments, residuals);
    if (rc != DANN_OK) return rc;
    DANN_HIP(hipMemcpy(centers, dcen.p, (size_t)ncenters * dim * 4, hipMemcpyDeviceToHost));
    return DANN_OK;
} DANN_CATCH_ALL

extern "C" int32_t dann_sq8_train(int32_t device, const float* data, uint64_t n, uint32_t dim, double standard_deviations,
                                  float* shift, float* scale, float* mean_norm) try {
    using namespace dann;
    if (!data || !shift || !scale || n == 0 || dim == 0) return DANN_EINVAL;
    if (!(standard_deviations > 0.0)) {  // Positive<f64>
        set_error("standard_deviations must be positive");
        return DANN_EINVAL;
    }
    if (device >= 0) DANN_HIP(hipSetDevice(device));
    Buf dx, dm, dv, dn, dmn;
    DANN_HIP(hipMalloc(&dx.p, n * dim * 4));
    DANN_HIP(hipMalloc(&dm.p, (size_t)dim * 8));
    DANN_HIP(hipMalloc(&dv.p, (size_t)dim * 8));
    DANN_HIP(hipMalloc(&dn.p, n * 8));
    DANN_HIP(hipMalloc(&dmn.p, 8));
    DANN_HIP(hipMemcpy(dx.p, data, n * dim * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(sq_col_sum_kernel, dim3((dim + 63) / 64), dim3(64), 0, 0, (const float*)dx.p, n, dim,
                       (const double*)nullptr, (double*)dm.p);
    hipLaunchKernelGGL(sq_col_sum_kernel, dim3((dim + 63) / 64), dim3(64), 0, 0, (const float*)dx.p, n, dim,
                       (const double*)dm.p, (double*)dv.p);
    hipLaunchKernelGGL(sq_row_norm_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, (const float*)dx.p, n, dim,
                       (double*)dn.p);
    hipLaunchKernelGGL(sq_chain_sum_kernel, dim3(1), dim3(256), 0, 0, (const double*)dn.p, n, (double*)dmn.p);
    DANN_HIP(hipGetLastError());
    std::vector<double> means(dim), var(dim);
    double mn = 0.0;
    DANN_HIP(hipMemcpy(means.data(), dm.p, (size_t)dim * 8, hipMemcpyDeviceToHost));
    DANN_HIP(hipMemcpy(var.data(), dv.p, (size_t)dim * 8, hipMemcpyDeviceToHost));
    DANN_HIP(hipMemcpy(&mn, dmn.p, 8, hipMemcpyDeviceToHost));
    double mx = 0.0;
    for (uint32_t d = 0; d < dim; ++d) mx = std::max(mx, var[d]);
    const double p = std::sqrt(mx) * standard_deviations;
    *scale = (float)(2.0 * p);
    for (uint32_t d = 0; d < dim; ++d) shift[d] = (float)(means[d] - p);
    if (mean_norm) *mean_norm = (float)mn;
    return DANN_OK;
} DANN_CATCH_ALL

extern "C" int32_t dann_sq8_compress(int32_t device, const float* x, uint32_t n, uint32_t dim, const float* shift,
                                     float scale, void* out) try {
    using namespace dann;
    if (!x || !shift || !out || dim == 0 || !(scale > 0.0f)) return DANN_EINVAL;
    if (n == 0) return DANN_OK;
    if (device >= 0) DANN_HIP(hipSetDevice(device));
    Buf dx, ds, dout;
    DANN_HIP(hipMalloc(&dx.p, (size_t)n * dim * 4));
    DANN_HIP(hipMalloc(&ds.p, (size_t)dim * 4));
    DANN_HIP(hipMalloc(&dout.p, (size_t)n * (dim + 4)));
    DANN_HIP(hipMemcpy(dx.p, x, (size_t)n * dim * 4, hipMemcpyHostToDevice));
    DANN_HIP(hipMemcpy(ds.p, shift, (size_t)dim * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(sq8_compress_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, (const float*)dx.p, n, dim,
                       (const float*)ds.p, scale, (uint8_t*)dout.p);
    DANN_HIP(hipGetLastError());
    DANN_HIP(hipMemcpy(out, dout.p, (size_t)n * (dim + 4), hipMemcpyDeviceToHost));
    return DANN_OK;
} DANN_CATCH_ALL

// k-means++ for every chunk in lockstep: per centre one update launch over all (row, chunk) pairs, the sequential f64
// totals, one host round trip for the caller's threshold draws, the sequential selection, the commit.
namespace dann {
namespace {
// k-means++ on device-resident rows; dcen (ncenters x dim, device) receives the seeds, `sel` the number of centres
// selected per chunk.  One host round trip per centre: the totals come back together with the previous centre's
// picks, the thresholds go out with the chunks' activity flags.
int32_t pq_kmeanspp_device(const float* dx, uint64_t n, uint32_t dim, const uint32_t* doff, uint32_t nchunks,
                           uint32_t ncenters, const dann_rng* rng, float* dcen, std::vector<uint32_t>& sel) {
    const uint64_t nblocks = (n + 15) / 16;
    Buf dnorm, dmins, dbs, dback, dfwd, dpicked, dlast, dlastn;
    struct Pinned {
        void* p = nullptr;
        ~Pinned() {
            if (p) (void)hipHostFree(p);
        }
    } hback, hfwd;
    // back: [totals: nchunks f64][chosen: nchunks i64]; forward: [thresholds: nchunks f64][active: nchunks bytes]
    const size_t back_bytes = (size_t)nchunks * 16, fwd_bytes = (size_t)nchunks * 9;
    DANN_HIP(hipMalloc(&dnorm.p, (size_t)nchunks * n * 4));
    DANN_HIP(hipMalloc(&dmins.p, (size_t)nchunks * n * 4));
    DANN_HIP(hipMalloc(&dbs.p, (size_t)nchunks * nblocks * 8));
    DANN_HIP(hipMalloc(&dback.p, back_bytes));
    DANN_HIP(hipMalloc(&dfwd.p, fwd_bytes));
    DANN_HIP(hipMalloc(&dpicked.p, (size_t)nchunks * n));
    DANN_HIP(hipMalloc(&dlast.p, (size_t)nchunks * dim * 4));
    DANN_HIP(hipMalloc(&dlastn.p, (size_t)nchunks * 4));
    DANN_HIP(hipHostMalloc(&hback.p, back_bytes, hipHostMallocDefault));
    DANN_HIP(hipHostMalloc(&hfwd.p, fwd_bytes, hipHostMallocDefault));
    double* const dtot = (double*)dback.p;
    int64_t* const dchosen = (int64_t*)((uint8_t*)dback.p + (size_t)nchunks * 8);
    double* const dthr = (double*)dfwd.p;
    uint8_t* const dact = (uint8_t*)dfwd.p + (size_t)nchunks * 8;
    double* const h_tot = (double*)hback.p;
    int64_t* const h_chosen = (int64_t*)((uint8_t*)hback.p + (size_t)nchunks * 8);
    double* const h_thr = (double*)hfwd.p;
    uint8_t* const h_active = (uint8_t*)hfwd.p + (size_t)nchunks * 8;
    DANN_HIP(hipMemsetAsync(dcen, 0, (size_t)ncenters * dim * 4, 0));
    DANN_HIP(hipMemsetAsync(dpicked.p, 0, (size_t)nchunks * n, 0));
    const dim3 rows_grid((uint32_t)((n + 255) / 256), nchunks);
    hipLaunchKernelGGL(pq_data_norms_kernel, rows_grid, dim3(256), 0, 0, dx, n, dim, doff, (float*)dnorm.p);
    {
        const uint64_t tot = (uint64_t)nchunks * n;
        hipLaunchKernelGGL(kpp_fill_kernel, dim3((uint32_t)((tot + 255) / 256)), dim3(256), 0, 0, (float*)dmins.p, tot,
                           __builtin_inff());
    }
    // first centre of every chunk: Uniform::new(0, n).sample(rng)
    const uint8_t one = 1;
    for (uint32_t c = 0; c < nchunks; ++c) {
        const uint64_t i = rng->uniform_index(rng->ctx, c, n);
        if (i >= n) {
            set_error("uniform_index callback returned %llu for n = %llu", (unsigned long long)i, (unsigned long long)n);
            return DANN_EINVAL;
        }
        h_chosen[c] = (int64_t)i;
        h_active[c] = 1;
        DANN_HIP(hipMemcpy((uint8_t*)dpicked.p + (size_t)c * n + i, &one, 1, hipMemcpyHostToDevice));
        sel[c] = 1;
    }
    DANN_HIP(hipMemcpy(dchosen, h_chosen, (size_t)nchunks * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kpp_commit_kernel, dim3(nchunks), dim3(64), 0, 0, dx, n, dim, doff, (const float*)dnorm.p,
                       (const int64_t*)dchosen, 0u, dcen, (float*)dlast.p, (float*)dlastn.p);
    const uint64_t limit = std::min<uint64_t>(ncenters, n);
    int32_t status = DANN_OK;
    // the picks of centre `at` (read back with the next centre's totals, or at the end) decide which chunks go on
    auto account = [&](uint64_t at) {
        for (uint32_t c = 0; c < nchunks; ++c) {
            if (!h_active[c]) continue;
            if (h_chosen[c] < 0) h_active[c] = 0;  // InsufficientDiversity: this chunk stops, its remaining centres stay zero
            else sel[c] = (uint32_t)at + 1;
        }
    };
    uint64_t pending = 0;  // the centre whose picks have not been read back yet (0: none)
    for (uint64_t cur = 1; cur < limit; ++cur) {
        hipLaunchKernelGGL(kpp_update_kernel, rows_grid, dim3(256), 0, 0, dx, n, dim, doff, (const float*)dnorm.p,
                           (const float*)dlast.p, (const float*)dlastn.p, (float*)dmins.p, (double*)dbs.p, nblocks);
        hipLaunchKernelGGL(kpp_total_par_kernel, dim3(nchunks), dim3(1024), 0, 0, (const double*)dbs.p, nblocks, dtot);
        DANN_HIP(hipMemcpy(hback.p, dback.p, back_bytes, hipMemcpyDeviceToHost));
        if (pending) account(pending);
        pending = 0;
        bool any = false;
        for (uint32_t c = 0; c < nchunks; ++c) {
            if (!h_active[c]) continue;
            const double s = h_tot[c];
            if (!(0.0 < s)) {  // Uniform::<f64>::new(0.0, s) -> EmptyRange: no pick -> InsufficientDiversity
                h_active[c] = 0;
                continue;
            }
            if (!std::isfinite(s)) {  // NonFinite -> FailureReason::SawInfinity (not recoverable)
                set_error("k-means++ (chunk %u): a value of infinity or NaN was observed", c);
                status = DANN_EINVAL;
                h_active[c] = 0;
                continue;
            }
            h_thr[c] = rng->uniform_f64(rng->ctx, c, s);
            any = true;
        }
        if (status != DANN_OK || !any) break;
        DANN_HIP(hipMemcpyAsync(dfwd.p, hfwd.p, fwd_bytes, hipMemcpyHostToDevice, 0));
        hipLaunchKernelGGL(kpp_select_par_kernel, dim3(nchunks), dim3(1024), 0, 0, (const float*)dmins.p, n,
                           (const double*)dthr, (const uint8_t*)dact, (uint8_t*)dpicked.p, dchosen);
        hipLaunchKernelGGL(kpp_commit_kernel, dim3(nchunks), dim3(64), 0, 0, dx, n, dim, doff, (const float*)dnorm.p,
                           (const int64_t*)dchosen, (uint32_t)cur, dcen, (float*)dlast.p, (float*)dlastn.p);
        pending = cur;
        // (the forward buffer is rewritten by the host only after the next blocking read-back: the copy has landed)
    }
    if (pending) {
        DANN_HIP(hipMemcpy(hback.p, dback.p, back_bytes, hipMemcpyDeviceToHost));
        account(pending);
    }
    DANN_HIP(hipDeviceSynchronize());
    return status;
}
int32_t pq_kmeanspp_args(const float* data, uint64_t n, uint32_t dim, const uint32_t* chunk_offsets, uint32_t nchunks,
                         const dann_rng* rng, const float* centers) {
    if (!data || !chunk_offsets || !centers || !rng || !rng->uniform_index || !rng->uniform_f64 || nchunks == 0 || dim == 0)
        return DANN_EINVAL;
    if (n > 0xFFFFFFFFull) return DANN_EINVAL;
    return pq_check_offsets(chunk_offsets, nchunks, dim, nullptr);
}
}  // namespace
}  // namespace dann

extern "C" int32_t dann_pq_kmeanspp(int32_t device, const float* data, uint64_t n, uint32_t dim,
                                    const uint32_t* chunk_offsets, uint32_t nchunks, uint32_t ncenters, const dann_rng* rng,
                                    float* centers, uint32_t* selected) try {
    using namespace dann;
    int32_t rc = pq_kmeanspp_args(data, n, dim, chunk_offsets, nchunks, rng, centers);
    if (rc != DANN_OK) return rc;
    memset(centers, 0, (size_t)ncenters * dim * 4);
    if (selected) memset(selected, 0, (size_t)nchunks * 4);
    if (n == 0 || ncenters == 0) return DANN_OK;  // DatasetTooSmall is recoverable: all centres stay zero
    if (device >= 0) DANN_HIP(hipSetDevice(device));
    Buf dx, doff, dcen;
    DANN_HIP(hipMalloc(&dx.p, n * dim * 4));
    DANN_HIP(hipMalloc(&doff.p, (size_t)(nchunks + 1) * 4));
    DANN_HIP(hipMalloc(&dcen.p, (size_t)ncenters * dim * 4));
    DANN_HIP(hipMemcpy(dx.p, data, n * dim * 4, hipMemcpyHostToDevice));
    DANN_HIP(hipMemcpy(doff.p, chunk_offsets, (size_t)(nchunks + 1) * 4, hipMemcpyHostToDevice));
    std::vector<uint32_t> sel(nchunks, 0);
    rc = pq_kmeanspp_device((const float*)dx.p, n, dim, (const uint32_t*)doff.p, nchunks, ncenters, rng, (float*)dcen.p, sel);
    DANN_HIP(hipMemcpy(centers, dcen.p, (size_t)ncenters * dim * 4, hipMemcpyDeviceToHost));
    if (selected) memcpy(selected, sel.data(), (size_t)nchunks * 4);
    return rc;
} DANN_CATCH_ALL

// LightPQTrainingParameters::train (product/train.rs:96-226): k-means++ seeding then the Lloyd iterations, per chunk;
// the rows are uploaded once and the seeds never leave the device
extern "C" int32_t dann_pq_train(int32_t device, const float* data, uint64_t n, uint32_t dim, const uint32_t* chunk_offsets,
                                 uint32_t nchunks, uint32_t ncenters, uint32_t lloyds_reps, const dann_rng* rng,
                                 float* pivots) try {
    using namespace dann;
    int32_t rc = pq_kmeanspp_args(data, n, dim, chunk_offsets, nchunks, rng, pivots);
    if (rc != DANN_OK) return rc;
    memset(pivots, 0, (size_t)ncenters * dim * 4);
    if (ncenters == 0) return DANN_EINVAL;
    if (n == 0) return DANN_EINVAL;
    if (device >= 0) DANN_HIP(hipSetDevice(device));
    Buf dx, doff, dcen;
    DANN_HIP(hipMalloc(&dx.p, n * dim * 4));
    DANN_HIP(hipMalloc(&doff.p, (size_t)(nchunks + 1) * 4));
    DANN_HIP(hipMalloc(&dcen.p, (size_t)ncenters * dim * 4));
    DANN_HIP(hipMemcpy(dx.p, data, n * dim * 4, hipMemcpyHostToDevice));
    DANN_HIP(hipMemcpy(doff.p, chunk_offsets, (size_t)(nchunks + 1) * 4, hipMemcpyHostToDevice));
    std::vector<uint32_t> sel(nchunks, 0);
    rc = pq_kmeanspp_device((const float*)dx.p, n, dim, (const uint32_t*)doff.p, nchunks, ncenters, rng, (float*)dcen.p, sel);
    if (rc != DANN_OK) return rc;
    rc = pq_lloyds_device((const float*)dx.p, n, dim, chunk_offsets, (const uint32_t*)doff.p, nchunks, ncenters,
                          (float*)dcen.p, lloyds_reps, nullptr, nullptr);
    if (rc != DANN_OK) return rc;
    DANN_HIP(hipMemcpy(pivots, dcen.p, (size_t)ncenters * dim * 4, hipMemcpyDeviceToHost));
    return DANN_OK;
} DANN_CATCH_ALL

// dann_debug.h: how the trainer's rolling f64 sums were evaluated since the last reset
extern "C" int32_t dann_debug_pq_rolling_sum_stats(int32_t device, uint64_t* out4, int32_t reset) try {
    using namespace dann;
    if (device >= 0) DANN_HIP(hipSetDevice(device));
    unsigned long long h[4] = {0, 0, 0, 0};
    if (out4) {
        DANN_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_seq_stats), sizeof h));
        for (int i = 0; i < 4; ++i) out4[i] = h[i];
    }
    if (reset) {
        const unsigned long long z[4] = {0, 0, 0, 0};
        DANN_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_seq_stats), z, sizeof z));
    }
    return DANN_OK;
} DANN_CATCH_ALL
