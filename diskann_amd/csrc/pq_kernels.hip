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
__global__ void kpp_total_kernel(const double* block_sums, uint64_t nblocks, double* totals) {
    const uint32_t c = blockIdx.x;
    if (threadIdx.x != 0) return;
    double s = 0.0;
    for (uint64_t b = 0; b < nblocks; ++b) s += block_sums[(uint64_t)c * nblocks + b];
    totals[c] = s;
}
// the D^2 draw (plusplus.rs:446-462): first row whose running f64 sum reaches the threshold, with a positive minimum,
// not picked before.  Sequential by definition; one thread per chunk, chunks in parallel.
__global__ void kpp_select_kernel(const float* mins, uint64_t n, const double* thresholds, const uint8_t* active,
                                  uint8_t* picked, int64_t* chosen) {
    const uint32_t c = blockIdx.x;
    if (threadIdx.x != 0) return;
    chosen[c] = -1;
    if (!active[c]) return;
    const double t = thresholds[c];
    double acc = 0.0;
    for (uint64_t i = 0; i < n; ++i) {
        const float m = mins[(uint64_t)c * n + i];
        acc += (double)m;
        if (acc >= t && m > 0.0f && !picked[(uint64_t)c * n + i]) {
            picked[(uint64_t)c * n + i] = 1;
            chosen[c] = (int64_t)i;
            return;
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
    if (i < n && assign[i] < ncenters) atomicAdd(&counts[assign[i]], 1u);
}
// update_centroids (lloyds.rs:273-296): f64 sums in row order; block = one centre of one chunk, thread = dimension.
// `order` lists the rows of each centre in increasing row order (stable radix sort by centre).
__global__ void pq_update_kernel(const float* data, uint32_t dim, uint32_t s0, uint32_t len, const uint32_t* order,
                                 const uint32_t* counts, const uint32_t* starts, float* centers) {
    const uint32_t j = blockIdx.x, d = threadIdx.x;
    if (d >= len) return;
    const uint32_t cnt = counts[j];
    const uint32_t* list = order + starts[j];
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
__global__ void pq_scan_counts_kernel(const uint32_t* counts, uint32_t ncenters, uint32_t* starts) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint32_t acc = 0;
        for (uint32_t j = 0; j < ncenters; ++j) {
            starts[j] = acc;
            acc += counts[j];
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

// dann_pq_pack_neighbors: one wavefront per node.  Row = [len][ids x R][pad to 16][16-byte code row of neighbour 0 .. R-1]
// (neighbours beyond the length, and ids beyond the index, get zero code rows: they are never candidates).
__global__ __launch_bounds__(256) void pq_pack_kernel(IndexView ix, uint8_t* pack, uint32_t stride, uint32_t codes_off) {
    const uint32_t node = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (node >= ix.nslots) return;
    const uint32_t R = ix.max_degree;
    const uint32_t* arow = ix.adj + (uint64_t)node * ix.adj_stride;
    uint8_t* prow = pack + (uint64_t)node * stride;
    const uint32_t len = arow[0] < R ? arow[0] : R;
    for (uint32_t w = lane; w < codes_off / 4u; w += 64u)
        reinterpret_cast<uint32_t*>(prow)[w] = w <= R ? arow[w] : 0u;  // (the stored length word, unclamped, as the adjacency row has it)
    for (uint32_t j = lane; j < R; j += 64u) {
        uint4 code = make_uint4(0u, 0u, 0u, 0u);
        if (j < len) {
            const uint32_t id = arow[1u + j];
            if (id < ix.nslots) code = *reinterpret_cast<const uint4*>(ix.rows + (uint64_t)id * ix.row_stride);
        }
        *reinterpret_cast<uint4*>(prow + codes_off + 16u * j) = code;
    }
    for (uint32_t o = codes_off + 16u * R + 4u * lane; o < stride; o += 256u) *reinterpret_cast<uint32_t*>(prow + o) = 0u;
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

extern "C" int32_t dann_pq_lloyds(int32_t device, const float* data, uint64_t n, uint32_t dim,
                                  const uint32_t* chunk_offsets, uint32_t nchunks, uint32_t ncenters, float* centers,
                                  uint32_t max_reps, uint32_t* assignments, float* residuals) try {
    using namespace dann;
    if (!data || !chunk_offsets || !centers || nchunks == 0 || dim == 0 || ncenters == 0) return DANN_EINVAL;
    if (n == 0 || n > 0xFFFFFFFFull) return DANN_EINVAL;
    if (chunk_offsets[0] != 0 || chunk_offsets[nchunks] != dim) {
        set_error("chunk offsets must start at 0 and end at dim");
        return DANN_EINVAL;
    }
    uint32_t maxlen = 0;
    for (uint32_t c = 0; c < nchunks; ++c) {
        if (chunk_offsets[c + 1] <= chunk_offsets[c]) return DANN_EINVAL;
        maxlen = std::max(maxlen, chunk_offsets[c + 1] - chunk_offsets[c]);
    }
    const size_t lds = ((size_t)ncenters * maxlen + ncenters) * 4;
    if (lds > 160 * 1024 || maxlen > 1024) {
        set_error("PQ chunk of %u dimensions x %u centres does not fit the 160 KiB LDS slab", maxlen, ncenters);
        return DANN_EUNSUPPORTED;
    }
    if (device >= 0) DANN_HIP(hipSetDevice(device));
    if (lds > 64 * 1024)
        DANN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pq_assign_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    Buf dx, doff, dcen, dcn, ddn, dasg, dbest, dres, dord, dord2, dkeys2, dcounts, dstarts, dtmp, dbad;
    DANN_HIP(hipMalloc(&dx.p, n * dim * 4));
    DANN_HIP(hipMalloc(&doff.p, (size_t)(nchunks + 1) * 4));
    DANN_HIP(hipMalloc(&dcen.p, (size_t)ncenters * dim * 4));
    DANN_HIP(hipMalloc(&dcn.p, (size_t)nchunks * ncenters * 4));
    DANN_HIP(hipMalloc(&ddn.p, (size_t)nchunks * n * 4));
    DANN_HIP(hipMalloc(&dasg.p, (size_t)nchunks * n * 4));
    DANN_HIP(hipMalloc(&dbest.p, (size_t)nchunks * n * 4));
    DANN_HIP(hipMalloc(&dres.p, (size_t)nchunks * 4));
    DANN_HIP(hipMalloc(&dord.p, n * 4));
    DANN_HIP(hipMalloc(&dord2.p, n * 4));
    DANN_HIP(hipMalloc(&dkeys2.p, n * 4));
    DANN_HIP(hipMalloc(&dcounts.p, (size_t)ncenters * 4));
    DANN_HIP(hipMalloc(&dstarts.p, (size_t)ncenters * 4));
    DANN_HIP(hipMalloc(&dbad.p, 4));
    DANN_HIP(hipMemset(dbad.p, 0, 4));
    DANN_HIP(hipMemcpy(dx.p, data, n * dim * 4, hipMemcpyHostToDevice));
    DANN_HIP(hipMemcpy(doff.p, chunk_offsets, (size_t)(nchunks + 1) * 4, hipMemcpyHostToDevice));
    DANN_HIP(hipMemcpy(dcen.p, centers, (size_t)ncenters * dim * 4, hipMemcpyHostToDevice));
    int key_bits = 1;
    while ((1u << key_bits) < ncenters && key_bits < 32) ++key_bits;
    size_t tmp_bytes = 0;
    DANN_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, (const uint32_t*)dasg.p, (uint32_t*)dkeys2.p,
                                                (const uint32_t*)dord.p, (uint32_t*)dord2.p, (int)n, 0, key_bits));
    DANN_HIP(hipMalloc(&dtmp.p, tmp_bytes + 16));
    const dim3 rows_grid((uint32_t)((n + 255) / 256), nchunks);
    hipLaunchKernelGGL(pq_data_norms_kernel, rows_grid, dim3(256), 0, 0, (const float*)dx.p, n, dim,
                       (const uint32_t*)doff.p, (float*)ddn.p);
    hipLaunchKernelGGL(pq_center_norms_kernel, dim3((ncenters + 255) / 256, nchunks), dim3(256), 0, 0,
                       (const float*)dcen.p, ncenters, dim, (const uint32_t*)doff.p, (float*)dcn.p);
    hipLaunchKernelGGL(pq_iota_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, (uint32_t*)dord.p, n);
    for (uint32_t rep = 0; rep < max_reps; ++rep) {
        hipLaunchKernelGGL(pq_assign_kernel, rows_grid, dim3(256), lds, 0, (const float*)dx.p, n, dim,
                           (const uint32_t*)doff.p, (const float*)dcen.p, ncenters, (const float*)dcn.p,
                           (const float*)ddn.p, (uint32_t*)dasg.p, (float*)dbest.p);
        hipLaunchKernelGGL(pq_residual_kernel, dim3(nchunks), dim3(256), 0, 0, (const float*)dbest.p, n, (float*)dres.p,
                           (uint32_t*)dbad.p, (const uint32_t*)dasg.p);
        uint32_t bad = 0;
        DANN_HIP(hipMemcpy(&bad, dbad.p, 4, hipMemcpyDeviceToHost));
        if (bad) {
            set_error("k-means assignment saw only NaN scores for some row (non-finite data or centres)");
            return DANN_EINVAL;
        }
        for (uint32_t c = 0; c < nchunks; ++c) {
            const uint32_t* asg = (const uint32_t*)dasg.p + (size_t)c * n;
            DANN_HIP(hipMemsetAsync(dcounts.p, 0, (size_t)ncenters * 4, 0));
            hipLaunchKernelGGL(pq_hist_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, asg, n, ncenters,
                               (uint32_t*)dcounts.p);
            hipLaunchKernelGGL(pq_scan_counts_kernel, dim3(1), dim3(1), 0, 0, (const uint32_t*)dcounts.p, ncenters,
                               (uint32_t*)dstarts.p);
            size_t tb = tmp_bytes;
            DANN_HIP(hipcub::DeviceRadixSort::SortPairs(dtmp.p, tb, asg, (uint32_t*)dkeys2.p, (const uint32_t*)dord.p,
                                                        (uint32_t*)dord2.p, (int)n, 0, key_bits));
            const uint32_t s0 = chunk_offsets[c], len = chunk_offsets[c + 1] - s0;
            hipLaunchKernelGGL(pq_update_kernel, dim3(ncenters), dim3((len + 63) / 64 * 64), 0, 0, (const float*)dx.p, dim,
                               s0, len, (const uint32_t*)dord2.p, (const uint32_t*)dcounts.p,
                               (const uint32_t*)dstarts.p, (float*)dcen.p);
        }
        if (rep != max_reps - 1)
            hipLaunchKernelGGL(pq_center_norms_kernel, dim3((ncenters + 255) / 256, nchunks), dim3(256), 0, 0,
                               (const float*)dcen.p, ncenters, dim, (const uint32_t*)doff.p, (float*)dcn.p);
        DANN_HIP(hipGetLastError());
    }
    DANN_HIP(hipMemcpy(centers, dcen.p, (size_t)ncenters * dim * 4, hipMemcpyDeviceToHost));
    if (assignments) DANN_HIP(hipMemcpy(assignments, dasg.p, (size_t)nchunks * n * 4, hipMemcpyDeviceToHost));
    if (residuals) DANN_HIP(hipMemcpy(residuals, dres.p, (size_t)nchunks * 4, hipMemcpyDeviceToHost));
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
extern "C" int32_t dann_pq_kmeanspp(int32_t device, const float* data, uint64_t n, uint32_t dim,
                                    const uint32_t* chunk_offsets, uint32_t nchunks, uint32_t ncenters, const dann_rng* rng,
                                    float* centers, uint32_t* selected) try {
    using namespace dann;
    if (!data || !chunk_offsets || !centers || !rng || !rng->uniform_index || !rng->uniform_f64 || nchunks == 0 || dim == 0)
        return DANN_EINVAL;
    if (n > 0xFFFFFFFFull) return DANN_EINVAL;
    if (chunk_offsets[0] != 0 || chunk_offsets[nchunks] != dim) {
        set_error("chunk offsets must start at 0 and end at dim");
        return DANN_EINVAL;
    }
    for (uint32_t c = 0; c < nchunks; ++c)
        if (chunk_offsets[c + 1] <= chunk_offsets[c]) return DANN_EINVAL;
    memset(centers, 0, (size_t)ncenters * dim * 4);
    if (selected) memset(selected, 0, (size_t)nchunks * 4);
    if (n == 0 || ncenters == 0) return DANN_OK;  // DatasetTooSmall is recoverable: all centres stay zero
    if (device >= 0) DANN_HIP(hipSetDevice(device));
    const uint64_t nblocks = (n + 15) / 16;
    Buf dx, doff, dcen, dnorm, dmins, dbs, dtot, dthr, dact, dpicked, dchosen, dlast, dlastn;
    DANN_HIP(hipMalloc(&dx.p, n * dim * 4));
    DANN_HIP(hipMalloc(&doff.p, (size_t)(nchunks + 1) * 4));
    DANN_HIP(hipMalloc(&dcen.p, (size_t)ncenters * dim * 4));
    DANN_HIP(hipMalloc(&dnorm.p, (size_t)nchunks * n * 4));
    DANN_HIP(hipMalloc(&dmins.p, (size_t)nchunks * n * 4));
    DANN_HIP(hipMalloc(&dbs.p, (size_t)nchunks * nblocks * 8));
    DANN_HIP(hipMalloc(&dtot.p, (size_t)nchunks * 8));
    DANN_HIP(hipMalloc(&dthr.p, (size_t)nchunks * 8));
    DANN_HIP(hipMalloc(&dact.p, nchunks));
    DANN_HIP(hipMalloc(&dpicked.p, (size_t)nchunks * n));
    DANN_HIP(hipMalloc(&dchosen.p, (size_t)nchunks * 8));
    DANN_HIP(hipMalloc(&dlast.p, (size_t)nchunks * dim * 4));
    DANN_HIP(hipMalloc(&dlastn.p, (size_t)nchunks * 4));
    DANN_HIP(hipMemcpy(dx.p, data, n * dim * 4, hipMemcpyHostToDevice));
    DANN_HIP(hipMemcpy(doff.p, chunk_offsets, (size_t)(nchunks + 1) * 4, hipMemcpyHostToDevice));
    DANN_HIP(hipMemset(dcen.p, 0, (size_t)ncenters * dim * 4));
    DANN_HIP(hipMemset(dpicked.p, 0, (size_t)nchunks * n));
    const dim3 rows_grid((uint32_t)((n + 255) / 256), nchunks);
    hipLaunchKernelGGL(pq_data_norms_kernel, rows_grid, dim3(256), 0, 0, (const float*)dx.p, n, dim,
                       (const uint32_t*)doff.p, (float*)dnorm.p);
    {
        const uint64_t tot = (uint64_t)nchunks * n;
        hipLaunchKernelGGL(kpp_fill_kernel, dim3((uint32_t)((tot + 255) / 256)), dim3(256), 0, 0, (float*)dmins.p, tot,
                           __builtin_inff());
    }
    // first centre of every chunk: Uniform::new(0, n).sample(rng)
    std::vector<int64_t> h_chosen(nchunks);
    std::vector<uint8_t> h_pick1(1, 1), h_active(nchunks, 1);
    std::vector<uint32_t> sel(nchunks, 0);
    for (uint32_t c = 0; c < nchunks; ++c) {
        const uint64_t i = rng->uniform_index(rng->ctx, c, n);
        if (i >= n) {
            set_error("uniform_index callback returned %llu for n = %llu", (unsigned long long)i, (unsigned long long)n);
            return DANN_EINVAL;
        }
        h_chosen[c] = (int64_t)i;
        DANN_HIP(hipMemcpy((uint8_t*)dpicked.p + (size_t)c * n + i, h_pick1.data(), 1, hipMemcpyHostToDevice));
        sel[c] = 1;
    }
    DANN_HIP(hipMemcpy(dchosen.p, h_chosen.data(), (size_t)nchunks * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kpp_commit_kernel, dim3(nchunks), dim3(64), 0, 0, (const float*)dx.p, n, dim, (const uint32_t*)doff.p,
                       (const float*)dnorm.p, (const int64_t*)dchosen.p, 0u, (float*)dcen.p, (float*)dlast.p,
                       (float*)dlastn.p);
    const uint64_t limit = std::min<uint64_t>(ncenters, n);
    std::vector<double> h_tot(nchunks), h_thr(nchunks);
    int32_t status = DANN_OK;
    for (uint64_t cur = 1; cur < limit; ++cur) {
        bool any = false;
        for (uint32_t c = 0; c < nchunks; ++c) any |= h_active[c] != 0;
        if (!any) break;
        hipLaunchKernelGGL(kpp_update_kernel, rows_grid, dim3(256), 0, 0, (const float*)dx.p, n, dim, (const uint32_t*)doff.p,
                           (const float*)dnorm.p, (const float*)dlast.p, (const float*)dlastn.p, (float*)dmins.p,
                           (double*)dbs.p, nblocks);
        hipLaunchKernelGGL(kpp_total_kernel, dim3(nchunks), dim3(64), 0, 0, (const double*)dbs.p, nblocks, (double*)dtot.p);
        DANN_HIP(hipMemcpy(h_tot.data(), dtot.p, (size_t)nchunks * 8, hipMemcpyDeviceToHost));
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
        }
        if (status != DANN_OK) break;
        DANN_HIP(hipMemcpy(dthr.p, h_thr.data(), (size_t)nchunks * 8, hipMemcpyHostToDevice));
        DANN_HIP(hipMemcpy(dact.p, h_active.data(), nchunks, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(kpp_select_kernel, dim3(nchunks), dim3(64), 0, 0, (const float*)dmins.p, n, (const double*)dthr.p,
                           (const uint8_t*)dact.p, (uint8_t*)dpicked.p, (int64_t*)dchosen.p);
        hipLaunchKernelGGL(kpp_commit_kernel, dim3(nchunks), dim3(64), 0, 0, (const float*)dx.p, n, dim,
                           (const uint32_t*)doff.p, (const float*)dnorm.p, (const int64_t*)dchosen.p, (uint32_t)cur,
                           (float*)dcen.p, (float*)dlast.p, (float*)dlastn.p);
        DANN_HIP(hipMemcpy(h_chosen.data(), dchosen.p, (size_t)nchunks * 8, hipMemcpyDeviceToHost));
        for (uint32_t c = 0; c < nchunks; ++c) {
            if (!h_active[c]) continue;
            if (h_chosen[c] < 0) h_active[c] = 0;  // InsufficientDiversity: this chunk stops, its remaining centres stay zero
            else sel[c] = (uint32_t)cur + 1;
        }
    }
    DANN_HIP(hipMemcpy(centers, dcen.p, (size_t)ncenters * dim * 4, hipMemcpyDeviceToHost));
    if (selected) memcpy(selected, sel.data(), (size_t)nchunks * 4);
    return status;
} DANN_CATCH_ALL

// LightPQTrainingParameters::train (product/train.rs:96-226): k-means++ seeding then the Lloyd iterations, per chunk
extern "C" int32_t dann_pq_train(int32_t device, const float* data, uint64_t n, uint32_t dim, const uint32_t* chunk_offsets,
                                 uint32_t nchunks, uint32_t ncenters, uint32_t lloyds_reps, const dann_rng* rng,
                                 float* pivots) try {
    int32_t rc = dann_pq_kmeanspp(device, data, n, dim, chunk_offsets, nchunks, ncenters, rng, pivots, nullptr);
    if (rc != DANN_OK) return rc;
    return dann_pq_lloyds(device, data, n, dim, chunk_offsets, nchunks, ncenters, pivots, lloyds_reps, nullptr, nullptr);
} DANN_CATCH_ALL
