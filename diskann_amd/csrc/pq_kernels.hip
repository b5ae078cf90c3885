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
#include <algorithm>

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

struct Buf {
    void* p = nullptr;
    ~Buf() {
        if (p) (void)hipFree(p);
    }
};

}  // namespace
}  // namespace dann

using namespace dann;

extern "C" {

int32_t dann_pq_build_lut(int32_t device, int32_t metric, const float* pivots, const uint32_t* chunk_offsets,
                          uint32_t nchunks, uint32_t dim, const float* queries, uint32_t nq, float* lut) {
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
}

int32_t dann_pq_scan(int32_t device, const float* lut, uint32_t nq, uint32_t nchunks, const uint8_t* codes,
                     uint64_t npoints, const uint32_t* ids, const uint64_t* offsets, float* out) {
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
}

}  // extern "C"

extern "C" int32_t dann_pq_compress(int32_t device, const float* pivots, uint32_t ncenters, const uint32_t* chunk_offsets,
                                    uint32_t nchunks, uint32_t dim, const float* rows, uint64_t n, uint8_t* codes) {
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
}

extern "C" int32_t dann_sq8_compress(int32_t device, const float* x, uint32_t n, uint32_t dim, const float* shift,
                                     float scale, void* out) {
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
}
