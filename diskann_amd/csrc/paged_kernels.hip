// paged_kernels.hip -- DiskANNIndex::paged_search + PagedSearch::next_page on the GPU
// (diskann/src/graph/index.rs:2075-2155, diskann/src/graph/search/paged.rs:53-149).
//
// A paged search keeps its scratch between calls: an *unbounded* sorted candidate list (the auto-resizable
// NeighborPriorityQueue, queue.rs:95-121: nothing is ever dropped), the visited set and the tail of the last
// computed page.  That state lives in HBM per query: the list as (id | visited bit, distance) arrays in two
// buffers (every merge writes the other buffer), a `base` offset that implements drain_best (queue.rs:174-180)
// without moving anything, an exact hash set of ids, and the page cache.  One wave drives one query:
//   next_page(k): serve from the cache; else resume search_internal (pop the first unvisited entry among the
//   first L + starts, expand it, evaluate unseen neighbours with the search-path distance groups, merge them
//   by rank: old entry moves up by #{new <= it}, new entry lands at #{old < it} + #{new before it}); then the
//   first k entries become the new page and are drained.
// The list is global-memory resident, so a hop costs O(list length / 64) -- paged search trades throughput for
// resumability; the batched Knn kernel (search_kernel_impl.h) is the throughput path.
#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "dann_device.h"
#include "dann_internal.h"

namespace dann {
namespace {

constexpr int kWave = 64;

struct PagedArgs {
    IndexView ix;
    const void* queries;  // nq rows, resident for the lifetime of the session
    uint32_t nq, l_value, k, cap;
    uint32_t* ids[2];     // nq x cap each; bit 31 = expanded
    float* d[2];
    uint32_t* which;      // nq: buffer holding the list
    uint32_t* base;       // nq: first live entry (drain_best)
    uint32_t* size;       // nq: live entries
    uint32_t* vt;         // nq x 2^vt_bits visited ids (kEmpty = free)
    uint32_t vt_bits;
    uint32_t* vt_count;   // nq
    uint32_t* cache_ids;  // nq x l_value: computed_result
    float* cache_d;
    uint32_t* cache_n;    // nq
    uint32_t* cache_next; // nq
    uint32_t* out_ids;    // nq x k
    float* out_d;
    uint32_t* out_n;      // nq
    dann_search_stats* stats;  // nq, cumulative cmps / hops
    uint32_t init;
};

__device__ __forceinline__ uint32_t ld_u32(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_f32(const float* p) {
    return __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int DT, int OP, bool NORM>
__global__ __launch_bounds__(kWave) void paged_kernel(PagedArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    using S = Scheme<DT, OP, false>;
    constexpr int G = S::G, GROUPS = kWave / G;
    constexpr bool kInt = S::kInt;
    using QT = typename std::conditional<kInt, uint8_t, float>::type;
    using RT = typename RowType<DT>::type;
    const IndexView& ix = a.ix;
    const uint32_t lane = threadIdx.x, qi = blockIdx.x;
    const uint32_t R = ix.max_degree;
    const uint32_t rc = (R + 63u) & ~63u;
    uint32_t* cand_id = reinterpret_cast<uint32_t*>(smem);
    float* cand_d = reinterpret_cast<float*>(smem + (size_t)rc * 4);
    QT* qs = reinterpret_cast<QT*>(smem + (size_t)rc * 8);
    const SqParams sqp{ix.sq_k, ix.sq_shift_norm_sq};
    {
        const uint8_t* qsrc = reinterpret_cast<const uint8_t*>(a.queries) + (uint64_t)qi * ix.layer_bytes;
        if constexpr (kInt) {
            for (uint32_t i = lane; i < ix.layer_bytes; i += kWave) reinterpret_cast<uint8_t*>(qs)[i] = qsrc[i];
        } else {
            const RT* src = reinterpret_cast<const RT*>(qsrc);
            for (uint32_t i = lane; i < ix.dim; i += kWave) reinterpret_cast<float*>(qs)[i] = load1(src + i);
        }
    }
    __syncthreads();
    const int g = lane / G, v = lane % G;
    uint32_t cur = a.which[qi], base = a.base[qi], size = a.size[qi], vcount = a.vt_count[qi];
    uint32_t status = a.stats[qi].status, cmps = a.stats[qi].cmps, hops = a.stats[qi].hops;
    uint32_t* vt = a.vt + ((uint64_t)qi << a.vt_bits);
    const uint32_t vmask = (1u << a.vt_bits) - 1u, vshift = 32u - a.vt_bits;
    auto list_ids = [&](uint32_t w) { return a.ids[w] + (uint64_t)qi * a.cap; };
    auto list_d = [&](uint32_t w) { return a.d[w] + (uint64_t)qi * a.cap; };

    // visited.insert(id): exact set, linear probing, agent-scope CAS (the table outlives the launch)
    auto visit = [&](uint32_t id) -> bool {
        uint32_t h = (id * 2654435761u) >> vshift;
        for (;;) {
            const uint32_t old = atomicCAS(&vt[h], kEmpty, id);
            if (old == kEmpty) return true;
            if (old == id) return false;
            h = (h + 1) & vmask;
        }
    };
    // unseen, in-bounds neighbours of `node` -> cand_id[0..nc)   (provider.rs:448-454)
    auto expand = [&](uint32_t node) -> uint32_t {
        const uint32_t* arow = ix.adj + (uint64_t)node * ix.adj_stride;
        uint32_t len = arow[0];
        len = len < R ? len : R;
        uint32_t nc = 0;
        if (vcount + len > ((vmask + 1u) >> 1) + ((vmask + 1u) >> 2)) {  // keep the table under 75 %
            status = (uint32_t)(-DANN_EOVERFLOW);
            return 0;
        }
        for (uint32_t j0 = 0; j0 < len; j0 += kWave) {
            const uint32_t j = j0 + lane;
            const uint32_t id = j < len ? arow[1 + j] : kEmpty;
            const bool isnew = id != kEmpty && visit(id);
            // unreadable slots (inline tag below PUBLISHED) are skipped after the visited insert (provider.rs:681-686)
            const bool keep = isnew && id < ix.nslots &&
                              (!ix.tag_off || ix.rows[(uint64_t)id * ix.row_stride + ix.tag_off] >= 254);
            const uint64_t nm = ballot64(isnew), km = ballot64(keep);
            if (keep) cand_id[nc + mbcnt(km)] = id;
            nc += (uint32_t)__popcll(km);
            vcount += (uint32_t)__popcll(nm);
        }
        return nc;
    };
    auto gather = [&](uint32_t nc) {
        constexpr int U = 4;
        for (uint32_t c0 = 0; c0 < nc; c0 += GROUPS * U) {
            const uint8_t* rows[U];
            bool act[U];
            float o[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t c = c0 + u * GROUPS + g;
                act[u] = c < nc;
                rows[u] = ix.rows + (uint64_t)(act[u] ? cand_id[c] : 0u) * ix.row_stride;
            }
            group_distance_many<DT, OP, false, U, false>(qs, rows, act, (int)ix.dim, v, o);  // G-lane groups
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t c = c0 + u * GROUPS + g;
                if (act[u] && v == 0)
                    cand_d[c] = finish_distance<DT, OP, NORM>(o[u], reinterpret_cast<const uint8_t*>(qs), rows[u], ix.dim, sqp);
            }
        }
    };
    // insert cand[m0 .. m0+n) (n <= 64) into the unbounded list: same rank rules as the bounded merge
    // (queue.rs:107-171: NaN ignored, a new element goes before equal old ones), nothing dropped
    auto merge = [&](uint32_t m0, uint32_t n) {
        const bool has = lane < n;
        const float nd = has ? cand_d[m0 + lane] : 0.0f;
        const uint32_t nid = has ? cand_id[m0 + lane] : kEmpty;
        const bool nvalid = has && !(nd != nd);
        const uint64_t km = ballot64(nvalid);
        const uint32_t nv = (uint32_t)__popcll(km);
        if (nv == 0) return;
        if (size + nv > a.cap) {
            status = (uint32_t)(-DANN_EOVERFLOW);
            return;
        }
        uint32_t before = 0, lb = 0;
        for (uint64_t mm = km; mm; mm &= mm - 1) {
            const int j = __builtin_ctzll(mm);
            const float dj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, nd), j));
            before += ((dj < nd) | ((dj == nd) & ((uint32_t)j > lane))) ? 1u : 0u;
        }
        const uint32_t* oi = list_ids(cur) + base;
        const float* od = list_d(cur) + base;
        uint32_t* ni = list_ids(cur ^ 1u);
        float* ndp = list_d(cur ^ 1u);
        for (uint32_t c0 = 0; c0 < size; c0 += kWave) {
            const uint32_t p = c0 + lane;
            const bool in = p < size;
            const float d0 = in ? ld_f32(od + p) : 0.0f;
            const uint32_t i0 = in ? ld_u32(oi + p) : 0u;
            uint32_t shift = 0;
            for (uint64_t mm = km; mm; mm &= mm - 1) {
                const int j = __builtin_ctzll(mm);
                const float dj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, nd), j));
                shift += (in & (dj <= d0)) ? 1u : 0u;
                const uint32_t c = (uint32_t)__popcll(ballot64(in & (d0 < dj)));
                lb += ((int)lane == j) ? c : 0u;
            }
            if (in) {
                ni[p + shift] = i0;
                ndp[p + shift] = d0;
            }
        }
        if (nvalid) {
            ni[before + lb] = nid;
            ndp[before + lb] = nd;
        }
        size += nv;
        cur ^= 1u;
        base = 0;
        drain();
    };

    if (a.init) {
        // paged_search_with_init_ids (index.rs:2109-2144): the start points seed the visited set and are
        // expanded; only their neighbours become candidates
        for (uint32_t i = lane; i < ix.nstart; i += kWave) visit(ix.capacity + i);
        vcount = ix.nstart;
        for (uint32_t s = 0; s < ix.nstart && !status; ++s) {
            __syncthreads();
            const uint32_t nc = expand(ix.capacity + s);
            __syncthreads();
            gather(nc);
            __syncthreads();
            for (uint32_t m0 = 0; m0 < nc; m0 += kWave) merge(m0, (nc - m0) < (uint32_t)kWave ? (nc - m0) : (uint32_t)kWave);
        }
    } else if (!status) {
        uint32_t* out_ids = a.out_ids + (uint64_t)qi * a.k;
        float* out_d = a.out_d + (uint64_t)qi * a.k;
        uint32_t* cids = a.cache_ids + (uint64_t)qi * a.l_value;
        float* cds = a.cache_d + (uint64_t)qi * a.l_value;
        uint32_t cn = a.cache_n[qi], cnext = a.cache_next[qi];
        // 1. drain already-computed results (paged.rs:67-83)
        uint32_t n_out = 0;
        {
            const uint32_t avail = cn > cnext ? cn - cnext : 0u;
            const uint32_t take = a.k < avail ? a.k : avail;
            for (uint32_t i = lane; i < take; i += kWave) {
                out_ids[i] = cids[cnext + i];
                out_d[i] = cds[cnext + i];
            }
            cnext += take;
            n_out = take;
        }
        if (n_out < a.k) {
            // 2. resume search_internal with beam width 1 (paged.rs:86-94, index.rs:1960-1990)
            const uint32_t search_l = a.l_value + ix.nstart;
            for (;;) {
                const uint32_t lim = search_l < size ? search_l : size;
                uint32_t pos = kEmpty;
                for (uint32_t c0 = 0; c0 < lim && pos == kEmpty; c0 += kWave) {
                    const uint32_t p = c0 + lane;
                    const bool open = p < lim && !(ld_u32(list_ids(cur) + base + p) & kVisitedBit);
                    const uint64_t m = ballot64(open);
                    if (m) pos = c0 + (uint32_t)__builtin_ctzll(m);
                }
                if (pos == kEmpty) break;
                const uint32_t node = ld_u32(list_ids(cur) + base + pos);
                if (lane == 0) list_ids(cur)[base + pos] = node | kVisitedBit;
                drain();
                __syncthreads();
                const uint32_t nc = expand(node);
                if (status) break;
                __syncthreads();
                gather(nc);
                __syncthreads();
                for (uint32_t m0 = 0; m0 < nc && !status; m0 += kWave)
                    merge(m0, (nc - m0) < (uint32_t)kWave ? (nc - m0) : (uint32_t)kWave);
                if (status) break;
                cmps += nc;
                hops += 1;
            }
            // 3. filter_search_candidates + drain_best (paged.rs:96-103, 126-149): start points never enter the
            //    list (they were seeded as visited), so the page is the first k of the first L + starts entries
            if (!status) {
                const uint32_t lim = search_l < size ? search_l : size;
                const uint32_t total = a.k < lim ? a.k : lim;
                for (uint32_t i = lane; i < total; i += kWave) {
                    cids[i] = ld_u32(list_ids(cur) + base + i) & ~kVisitedBit;
                    cds[i] = ld_f32(list_d(cur) + base + i);
                }
                drain();
                base += total;
                size -= total;
                cn = total;
                cnext = 0;
                const uint32_t left = (a.k - n_out) < cn ? (a.k - n_out) : cn;
                for (uint32_t i = lane; i < left; i += kWave) {
                    out_ids[n_out + i] = ld_u32(cids + i);
                    out_d[n_out + i] = ld_f32(cds + i);
                }
                cnext = left;
                n_out += left;
            }
        }
        for (uint32_t i = n_out + lane; i < a.k; i += kWave) {
            out_ids[i] = kEmpty;
            out_d[i] = __builtin_inff();
        }
        if (lane == 0) {
            a.out_n[qi] = status ? 0u : n_out;
            a.cache_n[qi] = cn;
            a.cache_next[qi] = cnext;
        }
    }
    if (lane == 0) {
        a.which[qi] = cur;
        a.base[qi] = base;
        a.size[qi] = size;
        a.vt_count[qi] = vcount;
        dann_search_stats st;
        st.cmps = cmps;
        st.hops = hops;
        st.result_count = 0;
        st.status = status;
        st.written = 0;
        a.stats[qi] = st;
    }
}

template <int DT, int OP, bool NORM>
int32_t launch_paged_t(const PagedArgs& a, hipStream_t stream) {
    const bool is_int = DT == DT_U8 || DT == DT_I8 || DT == DT_SQ8;
    const uint32_t rc = (a.ix.max_degree + 63u) & ~63u;
    const size_t lds = (size_t)rc * 8 + (((is_int ? a.ix.layer_bytes : a.ix.dim * 4u) + 15u) & ~15u);
    auto kern = paged_kernel<DT, OP, NORM>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute");
    }
    hipLaunchKernelGGL(kern, dim3(a.nq), dim3(kWave), lds, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "paged_kernel launch");
    return DANN_OK;
}

template <int DT>
int32_t launch_paged_dt(const PagedArgs& a, hipStream_t stream) {
    int op;
    bool norm;
    if (!resolve_metric(a.ix.dtype, a.ix.metric, &op, &norm)) return DANN_EUNSUPPORTED;
    if (op == OP_L2) {
        if constexpr (DT == DT_SQ8) {
            if (norm) return launch_paged_t<DT, OP_L2, true>(a, stream);
        }
        return launch_paged_t<DT, OP_L2, false>(a, stream);
    }
    if (op == OP_IP) {
        if constexpr (DT == DT_F32 || DT == DT_F16) {
            if (norm) return launch_paged_t<DT, OP_IP, true>(a, stream);
        }
        return launch_paged_t<DT, OP_IP, false>(a, stream);
    }
    if constexpr (DT != DT_SQ8) return launch_paged_t<DT, OP_COS, false>(a, stream);
    return DANN_EUNSUPPORTED;
}

int32_t launch_paged(const PagedArgs& a, hipStream_t stream) {
    switch (a.ix.dtype) {
        case DT_F32: return launch_paged_dt<DT_F32>(a, stream);
        case DT_F16: return launch_paged_dt<DT_F16>(a, stream);
        case DT_U8: return launch_paged_dt<DT_U8>(a, stream);
        case DT_I8: return launch_paged_dt<DT_I8>(a, stream);
        case DT_SQ8: return launch_paged_dt<DT_SQ8>(a, stream);
    }
    set_error("paged search is not defined for dtype %d", a.ix.dtype);
    return DANN_EUNSUPPORTED;
}

struct Dev {
    void* p = nullptr;
    ~Dev() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t n, int fill = -1) {
        hipError_t e = hipMalloc(&p, n ? n : 1);
        if (e == hipSuccess && fill >= 0) e = hipMemset(p, fill, n ? n : 1);
        return e;
    }
};

}  // namespace
}  // namespace dann

using namespace dann;

struct dann_paged {
    dann_index* idx = nullptr;
    PagedArgs a;
    Dev q, ids0, ids1, d0, d1, which, base, size, vt, vtc, cids, cd, cn, cnext, stats;
    Dev out_ids, out_d, out_n;
    uint32_t out_k = 0;
};

extern "C" {

int32_t dann_paged_begin(dann_index* idx, const void* queries, uint32_t nq, uint32_t l_value, uint32_t list_cap,
                         dann_paged** out) try {
    if (!idx || !out) return DANN_EINVAL;
    *out = nullptr;
    if (!queries || nq == 0 || l_value == 0) {
        set_error("paged search needs queries and a non-zero l_value");
        return DANN_EINVAL;
    }
    if (idx->cfg.dtype == DT_PQ) {
        set_error("paged search is not defined for DANN_PQ rows");
        return DANN_EUNSUPPORTED;
    }
    // read-only on the index: sessions of different callers run side by side, each call on a leased context
    std::shared_lock<std::shared_mutex> rd(idx->rw);
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (prev != idx->device) DANN_HIP(hipSetDevice(idx->device));
    ::dann::CtxLease lease(idx);
    if (lease.status != DANN_OK) return lease.status;
    hipStream_t stream = lease.ctx->stream;
    dann_paged* s = new dann_paged();
    s->idx = idx;
    const uint32_t nslots = idx->nslots;
    uint32_t cap = list_cap ? list_cap : std::min<uint32_t>(nslots, std::max<uint32_t>(16384, 64 * l_value));
    cap = std::max<uint32_t>(std::min<uint32_t>(cap, nslots), 64);
    uint32_t vbits = 8;
    while ((1ull << vbits) * 3 / 4 < (uint64_t)cap + idx->cfg.num_start_points + idx->cfg.max_degree + 64 && vbits < 31)
        ++vbits;
    PagedArgs& a = s->a;
    a.ix = idx->view();
    a.nq = nq;
    a.l_value = l_value;
    a.k = 0;
    a.cap = cap;
    a.vt_bits = vbits;
    const size_t qb = idx->layer_bytes;
    auto fail = [&](hipError_t e) {
        delete s;
        return hip_fail(e, "paged session allocation");
    };
    hipError_t e;
    if ((e = s->q.alloc((size_t)nq * qb + 16)) != hipSuccess) return fail(e);
    if ((e = s->ids0.alloc((size_t)nq * cap * 4)) != hipSuccess) return fail(e);
    if ((e = s->ids1.alloc((size_t)nq * cap * 4)) != hipSuccess) return fail(e);
    if ((e = s->d0.alloc((size_t)nq * cap * 4)) != hipSuccess) return fail(e);
    if ((e = s->d1.alloc((size_t)nq * cap * 4)) != hipSuccess) return fail(e);
    if ((e = s->which.alloc((size_t)nq * 4, 0)) != hipSuccess) return fail(e);
    if ((e = s->base.alloc((size_t)nq * 4, 0)) != hipSuccess) return fail(e);
    if ((e = s->size.alloc((size_t)nq * 4, 0)) != hipSuccess) return fail(e);
    if ((e = s->vt.alloc(((size_t)nq << vbits) * 4, 0xFF)) != hipSuccess) return fail(e);
    if ((e = s->vtc.alloc((size_t)nq * 4, 0)) != hipSuccess) return fail(e);
    if ((e = s->cids.alloc((size_t)nq * l_value * 4)) != hipSuccess) return fail(e);
    if ((e = s->cd.alloc((size_t)nq * l_value * 4)) != hipSuccess) return fail(e);
    if ((e = s->cn.alloc((size_t)nq * 4, 0)) != hipSuccess) return fail(e);
    if ((e = s->cnext.alloc((size_t)nq * 4, 0)) != hipSuccess) return fail(e);
    if ((e = s->stats.alloc((size_t)nq * sizeof(dann_search_stats), 0)) != hipSuccess) return fail(e);
    if ((e = hipMemcpy(s->q.p, queries, (size_t)nq * qb, hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    a.queries = s->q.p;
    a.ids[0] = (uint32_t*)s->ids0.p;
    a.ids[1] = (uint32_t*)s->ids1.p;
    a.d[0] = (float*)s->d0.p;
    a.d[1] = (float*)s->d1.p;
    a.which = (uint32_t*)s->which.p;
    a.base = (uint32_t*)s->base.p;
    a.size = (uint32_t*)s->size.p;
    a.vt = (uint32_t*)s->vt.p;
    a.vt_count = (uint32_t*)s->vtc.p;
    a.cache_ids = (uint32_t*)s->cids.p;
    a.cache_d = (float*)s->cd.p;
    a.cache_n = (uint32_t*)s->cn.p;
    a.cache_next = (uint32_t*)s->cnext.p;
    a.out_ids = nullptr;
    a.out_d = nullptr;
    a.out_n = nullptr;
    a.stats = (dann_search_stats*)s->stats.p;
    a.init = 1;
    int32_t rc = launch_paged(a, stream);
    if (rc == DANN_OK && (e = hipStreamSynchronize(stream)) != hipSuccess) rc = hip_fail(e, "paged begin");
    if (rc != DANN_OK) {
        delete s;
        return rc;
    }
    a.init = 0;
    *out = s;
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_paged_next(dann_paged* s, uint32_t k, uint32_t* out_ids, float* out_dists, uint32_t* out_counts) try {
    if (!s || !out_ids || !out_dists) return DANN_EINVAL;
    if (k == 0) {
        set_error("k should be greater than 0");  // paged.rs:62-64
        return DANN_EINVAL;
    }
    if (k > s->a.l_value) {
        set_error("k should be less than or equal to search_param_l");  // paged.rs:57-61
        return DANN_EINVAL;
    }
    dann_index* idx = s->idx;
    std::shared_lock<std::shared_mutex> rd(idx->rw);
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (prev != idx->device) DANN_HIP(hipSetDevice(idx->device));
    ::dann::CtxLease lease(idx);
    if (lease.status != DANN_OK) return lease.status;
    hipStream_t stream = lease.ctx->stream;
    const uint32_t nq = s->a.nq;
    if (s->out_k < k) {
        if (s->out_ids.p) (void)hipFree(s->out_ids.p);
        if (s->out_d.p) (void)hipFree(s->out_d.p);
        s->out_ids.p = s->out_d.p = nullptr;
        DANN_HIP(hipMalloc(&s->out_ids.p, (size_t)nq * k * 4));
        DANN_HIP(hipMalloc(&s->out_d.p, (size_t)nq * k * 4));
        if (!s->out_n.p) DANN_HIP(hipMalloc(&s->out_n.p, (size_t)nq * 4));
        s->out_k = k;
    }
    s->a.ix = idx->view();
    s->a.k = k;
    s->a.out_ids = (uint32_t*)s->out_ids.p;
    s->a.out_d = (float*)s->out_d.p;
    s->a.out_n = (uint32_t*)s->out_n.p;
    int32_t rc = launch_paged(s->a, stream);
    if (rc != DANN_OK) return rc;
    std::vector<dann_search_stats> st(nq);
    std::vector<uint32_t> counts(nq);
    DANN_HIP(hipMemcpyAsync(out_ids, s->out_ids.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, stream));
    DANN_HIP(hipMemcpyAsync(out_dists, s->out_d.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, stream));
    DANN_HIP(hipMemcpyAsync(counts.data(), s->out_n.p, (size_t)nq * 4, hipMemcpyDeviceToHost, stream));
    DANN_HIP(hipMemcpyAsync(st.data(), s->stats.p, (size_t)nq * sizeof(dann_search_stats), hipMemcpyDeviceToHost,
                            stream));
    DANN_HIP(hipStreamSynchronize(stream));
    if (out_counts) memcpy(out_counts, counts.data(), (size_t)nq * 4);
    for (uint32_t i = 0; i < nq; ++i)
        if (st[i].status) {
            set_error("paged query %u: candidate list (%u entries) or visited set exhausted; raise list_cap", i, s->a.cap);
            return DANN_EOVERFLOW;
        }
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_paged_end(dann_paged* s) try {
    if (!s) return DANN_EINVAL;
    delete s;
    return DANN_OK;
} DANN_CATCH_ALL

}  // extern "C"
