/*
 * oracle/dann_oracle.cpp -- CPU restatement of the reference hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see dann_oracle.h).  Nothing under diskann_amd/ may
 * link, import or call this file.  Every function cites the reference file:line it
 * follows (paths relative to the reference root, microsoft/DiskANN Rust workspace
 * v0.56).  The arithmetic follows the reference's x86-64-v3 ("V3") code path, which
 * is the workspace build default (.cargo/config.toml: target-cpu=x86-64-v3).
 *
 * Where the reference leaves behaviour unspecified (tie order of the unstable sort in
 * internal/sorted_neighbors.rs:36-40) the oracle fixes a rule and says so:
 * the candidate sort of RobustPrune follows Rust's own order (tie rule 6, rust_unstable_sort.h: under it every
 * reference grid_insert golden is reproduced exactly, which is what pins the build half of this file; the product's
 * default, DANN_TIE_RUST); tie rule 0 = ascending distance, ties by original position in the pool (the product's
 * DANN_TIE_POSITION).  The post-processing sorts of the filtered searches are restated as stable sorts.
 */
#include "dann_oracle.h"
#include "rust_unstable_sort.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#if defined(__AVX2__) && defined(__FMA__) && defined(__F16C__)
#include <immintrin.h>
#define ORC_HAVE_AVX2 1
#else
#define ORC_HAVE_AVX2 0
#endif

namespace {

/* ======================================================================
 * f16 <-> f32  (half 2.6 semantics == IEEE; diskann-wide/tests/float16_conversion.rs)
 * ====================================================================== */
inline float f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal: normalise */
            int e = -1;
            do {
                man <<= 1;
                ++e;
            } while ((man & 0x400u) == 0);
            man &= 0x3FFu;
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7F800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

/* round-to-nearest-even (float16_conversion.rs header comment) */
inline uint16_t f32_to_f16(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t exp = (x >> 23) & 0xFFu;
    uint32_t man = x & 0x7FFFFFu;
    if (exp == 0xFF) return (uint16_t)(sign | 0x7C00u | (man ? (0x200u | (man >> 13)) : 0));
    int e = (int)exp - 127 + 15;
    if (e >= 31) return (uint16_t)(sign | 0x7C00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        man |= 0x800000u;
        int shift = 14 - e;
        uint32_t half_man = man >> shift;
        uint32_t rem = man & ((1u << shift) - 1);
        uint32_t halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (half_man & 1))) ++half_man;
        return (uint16_t)(sign | half_man);
    }
    uint32_t half_man = man >> 13;
    uint32_t rem = man & 0x1FFFu;
    uint16_t out = (uint16_t)(sign | ((uint32_t)e << 10) | half_man);
    if (rem > 0x1000u || (rem == 0x1000u && (half_man & 1))) ++out; /* carries into exponent correctly */
    return out;
}

/* ======================================================================
 * SIMD emulation: f32x8 lanes, Strategy{4x1,4x2,2x4}, sum_tree
 * ====================================================================== */
struct F8 {
    float v[8];
};
inline F8 f8_zero() {
    F8 r;
    for (int i = 0; i < 8; ++i) r.v[i] = 0.0f;
    return r;
}
inline F8 f8_add(const F8& a, const F8& b) {
    F8 r;
    for (int i = 0; i < 8; ++i) r.v[i] = a.v[i] + b.v[i];
    return r;
}
/* f32x8::sum_tree, diskann-wide/src/arch/x86_64/v3/f32x8_.rs:185-212 */
inline float f8_sum_tree(const F8& x) {
    float q0 = x.v[0] + x.v[4], q1 = x.v[1] + x.v[5], q2 = x.v[2] + x.v[6], q3 = x.v[3] + x.v[7];
    float d0 = q0 + q2, d1 = q1 + q3;
    return d0 + d1;
}

/* element loaders: first `n` (<=8) elements, zero padded (load_simd_first) */
inline void load8(const float* p, int n, float* out) {
    for (int i = 0; i < 8; ++i) out[i] = i < n ? p[i] : 0.0f;
}
inline void load8(const uint16_t* p, int n, float* out) {
    for (int i = 0; i < 8; ++i) out[i] = i < n ? f16_to_f32(p[i]) : 0.0f;
}

/* Accumulator kinds.  `accumulate` follows the SIMDSchema impls in
 * diskann-vector/src/distance/simd.rs (L2 :817-845, IP :1588-1616,
 * CosineStateless :2430-2461 + FullCosineAccumulator :2281-2374). */
struct AccL2 {
    F8 s;
    void init() { s = f8_zero(); }
    void acc(const float* x, const float* y) {
        for (int i = 0; i < 8; ++i) {
            float c = x[i] - y[i];
            s.v[i] = std::fmaf(c, c, s.v[i]);
        }
    }
    void combine(const AccL2& o) { s = f8_add(s, o.s); }
    float reduce() const { return f8_sum_tree(s); }
};
struct AccIP {
    F8 s;
    void init() { s = f8_zero(); }
    void acc(const float* x, const float* y) {
        for (int i = 0; i < 8; ++i) s.v[i] = std::fmaf(x[i], y[i], s.v[i]);
    }
    void combine(const AccIP& o) { s = f8_add(s, o.s); }
    float reduce() const { return f8_sum_tree(s); }
};
inline float cosine_finish(float normx, float normy, float prod) {
    /* FullCosineAccumulator::sum, simd.rs:2329-2362 */
    float denominator = std::sqrt(normx) * std::sqrt(normy);
    if (normx < std::numeric_limits<float>::min() || normy < std::numeric_limits<float>::min()) return 0.0f;
    float v = prod / denominator;
    /* (-1.0f32).max(1.0f32.min(v)) : Rust min/max return the non-NaN operand */
    float m = std::isnan(v) ? 1.0f : (v < 1.0f ? v : 1.0f);
    return m > -1.0f ? m : -1.0f;
}
struct AccCos {
    F8 nx, ny, xy;
    void init() { nx = ny = xy = f8_zero(); }
    void acc(const float* x, const float* y) {
        for (int i = 0; i < 8; ++i) {
            nx.v[i] = std::fmaf(x[i], x[i], nx.v[i]);
            ny.v[i] = std::fmaf(y[i], y[i], ny.v[i]);
            xy.v[i] = std::fmaf(x[i], y[i], xy.v[i]);
        }
    }
    void combine(const AccCos& o) {
        nx = f8_add(nx, o.nx);
        ny = f8_add(ny, o.ny);
        xy = f8_add(xy, o.xy);
    }
    float reduce() const { return cosine_finish(f8_sum_tree(nx), f8_sum_tree(ny), f8_sum_tree(xy)); }
};

/* simd_op (simd.rs:686-747) with MainLoop Strategy4x1/4x2 (NACC=4) or 2x4 (NACC=2)
 * (simd.rs:321-483).  All three strategies assign the g-th 8-wide block to
 * accumulator g % NACC in increasing g; full epilogue blocks continue the same
 * pattern; the partial block is accumulated into the combined accumulator. */
template <int NACC, class Acc, class XT, class YT>
float simd_op_f(const XT* x, const YT* y, size_t len) {
    Acc acc[NACC];
    for (int a = 0; a < NACC; ++a) acc[a].init();
    size_t blocks = len / 8;
    float bx[8], by[8];
    for (size_t g = 0; g < blocks; ++g) {
        load8(x + 8 * g, 8, bx);
        load8(y + 8 * g, 8, by);
        acc[g % NACC].acc(bx, by);
    }
    if (NACC == 4) {
        acc[0].combine(acc[1]);
        acc[2].combine(acc[3]);
        acc[0].combine(acc[2]);
    } else if (NACC == 2) {
        acc[0].combine(acc[1]);
    }
    size_t rem = len % 8;
    if (rem) {
        load8(x + 8 * blocks, (int)rem, bx);
        load8(y + 8 * blocks, (int)rem, by);
        acc[0].acc(bx, by);
    }
    return acc[0].reduce();
}

/* integer kernels: exact i32 accumulation, converted once
 * (simd.rs:1192-1225 L2, :1947-1979/:2109-2143 IP, :2790-2827/:2996-3033 cosine) */
template <class T>
float int_l2(const T* x, const T* y, size_t len) {
    int32_t s = 0;
    for (size_t i = 0; i < len; ++i) {
        int32_t c = (int32_t)x[i] - (int32_t)y[i];
        s = (int32_t)((uint32_t)s + (uint32_t)(c * c));
    }
    return (float)s;
}
template <class T>
float int_ip(const T* x, const T* y, size_t len) {
    int32_t s = 0;
    for (size_t i = 0; i < len; ++i) s = (int32_t)((uint32_t)s + (uint32_t)((int32_t)x[i] * (int32_t)y[i]));
    return (float)s;
}
template <class T>
float int_cos(const T* x, const T* y, size_t len) {
    int32_t nx = 0, ny = 0, xy = 0;
    for (size_t i = 0; i < len; ++i) {
        int32_t a = x[i], b = y[i];
        nx = (int32_t)((uint32_t)nx + (uint32_t)(a * a));
        ny = (int32_t)((uint32_t)ny + (uint32_t)(b * b));
        xy = (int32_t)((uint32_t)xy + (uint32_t)(a * b));
    }
    return cosine_finish((float)nx, (float)ny, (float)xy);
}

/* PostOp (implementations.rs:215-401): SimilarityScore conventions */
inline float post_op(int metric, float raw) {
    switch (metric) {
        case ORC_L2: return raw;
        case ORC_INNER_PRODUCT: return -raw;
        default: return 1.0f - raw; /* Cosine / CosineNormalized */
    }
}

/* pair (T x T) kernels -- DistanceProvider::distance_comparer, V3
 * (distance_provider.rs:265-337): f32xf32 L2/IP Strategy4x1, cosine 2x4;
 * f16xf16 L2/IP/cosine all Strategy2x4 on V3 (simd.rs:989,1752,2591);
 * integers exact; integer CosineNormalized -> Cosine (:274-297). */
float pair_raw(int dtype, int metric, const void* x, const void* y, size_t dim) {
    switch (dtype) {
        case ORC_F32: {
            const float* a = (const float*)x;
            const float* b = (const float*)y;
            if (metric == ORC_L2) return simd_op_f<4, AccL2>(a, b, dim);
            if (metric == ORC_COSINE) return simd_op_f<2, AccCos>(a, b, dim);
            return simd_op_f<4, AccIP>(a, b, dim);
        }
        case ORC_F16: {
            const uint16_t* a = (const uint16_t*)x;
            const uint16_t* b = (const uint16_t*)y;
            if (metric == ORC_L2) return simd_op_f<2, AccL2>(a, b, dim);
            if (metric == ORC_COSINE) return simd_op_f<2, AccCos>(a, b, dim);
            return simd_op_f<2, AccIP>(a, b, dim);
        }
        case ORC_U8: {
            const uint8_t* a = (const uint8_t*)x;
            const uint8_t* b = (const uint8_t*)y;
            if (metric == ORC_L2) return int_l2(a, b, dim);
            if (metric == ORC_INNER_PRODUCT) return int_ip(a, b, dim);
            return int_cos(a, b, dim);
        }
        case ORC_I8: {
            const int8_t* a = (const int8_t*)x;
            const int8_t* b = (const int8_t*)y;
            if (metric == ORC_L2) return int_l2(a, b, dim);
            if (metric == ORC_INNER_PRODUCT) return int_ip(a, b, dim);
            return int_cos(a, b, dim);
        }
    }
    return std::numeric_limits<float>::quiet_NaN();
}

/* query kernels -- Full<T>::query_distance (diskann-inmem/src/layers/full.rs:351-504):
 * f32: same schemas as the pair kernels; f16: f32(query) x f16(row) with
 * Strategy4x2 (== 4 accumulators) for L2/IP and 2x4 for cosine (simd.rs:1121,1878,2716). */
float query_raw(int dtype, int metric, const float* q32, const void* q, const void* row, size_t dim) {
    if (dtype == ORC_F16) {
        const uint16_t* b = (const uint16_t*)row;
        if (metric == ORC_L2) return simd_op_f<4, AccL2>(q32, b, dim);
        if (metric == ORC_COSINE) return simd_op_f<2, AccCos>(q32, b, dim);
        return simd_op_f<4, AccIP>(q32, b, dim);
    }
    return pair_raw(dtype, metric, q, row, dim);
}

#if ORC_HAVE_AVX2
/* AVX2 twins of the f32 L2 / IP kernels (the reference's actual instruction mix:
 * vsubps + vfmadd231ps, 4 accumulators; RFC rfcs/01206-inmem2.md:164-215). */
inline float hsum_tree(__m256 x) {
    __m128 hi = _mm256_extractf128_ps(x, 1);
    __m128 lo = _mm256_castps256_ps128(x);
    __m128 q = _mm_add_ps(lo, hi);
    __m128 d = _mm_add_ps(q, _mm_movehl_ps(q, q));
    __m128 s = _mm_add_ss(d, _mm_shuffle_ps(d, d, 0x1));
    return _mm_cvtss_f32(s);
}
inline __m256 loadx(const float* p) { return _mm256_loadu_ps(p); }
inline __m256 loadx(const uint16_t* p) { return _mm256_cvtph_ps(_mm_loadu_si128((const __m128i*)p)); }
inline __m256 load_first(const float* p, int n) {
    alignas(32) float t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) t[i] = p[i];
    return _mm256_load_ps(t);
}
inline __m256 load_first(const uint16_t* p, int n) {
    alignas(16) uint16_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) t[i] = p[i];
    return _mm256_cvtph_ps(_mm_load_si128((const __m128i*)t));
}
template <bool IS_L2, class YT>
float avx2_op4(const float* x, const YT* y, size_t len) {
    __m256 s[4] = {_mm256_setzero_ps(), _mm256_setzero_ps(), _mm256_setzero_ps(), _mm256_setzero_ps()};
    size_t blocks = len / 8, g = 0;
    for (; g + 4 <= blocks; g += 4) {
        for (int a = 0; a < 4; ++a) {
            __m256 xv = _mm256_loadu_ps(x + 8 * (g + a));
            __m256 yv = loadx(y + 8 * (g + a));
            if (IS_L2) {
                __m256 c = _mm256_sub_ps(xv, yv);
                s[a] = _mm256_fmadd_ps(c, c, s[a]);
            } else {
                s[a] = _mm256_fmadd_ps(xv, yv, s[a]);
            }
        }
    }
    for (int a = 0; g < blocks; ++g, ++a) {
        __m256 xv = _mm256_loadu_ps(x + 8 * g);
        __m256 yv = loadx(y + 8 * g);
        if (IS_L2) {
            __m256 c = _mm256_sub_ps(xv, yv);
            s[a] = _mm256_fmadd_ps(c, c, s[a]);
        } else {
            s[a] = _mm256_fmadd_ps(xv, yv, s[a]);
        }
    }
    __m256 t = _mm256_add_ps(_mm256_add_ps(s[0], s[1]), _mm256_add_ps(s[2], s[3]));
    size_t rem = len % 8;
    if (rem) {
        __m256 xv = load_first(x + 8 * blocks, (int)rem);
        __m256 yv = load_first(y + 8 * blocks, (int)rem);
        if (IS_L2) {
            __m256 c = _mm256_sub_ps(xv, yv);
            t = _mm256_fmadd_ps(c, c, t);
        } else {
            t = _mm256_fmadd_ps(xv, yv, t);
        }
    }
    return hsum_tree(t);
}
#endif

float query_raw_fast(int dtype, int metric, const float* q32, const void* q, const void* row, size_t dim) {
#if ORC_HAVE_AVX2
    if (metric == ORC_L2 || metric == ORC_INNER_PRODUCT || metric == ORC_COSINE_NORMALIZED) {
        if (dtype == ORC_F32) {
            return metric == ORC_L2 ? avx2_op4<true>((const float*)q, (const float*)row, dim)
                                    : avx2_op4<false>((const float*)q, (const float*)row, dim);
        }
        if (dtype == ORC_F16) {
            return metric == ORC_L2 ? avx2_op4<true>(q32, (const uint16_t*)row, dim)
                                    : avx2_op4<false>(q32, (const uint16_t*)row, dim);
        }
    }
#endif
    return query_raw(dtype, metric, q32, q, row, dim);
}

inline int eff_metric(int dtype, int metric) {
    /* integer types treat CosineNormalized as Cosine (distance_provider.rs:274-297, full.rs:470,499) */
    if ((dtype == ORC_U8 || dtype == ORC_I8) && metric == ORC_COSINE_NORMALIZED) return ORC_COSINE;
    return metric;
}
inline size_t elem_size(int dtype) { return dtype == ORC_F32 ? 4 : dtype == ORC_F16 ? 2 : 1; }
inline size_t layer_bytes(int dtype, size_t dim) { return dim * elem_size(dtype) + (dtype == ORC_SQ8 ? 4 : 0); }
inline size_t query_bytes(const orc_index* ix) { return ix->dtype == ORC_PQ ? (size_t)ix->dim * 4 : layer_bytes(ix->dtype, ix->dim); }

/* CompensatedSquaredL2 / CompensatedIP / CompensatedCosineNormalized
 * (diskann-quantization/src/scalar/vectors.rs:216-245, 306-370, 403-465) on 8-bit codes with the
 * trailing f32 compensation. */
float sq8_similarity(int metric, const uint8_t* x, const uint8_t* y, uint32_t dim, float scale, float shift_norm_sq) {
    const float ibs = 1.0f / 255.0f;
    const float bit_scale = ibs * ibs;
    const float scale_sq = scale * scale;
    float cx, cy;
    std::memcpy(&cx, x + dim, 4);
    std::memcpy(&cy, y + dim, 4);
    if (metric == ORC_INNER_PRODUCT) {
        uint32_t p = 0;
        for (uint32_t i = 0; i < dim; ++i) p += (uint32_t)x[i] * (uint32_t)y[i];
        float r = std::fmaf(bit_scale * scale_sq, (float)p, shift_norm_sq) + (cy + cx);
        return -r;
    }
    uint32_t s = 0;
    for (uint32_t i = 0; i < dim; ++i) {
        int32_t c = (int32_t)x[i] - (int32_t)y[i];
        s += (uint32_t)(c * c);
    }
    float l2 = bit_scale * scale_sq * (float)s;
    if (metric == ORC_L2) return l2;
    float sim = 1.0f - l2 / 2.0f; /* CosineNormalized */
    return 1.0f - sim;
}

/* hashbrown::HashSet<u32> stand-in for the visited set (glue.rs:542-549): an exact set with open addressing
 * (power-of-two slots, linear probing, grown at 7/8 load) so that the timed CPU baseline is not handicapped by
 * std::unordered_set's node allocations.  Slot value = id + 1, 0 = free. */
struct IdSet {
    std::vector<uint64_t> slots;
    size_t count = 0, mask = 0;
    struct Result {
        bool second;
    };
    static uint64_t mix(uint32_t id) { return (uint64_t)id * 0x9E3779B97F4A7C15ull; }
    void rehash(size_t cap) {
        std::vector<uint64_t> old;
        old.swap(slots);
        slots.assign(cap, 0);
        mask = cap - 1;
        for (uint64_t v : old)
            if (v) {
                size_t h = (size_t)(mix((uint32_t)(v - 1)) >> 32) & mask;
                while (slots[h]) h = (h + 1) & mask;
                slots[h] = v;
            }
    }
    void reserve(size_t n) {
        size_t cap = 16;
        while (cap * 7 / 8 < n) cap <<= 1;
        if (cap > slots.size()) rehash(cap);
    }
    Result insert(uint32_t id) {
        if (slots.empty() || (count + 1) * 8 > slots.size() * 7) rehash(slots.empty() ? 16 : slots.size() * 2);
        const uint64_t v = (uint64_t)id + 1;
        size_t h = (size_t)(mix(id) >> 32) & mask;
        while (slots[h]) {
            if (slots[h] == v) return {false};
            h = (h + 1) & mask;
        }
        slots[h] = v;
        ++count;
        return {true};
    }
    void clear() {
        std::fill(slots.begin(), slots.end(), 0);
        count = 0;
    }
};

/* ======================================================================
 * NeighborPriorityQueue  (diskann/src/neighbor/queue.rs:68-475)
 * ====================================================================== */
struct Queue {
    size_t capacity, search_l, cursor = 0;
    bool auto_resizable = false; /* auto_resizable_with_search_param_l (queue.rs:95-105): never drops */
    std::vector<uint32_t> ids;
    std::vector<uint8_t> visited;
    std::vector<float> dist;
    explicit Queue(size_t l) : capacity(l), search_l(l) {
        ids.reserve(l + 1);
        visited.reserve(l + 1);
        dist.reserve(l + 1);
    }
    size_t size() const { return ids.size(); }
    /* get_lower_bound :229-280 -- first index with dist >= d */
    size_t lower_bound(float d) const {
        return (size_t)(std::lower_bound(dist.begin(), dist.end(), d) - dist.begin());
    }
    /* insert :130-171 */
    void insert(uint32_t id, float d) {
        if (std::isnan(d)) return;
        size_t n = size();
        if (auto_resizable) {
            if (n == capacity) capacity += std::max<size_t>(1, capacity >> 1); /* reserve: 1.5x (:117-121) */
        } else if (n == capacity && n > 0 && dist[n - 1] < d) {
            return;
        }
        if (capacity == 0) return;
        size_t pos = n > 0 ? lower_bound(d) : 0;
        if (n == capacity) {
            ids.pop_back();
            visited.pop_back();
            dist.pop_back();
        }
        ids.insert(ids.begin() + pos, id);
        visited.insert(visited.begin() + pos, 0);
        dist.insert(dist.begin() + pos, d);
        if (pos < cursor) cursor = pos;
    }
    /* drain_best :174-180 */
    void drain_best(size_t count) {
        count = std::min(count, size());
        ids.erase(ids.begin(), ids.begin() + count);
        visited.erase(visited.begin(), visited.begin() + count);
        dist.erase(dist.begin(), dist.begin() + count);
        cursor = 0;
    }
    /* has_notvisited_node :316-318 */
    bool has_notvisited() const { return cursor < std::min(search_l, size()); }
    /* closest_notvisited :297-313 */
    bool pop(uint32_t* id, float* d) {
        if (!has_notvisited()) return false;
        size_t cur = cursor;
        visited[cur] = 1;
        ++cursor;
        while (cursor < size() && visited[cursor]) ++cursor;
        *id = ids[cur];
        *d = dist[cur];
        return true;
    }
};

/* ======================================================================
 * index views
 * ====================================================================== */
struct View {
    const orc_index* ix;
    int metric;
    size_t esz;
    explicit View(const orc_index* i) : ix(i), metric(eff_metric(i->dtype, i->metric)), esz(elem_size(i->dtype)) {}
    uint32_t nslots() const { return ix->capacity + ix->nstart; }
    const uint8_t* row(uint32_t id) const { return ix->rows + (size_t)id * ix->row_stride; }
    uint32_t* adj_row(uint32_t id) const { return ix->adj + (size_t)id * (ix->max_degree + 1); }
    /* Reader::read_in_bounds (store.rs:693-723): the slot's inline tag byte sits right after the payload and the
     * row is readable iff Tag::can_read, i.e. tag >= PUBLISHED = 254 (tag.rs:86-133).  tag_offset == 0: a store
     * without tags (every slot readable). */
    bool readable(uint32_t id) const { return ix->tag_offset == 0 || row(id)[ix->tag_offset] >= 254; }
    /* Neighbors::get, neighbors.rs:124-163 (length clamped to max_degree) */
    uint32_t get_neighbors(uint32_t id, const uint32_t** out) const {
        uint32_t* r = adj_row(id);
        *out = r + 1;
        return std::min(r[0], ix->max_degree);
    }
    /* Neighbors::set, neighbors.rs:207-224 */
    int set_neighbors(uint32_t id, const uint32_t* n, uint32_t len) const {
        if (id >= nslots()) return -3;
        if (len > ix->max_degree) return -4;
        uint32_t* r = adj_row(id);
        std::memcpy(r + 1, n, (size_t)len * 4);
        r[0] = len;
        return 0;
    }
    /* PruneAccessor::append_vector with clamp, provider.rs:795-822 */
    void append_neighbors(uint32_t id, const uint32_t* n, uint32_t len) const {
        uint32_t* r = adj_row(id);
        uint32_t cur = std::min(r[0], ix->max_degree);
        uint32_t slack = ix->max_degree - cur;
        uint32_t take = std::min(len, slack);
        std::memcpy(r + 1 + cur, n, (size_t)take * 4);
        r[0] = cur + take;
    }
    float pair(uint32_t a, uint32_t b) const {
        if (ix->dtype == ORC_SQ8) return sq8_similarity(metric, row(a), row(b), ix->dim, ix->sq_scale, ix->sq_shift_norm_sq);
        return post_op(metric, pair_raw(ix->dtype, metric, row(a), row(b), ix->dim));
    }
};

struct QueryCtx {
    const View& v;
    const void* q;
    std::vector<float> q32; /* f16 query widened once, full.rs:421-423 */
    std::vector<float> lut; /* PQ: populate_chunk_distances_impl, once per query */
    bool fast;
    QueryCtx(const View& view, const void* query, bool fast_) : v(view), q(query), fast(fast_) {
        if (v.ix->dtype == ORC_F16) {
            q32.resize(v.ix->dim);
            const uint16_t* h = (const uint16_t*)query;
            for (uint32_t i = 0; i < v.ix->dim; ++i) q32[i] = f16_to_f32(h[i]);
        }
        if (v.ix->dtype == ORC_PQ) {
            lut.resize((size_t)v.ix->pq_chunks * 256);
            orc_pq_build_lut(v.metric, v.ix->pq_pivots, nullptr, v.ix->pq_offsets, v.ix->pq_chunks, v.ix->dim,
                             (const float*)query, lut.data());
        }
    }
    float eval(uint32_t id) const {
        if (v.ix->dtype == ORC_PQ) return orc_pq_lookup(lut.data(), v.row(id), v.ix->pq_chunks);
        if (v.ix->dtype == ORC_SQ8)
            return sq8_similarity(v.metric, (const uint8_t*)q, v.row(id), v.ix->dim, v.ix->sq_scale, v.ix->sq_shift_norm_sq);
        float raw = fast ? query_raw_fast(v.ix->dtype, v.metric, q32.data(), q, v.row(id), v.ix->dim)
                         : query_raw(v.ix->dtype, v.metric, q32.data(), q, v.row(id), v.ix->dim);
        return post_op(v.metric, raw);
    }
};

struct SearchOut {
    uint32_t cmps = 0, hops = 0;
    std::vector<std::pair<uint32_t, float>>* record = nullptr;
};

inline void prefetch_row(const uint8_t* p, size_t bytes) {
    const size_t lines = (bytes + 63) / 64;
    if (!lines) return;
    _mm_prefetch((const char*)p + 64 * (lines - 1), _MM_HINT_T0);
    for (size_t i = 0; i + 1 < lines; ++i) _mm_prefetch((const char*)p + 64 * i, _MM_HINT_T0);
}

/* DiskANNIndex::search_internal (index.rs:1933-2000) through the inmem2
 * SearchAccessor (provider.rs:408-480).  The visited set is hashbrown::HashSet<u32>
 * (scratch.rs:48, glue.rs:542-549) == any exact set. */
void search_internal(const QueryCtx& qc, Queue& best, IdSet& visited, uint32_t beam_width,
                     SearchOut& out) {
    const View& v = qc.v;
    const orc_index* ix = v.ix;
    /* start_point_distances: frozen slots [capacity, capacity+nstart) */
    for (uint32_t p = ix->capacity; p < ix->capacity + ix->nstart; ++p) {
        visited.insert(p);
        best.insert(p, qc.eval(p));
        out.cmps += 1;
    }
    std::vector<uint32_t> beam, ids;
    std::vector<std::pair<uint32_t, float>> neighbors;
    const size_t row_bytes = v.ix->dtype == ORC_PQ ? v.ix->pq_chunks : layer_bytes(v.ix->dtype, v.ix->dim);
    if (beam_width == 0) beam_width = 1;
    while (best.has_notvisited()) {
        beam.clear();
        uint32_t id;
        float d;
        while (beam.size() < beam_width && best.pop(&id, &d)) {
            if (out.record) out.record->emplace_back(id, d);
            beam.push_back(id);
        }
        neighbors.clear();
        for (uint32_t b : beam) {
            const uint32_t* adj;
            uint32_t n = v.get_neighbors(b, &adj);
            /* retain(pred.eval_mut(i) && in_bounds(i)), provider.rs:453-454, then expand_beam_inner with its
             * software prefetch `lookahead` (8) rows ahead, last cache line first (provider.rs:581-602, 620-690) */
            ids.clear();
            for (uint32_t j = 0; j < n; ++j) {
                uint32_t nb = adj[j];
                if (visited.insert(nb).second && nb < v.nslots() && v.readable(nb)) ids.push_back(nb);
            }
            const size_t len = ids.size(), look = std::min<size_t>(8, len);
            for (size_t j = 0; j < look; ++j) prefetch_row(v.row(ids[j]), row_bytes);
            size_t ahead = look == 0 ? len : look;
            for (size_t j = 0; j < len; ++j) {
                if (ahead != len) prefetch_row(v.row(ids[ahead++]), row_bytes);
                neighbors.emplace_back(ids[j], qc.eval(ids[j]));
            }
        }
        for (auto& nb : neighbors) best.insert(nb.first, nb.second);
        out.cmps += (uint32_t)neighbors.size();
        out.hops += (uint32_t)beam.size();
    }
}

int32_t search_one(const orc_index* ix, const void* query, uint32_t l_value, uint32_t beam_width, uint32_t k,
                   uint32_t* out_ids, float* out_dists, uint32_t* stats,
                   std::vector<std::pair<uint32_t, float>>* record, bool fast) {
    if (!ix || !query || l_value == 0 || beam_width == 0) return -1;
    View v(ix);
    QueryCtx qc(v, query, fast);
    /* queue capacity == search_l == L + num_start_points (scratch.rs:199-207) */
    Queue best((size_t)l_value + ix->nstart);
    IdSet visited;
    visited.reserve((size_t)(1.1 * ix->max_degree * 1.3 * l_value) + 1);
    SearchOut so;
    so.record = record;
    search_internal(qc, best, visited, beam_width, so);
    for (uint32_t i = 0; i < k; ++i) {
        out_ids[i] = 0xFFFFFFFFu;
        out_dists[i] = std::numeric_limits<float>::infinity();
    }
    /* Translate::post_process (provider.rs:907-949): drop ids without an external
     * mapping (== frozen start points), stop when the buffer is full.  `count`
     * reproduces the reference's accounting (push() returns Full on the last slot). */
    uint32_t written = 0, ref_count = 0;
    size_t n = std::min(best.search_l, best.size());
    for (size_t i = 0; i < n && k > 0; ++i) {
        uint32_t id = best.ids[i];
        if (id >= ix->capacity) continue;
        out_ids[written] = id;
        out_dists[written] = best.dist[i];
        ++written;
        if (written == k) break;
        ++ref_count;
    }
    if (stats) {
        stats[0] = so.cmps;
        stats[1] = so.hops;
        stats[2] = ref_count;
    }
    return (int32_t)written;
}

/* ======================================================================
 * RobustPrune
 * ====================================================================== */
struct PNeighbor {
    uint32_t id;
    float d;
    uint32_t pos;
};

/* SortedNeighbors::new (internal/sorted_neighbors.rs:26-44).  The reference sorts with select_nth_unstable_by +
 * sort_unstable_by: the order of candidates at EQUAL distance is whatever Rust's unstable sort leaves (its source is
 * not in this image).  The oracle's rule is "ties by pool position" (rule 0); the other rules exist to measure how far
 * a tie order can move the reference's grid_insert counters (tests/test_oracle_build.py, tie envelope) -- the
 * default is rule 6 (the product's DANN_TIE_RUST); rule 0 is the product's DANN_TIE_POSITION:
 *   0 pool position ascending (stable)   1 pool position descending   2 id ascending   3 id descending
 *   4 a seeded shuffle of the tied entries (a fresh permutation per sort)
 *   5 a hypothesis about small pools: when the whole pool is kept (max >= len) select_nth_unstable_by(len - 1) swaps
 *     the FIRST maximum with the last element, and a prefix of <= 20 entries is sorted by insertion (stable on the
 *     order after that swap); longer prefixes fall back to rule 0.  Public descriptions of Rust >= 1.81's sort, not
 *     checked against its source.
 *   6 Rust's own order: select_nth_unstable_by + sort_unstable_by as restated in rust_unstable_sort.h (ipnsort and
 *     its selection, for 8-byte Copy elements).  With it the oracle reproduces every counter of the reference's
 *     twelve tie-heavy grid_insert goldens exactly (tests/test_oracle_build.py). */
static int g_tie_rule = 6; /* the reference's own order; 0 = the product's DANN_TIE_POSITION */
static uint64_t g_rust_fallbacks = 0;
static uint64_t g_tie_state = 0x9E3779B97F4A7C15ull;
void tie_rule_set(int32_t rule, uint64_t seed) {
    g_tie_rule = rule;
    g_tie_state = seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
}
static inline uint64_t tie_next() { /* splitmix64 */
    uint64_t z = (g_tie_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
void sort_pool(std::vector<PNeighbor>& pool, size_t max) {
    for (size_t i = 0; i < pool.size(); ++i) pool[i].pos = (uint32_t)i;
    /* fast_distance: partial_cmp, NaN compares Equal (neighbor/mod.rs:150-154) */
    auto by_d = [](const PNeighbor& a, const PNeighbor& b) { return a.d < b.d; };
    switch (g_tie_rule) {
        case 1: std::reverse(pool.begin(), pool.end()); break;
        case 2: std::stable_sort(pool.begin(), pool.end(), [](const PNeighbor& a, const PNeighbor& b) { return a.id < b.id; }); break;
        case 3: std::stable_sort(pool.begin(), pool.end(), [](const PNeighbor& a, const PNeighbor& b) { return a.id > b.id; }); break;
        case 4:
            for (size_t i = pool.size(); i > 1; --i) std::swap(pool[i - 1], pool[tie_next() % i]);
            break;
        case 5:
            if (max >= pool.size() && pool.size() >= 2 && pool.size() <= 21) {
                size_t mx = 0;
                for (size_t i = 1; i < pool.size(); ++i)
                    if (pool[mx].d < pool[i].d) mx = i; /* acc kept unless acc < t: the first maximum */
                std::swap(pool[mx], pool[pool.size() - 1]);
                std::stable_sort(pool.begin(), pool.end() - 1, by_d);
                return;
            }
            break;
        case 6:
            g_rust_fallbacks += rust_sort::sorted_neighbors(pool, max, by_d) ? 1 : 0;
            return;
        default: break;
    }
    std::stable_sort(pool.begin(), pool.end(), by_d);
    if (pool.size() > max) pool.resize(max);
}

/* PruneKind::update_occlude_factor (config/mod.rs:80-103) */
inline float update_occlude(bool occluding, float d_ik, float d_jk, float cur, float alpha) {
    if (!occluding) {
        if (d_jk == 0.0f) return std::numeric_limits<float>::max();
        float r = d_ik / d_jk;
        /* f32::max: NaN-ignoring */
        if (std::isnan(r)) return cur;
        if (std::isnan(cur)) return r;
        return cur > r ? cur : r;
    }
    if (d_jk < alpha * d_ik) return alpha + 0.01f;
    return cur;
}

/* occlude_list (index.rs:2565-2650) + prune::robust_prune (internal/prune.rs:106-259).
 * `pool` is sorted; `location` is masked out (index.rs:2607-2613). */
uint32_t occlude_list(const View& v, const orc_build_config* cfg, uint32_t location,
                      const std::vector<PNeighbor>& pool, bool force_saturate, std::vector<uint32_t>& out,
                      uint64_t* pair_evals) {
    out.clear();
    if (pool.empty()) return 0;
    const float alpha = cfg->alpha;
    const size_t degree = cfg->pruned_degree;
    const bool occluding = (v.metric == ORC_INNER_PRODUCT);
    struct State {
        float occ = 0.0f;
        uint16_t last_checked = 0, neighbor = 0;
    };
    std::vector<State> st(pool.size());
    auto present = [&](size_t i) { return pool[i].id != location && pool[i].id < v.nslots(); };
    float current_alpha = 1.0f;
    const float inc = alpha < 1.2f ? alpha : 1.2f;
    size_t found = 0;
    while (found < degree) {
        for (size_t i = 0; i < pool.size(); ++i) {
            if (found >= degree) break;
            float occ = st[i].occ;
            uint16_t last = st[i].last_checked;
            if (occ > current_alpha) continue;
            if (!present(i)) {
                st[i].occ = std::numeric_limits<float>::max();
                continue;
            }
            while ((size_t)last != found) {
                size_t rp = st[last].neighbor;
                ++last;
                if (rp >= i) {
                    st[i].last_checked = last;
                    continue;
                }
                float d;
                if (present(rp)) {
                    d = v.pair(pool[i].id, pool[rp].id);
                    if (pair_evals) ++*pair_evals;
                } else {
                    d = std::numeric_limits<float>::max();
                }
                occ = update_occlude(occluding, pool[i].d, d, occ, current_alpha);
                if (occ > current_alpha) break;
            }
            st[i].last_checked = last;
            if (occ > current_alpha) {
                st[i].occ = occ;
                continue;
            }
            st[i].occ = std::numeric_limits<float>::max();
            st[found].neighbor = (uint16_t)i;
            ++found;
        }
        if (current_alpha == alpha) break;
        float next = current_alpha * inc;
        current_alpha = next < alpha ? next : alpha;
    }
    for (size_t n = 0; n < found; ++n) out.push_back(pool[st[n].neighbor].id);
    if (force_saturate || (cfg->saturate_after_prune && alpha > 1.0f)) {
        for (const auto& p : pool) {
            if (out.size() >= degree) break;
            if (p.id != location && std::find(out.begin(), out.end(), p.id) == out.end()) out.push_back(p.id);
        }
    }
    return (uint32_t)out.size();
}

/* robust_prune_list (index.rs:2397-2454): distances location -> each list member,
 * then sort + occlude. */
void robust_prune_list(const View& v, const orc_build_config* cfg, uint32_t location,
                       const std::vector<uint32_t>& list, bool force_saturate, std::vector<uint32_t>& out,
                       uint64_t* counters) {
    out.clear();
    if (list.empty()) return;
    std::vector<PNeighbor> pool;
    pool.reserve(list.size());
    for (uint32_t id : list) {
        if (id == location || id >= v.nslots()) continue;
        pool.push_back({id, v.pair(location, id), 0});
        if (counters) ++counters[1];
    }
    sort_pool(pool, cfg->max_occlusion_size);
    occlude_list(v, cfg, location, pool, force_saturate, out, counters ? &counters[1] : nullptr);
}

/* add_edge_and_prune (index.rs:2264-2341) */
void add_edge_and_prune(const View& v, const orc_build_config* cfg, const uint32_t* targets, uint32_t nt,
                        uint32_t source, uint64_t* counters) {
    const uint32_t* adj;
    uint32_t n = v.get_neighbors(source, &adj);
    if (counters) ++counters[4];
    std::vector<uint32_t> list(adj, adj + n);
    uint32_t added = 0;
    for (uint32_t t = 0; t < nt; ++t) { /* AdjacencyList::extend_from_slice keeps ids unique */
        if (std::find(list.begin(), list.end(), targets[t]) == list.end()) {
            list.push_back(targets[t]);
            ++added;
        }
    }
    if (added == 0) return;
    if (list.size() <= cfg->max_degree) {
        v.append_neighbors(source, list.data() + (list.size() - added), added);
        if (counters) ++counters[3];
    } else {
        std::vector<uint32_t> pruned;
        robust_prune_list(v, cfg, source, list, false, pruned, counters);
        v.set_neighbors(source, pruned.data(), (uint32_t)pruned.size());
        if (counters) ++counters[2];
    }
}

void record_to_pool(const std::vector<std::pair<uint32_t, float>>& rec, std::vector<PNeighbor>& pool) {
    pool.clear();
    pool.reserve(rec.size());
    for (auto& r : rec) pool.push_back({r.first, r.second, 0});
}

}  // namespace

/* ======================================================================
 * C interface
 * ====================================================================== */
extern "C" {

float orc_f16_to_f32(uint16_t h) { return f16_to_f32(h); }
uint16_t orc_f32_to_f16(float f) { return f32_to_f16(f); }

float orc_distance(int32_t dtype, int32_t metric, const void* x, const void* y, size_t dim) {
    int m = eff_metric(dtype, metric);
    return post_op(m, pair_raw(dtype, m, x, y, dim));
}

float orc_query_distance(int32_t dtype, int32_t metric, const void* query, const void* row, size_t dim) {
    int m = eff_metric(dtype, metric);
    std::vector<float> q32;
    if (dtype == ORC_F16) {
        q32.resize(dim);
        for (size_t i = 0; i < dim; ++i) q32[i] = f16_to_f32(((const uint16_t*)query)[i]);
    }
    return post_op(m, query_raw(dtype, m, q32.data(), query, row, dim));
}

float orc_query_distance_fast(int32_t dtype, int32_t metric, const void* query, const void* row, size_t dim) {
    int m = eff_metric(dtype, metric);
    std::vector<float> q32;
    if (dtype == ORC_F16) {
        q32.resize(dim);
        for (size_t i = 0; i < dim; ++i) q32[i] = f16_to_f32(((const uint16_t*)query)[i]);
    }
    return post_op(m, query_raw_fast(dtype, m, q32.data(), query, row, dim));
}

/* diskann-vector/src/distance/reference.rs: plain scalar loops, f32 accumulation
 * (floats), exact integers. */
float orc_distance_scalar_ref(int32_t dtype, int32_t metric, const void* x, const void* y, size_t dim) {
    int m = eff_metric(dtype, metric);
    auto get = [&](const void* p, size_t i) -> float {
        switch (dtype) {
            case ORC_F32: return ((const float*)p)[i];
            case ORC_F16: return f16_to_f32(((const uint16_t*)p)[i]);
            case ORC_U8: return (float)((const uint8_t*)p)[i];
            default: return (float)((const int8_t*)p)[i];
        }
    };
    if (dtype == ORC_U8 || dtype == ORC_I8) return post_op(m, pair_raw(dtype, m, x, y, dim));
    double l2 = 0, ip = 0, nx = 0, ny = 0;
    for (size_t i = 0; i < dim; ++i) {
        double a = get(x, i), b = get(y, i);
        l2 += (a - b) * (a - b);
        ip += a * b;
        nx += a * a;
        ny += b * b;
    }
    if (m == ORC_L2) return (float)l2;
    if (m == ORC_INNER_PRODUCT) return (float)-ip;
    if (m == ORC_COSINE_NORMALIZED) return (float)(1.0 - ip);
    if (nx < (double)std::numeric_limits<float>::min() || ny < (double)std::numeric_limits<float>::min())
        return 1.0f;
    double c = ip / (std::sqrt(nx) * std::sqrt(ny));
    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
    return (float)(1.0 - c);
}

int32_t orc_search(const orc_index* ix, const void* query, uint32_t l_value, uint32_t beam_width, uint32_t k,
                   uint32_t* out_ids, float* out_dists, uint32_t* stats, uint32_t* rec_ids, float* rec_dists,
                   uint32_t rec_cap, uint32_t* rec_n) {
    std::vector<std::pair<uint32_t, float>> rec;
    int32_t r = search_one(ix, query, l_value, beam_width, k, out_ids, out_dists, stats,
                           (rec_ids && rec_n) ? &rec : nullptr, false);
    if (rec_ids && rec_n) {
        *rec_n = (uint32_t)rec.size();
        for (size_t i = 0; i < rec.size() && i < rec_cap; ++i) {
            rec_ids[i] = rec[i].first;
            if (rec_dists) rec_dists[i] = rec[i].second;
        }
    }
    return r;
}

int32_t orc_search_batch(const orc_index* ix, const void* queries, uint32_t nq, uint32_t l_value,
                         uint32_t beam_width, uint32_t k, uint32_t* out_ids, float* out_dists,
                         uint32_t* out_counts, uint32_t* stats, uint32_t threads, int32_t fast,
                         uint64_t* per_query_ns) {
    if (!ix || !queries) return -1;
    if (threads == 0) threads = 1;
    size_t qbytes = query_bytes(ix);
    std::vector<int32_t> status(threads, 0);
    auto work = [&](uint32_t t) {
        /* PartitionIter: contiguous ranges (search/api.rs:410-419) */
        /* partition_impl, diskann/src/utils/async_tools.rs:351-365 */
        uint64_t kk = nq / threads, mm = nq - kk * threads, lo, hi;
        if (t >= mm) {
            lo = mm * (kk + 1) + (t - mm) * kk;
            hi = lo + kk;
        } else {
            lo = (uint64_t)t * (kk + 1);
            hi = lo + kk + 1;
        }
        for (uint64_t q = lo; q < hi; ++q) {
            auto t0 = std::chrono::steady_clock::now();
            int32_t r = search_one(ix, (const uint8_t*)queries + q * qbytes, l_value, beam_width, k,
                                   out_ids + q * k, out_dists + q * k, stats ? stats + q * 3 : nullptr, nullptr,
                                   fast != 0);
            auto t1 = std::chrono::steady_clock::now();
            if (per_query_ns)
                per_query_ns[q] = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
            if (r < 0) {
                status[t] = r;
                return;
            }
            if (out_counts) out_counts[q] = (uint32_t)r;
        }
    };
    if (threads == 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (uint32_t t = 0; t < threads; ++t) pool.emplace_back(work, t);
        for (auto& th : pool) th.join();
    }
    for (int32_t s : status)
        if (s < 0) return s;
    return 0;
}

int32_t orc_range_search(const orc_index* ix, const void* query, uint32_t starting_l, uint32_t beam_width,
                         float radius, int32_t has_inner, float inner_radius, float initial_slack,
                         float range_slack, uint64_t max_returned, uint32_t* out_ids, float* out_dists,
                         uint64_t out_cap, uint32_t* stats) {
    if (!ix || !query || starting_l == 0 || beam_width == 0) return -1;
    View v(ix);
    QueryCtx qc(v, query, false);
    Queue best((size_t)starting_l + ix->nstart);
    IdSet visited;
    SearchOut so;
    search_internal(qc, best, visited, beam_width, so);
    const uint32_t init_cmps = so.cmps, init_hops = so.hops;
    const uint64_t max_ret = max_returned ? max_returned : ~0ull;
    std::vector<std::pair<uint32_t, float>> in_range;
    size_t n = std::min(std::min(best.search_l, best.size()), (size_t)starting_l);
    for (size_t i = 0; i < n; ++i)
        if (best.dist[i] <= radius) in_range.emplace_back(best.ids[i], best.dist[i]);
    visited.clear();
    for (auto& nb : in_range) visited.insert(nb.first);
    bool second = false;
    uint32_t hops = init_hops, cum_hops = so.hops;
    if (in_range.size() >= (size_t)((float)starting_l * initial_slack) && in_range.size() < max_ret) {
        second = true;
        size_t front = 0; /* range_frontier == in_range in arrival order */
        std::vector<uint32_t> beam;
        std::vector<std::pair<uint32_t, float>> neighbors;
        while (front < in_range.size() && in_range.size() < max_ret) {
            beam.clear();
            while (front < in_range.size() && beam.size() < beam_width) beam.push_back(in_range[front++].first);
            neighbors.clear();
            for (uint32_t b : beam) {
                const uint32_t* adj;
                uint32_t len = v.get_neighbors(b, &adj);
                for (uint32_t j = 0; j < len; ++j) {
                    uint32_t nb = adj[j];
                    if (visited.insert(nb).second && nb < v.nslots() && v.readable(nb)) neighbors.emplace_back(nb, qc.eval(nb));
                }
            }
            for (auto& nb : neighbors)
                if (nb.second <= radius * range_slack && in_range.size() < max_ret) in_range.push_back(nb);
            cum_hops += (uint32_t)beam.size();
        }
        hops = init_hops + cum_hops;
    }
    uint64_t written = 0;
    for (auto& nb : in_range) {
        if (nb.first >= ix->capacity) continue; /* start points have no external id */
        if (has_inner && nb.second <= inner_radius) continue;
        if (!(nb.second <= radius)) continue;
        if (written >= out_cap) break;
        out_ids[written] = nb.first;
        out_dists[written] = nb.second;
        ++written;
    }
    if (stats) {
        stats[0] = init_cmps;
        stats[1] = hops;
        stats[2] = (uint32_t)written;
        stats[3] = second ? 1u : 0u;
    }
    return (int32_t)written;
}

/* ======================================================================
 * filtered searches (diskann/src/graph/search/{inline_filter_search,multihop_filter_search,
 * filtered_range_search}.rs through graph/ext/labeled.rs).  The QueryLabelProvider is a
 * bitmap over slot ids: is_match(i) = bit i of filter_bits (start points included).
 * ====================================================================== */
namespace {
inline bool is_match(const uint32_t* bits, uint32_t nslots, uint32_t id) {
    return id < nslots && ((bits[id >> 5] >> (id & 31)) & 1u);
}
/* compute_adaptive_l, inline_filter_search.rs:283-301 (f64, truncating casts) */
size_t compute_adaptive_l(size_t base_l, size_t visited, size_t matched, double max_multiplier) {
    if (matched == 0 || visited == 0) return (size_t)((double)base_l * max_multiplier);
    double specificity = (double)matched / (double)visited;
    double multiplier;
    if (specificity >= 0.5) multiplier = 1.0;
    else if (specificity >= 0.1) multiplier = 2.0;
    else multiplier = std::pow(2.0, -std::log10(specificity));
    multiplier = std::min(std::max(multiplier, 1.0), max_multiplier);
    return (size_t)((double)base_l * multiplier);
}
struct Matched {
    uint32_t id;
    float d;
};
/* `v.sort_unstable_by(neighbor::ord::fast_distance)` of the filtered searches' post-processing
 * (inline_filter_search.rs:274, multihop_filter_search.rs:207): under the default tie rule (6) Rust's own unstable sort
 * as restated in rust_unstable_sort.h -- equal distances end up where ipnsort leaves them --, under the other rules a
 * stable sort (equal distances keep push order: the product's DANN_TIE_POSITION). */
static void sort_matched_by_distance(std::vector<Matched>& v) {
    auto by_d = [](const Matched& a, const Matched& b) { return a.d < b.d; };
    if (g_tie_rule == 6) {
        rust_sort::Impl<Matched, decltype(by_d)> srt(by_d);
        srt.sort_unstable(v.data(), v.size());
    } else {
        std::stable_sort(v.begin(), v.end(), by_d);
    }
}
/* inline_filter_search_internal, inline_filter_search.rs:166-281.  matched is returned sorted by distance
 * (sort_matched_by_distance). */
void inline_internal(const QueryCtx& qc, Queue& best, IdSet& visited, uint32_t beam_width,
                     size_t l_search, const uint32_t* filter, uint32_t adaptive_samples, double adaptive_scale,
                     SearchOut& out, std::vector<Matched>& matched) {
    const View& v = qc.v;
    const orc_index* ix = v.ix;
    for (uint32_t p = ix->capacity; p < ix->capacity + ix->nstart; ++p) {
        float d = qc.eval(p);
        visited.insert(p);
        best.insert(p, d);
        if (is_match(filter, v.nslots(), p)) matched.push_back({p, d});
    }
    std::vector<uint32_t> beam;
    std::vector<std::pair<uint32_t, float>> one_hop;
    size_t sample_visited = 0, sample_matched = 0;
    bool l_adjusted = false;
    for (;;) {
        beam.clear();
        one_hop.clear();
        uint32_t id;
        float d;
        while (beam.size() < beam_width && best.pop(&id, &d)) beam.push_back(id);
        if (beam.empty()) break;
        for (uint32_t b : beam) {
            const uint32_t* adj;
            uint32_t n = v.get_neighbors(b, &adj);
            for (uint32_t j = 0; j < n; ++j) {
                uint32_t nb = adj[j];
                if (visited.insert(nb).second && nb < v.nslots() && v.readable(nb)) one_hop.emplace_back(nb, qc.eval(nb));
            }
        }
        for (auto& nb : one_hop) {
            if (is_match(filter, v.nslots(), nb.first)) {
                matched.push_back({nb.first, nb.second});
                ++sample_matched;
            }
            best.insert(nb.first, nb.second);
            ++sample_visited;
        }
        out.cmps += (uint32_t)one_hop.size();
        out.hops += (uint32_t)beam.size();
        if (adaptive_samples && !l_adjusted && sample_visited >= adaptive_samples) {
            l_adjusted = true;
            size_t new_l = compute_adaptive_l(l_search, sample_visited, sample_matched, adaptive_scale);
            if (new_l > l_search) { /* SearchScratch::resize -> NeighborPriorityQueue::reconfigure, queue.rs:339-353 */
                best.search_l = new_l;
                best.capacity = new_l;
                if (new_l < best.size()) {
                    best.ids.resize(new_l);
                    best.visited.resize(new_l);
                    best.dist.resize(new_l);
                    best.cursor = std::min(best.cursor, new_l);
                }
            }
        }
    }
    sort_matched_by_distance(matched);
}
}  // namespace

extern "C" {
int32_t orc_adaptive_l(uint32_t base_l, uint32_t visited, uint32_t matched, double max_multiplier) {
    return (int32_t)compute_adaptive_l(base_l, visited, matched, max_multiplier);
}

int32_t orc_inline_filter_search(const orc_index* ix, const void* query, uint32_t l_value, uint32_t beam_width,
                                 uint32_t k, const uint32_t* filter_bits, uint32_t adaptive_samples,
                                 double adaptive_scale, uint32_t* out_ids, float* out_dists, uint32_t* stats) {
    if (!ix || !query || !filter_bits || l_value == 0 || beam_width == 0) return -1;
    if (adaptive_samples && !(adaptive_scale >= 1.0)) return -1; /* AdaptiveLSearchError */
    View v(ix);
    QueryCtx qc(v, query, false);
    Queue best((size_t)l_value + ix->nstart);
    IdSet visited;
    SearchOut so;
    std::vector<Matched> matched;
    inline_internal(qc, best, visited, beam_width, l_value, filter_bits, adaptive_samples, adaptive_scale, so, matched);
    for (uint32_t i = 0; i < k; ++i) {
        out_ids[i] = 0xFFFFFFFFu;
        out_dists[i] = std::numeric_limits<float>::infinity();
    }
    /* matched_results.take(l_value) -> Translate (start points have no external id) -> first k */
    uint32_t written = 0;
    for (size_t i = 0; i < matched.size() && i < l_value && written < k; ++i) {
        if (matched[i].id >= ix->capacity) continue;
        out_ids[written] = matched[i].id;
        out_dists[written] = matched[i].d;
        ++written;
    }
    if (stats) {
        stats[0] = so.cmps;
        stats[1] = so.hops;
        stats[2] = written;
    }
    return (int32_t)written;
}

/* MultihopFilterSearch, multihop_filter_search.rs:46-244 */
int32_t orc_multihop_search(const orc_index* ix, const void* query, uint32_t l_value, uint32_t beam_width, uint32_t k,
                            const uint32_t* filter_bits, uint32_t* out_ids, float* out_dists, uint32_t* stats) {
    if (!ix || !query || !filter_bits || l_value == 0 || beam_width == 0) return -1;
    View v(ix);
    QueryCtx qc(v, query, false);
    Queue best((size_t)l_value + ix->nstart);
    IdSet visited;
    uint32_t cmps = 0, hops = 0;
    for (uint32_t p = ix->capacity; p < ix->capacity + ix->nstart; ++p) {
        visited.insert(p);
        best.insert(p, qc.eval(p)); /* rejected start points stay in the queue; dropped in post-processing */
    }
    std::vector<uint32_t> beam;
    std::vector<std::pair<uint32_t, float>> one_hop, two_hop;
    std::vector<Matched> cand;
    while (best.has_notvisited()) {
        beam.clear();
        one_hop.clear();
        cand.clear();
        two_hop.clear();
        uint32_t id;
        float d;
        while (beam.size() < beam_width && best.pop(&id, &d)) beam.push_back(id);
        for (uint32_t b : beam) {
            const uint32_t* adj;
            uint32_t n = v.get_neighbors(b, &adj);
            for (uint32_t j = 0; j < n; ++j) {
                uint32_t nb = adj[j];
                if (visited.insert(nb).second && nb < v.nslots() && v.readable(nb)) one_hop.emplace_back(nb, qc.eval(nb));
            }
        }
        for (auto& nb : one_hop) {
            if (is_match(filter_bits, v.nslots(), nb.first)) best.insert(nb.first, nb.second);
            else cand.push_back({nb.first, nb.second});
        }
        cmps += (uint32_t)one_hop.size();
        hops += (uint32_t)beam.size();
        /* closest rejected nodes first (sort_unstable_by: sort_matched_by_distance), at most max_degree / 2 of them */
        sort_matched_by_distance(cand);
        if (cand.size() > ix->max_degree / 2) cand.resize(ix->max_degree / 2);
        /* expand_beam_accept_only: pred.eval_mut = is_match(id) && visited.insert(id) (labeled.rs:284-291) */
        for (auto& c : cand) {
            const uint32_t* adj;
            uint32_t n = v.get_neighbors(c.id, &adj);
            for (uint32_t j = 0; j < n; ++j) {
                uint32_t nb = adj[j];
                if (is_match(filter_bits, v.nslots(), nb) && visited.insert(nb).second && nb < v.nslots() && v.readable(nb))
                    two_hop.emplace_back(nb, qc.eval(nb));
            }
        }
        for (auto& nb : two_hop) best.insert(nb.first, nb.second);
        cmps += (uint32_t)two_hop.size();
        hops += (uint32_t)cand.size();
    }
    for (uint32_t i = 0; i < k; ++i) {
        out_ids[i] = 0xFFFFFFFFu;
        out_dists[i] = std::numeric_limits<float>::infinity();
    }
    uint32_t written = 0;
    size_t n = std::min(std::min(best.search_l, best.size()), (size_t)l_value + ix->nstart);
    size_t taken = 0;
    for (size_t i = 0; i < n && written < k; ++i) {
        uint32_t id = best.ids[i];
        /* .filter(not a rejected start point).take(l_value) then Translate drops every start point */
        if (id >= ix->capacity && !is_match(filter_bits, v.nslots(), id)) continue;
        if (taken++ >= l_value) break;
        if (id >= ix->capacity) continue;
        out_ids[written] = id;
        out_dists[written] = best.dist[i];
        ++written;
    }
    if (stats) {
        stats[0] = cmps;
        stats[1] = hops;
        stats[2] = written;
    }
    return (int32_t)written;
}

/* FilteredRange, filtered_range_search.rs:111-330 */
int32_t orc_filtered_range_search(const orc_index* ix, const void* query, uint32_t starting_l, uint32_t beam_width,
                                  float radius, int32_t has_inner, float inner_radius, float initial_slack,
                                  float range_slack, uint64_t max_returned, const uint32_t* filter_bits,
                                  uint32_t* out_ids, float* out_dists, uint64_t out_cap, uint32_t* stats) {
    if (!ix || !query || !filter_bits || starting_l == 0 || beam_width == 0) return -1;
    View v(ix);
    QueryCtx qc(v, query, false);
    Queue best((size_t)starting_l + ix->nstart);
    IdSet visited;
    SearchOut so;
    std::vector<Matched> matched;
    inline_internal(qc, best, visited, beam_width, starting_l, filter_bits, 0, 1.0, so, matched);
    const uint64_t max_ret = max_returned ? max_returned : ~0ull;
    std::vector<Matched> in_range;
    size_t n = std::min(std::min(best.search_l, best.size()), (size_t)starting_l);
    for (size_t i = 0; i < n; ++i)
        if (best.dist[i] <= radius) in_range.push_back({best.ids[i], best.dist[i]});
    for (auto& m : matched)
        if (m.d <= radius) in_range.push_back(m);
    /* fast_distance_total: distance then id; dedup_by equal ids (adjacent) */
    std::sort(in_range.begin(), in_range.end(),
              [](const Matched& a, const Matched& b) { return a.d < b.d || (a.d == b.d && a.id < b.id); });
    in_range.erase(std::unique(in_range.begin(), in_range.end(),
                               [](const Matched& a, const Matched& b) { return a.id == b.id; }),
                   in_range.end());
    std::vector<Matched> within;
    for (auto& m : matched)
        if (m.d <= radius) within.push_back(m);
    bool second = false;
    uint32_t cmps = so.cmps, hops = so.hops;
    if (in_range.size() >= (size_t)((float)starting_l * initial_slack) && within.size() < max_ret) {
        second = true;
        visited.clear();
        std::vector<uint32_t> frontier;
        for (auto& m : in_range) {
            visited.insert(m.id);
            frontier.push_back(m.id);
        }
        size_t front = 0;
        std::vector<uint32_t> beam;
        std::vector<std::pair<uint32_t, float>> neighbors;
        const float nav = radius * range_slack;
        while (front < frontier.size() && within.size() < max_ret) {
            beam.clear();
            while (front < frontier.size() && beam.size() < beam_width) beam.push_back(frontier[front++]);
            neighbors.clear();
            for (uint32_t b : beam) {
                const uint32_t* adj;
                uint32_t len = v.get_neighbors(b, &adj);
                for (uint32_t j = 0; j < len; ++j) {
                    uint32_t nb = adj[j];
                    if (visited.insert(nb).second && nb < v.nslots() && v.readable(nb)) neighbors.emplace_back(nb, qc.eval(nb));
                }
            }
            for (auto& nb : neighbors) {
                if (nb.second <= nav) {
                    frontier.push_back(nb.first);
                    if (nb.second <= radius && is_match(filter_bits, v.nslots(), nb.first) && within.size() < max_ret)
                        within.push_back({nb.first, nb.second});
                }
            }
            cmps += (uint32_t)neighbors.size();
            hops += (uint32_t)beam.size();
        }
    }
    uint64_t written = 0, taken = 0;
    for (auto& m : within) {
        if (taken++ >= max_ret) break;
        if (m.id >= ix->capacity) continue;
        if (has_inner && m.d <= inner_radius) continue;
        if (written >= out_cap) break;
        out_ids[written] = m.id;
        out_dists[written] = m.d;
        ++written;
    }
    if (stats) {
        stats[0] = cmps;
        stats[1] = hops;
        stats[2] = (uint32_t)written;
        stats[3] = second ? 1u : 0u;
    }
    return (int32_t)written;
}
}  // extern "C"

/* ======================================================================
 * paged search: DiskANNIndex::paged_search (index.rs:2075-2155) + PagedSearch::next_page (search/paged.rs:53-149)
 * ====================================================================== */
struct orc_paged {
    const orc_index* ix;
    View v;
    std::vector<uint8_t> qbytes;
    QueryCtx* qc;
    Queue best;
    IdSet visited;
    std::vector<std::pair<uint32_t, float>> computed;
    size_t next_index;
    uint32_t l_value;
    SearchOut so;
    orc_paged(const orc_index* i, size_t cap) : ix(i), v(i), qc(nullptr), best(cap), next_index(0), l_value(0) {}
};

orc_paged* orc_paged_begin(const orc_index* ix, const void* query, uint32_t l_value) {
    if (!ix || !query || l_value == 0) return nullptr;
    orc_paged* s = new orc_paged(ix, (size_t)l_value + ix->nstart);
    s->best.auto_resizable = true;
    s->qbytes.assign((const uint8_t*)query, (const uint8_t*)query + query_bytes(ix));
    s->qc = new QueryCtx(s->v, s->qbytes.data(), false);
    s->l_value = l_value;
    /* the start points seed the visited set and are expanded, but are not candidates themselves (:2117-2141) */
    for (uint32_t p = ix->capacity; p < ix->capacity + ix->nstart; ++p) s->visited.insert(p);
    std::vector<std::pair<uint32_t, float>> neighbors;
    for (uint32_t p = ix->capacity; p < ix->capacity + ix->nstart; ++p) {
        const uint32_t* adj;
        uint32_t n = s->v.get_neighbors(p, &adj);
        for (uint32_t j = 0; j < n; ++j) {
            uint32_t nb = adj[j];
            if (s->visited.insert(nb).second && nb < s->v.nslots() && s->v.readable(nb)) neighbors.emplace_back(nb, s->qc->eval(nb));
        }
    }
    for (auto& nb : neighbors) s->best.insert(nb.first, nb.second);
    s->computed.assign(l_value, std::make_pair(0u, 0.0f));
    s->next_index = l_value;
    return s;
}

int32_t orc_paged_next(orc_paged* s, uint32_t k, uint32_t* out_ids, float* out_dists) {
    if (!s || !out_ids || !out_dists) return -1;
    if (k > s->l_value || k == 0) return -1; /* "k should be less than or equal to search_param_l" / "> 0" */
    uint32_t n = 0;
    size_t avail = s->computed.size() > s->next_index ? s->computed.size() - s->next_index : 0;
    size_t from_cache = std::min<size_t>(k, avail);
    for (size_t i = 0; i < from_cache; ++i) {
        out_ids[n] = s->computed[s->next_index + i].first;
        out_dists[n] = s->computed[s->next_index + i].second;
        ++n;
    }
    s->next_index += from_cache;
    if (n == k) return (int32_t)n;
    /* resume: search_internal does not re-seed because the visited set is not empty (index.rs:1948-1958) */
    {
        const View& v = s->v;
        std::vector<std::pair<uint32_t, float>> neighbors;
        uint32_t id;
        float d;
        while (s->best.has_notvisited()) {
            if (!s->best.pop(&id, &d)) break;
            neighbors.clear();
            const uint32_t* adj;
            uint32_t len = v.get_neighbors(id, &adj);
            for (uint32_t j = 0; j < len; ++j) {
                uint32_t nb = adj[j];
                if (s->visited.insert(nb).second && nb < v.nslots() && v.readable(nb)) neighbors.emplace_back(nb, s->qc->eval(nb));
            }
            for (auto& nb : neighbors) s->best.insert(nb.first, nb.second);
            s->so.cmps += (uint32_t)neighbors.size();
            s->so.hops += 1;
        }
    }
    /* filter_search_candidates (:126-149) over best.iter() = the first min(search_l, size) entries */
    size_t total = 0, lim = std::min(s->best.search_l, s->best.size());
    std::vector<std::pair<uint32_t, float>> cand;
    for (size_t i = 0; i < lim; ++i) {
        ++total;
        if (s->best.ids[i] >= s->ix->capacity) continue;
        cand.emplace_back(s->best.ids[i], s->best.dist[i]);
        if (cand.size() >= k) break;
    }
    s->best.drain_best(total);
    s->computed = cand;
    s->next_index = 0;
    size_t leftover = std::min<size_t>(k - n, s->computed.size());
    for (size_t i = 0; i < leftover; ++i) {
        out_ids[n] = s->computed[i].first;
        out_dists[n] = s->computed[i].second;
        ++n;
    }
    s->next_index += leftover;
    return (int32_t)n;
}

void orc_paged_end(orc_paged* s) {
    if (!s) return;
    delete s->qc;
    delete s;
}

int32_t orc_expand_beam(const orc_index* ix, const void* query, const uint32_t* ids, uint32_t n,
                        uint32_t* out_ids, float* out_dists) {
    if (!ix || !query) return -1;
    View v(ix);
    QueryCtx qc(v, query, false);
    uint32_t m = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (ids[i] >= v.nslots()) return -3;
        if (!v.readable(ids[i])) continue; /* read_in_bounds -> None: skipped, not counted (provider.rs:681-686) */
        out_ids[m] = ids[i];
        out_dists[m] = qc.eval(ids[i]);
        ++m;
    }
    return (int32_t)m;
}

/* checker for gram_tiles_kernel (dann_debug_gram_tiles): one f32 fmaf chain over k = 0 .. dim-1 per entry (what
 * v_mfma_f32_32x32x2_f32 computes when its accumulator runs through the whole row; zero padding adds nothing) */
void orc_gram_chain(const float* rows, uint32_t n, uint32_t dim, float* out) {
    for (uint32_t i = 0; i < n; ++i)
        for (uint32_t j = 0; j < n; ++j) {
            float acc = 0.0f;
            for (uint32_t k = 0; k < dim; ++k) acc = std::fmaf(rows[(size_t)i * dim + k], rows[(size_t)j * dim + k], acc);
            out[(size_t)i * n + j] = acc;
        }
}

int32_t orc_prune_pool(const orc_index* ix, const orc_build_config* cfg, uint32_t location, uint32_t* pool_ids,
                       float* pool_dists, uint32_t pool_n, int32_t force_saturate, uint32_t* out_neighbors,
                       uint64_t* pair_evals) {
    if (!ix || !cfg) return -1;
    View v(ix);
    std::vector<PNeighbor> pool(pool_n);
    for (uint32_t i = 0; i < pool_n; ++i) pool[i] = {pool_ids[i], pool_dists[i], i};
    sort_pool(pool, cfg->max_occlusion_size);
    std::vector<uint32_t> out;
    occlude_list(v, cfg, location, pool, force_saturate != 0, out, pair_evals);
    for (size_t i = 0; i < pool.size(); ++i) { /* hand the sorted pool back for inspection */
        pool_ids[i] = pool[i].id;
        pool_dists[i] = pool[i].d;
    }
    std::copy(out.begin(), out.end(), out_neighbors);
    return (int32_t)out.size();
}

void orc_set_tie_rule(int32_t rule, uint64_t seed) { tie_rule_set(rule, seed); }

/* the restated Rust sort on its own (tests/test_oracle_rust_sort.py): mode 0 = SortedNeighbors::new(v, max) -- returns
 * the new length; mode 1 = sort_unstable_by over the whole slice; mode 2 = the <= 32-element small sort alone;
 * mode 3 = select_nth_unstable_by(max) alone.  -1 on bad arguments. */
int64_t orc_rust_sort(int32_t mode, uint32_t* ids, float* dists, uint64_t n, uint64_t max) {
    if ((n && (!ids || !dists)) || (mode == 2 && n > 32) || (mode == 3 && max >= n)) return -1;
    std::vector<PNeighbor> v(n);
    for (uint64_t i = 0; i < n; ++i) v[i] = {ids[i], dists[i], (uint32_t)i};
    auto by_d = [](const PNeighbor& a, const PNeighbor& b) { return a.d < b.d; };
    rust_sort::Impl<PNeighbor, decltype(by_d)> s(by_d);
    switch (mode) {
        case 0: g_rust_fallbacks += rust_sort::sorted_neighbors(v, (size_t)max, by_d) ? 1 : 0; break;
        case 1: s.sort_unstable(v.data(), v.size()); break;
        case 2: s.small_sort_network(v.data(), v.size()); break;
        case 3: s.select_nth_unstable(v.data(), v.size(), (size_t)max); g_rust_fallbacks += s.fallback_used ? 1 : 0; break;
        default: return -1;
    }
    for (size_t i = 0; i < v.size(); ++i) {
        ids[i] = v[i].id;
        dists[i] = v[i].d;
    }
    return (int64_t)v.size();
}
/* how often the selection's median-of-medians fallback (restated as a plain sort, rust_unstable_sort.h) was reached
 * since the library was loaded: a pin that relies on rule 6 asserts this stays 0 */
uint64_t orc_rust_sort_fallbacks(void) { return g_rust_fallbacks; }
/* path counters of the restated sort (rust_sort::Path order), `n` words copied; returns the number of paths */
uint32_t orc_rust_sort_paths(uint64_t* out, uint32_t n) {
    for (uint32_t i = 0; i < n && i < (uint32_t)rust_sort::P_COUNT; ++i) out[i] = rust_sort::path_counters()[i];
    return (uint32_t)rust_sort::P_COUNT;
}

/* ---- CPU distance micro-benchmark (bench.py cpu_distance_kernels; never used by a test as a checker) -----------------
 * The shape of diskann-benchmark-simd (src/lib.rs:716-771, examples/simd.json): ONE query against `nrows` contiguous
 * rows, `loops` times over -- everything stays in L1/L2, the number is the kernel's arithmetic rate.  random_order != 0
 * is the cache-defeating variant the GPU gather kernel is held against: the rows of a table far larger than the caches
 * are visited in a random order (one dependent-free load stream, hardware prefetchers useless).  f32 / f16 rows run the
 * AVX2 + FMA kernels of this file (bit-identical to the scalar emulation of the reference's V3 kernels); u8 / i8 rows a
 * plain integer loop the compiler vectorises.  `threads` workers share the table, each with its own query and order.
 * Returns distances per second over all threads (best of 3 timed passes); *checksum keeps the work observable. */
}  // extern "C"
template <bool SIGNED, bool L2>
static int32_t int_rows_op(const uint8_t* a, const uint8_t* b, size_t n) {
    int32_t s = 0;
    for (size_t i = 0; i < n; ++i) {
        const int32_t x = SIGNED ? (int32_t)(int8_t)a[i] : (int32_t)a[i], y = SIGNED ? (int32_t)(int8_t)b[i] : (int32_t)b[i];
        s += L2 ? (x - y) * (x - y) : x * y;
    }
    return s;
}
extern "C" {
double orc_bench_distance(int32_t dtype, int32_t metric, uint32_t dim, uint64_t nrows, uint32_t loops, int32_t random_order,
                          uint32_t threads, uint64_t seed, double* checksum) {
    if (!(dtype == ORC_F32 || dtype == ORC_F16 || dtype == ORC_U8 || dtype == ORC_I8) || dim == 0 || nrows == 0 || loops == 0)
        return -1.0;
    if (!(metric == ORC_L2 || metric == ORC_INNER_PRODUCT)) return -1.0;
    if (threads == 0) threads = 1;
    const size_t esz = elem_size(dtype), rb = (size_t)dim * esz;
    std::vector<uint8_t> table(nrows * rb + 64);
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + 1;
    auto next = [&]() {
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        return x;
    };
    if (dtype == ORC_F32) {
        float* t = reinterpret_cast<float*>(table.data());
        for (size_t i = 0; i < nrows * dim; ++i) t[i] = (float)((int32_t)(next() >> 40) - (1 << 23)) * (1.0f / (1 << 23));
    } else if (dtype == ORC_F16) {
        uint16_t* t = reinterpret_cast<uint16_t*>(table.data());
        for (size_t i = 0; i < nrows * dim; ++i) t[i] = orc_f32_to_f16((float)((int32_t)(next() >> 40) - (1 << 23)) * (1.0f / (1 << 23)));
    } else {
        for (size_t i = 0; i < nrows * rb; i += 8) {
            const uint64_t r = next();
            std::memcpy(table.data() + i, &r, std::min<size_t>(8, nrows * rb - i));
        }
    }
    std::vector<double> rate(threads, 0.0), sums(threads, 0.0);
    auto work = [&](uint32_t tid) {
        std::vector<float> q32(dim);
        std::vector<uint8_t> q(rb);
        uint64_t y = (seed + 77 * (tid + 1)) * 0xD1B54A32D192ED03ull + 1;
        auto nx = [&]() {
            y ^= y << 13;
            y ^= y >> 7;
            y ^= y << 17;
            return y;
        };
        for (uint32_t i = 0; i < dim; ++i) q32[i] = (float)((int32_t)(nx() >> 40) - (1 << 23)) * (1.0f / (1 << 23));
        if (dtype == ORC_F32) std::memcpy(q.data(), q32.data(), rb);
        else if (dtype == ORC_F16) for (uint32_t i = 0; i < dim; ++i) reinterpret_cast<uint16_t*>(q.data())[i] = orc_f32_to_f16(q32[i]);
        else for (size_t i = 0; i < rb; ++i) q[i] = (uint8_t)nx();
        std::vector<uint32_t> order;
        if (random_order) {
            order.resize(nrows);
            for (uint64_t i = 0; i < nrows; ++i) order[i] = (uint32_t)i;
            for (uint64_t i = nrows; i > 1; --i) std::swap(order[i - 1], order[nx() % i]);
        }
        double best = 0.0, acc = 0.0;
        for (int pass = 0; pass < 4; ++pass) { /* pass 0 warms */
            const auto t0 = std::chrono::steady_clock::now();
            for (uint32_t l = 0; l < loops; ++l)
                for (uint64_t r = 0; r < nrows; ++r) {
                    const uint8_t* row = table.data() + (random_order ? (size_t)order[r] : (size_t)r) * rb;
                    float d;
                    if (dtype == ORC_U8) d = (float)(metric == ORC_L2 ? int_rows_op<false, true>(q.data(), row, dim) : int_rows_op<false, false>(q.data(), row, dim));
                    else if (dtype == ORC_I8) d = (float)(metric == ORC_L2 ? int_rows_op<true, true>(q.data(), row, dim) : int_rows_op<true, false>(q.data(), row, dim));
                    else d = query_raw_fast(dtype, metric, q32.data(), q.data(), row, dim);
                    acc += d;
                }
            const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (pass > 0) best = std::max(best, (double)nrows * loops / secs);
        }
        rate[tid] = best;
        sums[tid] = acc;
    };
    std::vector<std::thread> pool;
    for (uint32_t t = 1; t < threads; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
    double total = 0.0, cs = 0.0;
    for (uint32_t t = 0; t < threads; ++t) {
        total += rate[t];
        cs += sums[t];
    }
    if (checksum) *checksum = cs;
    return total;
}

/* counters (five words): [0] query distances, [1] pair (prune) distances, [2] set_neighbors, [3] appends,
 * [4] get_neighbors calls -- one per expanded node of an insert search (the test provider's expand_beam,
 * graph/test/provider.rs:1207-1232) and one per add_edge_and_prune (index.rs:2276-2280); [2]-[4] are what the
 * reference's grid_insert goldens hold as insert_metrics.{set_neighbors, append_neighbors, get_neighbors} */
int32_t orc_insert(orc_index* ix, const orc_build_config* cfg, uint32_t slot, uint64_t* counters) {
    if (!ix || !cfg || slot >= ix->capacity) return -1;
    View v(ix);
    /* insert_search_accessor == search_accessor (glue.rs:932-939); beam width 1, L = l_build */
    QueryCtx qc(v, v.row(slot), false);
    Queue best((size_t)cfg->l_build + ix->nstart);
    IdSet visited;
    std::vector<std::pair<uint32_t, float>> rec;
    SearchOut so;
    so.record = &rec;
    search_internal(qc, best, visited, 1, so);
    if (counters) counters[0] += so.cmps, counters[4] += so.hops;
    std::vector<PNeighbor> pool;
    record_to_pool(rec, pool);
    sort_pool(pool, cfg->max_occlusion_size);
    std::vector<uint32_t> nbrs;
    occlude_list(v, cfg, slot, pool, false, nbrs, counters ? &counters[1] : nullptr);
    int rc = v.set_neighbors(slot, nbrs.data(), (uint32_t)nbrs.size());
    if (rc < 0) return rc;
    if (counters) ++counters[2];
    size_t nb = std::min<size_t>(nbrs.size(), cfg->max_backedges);
    for (size_t i = 0; i < nb; ++i) add_edge_and_prune(v, cfg, &slot, 1, nbrs[i], counters);
    return (int32_t)nbrs.size();
}

int32_t orc_multi_insert(orc_index* ix, const orc_build_config* cfg, const uint32_t* slots, uint32_t n,
                         uint64_t* counters) {
    if (!ix || !cfg) return -1;
    View v(ix);
    struct Pending {
        uint32_t source;
        std::vector<uint32_t> edges;
    };
    std::vector<Pending> edges(n);
    const size_t batch = n;
    size_t cand = cfg->intra_batch_candidates == ORC_IBC_ALL ? batch
                                                               : std::min<size_t>(cfg->intra_batch_candidates, batch);
    /* search_and_prune (index.rs:349-434) for each position, sequentially */
    for (uint32_t pos = 0; pos < n; ++pos) {
        uint32_t id = slots[pos];
        if (id >= ix->capacity) return -3;
        QueryCtx qc(v, v.row(id), false);
        Queue best((size_t)cfg->l_build + ix->nstart);
        IdSet visited;
        std::vector<std::pair<uint32_t, float>> rec;
        SearchOut so;
        so.record = &rec;
        search_internal(qc, best, visited, 1, so);
        if (counters) counters[0] += so.cmps, counters[4] += so.hops;
        /* robust_prune_with (index.rs:2476-2532): extras = around(ids, pos, cand)
         * (utils/async_tools.rs:51-131).  inmem2's PruneAccessor::fill is zero-copy
         * (provider.rs:757-765), so every id is retrievable. */
        if (cand != 0 && n > 1) {
            size_t len = std::min(cand, (size_t)n - 1);
            size_t half = (len + 1) / 2;
            size_t p = pos >= half ? pos - half : n - (half - pos);
            for (size_t r = 0; r < len; ++r) {
                size_t i = p;
                p = (p + 1 == n) ? 0 : p + 1;
                if (i == pos) {
                    i = p;
                    p = (p + 1 == n) ? 0 : p + 1;
                }
                rec.emplace_back(slots[i], v.pair(id, slots[i]));
                if (counters) ++counters[1];
            }
        }
        std::vector<PNeighbor> pool;
        record_to_pool(rec, pool);
        sort_pool(pool, cfg->max_occlusion_size);
        edges[pos].source = id;
        occlude_list(v, cfg, id, pool, false, edges[pos].edges, counters ? &counters[1] : nullptr);
    }
    /* aggregate_backedges (index.rs:123-143) */
    auto aggregate = [&](std::unordered_map<uint32_t, std::vector<uint32_t>>& map) {
        map.clear();
        for (auto& e : edges)
            for (uint32_t t : e.edges) map[t].push_back(e.source);
    };
    std::unordered_map<uint32_t, std::vector<uint32_t>> back;
    aggregate(back);
    /* bootstrap (index.rs:926-938, 597-645) */
    size_t resolved = std::max<size_t>(cand, 1);
    if (resolved < batch && (back.size() + 7) / 8 <= batch) {
        std::vector<Pending> next(n);
        for (uint32_t pos = 0; pos < n; ++pos) {
            std::vector<uint32_t> cands; /* AdjacencyList::from_iter_untrusted: dedup, keep first */
            auto push_unique = [&](uint32_t x) {
                if (std::find(cands.begin(), cands.end(), x) == cands.end()) cands.push_back(x);
            };
            for (uint32_t e : edges[pos].edges) push_unique(e);
            for (uint32_t o = 0; o < n; ++o)
                if (edges[o].source != edges[pos].source) push_unique(edges[o].source);
            /* from_iter_untrusted is sort_unstable + dedup (adjacencylist.rs:181-190): the list robust_prune_list walks
             * is in ascending id order (rule 6, and the product under DANN_TIE_RUST).  Under the position rules the order of
             * first occurrence is kept (the product's bootstrap kernel under DANN_TIE_POSITION; only ties can tell the
             * two apart). */
            if (g_tie_rule == 6) std::sort(cands.begin(), cands.end());
            next[pos].source = edges[pos].source;
            robust_prune_list(v, cfg, edges[pos].source, cands, true, next[pos].edges, counters);
        }
        edges.swap(next);
        aggregate(back);
    }
    /* set_neighbors_bulk (index.rs:948-962) */
    for (auto& e : edges) {
        int rc = v.set_neighbors(e.source, e.edges.data(), (uint32_t)e.edges.size());
        if (rc < 0) return rc;
        if (counters) ++counters[2];
    }
    /* back-edges: each source once, targets sorted (index.rs:988-1003).  Sources are
     * independent, so the HashMap iteration order does not matter; iterate sorted. */
    std::vector<uint32_t> sources;
    sources.reserve(back.size());
    for (auto& kv : back) sources.push_back(kv.first);
    std::sort(sources.begin(), sources.end());
    for (uint32_t s : sources) {
        auto& t = back[s];
        std::sort(t.begin(), t.end());
        add_edge_and_prune(v, cfg, t.data(), (uint32_t)t.size(), s, counters);
    }
    return 0;
}

int64_t orc_medoid_f32(const float* data, uint64_t nrows, uint32_t dim, float* out_mean) {
    if (dim == 0 || nrows == 0) return -1;
    std::vector<double> sum(dim, 0.0);
    for (uint64_t r = 0; r < nrows; ++r)
        for (uint32_t c = 0; c < dim; ++c) sum[c] += (double)data[r * dim + c];
    std::vector<float> m(dim);
    for (uint32_t c = 0; c < dim; ++c) m[c] = (float)(sum[c] / (double)nrows);
    if (out_mean) std::copy(m.begin(), m.end(), out_mean);
    float min_dist = std::numeric_limits<float>::max();
    int64_t best = -1;
    for (uint64_t r = 0; r < nrows; ++r) {
        float d = query_raw_fast(ORC_F32, ORC_L2, nullptr, m.data(), data + r * dim, dim);
        if (d < min_dist) {
            min_dist = d;
            best = (int64_t)r;
        }
    }
    return best;
}

void orc_pq_build_lut(int32_t metric, const float* pivots, const float* centroid, const uint32_t* chunk_offsets,
                      uint32_t nchunks, uint32_t dim, const float* query, float* lut) {
    std::vector<float> q(query, query + dim);
    if (centroid)
        for (uint32_t i = 0; i < dim; ++i) q[i] = query[i] - centroid[i];
    for (uint32_t c = 0; c < 256; ++c) {
        for (uint32_t ch = 0; ch < nchunks; ++ch) {
            uint32_t s = chunk_offsets[ch], e = chunk_offsets[ch + 1];
            const float* piv = pivots + (size_t)c * dim + s;
            float raw = metric == ORC_L2 ? simd_op_f<4, AccL2>(q.data() + s, piv, e - s)
                                         : simd_op_f<4, AccIP>(q.data() + s, piv, e - s);
            lut[(size_t)ch * 256 + c] = metric == ORC_L2 ? raw : -raw;
        }
    }
}

float orc_pq_lookup(const float* lut, const uint8_t* code, uint32_t nchunks) {
    float accum = 0.0f;
    for (uint32_t ch = 0; ch < nchunks; ++ch) accum += lut[(size_t)ch * 256 + code[ch]];
    return accum;
}

/* ---- PQ compression: TransposedTable::compress_into -> Chunk::find_closest
 * (diskann-quantization/src/product/tables/transposed/table.rs:382-403, pivots.rs:253-345, 785-905).
 * score(j) = |p_j|^2 - (ip_j + ip_j), ip_j = fma chain over the chunk's dimensions in order, |p_j|^2 from
 * kmeans::square_norm (algorithms/kmeans/common.rs:8-62).  Minimum tracked per SIMD lane (j mod 8) with
 * strict `<`, lanes scanned in order with strict `<` from f32::MAX: lexicographic min of (score, j % 8, j). */
static float pq_square_norm(const float* x, size_t len) {
    F8 s = f8_zero();
    size_t i = 0;
    float v[8];
    if (i + 32 <= len) {
        F8 acc[4] = {f8_zero(), f8_zero(), f8_zero(), f8_zero()};
        while (i + 32 <= len) {
            for (int b = 0; b < 4; ++b) {
                load8(x + i + 8 * b, 8, v);
                for (int l = 0; l < 8; ++l) acc[b].v[l] = std::fma(v[l], v[l], acc[b].v[l]);
            }
            i += 32;
        }
        s = f8_add(f8_add(acc[0], acc[1]), f8_add(acc[2], acc[3]));
    }
    while (i + 8 <= len) {
        load8(x + i, 8, v);
        for (int l = 0; l < 8; ++l) s.v[l] = std::fma(v[l], v[l], s.v[l]);
        i += 8;
    }
    if (len - i) {
        load8(x + i, (int)(len - i), v);
        for (int l = 0; l < 8; ++l) s.v[l] = std::fma(v[l], v[l], s.v[l]);
    }
    return f8_sum_tree(s);
}

int32_t orc_pq_square_norms(const float* pivots, uint32_t ncenters, const uint32_t* chunk_offsets, uint32_t nchunks,
                            uint32_t dim, float* norms /* nchunks x ncenters */) {
    for (uint32_t c = 0; c < nchunks; ++c)
        for (uint32_t j = 0; j < ncenters; ++j)
            norms[(size_t)c * ncenters + j] =
                pq_square_norm(pivots + (size_t)j * dim + chunk_offsets[c], chunk_offsets[c + 1] - chunk_offsets[c]);
    return 0;
}

int64_t orc_pq_compress(const float* pivots, uint32_t ncenters, const uint32_t* chunk_offsets, uint32_t nchunks,
                        uint32_t dim, const float* rows, uint64_t n, uint8_t* codes) {
    if (!pivots || !chunk_offsets || !rows || !codes || ncenters == 0 || ncenters > 256) return -1;
    std::vector<float> norms((size_t)nchunks * ncenters);
    orc_pq_square_norms(pivots, ncenters, chunk_offsets, nchunks, dim, norms.data());
    for (uint64_t r = 0; r < n; ++r) {
        for (uint32_t c = 0; c < nchunks; ++c) {
            const uint32_t s0 = chunk_offsets[c], len = chunk_offsets[c + 1] - s0;
            const float* x = rows + r * dim + s0;
            float best_d[8];
            uint32_t best_i[8];
            for (int l = 0; l < 8; ++l) {
                best_d[l] = std::numeric_limits<float>::infinity();
                best_i[l] = 0xFFFFFFFFu;
            }
            for (uint32_t j = 0; j < ncenters; ++j) {
                const float* p = pivots + (size_t)j * dim + s0;
                float ip = 0.0f;
                for (uint32_t d = 0; d < len; ++d) ip = std::fma(x[d], p[d], ip);
                const float score = norms[(size_t)c * ncenters + j] - (ip + ip);
                if (score < best_d[j & 7]) {
                    best_d[j & 7] = score;
                    best_i[j & 7] = j;
                }
            }
            float md = std::numeric_limits<float>::max();
            uint32_t mi = 0xFFFFFFFFu;
            for (int l = 0; l < 8; ++l)
                if (best_d[l] < md) {
                    md = best_d[l];
                    mi = best_i[l];
                }
            if (!std::isfinite(md) || mi == 0xFFFFFFFFu) return -(int64_t)(2 + r * nchunks + c); /* InfinityOrNaN(chunk, row) */
            codes[r * nchunks + c] = (uint8_t)mi;
        }
    }
    return 0;
}

/* ---- PQ training: Lloyd's iterations of LightPQTrainingParameters::train per chunk
 * (diskann-quantization/src/product/train.rs:96-226 -> algorithms/kmeans/lloyds.rs:23-438).
 * The k-means++ seeding (kmeans/plusplus.rs, rand's StdRng) is the caller's: `centers` holds the initial
 * centres.  Assignment: score(c) = ((n_c - ip) - ip) + |x|^2 with ip an fma chain over the chunk's dimensions,
 * first strictly smaller wins in centre order (lloyds.rs:66-200, 262-270).  Residual: per SIMD lane (point index
 * mod 8) in point order, then sum_tree (:201, :254-257).  Update: f64 sums in row order, divided by max(count, 1)
 * (:273-296).  Norms of the centres are refreshed between iterations only (:334-350).
 * centers: ncenters x dim (chunk columns concatenated), assignments: nchunks x n (last assignment step),
 * residuals: nchunks. */
/* ---- k-means++ seeding of the PQ trainer: kmeans::plusplus::kmeans_plusplus_into_inner
 * (diskann-quantization/src/algorithms/kmeans/plusplus.rs:366-497) for every chunk, as LightPQTrainingParameters::train
 * calls it (product/train.rs:164-197).  The two random draws of the algorithm come from the caller's generator
 * (the reference seeds rand's StdRng per chunk, random.rs:33-44; that generator is not part of the reference tree):
 *   uniform_index(ctx, chunk, n)   == Uniform::new(0, n).sample(rng)           first centre
 *   uniform_f64(ctx, chunk, high)  == Uniform::<f64>::new(0.0, high).sample(rng)  D^2 threshold
 * update_distances (:239-311) with the BlockTransposed<f32, 16> micro-kernel (:87-237): per row one fma chain over
 * the chunk's columns in order, times -2, distance = (norm + |centre|^2) + that, minimum with strict <, and the sum of
 * the minima accumulated in f64 block by block (16 rows: lanes k and k + 8 paired, the 8 pair sums folded in order).
 * Returns 0, or -2 when a non-finite total appears (FailureReason::SawInfinity); selected[c] < ncenters reports the
 * recoverable failures (DatasetTooSmall / InsufficientDiversity: remaining centres stay zero). */
int32_t orc_pq_kmeanspp(const float* data, uint64_t n, uint32_t dim, const uint32_t* chunk_offsets, uint32_t nchunks,
                        uint32_t ncenters, const orc_rng* rng, float* centers, uint32_t* selected) {
    if (!data || !chunk_offsets || !centers || !rng || !rng->uniform_index || !rng->uniform_f64) return -1;
    for (uint32_t c = 0; c < nchunks; ++c) {
        const uint32_t s0 = chunk_offsets[c], len = chunk_offsets[c + 1] - s0;
        for (uint32_t j = 0; j < ncenters; ++j) std::fill(centers + (size_t)j * dim + s0, centers + (size_t)j * dim + s0 + len, 0.0f);
        uint32_t sel = 0;
        if (selected) selected[c] = 0;
        if (n == 0 || ncenters == 0) continue; /* DatasetTooSmall / nothing to do */
        std::vector<float> norms(n), mins(n, std::numeric_limits<float>::infinity());
        for (uint64_t r = 0; r < n; ++r) norms[r] = pq_square_norm(data + r * dim + s0, len);
        std::vector<uint8_t> picked(n, 0);
        uint64_t first = rng->uniform_index(rng->ctx, c, n);
        if (first >= n) return -1;
        std::copy(data + first * dim + s0, data + first * dim + s0 + len, centers + s0);
        picked[first] = 1;
        float prev_norm = norms[first];
        sel = 1;
        const uint64_t limit = std::min<uint64_t>(ncenters, n);
        for (uint64_t cur = 1; cur < limit; ++cur) {
            const float* last = centers + (cur - 1) * dim + s0;
            /* update_distances */
            double rolling = 0.0;
            for (uint64_t b0 = 0; b0 < n; b0 += 16) {
                float curd[16];
                for (int r = 0; r < 16; ++r) {
                    const uint64_t row = b0 + r;
                    if (row >= n) { /* finish_last: lanes past the end hold 0 and stay 0 */
                        curd[r] = 0.0f;
                        continue;
                    }
                    const float* x = data + row * dim + s0;
                    float acc = 0.0f;
                    for (uint32_t k = 0; k < len; ++k) acc = std::fma(x[k], last[k], acc);
                    acc = acc * -2.0f;
                    const float d = (norms[row] + prev_norm) + acc;
                    if (d < mins[row]) mins[row] = d;
                    curd[r] = mins[row];
                }
                double blk = 0.0;
                for (int k = 0; k < 8; ++k) blk += (double)curd[k] + (double)curd[k + 8];
                rolling += blk;
            }
            const double s = rolling;
            bool got = false;
            if (0.0 < s) { /* Uniform::<f64>::new(0.0, s): EmptyRange unless 0 < s */
                if (!std::isfinite(s)) {
                    if (selected) selected[c] = sel;
                    return -2; /* NonFinite -> SawInfinity */
                }
                const double threshold = rng->uniform_f64(rng->ctx, c, s);
                double acc = 0.0;
                for (uint64_t i = 0; i < n; ++i) {
                    acc += (double)mins[i];
                    if (acc >= threshold && mins[i] > 0.0f && !picked[i]) {
                        std::copy(data + i * dim + s0, data + i * dim + s0 + len, centers + cur * dim + s0);
                        picked[i] = 1;
                        prev_norm = norms[i];
                        sel = (uint32_t)cur + 1;
                        got = true;
                        break;
                    }
                }
            }
            if (!got) break; /* InsufficientDiversity */
        }
        if (selected) selected[c] = sel;
    }
    return 0;
}

int32_t orc_pq_lloyds(const float* data, uint64_t n, uint32_t dim, const uint32_t* chunk_offsets, uint32_t nchunks,
                      uint32_t ncenters, float* centers, uint32_t max_reps, uint32_t* assignments, float* residuals) {
    if (!data || !chunk_offsets || !centers || ncenters == 0 || n == 0) return -1;
    for (uint32_t c = 0; c < nchunks; ++c) {
        const uint32_t s0 = chunk_offsets[c], len = chunk_offsets[c + 1] - s0;
        std::vector<float> dnorm(n), cnorm(ncenters);
        for (uint64_t r = 0; r < n; ++r) dnorm[r] = pq_square_norm(data + r * dim + s0, len);
        for (uint32_t j = 0; j < ncenters; ++j) cnorm[j] = pq_square_norm(centers + (size_t)j * dim + s0, len);
        std::vector<uint32_t> assign(n, 0);
        float residual = 0.0f;
        for (uint32_t rep = 0; rep < max_reps; ++rep) {
            float lanes[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (uint64_t r = 0; r < n; ++r) {
                const float* x = data + r * dim + s0;
                float best = std::numeric_limits<float>::infinity();
                uint32_t bi = 0xFFFFFFFFu;
                for (uint32_t j = 0; j < ncenters; ++j) {
                    const float* p = centers + (size_t)j * dim + s0;
                    float ip = 0.0f;
                    for (uint32_t d = 0; d < len; ++d) ip = std::fma(p[d], x[d], ip);
                    const float sc = ((cnorm[j] - ip) - ip) + dnorm[r];
                    if (sc < best) {
                        best = sc;
                        bi = j;
                    }
                }
                assign[r] = bi;
                lanes[r & 7] = lanes[r & 7] + best;
            }
            residual = ((lanes[0] + lanes[4]) + (lanes[2] + lanes[6])) + ((lanes[1] + lanes[5]) + (lanes[3] + lanes[7]));
            std::vector<double> sums((size_t)ncenters * len, 0.0);
            std::vector<uint32_t> counts(ncenters, 0);
            for (uint64_t r = 0; r < n; ++r) {
                const uint32_t j = assign[r];
                if (j >= ncenters) return -2; /* every score NaN: the reference would index out of bounds */
                counts[j] += 1;
                for (uint32_t d = 0; d < len; ++d) sums[(size_t)j * len + d] += (double)data[r * dim + s0 + d];
            }
            for (uint32_t j = 0; j < ncenters; ++j) {
                const double cnt = (double)std::max<uint32_t>(counts[j], 1);
                for (uint32_t d = 0; d < len; ++d) centers[(size_t)j * dim + s0 + d] = (float)(sums[(size_t)j * len + d] / cnt);
            }
            if (rep != max_reps - 1)
                for (uint32_t j = 0; j < ncenters; ++j) cnorm[j] = pq_square_norm(centers + (size_t)j * dim + s0, len);
        }
        if (assignments) std::memcpy(assignments + (size_t)c * n, assign.data(), n * 4);
        if (residuals) residuals[c] = residual;
    }
    return 0;
}

/* ScalarQuantizationParameters::train (diskann-quantization/src/scalar/train.rs:33-52) with its statistics
 * helpers (utils.rs:109-140, 180-199): f64 sums in row order. */
void orc_sq8_train(const float* data, uint64_t n, uint32_t dim, double standard_deviations, float* shift,
                   float* scale, float* mean_norm) {
    std::vector<double> means(dim, 0.0), var(dim, 0.0);
    double norm_sum = 0.0;
    for (uint64_t r = 0; r < n; ++r) {
        const float* row = data + r * dim;
        for (uint32_t d = 0; d < dim; ++d) means[d] += (double)row[d];
        double sq = 0.0;
        for (uint32_t d = 0; d < dim; ++d) {
            const double x = (double)row[d];
            sq += x * x;
        }
        norm_sum = norm_sum + std::sqrt(sq);
    }
    const double mn = norm_sum / (double)n;
    for (uint32_t d = 0; d < dim; ++d) means[d] /= (double)n;
    for (uint64_t r = 0; r < n; ++r) {
        const float* row = data + r * dim;
        for (uint32_t d = 0; d < dim; ++d) {
            const double df = (double)row[d] - means[d];
            var[d] += df * df;
        }
    }
    double mx = 0.0;
    for (uint32_t d = 0; d < dim; ++d) {
        var[d] /= (double)n;
        mx = std::max(mx, var[d]); /* fold(0.0, f64::max) */
    }
    const double p = std::sqrt(mx) * standard_deviations;
    *scale = (float)(2.0 * p);
    for (uint32_t d = 0; d < dim; ++d) shift[d] = (float)(means[d] - p);
    if (mean_norm) *mean_norm = (float)mn;
}

void orc_sq8_compress(const float* x, uint32_t dim, const float* shift, float scale, uint8_t* code,
                      float* compensation) {
    const float inverse_scale = 255.0f / scale;
    float dot = 0.0f;
    for (uint32_t i = 0; i < dim; ++i) {
        float c = (x[i] - shift[i]) * inverse_scale;
        c = c < 0.0f ? 0.0f : (c > 255.0f ? 255.0f : c);
        c = std::round(c);
        dot = std::fmaf(c, shift[i], dot);
        code[i] = (uint8_t)c;
    }
    if (compensation) *compensation = scale * (1.0f / 255.0f) * dot;
}

float orc_sq8_distance(int32_t metric, const uint8_t* x, float cx, const uint8_t* y, float cy, uint32_t dim,
                       float scale, float shift_norm_sq) {
    const float ibs = 1.0f / 255.0f;
    const float bit_scale = ibs * ibs;
    const float scale_sq = scale * scale;
    if (metric == ORC_L2) {
        uint32_t s = 0;
        for (uint32_t i = 0; i < dim; ++i) {
            int32_t c = (int32_t)x[i] - (int32_t)y[i];
            s += (uint32_t)(c * c);
        }
        return bit_scale * scale_sq * (float)s;
    }
    uint32_t p = 0;
    for (uint32_t i = 0; i < dim; ++i) p += (uint32_t)x[i] * (uint32_t)y[i];
    float r = std::fmaf(bit_scale * scale_sq, (float)p, shift_norm_sq) + (cy + cx);
    return -r;
}

} /* extern "C" */
