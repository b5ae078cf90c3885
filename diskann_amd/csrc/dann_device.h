// dann_device.h -- CDNA4 (gfx950) device primitives for the DiskANN hot path.
//
// The distance functions reproduce, bit for bit, the association order of the
// reference's x86-64-v3 kernels (diskann-vector/src/distance/simd.rs:321-483 main-loop
// strategies, :686-747 simd_op, diskann-wide/src/arch/x86_64/v3/f32x8_.rs:185-212
// sum_tree), because the returned neighbour ids must equal the CPU path's at fixed L
// and near-ties in the L-queue are decided by the last ulp.
//
// Mapping onto a 64-wide wavefront.  The CPU keeps NACC accumulators of 8 f32 lanes; element
// e of a vector is accumulated (one FMA per visit, increasing e) into "chain"
// c = e mod (8*NACC).  A chain is a strictly sequential FMA sequence, so a whole chain must
// live in one GPU lane.  We give every candidate row a group of G = 2*NACC adjacent lanes;
// lane v of the group owns chains 4v..4v+3, i.e. per "trip" of 8*NACC elements it loads the
// 4 consecutive elements [trip + 4v, trip + 4v + 4) -- one 16-byte load for f32 rows,
// one 8-byte load for f16 rows -- so a group reads 128 (64) contiguous bytes per
// instruction and a wavefront reads 64/G rows at once.  The accumulator combine
// `(s0+s1)+(s2+s3)` and the 8-lane horizontal tree are three DPP adds plus three in-lane
// adds.  The result is valid in lane v == 0 of each group.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dann {

enum : int { DT_F32 = 0, DT_F16 = 1, DT_U8 = 2, DT_I8 = 3, DT_SQ8 = 4, DT_PQ = 5 };
enum : int { M_COSINE = 0, M_IP = 1, M_L2 = 2, M_COSN = 3 };
enum : int { OP_L2 = 0, OP_IP = 1, OP_COS = 2 };

constexpr uint32_t kEmpty = 0xFFFFFFFFu;
constexpr uint32_t kVisitedBit = 0x80000000u;

// DPP controls (cdna4 ISA: quad_perm = 0x00-0xFF, row_shl:n = 0x100+n)
constexpr int DPP_XOR1 = 0xB1;   // quad_perm:[1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;   // quad_perm:[2,3,0,1]
constexpr int DPP_SHL4 = 0x104;  // lane i <- lane i+4 (within a row of 16)

template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
    return __builtin_bit_cast(float,
                              __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int x) {
    return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true);
}

__device__ __forceinline__ uint64_t ballot64(bool p) { return __ballot(p); }
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
// number of set bits of `m` strictly below this lane
__device__ __forceinline__ uint32_t mbcnt(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// ---- element loads: 4 consecutive elements as f32 -----------------------------------
struct F4 {
    float x, y, z, w;
};
__device__ __forceinline__ F4 load4(const float* p) {
    float4 t = *reinterpret_cast<const float4*>(p);
    return {t.x, t.y, t.z, t.w};
}
__device__ __forceinline__ F4 load4(const __half* p) {
    // exact widening (v_cvt_f32_f16), as `half` -> f32 on the CPU
    uint2 t = *reinterpret_cast<const uint2*>(p);
    __half2 a = __builtin_bit_cast(__half2, t.x), b = __builtin_bit_cast(__half2, t.y);
    float2 fa = __half22float2(a), fb = __half22float2(b);
    return {fa.x, fa.y, fb.x, fb.y};
}
__device__ __forceinline__ float load1(const float* p) { return *p; }
__device__ __forceinline__ float load1(const __half* p) { return __half2float(*p); }
__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(__half x) { return __half2float(x); }


// Raw (unconverted) loads: converting inside a predicated block makes hipcc wait for each load
// before issuing the next (measured: 16 serialized round trips per f16 gather), so the loaders
// below only move bits; conversion happens in the compute phase.
template <typename RT>
struct Raw4;
template <>
struct Raw4<float> {
    float4 v;
    __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const float4*>(p); }
    __device__ __forceinline__ F4 get() const { return {v.x, v.y, v.z, v.w}; }
};
template <>
struct Raw4<__half> {
    uint2 v;
    __device__ __forceinline__ void load(const __half* p) { v = *reinterpret_cast<const uint2*>(p); }
    __device__ __forceinline__ F4 get() const {
        float2 a = __half22float2(__builtin_bit_cast(__half2, v.x)), b = __half22float2(__builtin_bit_cast(__half2, v.y));
        return {a.x, a.y, b.x, b.y};
    }
};

// FullCosineAccumulator::sum (simd.rs:2329-2362)
__device__ __forceinline__ float cosine_finish(float normx, float normy, float prod) {
    // NB: __fsqrt_rn is NOT correctly rounded on ROCm 7.2 (15 % of inputs off by 1 ulp on gfx950);
    // __builtin_sqrtf and operator/ are (measured exhaustively in scratch probes).
    float denominator = __builtin_sqrtf(normx) * __builtin_sqrtf(normy);
    if (normx < 1.17549435e-38f || normy < 1.17549435e-38f) return 0.0f;
    float v = prod / denominator;
    float m = (v != v) ? 1.0f : (v < 1.0f ? v : 1.0f);
    return m > -1.0f ? m : -1.0f;
}

// PostOp (diskann-vector/src/distance/implementations.rs:215-401)
template <int OP, bool NORMALIZED>
__device__ __forceinline__ float post_op(float raw) {
    if (OP == OP_L2) return raw;
    if (OP == OP_IP) return NORMALIZED ? 1.0f - raw : -raw;
    return 1.0f - raw;
}

// One FMA step of a schema (simd.rs L2 :817-845, IP :1588-1616, cosine :2296-2305)
template <int OP>
struct FAcc {
    float s[4], nx[4], ny[4];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] = nx[i] = ny[i] = 0.0f;
    }
    // element pairs as 2-vectors: v_pk_add_f32 / v_pk_fma_f32 (one IEEE subtract / fused multiply-add per element,
    // the same results as the scalar instructions, half as many of them)
    typedef float v2f __attribute__((ext_vector_type(2)));
    __device__ __forceinline__ void step(const F4& x, const F4& y) {
        const v2f xv[2] = {{x.x, x.y}, {x.z, x.w}}, yv[2] = {{y.x, y.y}, {y.z, y.w}};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            v2f sv = {s[2 * h], s[2 * h + 1]};
            if (OP == OP_L2) {
                const v2f c = xv[h] - yv[h];
                sv = __builtin_elementwise_fma(c, c, sv);
            } else if (OP == OP_IP) {
                sv = __builtin_elementwise_fma(xv[h], yv[h], sv);
            } else {
                v2f nxv = {nx[2 * h], nx[2 * h + 1]}, nyv = {ny[2 * h], ny[2 * h + 1]};
                nxv = __builtin_elementwise_fma(xv[h], xv[h], nxv);
                nyv = __builtin_elementwise_fma(yv[h], yv[h], nyv);
                sv = __builtin_elementwise_fma(xv[h], yv[h], sv);
                nx[2 * h] = nxv.x; nx[2 * h + 1] = nxv.y;
                ny[2 * h] = nyv.x; ny[2 * h + 1] = nyv.y;
            }
            s[2 * h] = sv.x; s[2 * h + 1] = sv.y;
        }
    }
};

// accumulator combine across the group + partial block + horizontal tree for one f32x8
// logical vector held as 4 floats in each of G lanes.  Returns the sum (valid in lane v==0).
template <int NACC, class PartialFn>
__device__ __forceinline__ float finish_vec(float (&a)[4], PartialFn&& partial) {
    // (s0+s1) [+ (s2+s3)]: lane v holds accumulator v/2, vector lanes 4*(v&1)+i
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = a[i] + dpp_f<DPP_XOR2>(a[i]);
    if (NACC == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = a[i] + dpp_f<DPP_SHL4>(a[i]);
    }
    partial(a);
    // sum_tree: ((x0+x4)+(x2+x6)) + ((x1+x5)+(x3+x7)); lane even holds x0..3, odd x4..7
    float q0 = a[0] + dpp_f<DPP_XOR1>(a[0]);
    float q1 = a[1] + dpp_f<DPP_XOR1>(a[1]);
    float q2 = a[2] + dpp_f<DPP_XOR1>(a[2]);
    float q3 = a[3] + dpp_f<DPP_XOR1>(a[3]);
    return (q0 + q2) + (q1 + q3);
}

// Distance between `q` (QT elements, any address space) and `row` (RT elements), both of
// length `dim`, computed by a group of G = 2*NACC lanes; `v` = lane index inside the group.
// DIM > 0 fixes the length at compile time (fully unrolled, all loads issued up front).
template <int NACC, int OP, int DIM, typename QT, typename RT>
__device__ __forceinline__ float group_distance_raw(const QT* __restrict__ q, const RT* __restrict__ row, int dim,
                                                    int v) {
    constexpr int G = 2 * NACC, TRIP = 4 * G;
    if (DIM > 0) dim = DIM;
    const int full_end = dim & ~7;
    FAcc<OP> acc;
    acc.init();
    if constexpr (DIM > 0) {
        constexpr int NT = (DIM / 8 * 8 + TRIP - 1) / TRIP;
        F4 ys[NT], xs[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            int e = t * TRIP + 4 * v;
            if (e < full_end) ys[t] = load4(row + e);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            int e = t * TRIP + 4 * v;
            if (e < full_end) xs[t] = load4(q + e);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            int e = t * TRIP + 4 * v;
            if (e < full_end) acc.step(xs[t], ys[t]);
        }
    } else {
        for (int e = 4 * v; e < full_end; e += TRIP) {
            F4 y = load4(row + e);
            F4 x = load4(q + e);
            acc.step(x, y);
        }
    }
    const int rem = dim & 7;
    RT ty[4];
    if (rem) {  // all four requests issued together (index clamped into the block)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int l = 4 * (v & 1) + i;
            ty[i] = row[full_end + (l < rem ? l : 0)];
        }
    }
    // partial block: zero-padded masked load accumulated into the *combined* vector
    // (simd.rs:735-745, SIMDSchema::epilogue :545-563)
    auto partial = [&](float(&a)[4], int which) {
        if (rem == 0) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int l = 4 * (v & 1) + i;
            float x = 0.0f, y = 0.0f;
            if (l < rem) {
                x = load1(q + full_end + l);
                y = to_f32(ty[i]);
            }
            if (OP == OP_L2) {
                float c = x - y;
                a[i] = __builtin_fmaf(c, c, a[i]);
            } else if (which == 0) {
                a[i] = __builtin_fmaf(x, y, a[i]);
            } else if (which == 1) {
                a[i] = __builtin_fmaf(x, x, a[i]);
            } else {
                a[i] = __builtin_fmaf(y, y, a[i]);
            }
        }
    };
    float s = finish_vec<NACC>(acc.s, [&](float(&a)[4]) { partial(a, 0); });
    if (OP == OP_COS) {
        float nx = finish_vec<NACC>(acc.nx, [&](float(&a)[4]) { partial(a, 1); });
        float ny = finish_vec<NACC>(acc.ny, [&](float(&a)[4]) { partial(a, 2); });
        return cosine_finish(nx, ny, s);
    }
    return s;
}


// Stored row x stored row (RobustPrune's primitive) with every load of a T-trip window -- both rows -- issued
// before the first FMA.  group_distance_raw's run-time loop waits for each trip's loads before issuing the next
// trip's: four dependent round trips per 128-d distance, which is what the lazy prune scan then spends its time on.
// Same arithmetic, element order and partial-block rule as group_distance_raw.
template <int NACC, int OP, typename RT>
__device__ __forceinline__ float group_distance_pair(const RT* __restrict__ x, const RT* __restrict__ y, int dim, int v) {
    constexpr int G = 2 * NACC, TRIP = 4 * G;
    const int full_end = dim & ~7, rem = dim & 7;
    FAcc<OP> acc;
    acc.init();
    RT tx[4], ty[4];
    if (rem) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int l = 4 * (v & 1) + i, lc = l < rem ? l : 0;  // clamped, unconditional: no branch, no wait
            tx[i] = x[full_end + lc];
            ty[i] = y[full_end + lc];
        }
    }
    constexpr int T = 4;
    for (int e0 = 4 * v; e0 < full_end; e0 += TRIP * T) {
        Raw4<RT> xs[T], ys[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int e = e0 + t * TRIP;
            if (e < full_end) {
                xs[t].load(x + e);
                ys[t].load(y + e);
            }
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int e = e0 + t * TRIP;
            if (e < full_end) acc.step(xs[t].get(), ys[t].get());
        }
    }
    auto partial = [&](float(&a)[4], int which) {
        if (rem == 0) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int l = 4 * (v & 1) + i;
            float xv = 0.0f, yv = 0.0f;
            if (l < rem) {
                xv = to_f32(tx[i]);
                yv = to_f32(ty[i]);
            }
            if (OP == OP_L2) {
                const float c = xv - yv;
                a[i] = __builtin_fmaf(c, c, a[i]);
            } else if (which == 0) {
                a[i] = __builtin_fmaf(xv, yv, a[i]);
            } else if (which == 1) {
                a[i] = __builtin_fmaf(xv, xv, a[i]);
            } else {
                a[i] = __builtin_fmaf(yv, yv, a[i]);
            }
        }
    };
    float s = finish_vec<NACC>(acc.s, [&](float(&a)[4]) { partial(a, 0); });
    if (OP == OP_COS) {
        float nx = finish_vec<NACC>(acc.nx, [&](float(&a)[4]) { partial(a, 1); });
        float ny = finish_vec<NACC>(acc.ny, [&](float(&a)[4]) { partial(a, 2); });
        return cosine_finish(nx, ny, s);
    }
    return s;
}


// Fixed-length variant with the query slice of this lane preloaded in registers
// (DIM % (8*NACC) == 0, so there is neither an epilogue block nor a partial block).
// `U` rows are processed together: all U*NT row loads are issued before the first FMA so
// one lane keeps U*NT 16-byte requests in flight.
template <int NACC, int OP, int DIM, int U, typename RT>
__device__ __forceinline__ void group_distance_pre(const F4 (&xs)[DIM / (8 * NACC)], const RT* const (&rows)[U],
                                                   const bool (&active)[U], int v, float (&out)[U]) {
    constexpr int G = 2 * NACC, TRIP = 4 * G, NT = DIM / TRIP;
    Raw4<RT> ys[U][NT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int t = 0; t < NT; ++t) ys[u][t].load(rows[u] + t * TRIP + 4 * v);  // unconditional (see group_distance_many)
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        FAcc<OP> acc;
        acc.init();
#pragma unroll
        for (int t = 0; t < NT; ++t) acc.step(xs[t], ys[u][t].get());
        float s = finish_vec<NACC>(acc.s, [](float(&)[4]) {});
        if (OP == OP_COS) {
            float nx = finish_vec<NACC>(acc.nx, [](float(&)[4]) {});
            float ny = finish_vec<NACC>(acc.ny, [](float(&)[4]) {});
            s = cosine_finish(nx, ny, s);
        }
        out[u] = s;
    }
}


// U rows at once with run-time length: per trip the U row loads are issued together, so a lane
// keeps U (x unroll) requests in flight instead of one.  Same arithmetic as group_distance_raw.
template <int NACC, int OP, int U, typename QT, typename RT>
__device__ __forceinline__ void group_distance_multi(const QT* __restrict__ q, const RT* const (&rows)[U],
                                                     const bool (&active)[U], int dim, int v, float (&out)[U]) {
    constexpr int G = 2 * NACC, TRIP = 4 * G;
    const int full_end = dim & ~7;
    const int rem = dim & 7;
    FAcc<OP> acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u].init();
    // the partial block's elements (dim % 8 != 0) are requested first, raw, so they travel with the main
    // loads instead of costing a second dependent round trip after them (dim = 100: 2.4x)
    RT ty[U][4];
    if (rem) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int l = 4 * (v & 1) + i;
#pragma unroll
            for (int u = 0; u < U; ++u) ty[u][i] = rows[u][full_end + (l < rem ? l : 0)];  // unconditional: no branch, no wait
        }
    }
    // T trips of loads are issued before the first FMA: U*T 16-byte requests in flight per lane
    constexpr int T = 4;
    for (int e0 = 4 * v; e0 < full_end; e0 += TRIP * T) {
        Raw4<RT> ys[T][U];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int e = e0 + t * TRIP;
            const int el = e < full_end ? e : e0;  // clamped: every request is issued, none behind a branch
#pragma unroll
            for (int u = 0; u < U; ++u) ys[t][u].load(rows[u] + el);
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int e = e0 + t * TRIP;
            if (e < full_end) {
                const F4 x = load4(q + e);
#pragma unroll
                for (int u = 0; u < U; ++u) acc[u].step(x, ys[t][u].get());  // inactive rows: computed, discarded
            }
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const bool act = active[u];
        auto partial = [&](float(&a)[4], int which) {
            if (rem == 0 || !act) return;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int l = 4 * (v & 1) + i;
                float x = 0.0f, y = 0.0f;
                if (l < rem) {
                    x = load1(q + full_end + l);
                    y = to_f32(ty[u][i]);
                }
                if (OP == OP_L2) {
                    float c = x - y;
                    a[i] = __builtin_fmaf(c, c, a[i]);
                } else if (which == 0) {
                    a[i] = __builtin_fmaf(x, y, a[i]);
                } else if (which == 1) {
                    a[i] = __builtin_fmaf(x, x, a[i]);
                } else {
                    a[i] = __builtin_fmaf(y, y, a[i]);
                }
            }
        };
        float s = finish_vec<NACC>(acc[u].s, [&](float(&a)[4]) { partial(a, 0); });
        if (OP == OP_COS) {
            float nx = finish_vec<NACC>(acc[u].nx, [&](float(&a)[4]) { partial(a, 1); });
            float ny = finish_vec<NACC>(acc[u].ny, [&](float(&a)[4]) { partial(a, 2); });
            s = cosine_finish(nx, ny, s);
        }
        out[u] = s;
    }
}


// ---- "wide" layout for 2-byte rows: one lane owns a whole 8-lane accumulator --------------------
// With f16 rows a 4-element slice is only 8 bytes; giving each lane 8 consecutive elements
// (one 16-byte load) and one CPU accumulator (lane w of a group of NACC lanes == accumulator w)
// halves the instructions per byte.  Chains stay in-lane, so the association order is unchanged:
// combine (s0+s1)+(s2+s3) across lanes (two DPP adds), partial block, sum_tree in-lane.
struct F8v {
    float e[8];
};
__device__ __forceinline__ F8v load8(const float* p) {
    float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    return {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
}
__device__ __forceinline__ F8v load8(const __half* p) {
    uint4 t = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
    F8v r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float2 f = __half22float2(__builtin_bit_cast(__half2, w[i]));
        r.e[2 * i] = f.x;
        r.e[2 * i + 1] = f.y;
    }
    return r;
}
template <typename RT>
__device__ __forceinline__ F8v cvt8(const uint4& t);
template <>
__device__ __forceinline__ F8v cvt8<__half>(const uint4& t) {
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
    F8v r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float2 f = __half22float2(__builtin_bit_cast(__half2, w[i]));
        r.e[2 * i] = f.x;
        r.e[2 * i + 1] = f.y;
    }
    return r;
}
template <int NACC>
__device__ __forceinline__ float finish_wide(float (&a)[8], const float (&px)[8], const float (&py)[8], int rem,
                                             int op_kind /*0 L2, 1 xy, 2 xx, 3 yy*/) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = a[i] + dpp_f<DPP_XOR1>(a[i]);
    if (NACC == 4) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = a[i] + dpp_f<DPP_XOR2>(a[i]);
    }
    if (rem) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float x = px[i], y = py[i];  // zero beyond `rem`
            if (op_kind == 0) {
                const float c = x - y;
                a[i] = __builtin_fmaf(c, c, a[i]);
            } else if (op_kind == 1) {
                a[i] = __builtin_fmaf(x, y, a[i]);
            } else if (op_kind == 2) {
                a[i] = __builtin_fmaf(x, x, a[i]);
            } else {
                a[i] = __builtin_fmaf(y, y, a[i]);
            }
        }
    }
    return ((a[0] + a[4]) + (a[2] + a[6])) + ((a[1] + a[5]) + (a[3] + a[7]));
}

#ifndef DANN_WIDE_TRIPS
#define DANN_WIDE_TRIPS 4
#endif
constexpr int kWideTrips = DANN_WIDE_TRIPS;
template <int NACC, int OP, int U, typename QT, typename RT>
__device__ __forceinline__ void group_distance_wide(const QT* __restrict__ q, const RT* const (&rows)[U],
                                                    const bool (&active)[U], int dim, int w, float (&out)[U]) {
    constexpr int TRIP = 8 * NACC;
    const int full_end = dim & ~7;
    float s[U][8], nx[U][8], ny[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) s[u][i] = nx[u][i] = ny[u][i] = 0.0f;
    // the partial block (dim % 8 != 0) is requested first, unconditionally (index clamped into the block), so
    // it travels with the main loads instead of costing `rem` dependent round trips after them
    const int rem = dim & 7;
    RT ry[U][8];
    if (rem) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) ry[u][i] = rows[u][full_end + (i < rem ? i : 0)];
    }
    constexpr int T = kWideTrips;
    for (int e0 = 8 * w; e0 < full_end; e0 += TRIP * T) {
        // All T*U 16-byte requests of a trip are issued back to back, unconditionally: a load under a per-lane
        // condition is compiled as a branch with its own s_waitcnt vmcnt(0), which leaves ONE request in
        // flight per wave (measured: 1 M x 768 f16 at 3.65 TB/s, the same time as the f32 rows).  Out-of-range
        // trips re-read the trip's first block and inactive rows point at row 0 (the caller's contract), so
        // every address is valid; their values are never used.
        uint4 raw[T][U];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int e = e0 + t * TRIP;
            const int el = e < full_end ? e : e0;
#pragma unroll
            for (int u = 0; u < U; ++u) raw[t][u] = *reinterpret_cast<const uint4*>(rows[u] + el);
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int e = e0 + t * TRIP;
            if (e >= full_end) continue;
            const F8v x = load8(q + e);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const F8v y = cvt8<RT>(raw[t][u]);  // inactive rows: computed and discarded
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (OP == OP_L2) {
                        const float c = x.e[i] - y.e[i];
                        s[u][i] = __builtin_fmaf(c, c, s[u][i]);
                    } else if (OP == OP_IP) {
                        s[u][i] = __builtin_fmaf(x.e[i], y.e[i], s[u][i]);
                    } else {
                        nx[u][i] = __builtin_fmaf(x.e[i], x.e[i], nx[u][i]);
                        ny[u][i] = __builtin_fmaf(y.e[i], y.e[i], ny[u][i]);
                        s[u][i] = __builtin_fmaf(x.e[i], y.e[i], s[u][i]);
                    }
                }
            }
        }
    }
    float px[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) px[i] = (rem && i < rem) ? load1(q + full_end + i) : 0.0f;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        float py[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) py[i] = (rem && i < rem) ? to_f32(ry[u][i]) : 0.0f;
        float r = finish_wide<NACC>(s[u], px, py, rem, OP == OP_L2 ? 0 : 1);
        if (OP == OP_COS) {
            float a = finish_wide<NACC>(nx[u], px, py, rem, 2);
            float b = finish_wide<NACC>(ny[u], px, py, rem, 3);
            r = cosine_finish(a, b, r);
        }
        out[u] = r;
    }
}

// ---- integer rows (u8 / i8): exact i32 accumulation, any order is bit-identical
// (simd.rs:1192-1225, 1947-1979, 2109-2143, 2790-2827, 2996-3033).  Group of 8 lanes,
// 16 bytes per lane per step, v_dot4 accumulate.  Valid in lane v == 0.
template <bool SIGNED>
__device__ __forceinline__ int dot4(uint32_t a, uint32_t b, int c) {
    if (SIGNED) return __builtin_amdgcn_sdot4((int)a, (int)b, c, false);
    return (int)__builtin_amdgcn_udot4(a, b, (uint32_t)c, false);
}
template <bool SIGNED>
__device__ __forceinline__ int elem(const uint8_t* p) {
    return SIGNED ? (int)(int8_t)*p : (int)*p;
}
__device__ __forceinline__ int group8_sum(int x) {
    x += dpp_i<DPP_XOR1>(x);
    x += dpp_i<DPP_XOR2>(x);
    x += dpp_i<DPP_SHL4>(x);
    return x;
}
template <int OP, bool SIGNED>
__device__ __forceinline__ float group_distance_int(const uint8_t* __restrict__ q, const uint8_t* __restrict__ row,
                                                    int dim, int v) {
    int xx = 0, yy = 0, xy = 0;
    const int vec_end = dim & ~15;
    for (int e = 16 * v; e < vec_end; e += 128) {
        uint4 x = *reinterpret_cast<const uint4*>(q + e);
        uint4 y = *reinterpret_cast<const uint4*>(row + e);
        const uint32_t xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xy = dot4<SIGNED>(xs[i], ys[i], xy);
            if (OP != OP_IP) {
                xx = dot4<SIGNED>(xs[i], xs[i], xx);
                yy = dot4<SIGNED>(ys[i], ys[i], yy);
            }
        }
    }
    for (int e = vec_end + v; e < dim; e += 8) {
        int a = elem<SIGNED>(q + e), b = elem<SIGNED>(row + e);
        xy += a * b;
        xx += a * a;
        yy += b * b;
    }
    xy = group8_sum(xy);
    if (OP == OP_IP) return (float)xy;
    xx = group8_sum(xx);
    yy = group8_sum(yy);
    if (OP == OP_L2) return (float)(int)((uint32_t)xx + (uint32_t)yy - 2u * (uint32_t)xy);
    return cosine_finish((float)xx, (float)yy, (float)xy);
}


template <int OP, bool SIGNED, int U>
__device__ __forceinline__ void group_distance_int_multi(const uint8_t* __restrict__ q,
                                                         const uint8_t* const (&rows)[U], const bool (&active)[U],
                                                         int dim, int v, float (&out)[U]) {
    int xx[U], yy[U], xy[U];
#pragma unroll
    for (int u = 0; u < U; ++u) xx[u] = yy[u] = xy[u] = 0;
    const int vec_end = dim & ~15;
    for (int e = 16 * v; e < vec_end; e += 128) {
        uint4 ys[U];
#pragma unroll
        for (int u = 0; u < U; ++u) ys[u] = *reinterpret_cast<const uint4*>(rows[u] + e);  // unconditional
        const uint4 x = *reinterpret_cast<const uint4*>(q + e);
        const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t yw[4] = {ys[u].x, ys[u].y, ys[u].z, ys[u].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xy[u] = dot4<SIGNED>(xs[i], yw[i], xy[u]);
                if (OP != OP_IP) {
                    xx[u] = dot4<SIGNED>(xs[i], xs[i], xx[u]);
                    yy[u] = dot4<SIGNED>(yw[i], yw[i], yy[u]);
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (active[u]) {
            for (int e = vec_end + v; e < dim; e += 8) {
                int a = elem<SIGNED>(q + e), b = elem<SIGNED>(rows[u] + e);
                xy[u] += a * b;
                xx[u] += a * a;
                yy[u] += b * b;
            }
        }
        int sxy = group8_sum(xy[u]);
        if (OP == OP_IP) {
            out[u] = (float)sxy;
            continue;
        }
        int sxx = group8_sum(xx[u]), syy = group8_sum(yy[u]);
        if (OP == OP_L2) out[u] = (float)(int)((uint32_t)sxx + (uint32_t)syy - 2u * (uint32_t)sxy);
        else out[u] = cosine_finish((float)sxx, (float)syy, (float)sxy);
    }
}

// Fixed 128-byte integer rows: the lane's 16 query bytes (`x`) and the query's squared norm (`xx`, the same for
// every row, summed once per search) stay in registers; per row only xy and yy are accumulated.  Integer sums:
// any order is bit-identical.  Valid in lane v == 0; rows[] must all be readable (see group_distance_many).
template <int OP, bool SIGNED, int U>
__device__ __forceinline__ void group_distance_int_pre(const uint4& x, int xx, const uint8_t* const (&rows)[U], int v,
                                                       float (&out)[U]) {
    uint4 ys[U];
#pragma unroll
    for (int u = 0; u < U; ++u) ys[u] = *reinterpret_cast<const uint4*>(rows[u] + 16 * v);
    const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t yw[4] = {ys[u].x, ys[u].y, ys[u].z, ys[u].w};
        int xy = 0, yy = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xy = dot4<SIGNED>(xs[i], yw[i], xy);
            if (OP != OP_IP) yy = dot4<SIGNED>(yw[i], yw[i], yy);
        }
        if (OP == OP_L2) {
            // |x - y|^2 = xx + sum(yy - 2 xy): one cross-lane sum of the lane's yy - 2 xy instead of two (integer sums modulo
            // 2^32: any order and grouping is bit-identical)
            const int st = group8_sum((int)((uint32_t)yy - 2u * (uint32_t)xy));
            out[u] = (float)(int)((uint32_t)xx + (uint32_t)st);
            continue;
        }
        const int sxy = group8_sum(xy);
        if (OP == OP_IP) {
            out[u] = (float)sxy;
            continue;
        }
        const int syy = group8_sum(yy);
        out[u] = cosine_finish((float)xx, (float)syy, (float)sxy);
    }
}
template <bool SIGNED>
__device__ __forceinline__ int group_norm_int_pre(const uint4& x) {
    int xx = dot4<SIGNED>(x.x, x.x, 0);
    xx = dot4<SIGNED>(x.y, x.y, xx);
    xx = dot4<SIGNED>(x.z, x.z, xx);
    xx = dot4<SIGNED>(x.w, x.w, xx);
    return group8_sum(xx);
}

// ---- dtype dispatch ------------------------------------------------------------------
// Query-side staging type and group width for the *search* path
// (Full<T>::query_distance, diskann-inmem/src/layers/full.rs:351-504):
//   f32 rows: f32 query, L2/IP Strategy4x1 (NACC 4), cosine Strategy2x4 (NACC 2)
//   f16 rows: query widened to f32 once, L2/IP Strategy4x2 (== NACC 4), cosine NACC 2
//   u8/i8  : integer query, exact
// and for the *pair* path (DistanceProvider::distance_comparer, V3):
//   f16 x f16: L2/IP/cosine all Strategy2x4 (NACC 2)  (simd.rs:989,1752,2591)
template <int DT, int OP, bool PAIR>
struct Scheme {
    static constexpr bool kInt = (DT == DT_U8 || DT == DT_I8 || DT == DT_SQ8 || DT == DT_PQ);
    static constexpr int NACC = (OP == OP_COS) ? 2 : ((DT == DT_F16 && PAIR) ? 2 : 4);
    static constexpr int G = kInt ? 8 : 2 * NACC;
    // search-path gather: 2-byte rows use the wide layout (one lane per accumulator)
    static constexpr bool kWide = (DT == DT_F16) && !PAIR;
    static constexpr int GS = kWide ? NACC : G;
};

template <int DT>
struct RowType {
    using type = float;
};
template <>
struct RowType<DT_F16> {
    using type = __half;
};
template <>
struct RowType<DT_U8> {
    using type = uint8_t;
};
template <>
struct RowType<DT_I8> {
    using type = uint8_t;
};
template <>
struct RowType<DT_SQ8> {
    using type = uint8_t;
};
template <>
struct RowType<DT_PQ> {
    using type = uint8_t;
};

// `q` is the staged query: f32 for float rows, raw bytes for integer rows.
template <int DT, int OP, bool PAIR, int DIM, typename QT>
__device__ __forceinline__ float group_distance(const QT* q, const uint8_t* row, int dim, int v) {
    if constexpr (DT == DT_U8 || DT == DT_I8 || DT == DT_SQ8) {
        return group_distance_int<OP, DT == DT_I8>(reinterpret_cast<const uint8_t*>(q), row, dim, v);
    } else {
        using RT = typename RowType<DT>::type;
        return group_distance_raw<Scheme<DT, OP, PAIR>::NACC, OP, DIM>(q, reinterpret_cast<const RT*>(row), dim, v);
    }
}


// stored row x stored row with the prune-path association (Scheme<DT, OP, true>)
template <int DT, int OP>
__device__ __forceinline__ float group_distance_rows(const uint8_t* x, const uint8_t* y, int dim, int v) {
    if constexpr (DT == DT_U8 || DT == DT_I8 || DT == DT_SQ8) {
        return group_distance_int<OP, DT == DT_I8>(x, y, dim, v);
    } else {
        using RT = typename RowType<DT>::type;
        return group_distance_pair<Scheme<DT, OP, true>::NACC, OP, RT>(reinterpret_cast<const RT*>(x),
                                                                       reinterpret_cast<const RT*>(y), dim, v);
    }
}

// U rows per lane group at once.  Contract: rows[u] is a readable row address even when !active[u] (callers pass
// row 0): the loads are issued unconditionally so that none of them sits behind a branch with its own wait.
// WIDE = false keeps the G-lane layout for 2-byte rows (kernels whose lane groups are Scheme::G wide).
template <int DT, int OP, bool PAIR, int U, bool WIDE = true, typename QT>
__device__ __forceinline__ void group_distance_many(const QT* q, const uint8_t* const (&rows)[U],
                                                    const bool (&active)[U], int dim, int v, float (&out)[U]) {
    if constexpr (DT == DT_U8 || DT == DT_I8 || DT == DT_SQ8) {
        group_distance_int_multi<OP, DT == DT_I8, U>(reinterpret_cast<const uint8_t*>(q), rows, active, dim, v, out);
    } else {
        using RT = typename RowType<DT>::type;
        const RT* typed[U];
#pragma unroll
        for (int u = 0; u < U; ++u) typed[u] = reinterpret_cast<const RT*>(rows[u]);
        if constexpr (Scheme<DT, OP, PAIR>::kWide && WIDE)
            group_distance_wide<Scheme<DT, OP, PAIR>::NACC, OP, U>(q, typed, active, dim, v, out);
        else
            group_distance_multi<Scheme<DT, OP, PAIR>::NACC, OP, U>(q, typed, active, dim, v, out);
    }
}

// simd_op for f32 x f32, Strategy4x1, V3 (simd.rs:321-363, 686-747); IS_L2 ? L2 : IP
template <bool IS_L2>
__device__ float simd_op_seq(const float* x, const float* y, uint32_t len) {
    float acc[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int l = 0; l < 8; ++l) acc[a][l] = 0.0f;
    const uint32_t blocks = len / 8;
    for (uint32_t g = 0; g < blocks; ++g) {
        const int a = g & 3;
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            const float xv = x[8 * g + l], yv = y[8 * g + l];
#pragma unroll
            for (int aa = 0; aa < 4; ++aa) {
                if (aa == a) {
                    if (IS_L2) {
                        const float c = xv - yv;
                        acc[aa][l] = __builtin_fmaf(c, c, acc[aa][l]);
                    } else {
                        acc[aa][l] = __builtin_fmaf(xv, yv, acc[aa][l]);
                    }
                }
            }
        }
    }
    float s[8];
#pragma unroll
    for (int l = 0; l < 8; ++l) s[l] = (acc[0][l] + acc[1][l]) + (acc[2][l] + acc[3][l]);
    const uint32_t rem = len & 7u;
    if (rem) {
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            const float xv = (uint32_t)l < rem ? x[8 * blocks + l] : 0.0f;
            const float yv = (uint32_t)l < rem ? y[8 * blocks + l] : 0.0f;
            if (IS_L2) {
                const float c = xv - yv;
                s[l] = __builtin_fmaf(c, c, s[l]);
            } else {
                s[l] = __builtin_fmaf(xv, yv, s[l]);
            }
        }
    }
    return ((s[0] + s[4]) + (s[2] + s[6])) + ((s[1] + s[5]) + (s[3] + s[7]));
}


// scalar-quantiser parameters of an SQ-8 index
struct SqParams {
    float k;              // (1/255)^2 * scale^2, evaluated in f32 in the reference's order
    float shift_norm_sq;  // ||shift||^2
};

__device__ __forceinline__ float load_f32_unaligned(const uint8_t* p) {
    uint32_t u = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    return __builtin_bit_cast(float, u);
}

// raw kernel result -> SimilarityScore.  Full-precision rows: PostOp; SQ-8 rows: the compensated
// epilogues of diskann-quantization/src/scalar/vectors.rs:216-245 (L2), :306-370 (IP),
// :403-465 (CosineNormalized); `x`, `y` are the two rows (code bytes + trailing f32 compensation).
template <int DT, int OP, bool NORM>
__device__ __forceinline__ float finish_distance(float raw, const uint8_t* x, const uint8_t* y, uint32_t dim,
                                                 const SqParams& sq) {
    if constexpr (DT != DT_SQ8) {
        return post_op<OP, NORM>(raw);
    } else if constexpr (OP == OP_L2) {
        const float l2 = sq.k * raw;
        if (!NORM) return l2;
        const float sim = 1.0f - l2 / 2.0f;
        return 1.0f - sim;
    } else {
        const float cx = load_f32_unaligned(x + dim), cy = load_f32_unaligned(y + dim);
        const float r = __builtin_fmaf(sq.k, raw, sq.shift_norm_sq) + (cy + cx);
        return -r;
    }
}

// (dtype, metric) -> (OP, NORMALIZED).  Integers treat CosineNormalized as Cosine
// (distance_provider.rs:274-297, full.rs:470,499); SQ-8 CosineNormalized is L2-based.
// Returns false for unsupported combinations (SQ-8 has no plain Cosine).
__host__ __device__ inline bool resolve_metric(int dtype, int metric, int* op, bool* norm) {
    *norm = false;
    if (dtype == DT_PQ) {
        if (metric == M_L2) { *op = OP_L2; return true; }
        if (metric == M_IP) { *op = OP_IP; return true; }
        return false;
    }
    if (dtype == DT_SQ8) {
        if (metric == M_L2) { *op = OP_L2; return true; }
        if (metric == M_IP) { *op = OP_IP; return true; }
        if (metric == M_COSN) { *op = OP_L2; *norm = true; return true; }
        return false;
    }
    if (metric == M_L2) { *op = OP_L2; return true; }
    if (metric == M_IP) { *op = OP_IP; return true; }
    if (metric == M_COSINE) { *op = OP_COS; return true; }
    if (dtype == DT_U8 || dtype == DT_I8) { *op = OP_COS; return true; }
    *op = OP_IP;
    *norm = true;  // CosineNormalized on float rows = 1 - <x, y>
    return true;
}

}  // namespace dann
