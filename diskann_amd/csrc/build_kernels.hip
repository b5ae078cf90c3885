// build_kernels.hip -- RobustPrune and batched Vamana insert on the GPU.
//
// Replaces, for a batch of new points, the reference call chain
//   DiskANNIndex::multi_insert                 diskann/src/graph/index.rs:815-1030
//     search_and_prune (beam 1, L = l_build)   index.rs:349-434  (search: search_kernels.hip)
//     robust_prune_with                        index.rs:2476-2532
//       SortedNeighbors::new                   graph/internal/sorted_neighbors.rs:26-44
//       occlude_list                           index.rs:2565-2650
//       prune::robust_prune                    graph/internal/prune.rs:106-259
//       PruneKind::update_occlude_factor       graph/config/mod.rs:80-103
//     aggregate_backedges                      index.rs:123-143
//     multi_insert_bootstrap_leaf              index.rs:597-645
//     set_neighbors_bulk                       index.rs:948-962
//     add_edge_and_prune / robust_prune_list   index.rs:2264-2341, 2397-2454
// with the adjacency lists it produces identical to the CPU restatement -- and, by default, to the
// reference itself on pools with EQUAL distances as well: such a pool is put in the order the reference's
// own select_nth_unstable_by + sort_unstable_by leave it in (rust_order.h, dann_set_prune_tie_order).
//
// One wavefront owns one point.  The candidate pool is sorted in LDS (bitonic, 64-bit
// keys = order-preserving distance bits : pool position; a pool whose sorted keys show equal
// distances is then walked by one lane, rust_order_by_lane0).  The alpha sweep keeps 64/G
// candidates in flight, one per G-lane distance group (the same bit-exact groups as the
// search path): each walks the selected list in the reference's order, is rejected at the
// first entry that pushes its factor past alpha, and is selected only once it is the oldest
// candidate in flight and has seen the whole list -- so `last_checked`, and with it the
// inner-product "Occluding" rule that depends on *when* a pair is evaluated, are reproduced
// exactly, and no distance is evaluated that the sequential sweep would not evaluate
// (prune_sorted_pool; rounds 1-4 visited one candidate at a time with 64/G speculative
// distances per round and were bound by the scalar unit: profiles/r05_prune_counters.txt).
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <mutex>
#include <vector>

#include "dann_device.h"
#include "dann_internal.h"
#include "rust_order.h"

namespace dann {
namespace {

constexpr int kWave = 64;
constexpr uint32_t kMaxPool = 4096;

struct PruneCfg {
    uint32_t pruned_degree, max_degree, max_occlusion;
    float alpha;
    uint32_t saturate_after_prune;
    uint32_t tie_order;  // dann_set_prune_tie_order: DANN_TIE_POSITION (bitonic sort on (distance, pool position)) or
                         // DANN_TIE_RUST (rust_order.h, one lane; the bootstrap's candidates in ascending id order)
    // optional device counters (u64): [0] pair distances of the sweeps (row kernel), [1] list / extra distances
    // d(location, c), [2] rows that went through an MFMA Gram, [3] Gram entries computed (x dim x 2 = MFMA flop),
    // [4] pair distances the lazy scans of the Gram sweeps asked for, [5] those answered by an exact re-evaluation,
    // [6] back-edge prunes through the MFMA path, [7] back-edge prunes of lists too long for it.
    // kStatStripes copies of the eight, one 64-byte line each, a workgroup adds to the copy of its index (stat_stripe):
    // every workgroup of a prune launch ends with a few of these adds, and atomics on ONE address are served one after
    // another -- 354 k of them per launch were 3 ms of backedge_scan_kernel's 4.4 ms at 10 M points
    // (profiles/r05_scan_atomics.txt).  dann_build_counters sums the copies.  One more word behind the stripes
    // (counters[kStatStripes * 8]): tied pools whose DANN_TIE_RUST walk reached the selection's fallback (rust_order.h).
    unsigned long long* counters;
};
constexpr uint32_t kStatStripes = 256;
__device__ __forceinline__ unsigned long long* stat_stripe(unsigned long long* counters) {
    return counters + (size_t)(blockIdx.x & (kStatStripes - 1u)) * 8u;
}

struct PoolLds {
    uint32_t keys_off, pid_off, pd_off, sid_off, sd_off, occ_off, last_off, sel_off, total;
};

__host__ __device__ inline PoolLds pool_lds_layout(uint32_t pcap, uint32_t degree) {
    PoolLds l;
    uint32_t off = 0;
    l.keys_off = off;
    off += pcap * 8u;
    l.pid_off = off;
    off += pcap * 4u;
    l.pd_off = off;
    off += pcap * 4u;
    l.sid_off = off;
    off += pcap * 4u;
    l.sd_off = off;
    off += pcap * 4u;
    l.occ_off = off;
    off += pcap * 4u;
    l.last_off = off;
    off += ((pcap * 2u) + 15u) & ~15u;
    l.sel_off = off;
    off += ((degree + 1u) * 4u + 15u) & ~15u;
    l.total = off;
    return l;
}

// What pool_sweep_kernel's batched sweep keeps in LDS: the sorted list (ids, distances), its sweep state and the norms of the
// Gram rows -- not the unsorted pool and the sort keys of the kernels that build the list (3 KB of 7.9 KB at 256 slots: with
// them the sweep ran 12 lists per CU, without them 16).  Same field names, so sweep_gram_batched reads either layout.
__host__ __device__ inline PoolLds sweep_lds_layout(uint32_t pcap, uint32_t degree) {
    PoolLds l;
    uint32_t off = 0;
    l.keys_off = off;  // the norms of the Gram rows (at most 256 rows)
    off += (pcap < 256u ? pcap : 256u) * 4u;
    l.pid_off = l.pd_off = 0;  // (not part of this layout)
    l.sid_off = off;
    off += pcap * 4u;
    l.sd_off = off;
    off += pcap * 4u;
    l.occ_off = off;
    off += pcap * 4u;
    l.last_off = off;
    off += ((pcap * 2u) + 15u) & ~15u;
    l.sel_off = off;
    off += ((degree + 1u) * 4u + 15u) & ~15u;
    l.total = off;
    return l;
}

__device__ __forceinline__ uint64_t sort_key(float d, uint32_t pos) {
    uint32_t u = __builtin_bit_cast(uint32_t, d + 0.0f);  // -0.0 -> +0.0: `<` treats them as equal
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((uint64_t)u << 32) | pos;
}

// PruneKind::update_occlude_factor (config/mod.rs:80-103)
template <int OP>
__device__ __forceinline__ float update_occlude(float d_ik, float d_jk, float cur, float alpha, bool occluding) {
    if (!occluding) {
        if (d_jk == 0.0f) return 3.402823466e+38f;
        float r = d_ik / d_jk;
        if (r != r) return cur;  // f32::max ignores NaN
        if (cur != cur) return r;
        return cur > r ? cur : r;
    }
    if (d_jk < alpha * d_ik) return alpha + 0.01f;
    return cur;
}

// DANN_TIE_RUST: the pool order of SortedNeighbors::new as the reference's own sort leaves it (rust_order.h).  Lane 0 walks
// the algorithm over the pool positions, kept in the `last` array (16 bits per slot, zeroed by the caller's final loop
// after it has read them); stacks and merge buffer in the region of the sort keys, whose contents (the network's output)
// are dead once the walk is decided.  Pools of fewer than 64 slots have less than kWorkBytes there -- and need none of it
// beyond the 64-byte merge buffer (quicksort starts at 33 entries).  The caller synchronises the wave afterwards.
//
// The walk is only needed where it can matter: a pool whose distances are all distinct (and ordered: no NaN) has one
// sorted order, and select_nth_unstable_by + sort_unstable_by + truncate leave exactly its head -- what the sorting
// network has already produced.  pool_has_ties looks at neighbouring keys of the network's output (wave-uniform result);
// on continuous data nearly every pool takes the network alone, on byte data and lattices the tied pools are walked.
__device__ __forceinline__ bool pool_has_ties(const uint64_t* keys, uint32_t P, bool lane_saw_nan) {
    const uint32_t lane = threadIdx.x & 63u;
    bool tie = lane_saw_nan;
    for (uint32_t i = lane; i + 1 < P; i += kWave) tie |= (uint32_t)(keys[i] >> 32) == (uint32_t)(keys[i + 1] >> 32);
    return ballot64(tie) != 0;
}
__device__ __forceinline__ void rust_order_by_lane0(const PruneCfg& cfg, uint32_t P, uint8_t* smem, const PoolLds& L) {
    uint16_t* ord = reinterpret_cast<uint16_t*>(smem + L.last_off);
    const float* pd = reinterpret_cast<const float*>(smem + L.pd_off);
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t i = lane; i < P; i += kWave) ord[i] = (uint16_t)i;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // (the selection's median-of-medians fallback -- sixteen unlucky partitions in a row -- is restated as a sort of the
    // range, which may leave equal keys in another arrangement than core's: counted, so that builds and tests can assert
    // it never happened: dann_build_counters()[10])
    if (lane == 0 && rust_order::sorted_neighbors(ord, pd, P, cfg.max_occlusion, smem + L.keys_off) && cfg.counters)
        atomicAdd(cfg.counters + (size_t)kStatStripes * 8u, 1ull);
}

// Sort pid/pd[0..P) (already in LDS) by (distance, position), truncate to max_occlusion,
// run occlude_list for `location` and write [len, ids...] to `out`.
template <int DT, int OP, bool NORM>
__device__ void prune_sorted_pool(const IndexView& ix, const PruneCfg& cfg, uint32_t location, uint32_t P,
                                  uint32_t pcap, uint8_t* smem, const PoolLds& L, bool force_saturate, uint32_t* out) {
    using S = Scheme<DT, OP, true>;
    constexpr int G = S::G;
    const uint32_t lane = threadIdx.x;
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem + L.keys_off);
    const uint32_t* pid = reinterpret_cast<const uint32_t*>(smem + L.pid_off);
    const float* pd = reinterpret_cast<const float*>(smem + L.pd_off);
    uint32_t* sid = reinterpret_cast<uint32_t*>(smem + L.sid_off);
    float* sd = reinterpret_cast<float*>(smem + L.sd_off);
    float* occ = reinterpret_cast<float*>(smem + L.occ_off);
    uint16_t* last = reinterpret_cast<uint16_t*>(smem + L.last_off);
    uint32_t* sel = reinterpret_cast<uint32_t*>(smem + L.sel_off);

    // ---- SortedNeighbors::new -------------------------------------------------------
    bool saw_nan = false;
    for (uint32_t i = lane; i < pcap; i += kWave) {
        const float d = i < P ? pd[i] : 0.0f;
        saw_nan |= d != d;
        keys[i] = i < P ? sort_key(d, i) : ~0ull;
    }
    __syncthreads();
    for (uint32_t k = 2; k <= pcap; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = lane; t < (pcap >> 1); t += kWave) {
                const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u));
                const uint32_t p = i | j;
                const uint64_t a = keys[i], b = keys[p];
                const bool up = (i & k) == 0;
                if ((a > b) == up) {
                    keys[i] = b;
                    keys[p] = a;
                }
            }
            __syncthreads();
        }
    }
    // DANN_TIE_RUST: a pool with equal (or unordered) distances is ordered by the reference's own sort instead
    const bool rust = cfg.tie_order != 0 && pool_has_ties(keys, P, saw_nan);
    if (rust) {
        __syncthreads();
        rust_order_by_lane0(cfg, P, smem, L);
        __syncthreads();
    }
    const uint32_t N = P < cfg.max_occlusion ? P : cfg.max_occlusion;
    for (uint32_t i = lane; i < N; i += kWave) {
        const uint32_t pos = rust ? (uint32_t)last[i] : (uint32_t)keys[i];
        const uint32_t id = pid[pos];
        sid[i] = id;
        sd[i] = pd[pos];
        occ[i] = (id == location || id >= ix.nslots) ? 3.402823466e+38f : 0.0f;  // excluded / not retrievable: never visited
        last[i] = 0;
    }
    __syncthreads();

    // ---- prune::robust_prune ----------------------------------------------------------
    const uint32_t degree = cfg.pruned_degree;
    const bool occluding = (ix.metric == M_IP);  // PruneKind::from_metric (config/mod.rs:69-75)
    const float alpha = cfg.alpha;
    const float inc = alpha < 1.2f ? alpha : 1.2f;
    float cur_alpha = 1.0f;
    uint32_t found = 0, npairs = 0;
    const int g = lane / G, v = lane % G;
    // The reference visits the sorted candidates one after another, each against the entries selected so far
    // (prune.rs:196-232).  Here GROUPS of them are in flight at once, one per lane group, each walking sel[] on its
    // own, one entry per step of the wave:
    //   * candidates are handed out in list order (the pass's worklist: those whose factor does not exceed cur_alpha);
    //   * a candidate is REJECTED as soon as its running factor exceeds cur_alpha -- against a prefix of sel[], which is
    //     append-only, so it is the prefix the sequential visit would have walked;
    //   * a candidate that has seen all of sel[] is SELECTED only when it is the oldest one in flight: every earlier
    //     candidate is resolved by then, so sel[] is what the sequential visit would have found, and later candidates
    //     in flight meet the new entry at its place in the order (they are at or before the old end of sel[]).
    // Same factors (each group applies update_occlude in sel[] order), same last_checked, same sel[]; no distance is
    // evaluated speculatively, and the per-candidate scalar control flow of the one-at-a-time form (45 k scalar
    // instructions per prune, one per CU-cycle: profiles/r05_prune_counters.txt) is paid once per GROUPS candidates.
    uint16_t* work = reinterpret_cast<uint16_t*>(smem + L.keys_off);  // the sort keys are dead by now
    const uint32_t gbit = (uint32_t)(g * G);
    if (N > 0) {
        while (found < degree) {
            uint32_t M = 0;
            for (uint32_t base = 0; base < N; base += kWave) {
                const uint32_t i = base + lane;
                const bool act = i < N && !(occ[i] > cur_alpha);
                const uint64_t m = ballot64(act);
                if (act) work[M + mbcnt(m)] = (uint16_t)i;
                M += (uint32_t)__builtin_popcountll(m);
            }
            __syncthreads();
            uint32_t next = 0;
            bool busy = false;
            uint32_t c = 0, w = 0, l = 0;
            float o = 0.0f, di = 0.0f;
            const uint8_t* xi = ix.rows;
            for (;;) {
                // idle groups take the next candidates of the worklist, in group order
                const uint64_t im = ballot64(!busy && v == 0);
                if (im != 0 && next < M) {
                    const uint32_t wi = next + (uint32_t)__builtin_popcountll(im & ((1ull << gbit) - 1ull));
                    if (!busy && wi < M) {
                        w = wi;
                        c = work[wi];
                        l = last[c];
                        o = occ[c];
                        di = sd[c];
                        xi = ix.rows + (uint64_t)sid[c] * ix.row_stride;
                        busy = true;
                    }
                    const uint32_t nidle = (uint32_t)__builtin_popcountll(im);
                    next = next + nidle < M ? next + nidle : M;
                }
                if (ballot64(busy) == 0) break;  // worklist done, nothing in flight
                uint32_t wmin = busy ? w : 0xFFFFFFFFu;  // the oldest candidate in flight
#pragma unroll
                for (int off = 32; off >= G; off >>= 1) {
                    const uint32_t t = (uint32_t)__shfl_xor((int)wmin, off);
                    wmin = t < wmin ? t : wmin;
                }
                bool rej = false, evaluated = false;
                if (busy && l < found) {
                    const uint32_t rp = sel[l];
                    ++l;
                    if (rp < c) {
                        const uint8_t* y = ix.rows + (uint64_t)sid[rp] * ix.row_stride;
                        const float d = finish_distance<DT, OP, NORM>(group_distance_rows<DT, OP>(xi, y, (int)ix.dim, v), xi, y,
                                                                      ix.dim, SqParams{ix.sq_k, ix.sq_shift_norm_sq});
                        o = update_occlude<OP>(di, d, o, cur_alpha, occluding);  // (meaningful in the group's lane 0)
                        rej = o > cur_alpha;
                        evaluated = v == 0;
                    }
                }
                npairs += (uint32_t)__builtin_popcountll(ballot64(evaluated));
                const bool grej = (ballot64(rej && v == 0) >> gbit) & 1ull;
                const bool ready = busy && !grej && l == found && w == wmin;  // at most one group
                const bool commit = ballot64(ready) != 0;
                if (grej || ready) {
                    if (v == 0) {
                        last[c] = (uint16_t)l;
                        occ[c] = grej ? o : 3.402823466e+38f;
                        if (ready) sel[found] = c;
                    }
                    busy = false;
                }
                if (commit) ++found;
                __syncthreads();
                if (found >= degree) break;
            }
            if (found >= degree) break;
            if (cur_alpha == alpha) break;
            const float next_alpha = cur_alpha * inc;
            cur_alpha = next_alpha < alpha ? next_alpha : alpha;
        }
    }
    // ---- neighbours + optional saturation (index.rs:2626-2649) ---------------------------
    __syncthreads();
    uint32_t nout = found;
    if (force_saturate || (cfg.saturate_after_prune && alpha > 1.0f)) {
        // sequential in pool order; AdjacencyList::push filters duplicates
        for (uint32_t i = 0; i < N && nout < degree; ++i) {
            const uint32_t id = sid[i];
            if (id == location) continue;
            bool dup = false;
            for (uint32_t n = lane; n < nout; n += kWave) dup |= (sid[sel[n]] == id);
            if (ballot64(dup)) continue;
            if (lane == 0) sel[nout] = i;
            ++nout;
            __syncthreads();
        }
    }
    __syncthreads();
    for (uint32_t n = lane; n < nout; n += kWave) out[1 + n] = sid[sel[n]];
    if (lane == 0) {
        out[0] = nout;
        if (cfg.counters) atomicAdd(&stat_stripe(cfg.counters)[0], (unsigned long long)npairs);
    }
}

// ---- kernel A: prune caller-provided pools (phase 2 of multi_insert, dann_prune_batch) -----
struct PoolArgs {
    IndexView ix;
    PruneCfg cfg;
    const uint32_t* locs;      // n locations
    const uint32_t* pool_ids;  // pools
    const float* pool_d;
    const uint64_t* offsets;   // n+1 (ragged) or null
    uint32_t stride;           // when offsets == null: pool i at i*stride, count counts[i]
    const uint32_t* counts;
    uint32_t cand;             // intra-batch candidates per item (0 = none); batch = locs[0..n)
    uint32_t n;
    uint32_t pos0;             // batch position of work item 0 (a rank may own a slice of the batch)
    uint32_t pcap;
    int32_t force_saturate;
    uint32_t* out;             // n x out_stride
    uint32_t out_stride;
    uint32_t* err;             // set to 1 on pool overflow
};

template <int DT, int OP, bool NORM>
__global__ __launch_bounds__(kWave) void pool_prune_kernel(PoolArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    using S = Scheme<DT, OP, true>;
    using RT = typename RowType<DT>::type;
    constexpr int G = S::G, GROUPS = kWave / G;
    const uint32_t lane = threadIdx.x, wi = blockIdx.x, item = a.pos0 + blockIdx.x;
    const PoolLds L = pool_lds_layout(a.pcap, a.cfg.pruned_degree);
    uint32_t* pid = reinterpret_cast<uint32_t*>(smem + L.pid_off);
    float* pd = reinterpret_cast<float*>(smem + L.pd_off);
    const uint32_t loc = a.locs[item];
    uint64_t lo;
    uint32_t cnt;
    if (a.offsets) {
        lo = a.offsets[wi];
        cnt = (uint32_t)(a.offsets[wi + 1] - lo);
    } else {
        lo = (uint64_t)wi * a.stride;
        cnt = a.counts[wi];
    }
    // extras = around(ids, position, cand) (utils/async_tools.rs:51-131)
    uint32_t nex = 0;
    if (a.cand != 0 && a.n > 1) nex = a.cand < a.n - 1 ? a.cand : a.n - 1;
    uint32_t* out = a.out + (uint64_t)wi * a.out_stride;
    if (cnt + nex > a.pcap) {
        if (lane == 0) {
            *a.err = 1;
            out[0] = 0;
        }
        return;
    }
    for (uint32_t i = lane; i < cnt; i += kWave) {
        pid[i] = a.pool_ids[lo + i];
        pd[i] = a.pool_d[lo + i];
    }
    if (nex) {
        const uint32_t half = (nex + 1) / 2;
        const uint32_t start = item >= half ? item - half : a.n - (half - item);
        const RT* x = reinterpret_cast<const RT*>(a.ix.rows + (uint64_t)loc * a.ix.row_stride);
        const int g = lane / G, v = lane % G;
        for (uint32_t r0 = 0; r0 < nex; r0 += GROUPS) {
            const uint32_t r = r0 + g;
            if (r < nex) {
                // r-th yielded position: walk from `start`, skipping `item`
                uint32_t p = start + r;
                // positions wrap; the skipped element shifts everything after it by one
                const uint32_t dist_to_item = item >= start ? item - start : item + a.n - start;
                if (r >= dist_to_item) p += 1;
                p %= a.n;
                const uint32_t id = a.locs[p];
                const uint8_t* y = a.ix.rows + (uint64_t)id * a.ix.row_stride;
                float d = finish_distance<DT, OP, NORM>(group_distance_rows<DT, OP>(reinterpret_cast<const uint8_t*>(x), y, (int)a.ix.dim, v),
                                                        reinterpret_cast<const uint8_t*>(x), y, a.ix.dim,
                                                        SqParams{a.ix.sq_k, a.ix.sq_shift_norm_sq});
                if (v == 0) {
                    pid[cnt + r] = id;
                    pd[cnt + r] = d;
                }
            }
        }
    }
    __syncthreads();
    prune_sorted_pool<DT, OP, NORM>(a.ix, a.cfg, loc, cnt + nex, a.pcap, smem, L, a.force_saturate != 0, out);
}

// ---- kernel B: robust_prune_list over explicit candidate lists ---------------------------
// (bootstrap: list = own edges ∪ other batch members; index.rs:597-645)
struct ListArgs {
    IndexView ix;
    PruneCfg cfg;
    const uint32_t* locs;
    const uint32_t* pending;  // n x pend_stride: [len, ids...] current edges of each item
    uint32_t pend_stride;
    uint32_t n;
    uint32_t pcap;
    uint32_t* out;
    uint32_t out_stride;
    uint32_t* err;
};

template <int DT, int OP, bool NORM>
__device__ void fill_list_distances(const IndexView& ix, uint32_t loc, uint32_t* pid, float* pd, uint32_t cnt) {
    using S = Scheme<DT, OP, true>;
    using RT = typename RowType<DT>::type;
    constexpr int G = S::G, GROUPS = kWave / G;
    const uint32_t lane = threadIdx.x;
    const int g = lane / G, v = lane % G;
    const RT* x = reinterpret_cast<const RT*>(ix.rows + (uint64_t)loc * ix.row_stride);
    for (uint32_t r0 = 0; r0 < cnt; r0 += GROUPS) {
        const uint32_t r = r0 + g;
        if (r < cnt) {
            const uint8_t* y = ix.rows + (uint64_t)pid[r] * ix.row_stride;
            float d = finish_distance<DT, OP, NORM>(group_distance_rows<DT, OP>(reinterpret_cast<const uint8_t*>(x), y, (int)ix.dim, v),
                                                    reinterpret_cast<const uint8_t*>(x), y, ix.dim,
                                                    SqParams{ix.sq_k, ix.sq_shift_norm_sq});
            if (v == 0) pd[r] = d;
        }
    }
}

template <int DT, int OP, bool NORM>
__global__ __launch_bounds__(kWave) void bootstrap_kernel(ListArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t lane = threadIdx.x, item = blockIdx.x;
    const PoolLds L = pool_lds_layout(a.pcap, a.cfg.pruned_degree);
    uint32_t* pid = reinterpret_cast<uint32_t*>(smem + L.pid_off);
    float* pd = reinterpret_cast<float*>(smem + L.pd_off);
    const uint32_t loc = a.locs[item];
    const uint32_t* mine = a.pending + (uint64_t)item * a.pend_stride;
    const uint32_t ne = mine[0];
    uint32_t* out = a.out + (uint64_t)item * a.out_stride;
    if (ne + a.n > a.pcap) {
        if (lane == 0) {
            *a.err = 1;
            out[0] = 0;
        }
        return;
    }
    // AdjacencyList::from_iter_untrusted(edges ++ other sources): first occurrence wins.
    // Own edges are unique and never contain `loc`; a batch member already present in the
    // edges is skipped.
    for (uint32_t i = lane; i < ne; i += kWave) pid[i] = mine[1 + i];
    __syncthreads();
    uint32_t cnt = ne;
    for (uint32_t p0 = 0; p0 < a.n; p0 += kWave) {
        const uint32_t p = p0 + lane;
        uint32_t id = kEmpty;
        bool take = false;
        if (p < a.n && p != item) {
            id = a.locs[p];
            take = true;
            for (uint32_t e = 0; e < ne; ++e) take &= (pid[e] != id);
        }
        const uint64_t m = ballot64(take);
        if (take) pid[cnt + mbcnt(m)] = id;
        cnt += (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (a.cfg.tie_order != 0) {
        // from_iter_untrusted is sort_unstable + dedup (adjacencylist.rs:181-190): robust_prune_list walks the candidates
        // in ascending id order, and the reference's sort sees them arrive in that order.  The ids are distinct: an id's
        // rank is the number of smaller ones.
        uint32_t* tmp = reinterpret_cast<uint32_t*>(smem + L.keys_off);
        for (uint32_t i = lane; i < cnt; i += kWave) {
            const uint32_t id = pid[i];
            uint32_t r = 0;
            for (uint32_t j = 0; j < cnt; ++j) r += pid[j] < id ? 1u : 0u;
            tmp[r] = id;
        }
        __syncthreads();
        for (uint32_t i = lane; i < cnt; i += kWave) pid[i] = tmp[i];
        __syncthreads();
    }
    fill_list_distances<DT, OP, NORM>(a.ix, loc, pid, pd, cnt);
    __syncthreads();
    prune_sorted_pool<DT, OP, NORM>(a.ix, a.cfg, loc, cnt, a.pcap, smem, L, true, out);
}

// ---- kernel C: back-edges, one wave per distinct target (add_edge_and_prune) ----------------
struct BackArgs {
    IndexView ix;
    PruneCfg cfg;
    const uint64_t* keys;      // sorted (target << 32 | source), `nkeys` valid entries first
    const uint32_t* seg_start; // nseg segment start indices (arbitrary order)
    const uint32_t* seg_len;   // length of the segment starting at index i (indexed by start)
    uint32_t nseg;
    uint32_t nkeys;
    uint32_t pcap;
    uint32_t* err;
    uint32_t count_long = 0;   // this launch prunes the lists too long for the MFMA path: counted in counters[7]
    const uint32_t* work;      // optional: segment indices to process (targets whose list needs a prune)
};

// ---- kernel C0: every distinct target of the batch, one wave each: append the new sources if the list still fits
// (provider.rs:795-822), otherwise queue the target for a prune kernel.  No distances, 1 KiB of LDS: this is the
// launch that covers ~10^6 targets per batch, the prune kernels only see the few percent that overflow.
struct ScanArgs {
    IndexView ix;
    uint32_t cfg_max_degree;   // "max_degree_with_slack": a list longer than this is pruned
    const uint64_t* keys;
    const uint32_t* seg_start;
    const uint32_t* seg_len;
    uint32_t nseg;
    uint32_t short_cap;        // lists up to this length go to work_short, longer ones to work_long
    uint32_t* work_short;
    uint32_t* work_long;
    uint32_t* counts;          // [0] #short [1] #long [2] longest list among the long ones
    uint32_t rank = 0, world = 1;  // owner-partitioned commit: only targets with id % world == rank are queued here
};

// Partitioned commit (multi-GPU build): lists that fit are appended by every replica (cheap, deterministic); a list
// that has to be pruned -- the expensive part -- is queued only on the rank that owns the target.  The owners export
// the resulting rows, the others apply them (dann_insert_batch_commit_part / dann_apply_neighbor_rows_device).
__global__ void export_rows_kernel(IndexView ix, const uint64_t* keys, const uint32_t* seg_start, const uint32_t* work_short,
                                   uint32_t nshort, const uint32_t* work_long, uint32_t nlong, uint32_t* out) {
    const uint32_t i = blockIdx.x, lane = threadIdx.x;
    if (i >= nshort + nlong) return;
    const uint32_t seg = i < nshort ? work_short[i] : work_long[i - nshort];
    const uint32_t tgt = (uint32_t)(keys[seg_start[seg]] >> 32);
    const uint32_t* arow = ix.adj + (uint64_t)tgt * ix.adj_stride;
    uint32_t* o = out + (uint64_t)i * (ix.max_degree + 2u);
    if (lane == 0) o[0] = tgt;
    for (uint32_t e = lane; e < ix.max_degree + 1u; e += blockDim.x) o[1 + e] = arow[e];
}

__global__ void apply_rows_kernel(IndexView ix, const uint32_t* rows, uint32_t count) {
    const uint32_t i = blockIdx.x, lane = threadIdx.x;
    if (i >= count) return;
    const uint32_t* r = rows + (uint64_t)i * (ix.max_degree + 2u);
    const uint32_t tgt = r[0];
    if (tgt >= ix.nslots) return;
    uint32_t* arow = ix.adj + (uint64_t)tgt * ix.adj_stride;
    for (uint32_t e = lane; e < ix.max_degree + 1u; e += blockDim.x) arow[e] = e == 0 ? (r[1] < ix.max_degree ? r[1] : ix.max_degree) : r[1 + e];
}

// Four targets per wavefront, 16 lanes each.  A target's scan is a chain of dependent loads (segment start -> first key /
// segment length -> adjacency row) and a few compares: one wavefront per target kept 8 192 resident waves waiting on
// that chain (190 ms of the 1 M x 768 build, 900 k targets per batch); four chains per wave run side by side.  What is
// written -- the appended ids in source order, the worklists' contents -- is what the one-target form wrote.
__global__ __launch_bounds__(kWave) void backedge_scan_kernel(ScanArgs a) {
    __shared__ uint32_t newid[4][64];
    const uint32_t lane = threadIdx.x, sub = lane >> 4, sl = lane & 15u;
    const uint32_t seg_raw = blockIdx.x * 4u + sub;
    const bool have = seg_raw < a.nseg;
    const uint32_t seg = have ? seg_raw : a.nseg - 1u;  // (an idle group repeats the last segment's loads, writes nothing)
    const uint32_t start = a.seg_start[seg];
    const uint32_t src = (uint32_t)(a.keys[start] >> 32);
    uint32_t* arow = a.ix.adj + (uint64_t)src * a.ix.adj_stride;
    uint32_t len = arow[0];
    len = len < a.ix.max_degree ? len : a.ix.max_degree;
    const uint32_t nsrc = a.seg_len[start];
    const bool owned = a.world <= 1u || src % a.world == a.rank;
    const bool hub = nsrc > (uint32_t)kWave;
    if (have && hub && sl == 0 && owned) {
        // a hub hit by many back-edges: no serial scan here, the prune kernels de-duplicate 64 sources at a time
        // (len + #sources bounds the list; they also handle the case that everything still fits)
        const uint32_t bound = len + nsrc;
        if (bound <= a.short_cap) {
            a.work_short[atomicAdd(&a.counts[0], 1u)] = seg;
        } else {
            a.work_long[atomicAdd(&a.counts[1], 1u)] = seg;
            atomicMax(&a.counts[2], bound);
        }
    }
    const bool scan = have && !hub;
    // the first 64 list entries and the (at most 64) sources of the group's target: four per lane
    uint32_t mine[4], srcs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t e = sl + 16u * (uint32_t)j;
        mine[j] = (scan && e < len) ? arow[1 + e] : kEmpty;
        srcs[j] = (scan && e < nsrc) ? (uint32_t)a.keys[start + e] : kEmpty;
    }
    // AdjacencyList::extend_from_slice: a source already in the list is skipped (sources of one target are distinct)
    uint32_t kmax = 0;
    {
        uint32_t v = scan ? nsrc : 0u;  // the longest scan among the four groups
        v = max(v, (uint32_t)__shfl_xor((int)v, 16));
        v = max(v, (uint32_t)__shfl_xor((int)v, 32));
        kmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    }
    uint32_t nnew = 0;
    for (uint32_t k = 0; k < kmax; ++k) {
        const uint32_t from = (lane & 48u) | (k & 15u);
        uint32_t id = kEmpty;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if ((k >> 4) == (uint32_t)j) id = (uint32_t)__shfl((int)srcs[j], (int)from);
        const bool on = scan && k < nsrc;
        bool dup = (mine[0] == id) | (mine[1] == id) | (mine[2] == id) | (mine[3] == id);
        for (uint32_t e = kWave + sl; e < len; e += 16u) dup |= (arow[1 + e] == id);  // (degrees beyond 64)
        const uint32_t dm = (uint32_t)(ballot64(dup) >> (16u * sub)) & 0xFFFFu;
        if (on && dm == 0u) {
            if (sl == 0) newid[sub][nnew] = id;
            ++nnew;
        }
    }
    __syncthreads();
    if (!scan || nnew == 0) return;
    const uint32_t cnt = len + nnew;
    if (cnt <= a.cfg_max_degree) {
        const uint32_t slack = a.ix.max_degree - len;
        const uint32_t take = nnew < slack ? nnew : slack;
        for (uint32_t i = sl; i < take; i += 16u) arow[1 + len + i] = newid[sub][i];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (sl == 0) {
            arow[0] = len + take;
        }
        return;
    }
    if (sl == 0 && owned) {
        if (cnt <= a.short_cap && cnt > a.cfg_max_degree) {
            a.work_short[atomicAdd(&a.counts[0], 1u)] = seg;
        } else {
            a.work_long[atomicAdd(&a.counts[1], 1u)] = seg;
            atomicMax(&a.counts[2], cnt);
        }
    }
}

template <int DT, int OP, bool NORM>
__global__ __launch_bounds__(kWave) void backedge_kernel(BackArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t lane = threadIdx.x, seg = a.work ? a.work[blockIdx.x] : blockIdx.x;
    const PoolLds L = pool_lds_layout(a.pcap, a.cfg.pruned_degree);
    uint32_t* pid = reinterpret_cast<uint32_t*>(smem + L.pid_off);
    float* pd = reinterpret_cast<float*>(smem + L.pd_off);
    const uint32_t start = a.seg_start[seg];
    const uint32_t src = (uint32_t)(a.keys[start] >> 32);
    // adjacency of `src` (Neighbors::get)
    uint32_t* arow = a.ix.adj + (uint64_t)src * a.ix.adj_stride;
    uint32_t len = arow[0];
    len = len < a.ix.max_degree ? len : a.ix.max_degree;
    for (uint32_t i = lane; i < len; i += kWave) pid[i] = arow[1 + i];
    __syncthreads();
    // extend_from_slice(sorted targets): unique append
    uint32_t cnt = len;
    bool overflow = false;
    const uint32_t end = start + a.seg_len[start];
    for (uint32_t k0 = start; k0 < end; k0 += kWave) {
        const uint32_t k = k0 + lane;
        const bool mine = k < end;
        const uint32_t id = mine ? (uint32_t)a.keys[k] : kEmpty;
        bool take = mine;
        if (mine)
            for (uint32_t e = 0; e < len; ++e) take &= (pid[e] != id);
        const uint64_t tm = ballot64(take);
        const uint32_t ntake = (uint32_t)__popcll(tm);
        if (cnt + ntake > a.pcap) {
            overflow = true;
            break;
        }
        if (take) pid[cnt + mbcnt(tm)] = id;
        cnt += ntake;
    }
    if (overflow) {
        if (lane == 0) *a.err = 1;
        return;
    }
    const uint32_t added = cnt - len;
    if (added == 0) return;
    __syncthreads();
    if (cnt <= a.cfg.max_degree) {
        // append_vector with the provider-capacity clamp (provider.rs:795-822)
        const uint32_t slack = a.ix.max_degree - len;
        const uint32_t take = added < slack ? added : slack;
        for (uint32_t i = lane; i < take; i += kWave) arow[1 + len + i] = pid[len + i];
        __syncthreads();
        if (lane == 0) {
            arow[0] = len + take;
        }
        return;
    }
    fill_list_distances<DT, OP, NORM>(a.ix, src, pid, pd, cnt);
    if (lane == 0 && a.cfg.counters) atomicAdd(&stat_stripe(a.cfg.counters)[1], (unsigned long long)cnt);
    __syncthreads();
    // the list lives in LDS, so the result can go straight into the adjacency row (nobody
    // else reads this row during the back-edge phase)
    prune_sorted_pool<DT, OP, NORM>(a.ix, a.cfg, src, cnt, a.pcap, smem, L, false, arow);
    if (lane == 0 && a.count_long && a.cfg.counters) atomicAdd(&stat_stripe(a.cfg.counters)[7], 1ull);
}


// ======================================================================================================
// MFMA path (f32 / f16 rows): pair similarities of one candidate list as a Gram matrix on the matrix cores.
//
// prune::robust_prune asks for d(c_i, c_j) between a candidate and the candidates already selected
// (prune.rs:196-232, `compute_distance`) -- for a back-edge prune nearly every pair of the list, for the pool prune of
// an inserted point ~2 200 pairs among the first ~140 sorted candidates.  Instead of evaluating them one dependent row
// pair at a time, the lower triangle of the list's Gram matrix G = C C^T is computed with v_mfma_f32_32x32x2_f32
// (gram_tiles_kernel) and the sweep derives from it an *approximation* of every pair distance together with a rigorous
// bound of its deviation from the reference's own f32 value:
//     L2:  d' = |x|^2 + |y|^2 - 2<x,y>,   IP: d' = -<x,y>,   CosineNormalized: d' = 1 - <x,y>
//     |d' - d_ref| <= E = c1 * (|x|^2 + |y|^2) + c2 * |d'|
// Every decision of the sweep is a comparison of d_ref with a threshold; where the interval [d' - E, d' + E]
// does not decide it, the pair is re-evaluated with the bit-exact row kernel.  The adjacency lists are therefore
// identical to the lazy path's (and the oracle's) by construction; tests/test_gpu_build.py checks it.
// ======================================================================================================
// The Gram arithmetic (gram_tiles_kernel): one f32 FMA chain over the whole row per entry, K = dim rounded up to 32 terms:
//       |G_ij - <x,y>| <= gamma_K sum|x_e y_e| <= K u (|x|^2 + |y|^2) / 2, u = 2^-24;  d' = (|x|^2 + |y|^2) - 2 G_ij
//       with the norms accumulated in f64 -> c1 = (K + 4) u (+ 5 % slack) covers G, the norms' rounding and the two f32
//       operations on terms of that size.
// c2 (relative to |d'|) covers the final subtraction and the reference's own f32 rounding, which grows with the row
// length: its L2 / IP kernels run dim / (8 NACC) terms per chain plus the combine and sum_tree adds and round x - y
// before squaring -> (dim / 8 + 16) u bounds every row type and strategy (never below the 3e-6 of the 128-d analysis).
// tests/test_gram_interval.py evaluates both sides on the CPU (pairs built to cancel) and checks the bound.
constexpr float kGramC2 = 3.0e-6f;
constexpr float kUnitRoundoff = 5.9604645e-8f;  // 2^-24
inline float gram_c2_for_dim(uint32_t dim) {
    const float c = ((float)(dim / 8u) + 16.0f) * kUnitRoundoff;
    return c > kGramC2 ? c : kGramC2;
}
inline float gram_c1_chained(uint32_t dim) { return 1.05f * ((float)((dim + 31u) & ~31u) + 4.0f) * kUnitRoundoff; }
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GramCtx {
    const float* g;      // LDS or global: g[p * ld + q] = <row p, row q>, p < nrows, q < ncols
    const float* nrm;    // |row p|^2, p < nrows
    uint32_t ld, nrows, ncols;
    bool by_sorted;      // rows are indexed by sorted pool order (pool prune) or by pool position (back-edge lists)
    float escale;        // 1.0; tests widen the error interval (DANN_DBG_GRAM_ESCALE) to drive every decision through the exact path
    float c1, c2;        // error interval E = c1 (|x|^2 + |y|^2) + c2 |d'| of this Gram's arithmetic
    bool count_rows;     // add this list to the Gram row / flop counters (the tiles kernel counts its own)
    bool prefetch;       // g lives in global memory: touch the next candidates' rows ahead of their look-ups
};

// prune_sorted_pool with the pair distances taken from the Gram matrix (exact re-check where the error interval
// does not decide).  Single wave (lanes 0..63 of the workgroup); the other waves have exited.
// one wave: LDS accesses of a single wave retire in program order; this is the compiler + counter fence between them
__device__ __forceinline__ void wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// SortedNeighbors::new for one wave (the other waves of the workgroup may be waiting at a barrier): pid/pd[0..P) ->
// sid/sd/occ/last[0..N), keys[i] keeps the pool position of sorted entry i.  Returns N.
__device__ uint32_t sort_pool_wave(const PruneCfg& cfg, uint32_t P, uint32_t pcap, uint8_t* smem, const PoolLds& L) {
    const uint32_t lane = threadIdx.x & 63u;
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem + L.keys_off);
    const uint32_t* pid = reinterpret_cast<const uint32_t*>(smem + L.pid_off);
    const float* pd = reinterpret_cast<const float*>(smem + L.pd_off);
    uint32_t* sid = reinterpret_cast<uint32_t*>(smem + L.sid_off);
    float* sd = reinterpret_cast<float*>(smem + L.sd_off);
    float* occ = reinterpret_cast<float*>(smem + L.occ_off);
    uint16_t* last = reinterpret_cast<uint16_t*>(smem + L.last_off);
    bool saw_nan = false;
    for (uint32_t i = lane; i < pcap; i += kWave) {
        const float d = i < P ? pd[i] : 0.0f;
        saw_nan |= d != d;
        keys[i] = i < P ? sort_key(d, i) : ~0ull;
    }
    wave_sync();
    for (uint32_t k = 2; k <= pcap; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = lane; t < (pcap >> 1); t += kWave) {
                const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u));
                const uint32_t p = i | j;
                const uint64_t a = keys[i], b = keys[p];
                const bool up = (i & k) == 0;
                if ((a > b) == up) {
                    keys[i] = b;
                    keys[p] = a;
                }
            }
            wave_sync();
        }
    }
    const uint32_t N = P < cfg.max_occlusion ? P : cfg.max_occlusion;
    if (cfg.tie_order != 0 && pool_has_ties(keys, P, saw_nan)) {  // (see prune_sorted_pool)
        wave_sync();
        rust_order_by_lane0(cfg, P, smem, L);
        wave_sync();  // the work area of the walk is dead: keys[i] = pool position of sorted entry i, as the sort leaves it
        for (uint32_t i = lane; i < N; i += kWave) keys[i] = last[i];
    }
    for (uint32_t i = lane; i < N; i += kWave) {
        const uint32_t pos = (uint32_t)keys[i];
        sid[i] = pid[pos];
        sd[i] = pd[pos];
        occ[i] = 0.0f;
        last[i] = 0;
    }
    wave_sync();
    return N;
}

// The sweep of prune::robust_prune over a pool already sorted by sort_pool_wave, pair distances from the Gram matrix
// where it covers the pair (exact re-check where the error interval does not decide, exact evaluation where it
// does not cover it).  One wave (lanes 0..63 of the workgroup).
template <int DT, int OP, bool NORM>
__device__ void sweep_sorted_pool_gram(const IndexView& ix, const PruneCfg& cfg, uint32_t location, uint32_t N,
                                       uint8_t* smem, const PoolLds& L, bool force_saturate, uint32_t* out,
                                       const GramCtx gc) {
    using S = Scheme<DT, OP, true>;
    constexpr int G = S::G;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t* keys = reinterpret_cast<const uint64_t*>(smem + L.keys_off);
    const uint32_t* sid = reinterpret_cast<const uint32_t*>(smem + L.sid_off);
    const float* sd = reinterpret_cast<const float*>(smem + L.sd_off);
    float* occ = reinterpret_cast<float*>(smem + L.occ_off);
    uint16_t* last = reinterpret_cast<uint16_t*>(smem + L.last_off);
    uint32_t* sel = reinterpret_cast<uint32_t*>(smem + L.sel_off);
    const uint32_t P = N;
    const uint32_t degree = cfg.pruned_degree;
    const bool occluding = (ix.metric == M_IP);
    const float alpha = cfg.alpha;
    const float inc = alpha < 1.2f ? alpha : 1.2f;
    const float kMax = 3.402823466e+38f;
    float cur_alpha = 1.0f;
    uint32_t found = 0, nexact = 0, nasked = 0;  // nasked: pair distances the reference's lazy scan asks for
    const int v = lane % G;
    const SqParams sqp{ix.sq_k, ix.sq_shift_norm_sq};
    // first selected entry in sel[a..b) (pool order filter rp < i) whose pair distance with candidate i makes
    // update_occlude exceed `at`; returns its index in sel, or b when there is none
    auto first_exceed = [&](uint32_t i, uint32_t a, uint32_t b, float at) -> uint32_t {
        const uint32_t pi = gc.by_sorted ? i : (uint32_t)keys[i];
        const bool irow = pi < gc.nrows;
        const float gii = irow ? gc.nrm[pi] : 0.0f;
        const float di = sd[i];
        const uint8_t* xi = ix.rows + (uint64_t)sid[i] * ix.row_stride;
        const float thr = at * di;  // occluding rule: d_jk < alpha * d_ik (config/mod.rs:98)
        for (uint32_t c0 = a; c0 < b; c0 += kWave) {
            const uint32_t c = c0 + lane;
            int cls = 0;  // 0: certainly not, 1: certainly exceeds, 2: the error interval does not decide
            uint32_t rp = 0;
            if (c < b) {
                rp = sel[c];
                if (rp < i) {
                    const uint32_t pj = gc.by_sorted ? rp : (uint32_t)keys[rp];
                    cls = 2;  // pairs the Gram does not cover are evaluated exactly
                    if (irow && pj < gc.ncols) {
                    const float gij = gc.g[pi * gc.ld + pj], gjj = gc.nrm[pj];
                    const float nsum = gii + gjj;
                    float dp;
                    if (OP == OP_L2) dp = nsum - 2.0f * gij;
                    else dp = NORM ? 1.0f - gij : -gij;
                    const float e = gc.escale * (gc.c1 * nsum + gc.c2 * __builtin_fabsf(dp));
                    const float lo = dp - e, hi = dp + e;
                    if (occluding) {
                        if (hi < thr) cls = 1;
                        else if (lo >= thr) cls = 0;
                    } else if (lo > 0.0f && di >= 0.0f) {
                        const float rmin = di / hi, rmax = di / lo;
                        if (rmin > at * 1.000001f) cls = 1;
                        else if (rmax < at * 0.999999f) cls = 0;
                    }
                    }
                }
            }
            uint64_t tu = ballot64(cls != 0);
            const uint64_t valid = ballot64(c < b && rp < i);  // the pairs of this block the lazy scan would evaluate
            auto asked_upto = [&](int f) { nasked += (uint32_t)__popcll(valid & (f >= 63 ? ~0ull : ((2ull << f) - 1ull))); };
            if (!tu) asked_upto(63);
            while (tu) {
                const int f = __builtin_ctzll(tu);
                if (__builtin_amdgcn_readlane(cls, f) == 1) {
                    asked_upto(f);
                    return c0 + (uint32_t)f;
                }
                // bit-exact pair distance (every lane group evaluates the same pair)
                ++nexact;
                const uint32_t rpf = (uint32_t)__builtin_amdgcn_readlane((int)rp, f);
                const uint8_t* y = ix.rows + (uint64_t)sid[rpf] * ix.row_stride;
                const float d = finish_distance<DT, OP, NORM>(group_distance_rows<DT, OP>(xi, y, (int)ix.dim, v), xi, y,
                                                              ix.dim, sqp);
                const float dg = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), 0));
                if (update_occlude<OP>(di, dg, 0.0f, at, occluding) > at) {
                    asked_upto(f);
                    return c0 + (uint32_t)f;
                }
                tu &= tu - 1;
                if (!tu) asked_upto(63);
            }
        }
        return b;
    };
    // Gram in global memory (three-kernel pool prune): every look-up of a candidate's row is a dependent round trip to
    // L2 / the Infinity Cache / HBM, and the sweep is a chain of them.  The rows of the next two candidates are touched
    // (one dword per 128-byte line of the row's <= 96 columns) while the current one is examined: pure prefetch into a
    // register nothing ever reads, never waited for (the loop's own loads drain the in-order queue behind it).
    uint32_t gpf = 0;
    auto touch_row = [&](uint32_t r) {
        if (!gc.prefetch || r >= gc.nrows) return;
        uint32_t col = lane < 3u ? lane * 32u : 0u;  // unconditional, clamped: no exec-mask block around the request
        col = col < gc.ncols ? col : 0u;
        const float* pr = gc.g + (size_t)r * gc.ld + col;
        asm volatile("global_load_dword %0, %1, off" : "=&v"(gpf) : "v"(pr));
    };
    if (N > 0) {
        while (found < degree) {
            touch_row(0);
            touch_row(1);
            for (uint32_t i = 0; i < N && found < degree; ++i) {
                asm volatile("" ::"v"(gpf));
                touch_row(i + 2);
                const float o = occ[i];
                uint32_t l = last[i];
                if (o == kMax) continue;                      // selected or excluded
                if (occluding && o > cur_alpha) continue;     // o is exact for the Occluding kind (alpha_then + 0.01)
                const uint32_t idi = sid[i];
                if (idi == location || idi >= ix.nslots) {
                    if (lane == 0) occ[i] = kMax;
                    continue;
                }
                // TriangleInequality kind: the stored maximum ratio is only ever compared with the current alpha;
                // it is re-derived from the entries already examined (sel[0..l))
                if (!occluding && l != 0 && first_exceed(i, 0, l, cur_alpha) != l) continue;
                bool rejected = false;
                if (l != found) {
                    const uint32_t at = first_exceed(i, l, found, cur_alpha);
                    if (at != found) {
                        l = at + 1;
                        rejected = true;
                    } else {
                        l = found;
                    }
                }
                wave_sync();
                if (lane == 0) {
                    last[i] = (uint16_t)l;
                    if (rejected) {
                        if (occluding) occ[i] = cur_alpha + 0.01f;
                    } else {
                        occ[i] = kMax;
                        sel[found] = i;
                    }
                }
                if (!rejected) ++found;
                wave_sync();
            }
            if (cur_alpha == alpha) break;
            const float next = cur_alpha * inc;
            cur_alpha = next < alpha ? next : alpha;
        }
    }
    if (gc.prefetch) {  // no prefetch load may still be in flight when its landing register is released
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" ::"v"(gpf));
    }
    wave_sync();
    uint32_t nout = found;
    if (force_saturate || (cfg.saturate_after_prune && alpha > 1.0f)) {
        for (uint32_t i = 0; i < N && nout < degree; ++i) {
            const uint32_t id = sid[i];
            if (id == location) continue;
            bool dup = false;
            for (uint32_t n = lane; n < nout; n += kWave) dup |= (sid[sel[n]] == id);
            if (ballot64(dup)) continue;
            if (lane == 0) sel[nout] = i;
            ++nout;
            wave_sync();
        }
    }
    wave_sync();
    for (uint32_t n = lane; n < nout; n += kWave) out[1 + n] = sid[sel[n]];
    if (lane == 0) {
        out[0] = nout;
        if (cfg.counters) {
            atomicAdd(&stat_stripe(cfg.counters)[0], (unsigned long long)nexact);
            if (gc.count_rows) {
                atomicAdd(&stat_stripe(cfg.counters)[2], (unsigned long long)P);
                atomicAdd(&stat_stripe(cfg.counters)[3], (unsigned long long)P * P);
            }
            atomicAdd(&stat_stripe(cfg.counters)[4], (unsigned long long)nasked);
            atomicAdd(&stat_stripe(cfg.counters)[5], (unsigned long long)nexact);
        }
    }
}

// The same sweep for lists whose Gram lives in global memory, eight candidates at a time.  In the form above every
// candidate costs a chain of dependent round trips (its queue state from LDS, then one Gram row from L2 / HBM, then the
// decision, then the state back to LDS): ~4 000 cycles each, 2.3 ms per pool prune of ~350 candidate visits.  Here a
// 64-candidate window of the sweep state is read once, the raw Gram rows of the next eight candidates that need a visit
// are requested together (two coalesced requests each: columns [0, 64) and [64, 128)), and the eight visits then run from
// registers: lane c stands for the c-th selected entry and picks G(i, sel_c) out of the row with one cross-lane read, so
// entries selected a moment ago need no new request.  The decisions, their order and the exact re-checks are those of
// sweep_sorted_pool_gram (same classes, same first-exceed rule); requires pruned_degree <= 64 (one lane per selected
// entry), a Gram indexed by sorted position and at most 128 Gram columns.
constexpr uint32_t kSweepRowsLds = 8u * 128u * 4u + 64u;  // LDS behind the pool layout: the raw Gram rows of one block of visits

template <int DT, int OP, bool NORM>
__device__ void sweep_gram_batched(const IndexView& ix, const PruneCfg& cfg, uint32_t location, uint32_t N,
                                   uint8_t* smem, const PoolLds& L, bool force_saturate, uint32_t* out, const GramCtx gc,
                                   float* rowsl) {
    constexpr int B = 8;
    using S = Scheme<DT, OP, true>;
    constexpr int G = S::G, GROUPS = kWave / G;
    const uint32_t lane = threadIdx.x & 63u;
    const int g = (int)(lane / G);
    const uint32_t* sid = reinterpret_cast<const uint32_t*>(smem + L.sid_off);
    const float* sd = reinterpret_cast<const float*>(smem + L.sd_off);
    float* occ = reinterpret_cast<float*>(smem + L.occ_off);
    uint16_t* last = reinterpret_cast<uint16_t*>(smem + L.last_off);
    uint32_t* sel = reinterpret_cast<uint32_t*>(smem + L.sel_off);
    float* nrml = reinterpret_cast<float*>(smem + L.keys_off);  // the sort keys are dead: the norms of the Gram rows
    for (uint32_t i = lane; i < gc.nrows; i += kWave) nrml[i] = gc.nrm[i];
    wave_sync();
    const uint32_t degree = cfg.pruned_degree;
    const bool occluding = (ix.metric == M_IP);
    const float alpha = cfg.alpha;
    const float inc = alpha < 1.2f ? alpha : 1.2f;
    const float kMax = 3.402823466e+38f;
    float cur_alpha = 1.0f;
    uint32_t found = 0, nexact = 0, nasked = 0;
    uint32_t my_sel = kEmpty;  // lane c < found: sorted position of the c-th selected entry
    float my_njj = 0.0f;       // its squared norm
    const int v = lane % G;
    const SqParams sqp{ix.sq_k, ix.sq_shift_norm_sq};
    const uint32_t col_lo = lane < gc.ncols ? lane : 0u, col_hi = lane + 64u < gc.ncols ? lane + 64u : 0u;
    while (N > 0 && found < degree) {
        uint32_t i0 = 0;
        while (i0 < N && found < degree) {
            // ---- a window of 64 candidates: who needs a visit in this pass
            const uint32_t wi = i0 + lane;
            float wocc = kMax;
            uint32_t wlast = 0, wsid = kEmpty;
            if (wi < N) {
                wocc = occ[wi];
                wlast = last[wi];
                wsid = sid[wi];
            }
            const bool need = wi < N && wocc != kMax && !(occluding && wocc > cur_alpha);
            uint64_t nm = ballot64(need);
            if (!nm) {
                i0 += kWave;
                continue;
            }
            uint32_t idx[B];
            uint32_t nb = 0;
#pragma unroll
            for (int j = 0; j < B; ++j) {
                if (nm) {
                    idx[j] = i0 + (uint32_t)__builtin_ctzll(nm);
                    nm &= nm - 1;
                    nb = (uint32_t)j + 1u;
                } else {
                    idx[j] = idx[j > 0 ? j - 1 : 0];
                }
            }
            // ---- their Gram rows, all requests up front (clamped, unconditional), parked in LDS: the visits below
            //      then run as an ordinary loop (one copy of the code) and pick G(i, sel_c) with one LDS read
            {
                float rlo[B], rhi[B];
#pragma unroll
                for (int j = 0; j < B; ++j) {
                    const uint32_t r = idx[j] < gc.nrows ? idx[j] : 0u;
                    const float* row = gc.g + (size_t)r * gc.ld;
                    rlo[j] = row[col_lo];
                    rhi[j] = row[col_hi];
                }
#pragma unroll
                for (int j = 0; j < B; ++j) {
                    rowsl[j * 128 + (int)lane] = rlo[j];
                    rowsl[j * 128 + 64 + (int)lane] = rhi[j];
                }
            }
            uint32_t* idxl = reinterpret_cast<uint32_t*>(rowsl + B * 128);  // the block's candidate indices
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < B; ++j) idxl[j] = idx[j];
            }
            wave_sync();
            // ---- the visits, in order
            for (uint32_t j = 0; j < nb && found < degree; ++j) {
                const uint32_t i = idxl[j];
                const int wl = (int)(i - i0);
                uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)wlast, wl);
                const uint32_t idi = (uint32_t)__builtin_amdgcn_readlane((int)wsid, wl);
                if (idi == location || idi >= ix.nslots) {  // excluded / not retrievable
                    if (lane == 0) occ[i] = kMax;
                    continue;
                }
                const bool irow = i < gc.nrows;
                const float gii = irow ? nrml[i] : 0.0f;
                const float di = sd[i];
                const float thr = cur_alpha * di;  // occluding rule: d_jk < alpha * d_ik (config/mod.rs:98)
                const uint8_t* xi = ix.rows + (uint64_t)idi * ix.row_stride;
                // G(i, sel_c) out of the parked row
                const float gsel = rowsl[j * 128u + (my_sel & 127u)];
                const bool have = lane < found && my_sel < i;  // the lazy scan skips selected entries that sort after i
                int cls = 0;  // 0: certainly not, 1: certainly exceeds, 2: the error interval does not decide / not covered
                if (have) {
                    cls = 2;
                    if (irow && my_sel < gc.ncols) {
                        const float gij = gsel;
                        const float nsum = gii + my_njj;
                        float dp;
                        if (OP == OP_L2) dp = nsum - 2.0f * gij;
                        else dp = NORM ? 1.0f - gij : -gij;
                        const float e = gc.escale * (gc.c1 * nsum + gc.c2 * __builtin_fabsf(dp));
                        const float lo = dp - e, hi = dp + e;
                        if (occluding) {
                            if (hi < thr) cls = 1;
                            else if (lo >= thr) cls = 0;
                        } else if (lo > 0.0f && di >= 0.0f) {
                            const float rmin = di / hi, rmax = di / lo;
                            if (rmin > cur_alpha * 1.000001f) cls = 1;
                            else if (rmax < cur_alpha * 0.999999f) cls = 0;
                        }
                    }
                }
                const uint64_t valid = ballot64(have), m1 = ballot64(cls == 1), m2 = ballot64(cls == 2);
                // first selected entry in sel[a..b) that makes update_occlude exceed the current alpha, or b
                auto first_exceed = [&](uint32_t a, uint32_t b) -> uint32_t {
                    const uint64_t below_b = b >= 64u ? ~0ull : ((1ull << b) - 1ull);
                    const uint64_t range = below_b & ~((1ull << a) - 1ull);
                    uint64_t tu = (m1 | m2) & range;
                    while (tu) {
                        const int f = __builtin_ctzll(tu);
                        const uint64_t upto = f >= 63 ? ~0ull : ((2ull << f) - 1ull);
                        if ((m1 >> f) & 1ull) {
                            nasked += (uint32_t)__popcll(valid & range & upto);
                            return (uint32_t)f;
                        }
                        // bit-exact pair distances, one lane group per pair: the undecided / uncovered entries that
                        // follow in scan order (up to the next certain one) are evaluated together -- the pairs a Gram
                        // block does not cover come in runs (the selected entries beyond its last column), and one
                        // evaluation costs a dependent fetch of two rows whether one group works or all of them.
                        // Consumed in scan order; the lazy scan stops at the first that exceeds (the rest is discarded).
                        int fs[GROUPS];
                        int ne = 0;
                        {
                            uint64_t t2 = tu;
#pragma unroll
                            for (int q = 0; q < GROUPS; ++q) {
                                fs[q] = 0;
                                if (t2 && ne == q) {
                                    const int f2 = __builtin_ctzll(t2);
                                    if (!((m1 >> f2) & 1ull)) {
                                        fs[q] = f2;
                                        ne = q + 1;
                                        t2 &= t2 - 1;
                                    }
                                }
                            }
                        }
                        uint32_t my_rp = 0;
#pragma unroll
                        for (int q = 0; q < GROUPS; ++q) {
                            const uint32_t rq = (uint32_t)__builtin_amdgcn_readlane((int)my_sel, fs[q]);
                            if (g == q) my_rp = rq;
                        }
                        const bool evalme = g < ne;
                        const uint8_t* y = ix.rows + (uint64_t)(evalme ? sid[my_rp] : idi) * ix.row_stride;
                        const float d = finish_distance<DT, OP, NORM>(group_distance_rows<DT, OP>(xi, y, (int)ix.dim, v), xi, y,
                                                                      ix.dim, sqp);
                        nexact += (uint32_t)ne;
                        for (int q = 0; q < ne; ++q) {
                            const float dg = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), q * G));
                            const int fq = fs[q];
                            if (update_occlude<OP>(di, dg, 0.0f, cur_alpha, occluding) > cur_alpha) {
                                const uint64_t uq = fq >= 63 ? ~0ull : ((2ull << fq) - 1ull);
                                nasked += (uint32_t)__popcll(valid & range & uq);
                                return (uint32_t)fq;
                            }
                            tu &= ~(1ull << fq);
                        }
                    }
                    nasked += (uint32_t)__popcll(valid & range);
                    return b;
                };
                // TriangleInequality kind: the stored maximum ratio is only ever compared with the current alpha; it is
                // re-derived from the entries already examined (sel[0..l))
                if (!occluding && l != 0 && first_exceed(0, l) != l) continue;
                bool rejected = false;
                if (l != found) {
                    const uint32_t at = first_exceed(l, found);
                    if (at != found) {
                        l = at + 1;
                        rejected = true;
                    } else {
                        l = found;
                    }
                }
                if (lane == 0) {
                    last[i] = (uint16_t)l;
                    if (rejected) {
                        if (occluding) occ[i] = cur_alpha + 0.01f;
                    } else {
                        occ[i] = kMax;
                        sel[found] = i;
                    }
                }
                if (!rejected) {
                    if (lane == found) {
                        my_sel = i;
                        my_njj = gii;
                    }
                    ++found;
                }
            }
            wave_sync();  // lane 0's state updates before the next window is read
            i0 = idxl[nb - 1u] + 1u;
        }
        if (cur_alpha == alpha) break;
        const float next = cur_alpha * inc;
        cur_alpha = next < alpha ? next : alpha;
    }
    wave_sync();
    uint32_t nout = found;
    if (force_saturate || (cfg.saturate_after_prune && alpha > 1.0f)) {
        for (uint32_t i = 0; i < N && nout < degree; ++i) {
            const uint32_t id = sid[i];
            if (id == location) continue;
            bool dup = false;
            for (uint32_t n = lane; n < nout; n += kWave) dup |= (sid[sel[n]] == id);
            if (ballot64(dup)) continue;
            if (lane == 0) sel[nout] = i;
            ++nout;
            wave_sync();
        }
    }
    wave_sync();
    for (uint32_t n = lane; n < nout; n += kWave) out[1 + n] = sid[sel[n]];
    if (lane == 0) {
        out[0] = nout;
        if (cfg.counters) {
            atomicAdd(&stat_stripe(cfg.counters)[0], (unsigned long long)nexact);
            atomicAdd(&stat_stripe(cfg.counters)[4], (unsigned long long)nasked);
            atomicAdd(&stat_stripe(cfg.counters)[5], (unsigned long long)nexact);
        }
    }
}

// ======================================================================================================
// Pool prune (phase 1 of multi_insert: robust_prune_with, index.rs:2476-2532) on the matrix cores, three kernels:
//   pool_sort_kernel   one wave per inserted point: pool + intra-batch extras -> SortedNeighbors::new, the sorted
//                      (id, distance) list goes to global memory;
//   gram_tiles_kernel  one 8-wave workgroup per point: the lower-triangular 32 x 32 tiles of
//                      G = C[0..ng) x C[0..mg)^T (the selected candidates come from the front of the sorted pool, and the
//                      sweep only ever asks for pairs (i, j) with j < i) with v_mfma_f32_32x32x2_f32, the tiles dealt out
//                      to the waves one at a time; rows streamed through two 32-column LDS slabs (the next slab's global
//                      loads in flight under the MFMAs, written to the other slab behind them, one barrier per slab);
//                      nothing else lives in this kernel: 128 registers, no scratch, two workgroups per CU (three for
//                      the short back-edge lists).  1 M x 768: 70 % of the matrix pipes' cycles in the 16 384-point
//                      launches (profiles/r04q_gram_tiles_pmc_*.csv), 91 TFLOP/s over the whole build;
//   pool_sweep_kernel  one wave per point at full occupancy: the sweep of prune::robust_prune with look-ups in G
//                      (global memory, L2 / Infinity-Cache resident) and the bit-exact row kernel where the error
//                      interval does not decide or the pair lies outside the block.
// The fused form (sort + Gram + sweep in one 4-wave workgroup, 75 KB of LDS) ran two sweeps per CU and lost to the row
// kernel (1 M x 768: 4.84 s vs 3.24 s); here the serial sweep and the Gram no longer share a workgroup.
// f16 rows are widened exactly while a slab is filled (the reference's f16 kernels widen to f32 and run f32 FMAs).
// ======================================================================================================
struct SortArgs {
    PoolArgs p;
    uint32_t* sid;   // n x pcap: ids in sorted pool order
    float* sd;       // n x pcap: their distances to the location
    uint32_t* sn;    // n: sorted pool length after the max_occlusion cut (0 = the pool overflowed)
};

template <int DT, int OP, bool NORM>
__global__ __launch_bounds__(kWave) void pool_sort_kernel(SortArgs sa) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    using S = Scheme<DT, OP, true>;
    constexpr int G = S::G, GROUPS = kWave / G;
    const PoolArgs& a = sa.p;
    const uint32_t lane = threadIdx.x, wi = blockIdx.x, item = a.pos0 + blockIdx.x;
    const PoolLds L = pool_lds_layout(a.pcap, a.cfg.pruned_degree);
    uint32_t* pid = reinterpret_cast<uint32_t*>(smem + L.pid_off);
    float* pd = reinterpret_cast<float*>(smem + L.pd_off);
    const uint32_t* sid = reinterpret_cast<const uint32_t*>(smem + L.sid_off);
    const float* sd = reinterpret_cast<const float*>(smem + L.sd_off);
    const uint32_t loc = a.locs[item];
    uint64_t lo;
    uint32_t cnt;
    if (a.offsets) {
        lo = a.offsets[wi];
        cnt = (uint32_t)(a.offsets[wi + 1] - lo);
    } else {
        lo = (uint64_t)wi * a.stride;
        cnt = a.counts[wi];
    }
    uint32_t nex = 0;
    if (a.cand != 0 && a.n > 1) nex = a.cand < a.n - 1 ? a.cand : a.n - 1;
    if (cnt + nex > a.pcap) {
        if (lane == 0) {
            *a.err = 1;
            a.out[(uint64_t)wi * a.out_stride] = 0;
            sa.sn[wi] = 0;
        }
        return;
    }
    for (uint32_t i = lane; i < cnt; i += kWave) {
        pid[i] = a.pool_ids[lo + i];
        pd[i] = a.pool_d[lo + i];
    }
    if (nex) {  // extras = around(ids, position, cand) (utils/async_tools.rs:51-131), as pool_prune_kernel
        const uint32_t half = (nex + 1) / 2;
        const uint32_t start = item >= half ? item - half : a.n - (half - item);
        const uint8_t* x = a.ix.rows + (uint64_t)loc * a.ix.row_stride;
        const int g = lane / G, v = lane % G;
        for (uint32_t r0 = 0; r0 < nex; r0 += GROUPS) {
            const uint32_t r = r0 + g;
            if (r < nex) {
                uint32_t p = start + r;
                const uint32_t dist_to_item = item >= start ? item - start : item + a.n - start;
                if (r >= dist_to_item) p += 1;
                p %= a.n;
                const uint32_t id = a.locs[p];
                const uint8_t* y = a.ix.rows + (uint64_t)id * a.ix.row_stride;
                const float d = finish_distance<DT, OP, NORM>(group_distance_rows<DT, OP>(x, y, (int)a.ix.dim, v), x, y,
                                                              a.ix.dim, SqParams{a.ix.sq_k, a.ix.sq_shift_norm_sq});
                if (v == 0) {
                    pid[cnt + r] = id;
                    pd[cnt + r] = d;
                }
            }
        }
        if (lane == 0 && a.cfg.counters) atomicAdd(&stat_stripe(a.cfg.counters)[1], (unsigned long long)nex);
    }
    wave_sync();
    const uint32_t N = sort_pool_wave(a.cfg, cnt + nex, a.pcap, smem, L);
    uint32_t* gs = sa.sid + (uint64_t)wi * a.pcap;
    float* gd = sa.sd + (uint64_t)wi * a.pcap;
    for (uint32_t i = lane; i < N; i += kWave) {
        gs[i] = sid[i];
        gd[i] = sd[i];
    }
    if (lane == 0) sa.sn[wi] = N;
}

struct TileArgs {
    IndexView ix;
    const uint32_t* sid;  // n x pcap sorted candidate ids (pool_sort_kernel)
    const uint32_t* sn;   // n
    uint32_t pcap;
    uint32_t ng, mg;      // Gram rows / columns per item: multiples of 32, ng <= 256, mg <= 96
    float* gram;          // n x ng x mg, entry (p, q) valid for q < p (and on the diagonal blocks)
    float* nrm;           // n x ng
    unsigned long long* counters;  // PruneCfg::counters: [2] += rows, [3] += tiles * 1024 (x dim x 2 = MFMA flop)
    const uint32_t* order = nullptr;  // optional: workgroup b works on item order[b] (longest lists first)
};

constexpr int kTileRowBlocks = 8;  // ng <= 256: wave w of the 8-wave workgroup owns row block w
constexpr int kTileColBlocks = 3;  // mg <= 96
constexpr int kTilePasses = kTileRowBlocks / 2;  // slab fill: 64 rows per pass

template <typename RT>
__device__ __forceinline__ float4 tile_load4(const uint8_t* p);  // 4 consecutive elements, widened exactly
template <>
__device__ __forceinline__ float4 tile_load4<float>(const uint8_t* p) {
    return *reinterpret_cast<const float4*>(p);
}
template <>
__device__ __forceinline__ float4 tile_load4<__half>(const uint8_t* p) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    const float2 a = __half22float2(__builtin_bit_cast(__half2, t.x)), b = __half22float2(__builtin_bit_cast(__half2, t.y));
    return float4{a.x, a.y, b.x, b.y};
}
template <typename RT>
__device__ __forceinline__ float4 tile_fetch4(const uint8_t* row, uint32_t k, uint32_t dim);
template <>
__device__ __forceinline__ float4 tile_fetch4<float>(const uint8_t* row, uint32_t k, uint32_t dim) {
    const float* r = reinterpret_cast<const float*>(row);
    float4 q = {0.f, 0.f, 0.f, 0.f};
    if (k + 3u < dim) {
        q = *reinterpret_cast<const float4*>(r + k);
    } else if (k < dim) {
        q.x = r[k];
        if (k + 1u < dim) q.y = r[k + 1u];
        if (k + 2u < dim) q.z = r[k + 2u];
    }
    return q;
}
template <>
__device__ __forceinline__ float4 tile_fetch4<__half>(const uint8_t* row, uint32_t k, uint32_t dim) {
    const __half* r = reinterpret_cast<const __half*>(row);
    float4 q = {0.f, 0.f, 0.f, 0.f};
    if (k + 3u < dim) {
        const uint2 t = *reinterpret_cast<const uint2*>(r + k);
        const float2 a = __half22float2(__builtin_bit_cast(__half2, t.x)), b = __half22float2(__builtin_bit_cast(__half2, t.y));
        q = {a.x, a.y, b.x, b.y};
    } else if (k < dim) {
        q.x = __half2float(r[k]);
        if (k + 1u < dim) q.y = __half2float(r[k + 1u]);
        if (k + 2u < dim) q.z = __half2float(r[k + 2u]);
    }
    return q;
}

// one 32-column slab of the NT tiles of a wave (tile t: row-block operand at ao[t], column-block operand at bo[t], float
// offsets into the slab).  The operands come from LDS in chunks of CK k-pairs, the next chunk requested before the MFMAs of
// the current one are issued.  The scheduling barriers pin that order: left to itself the compiler puts every read directly
// in front of its MFMAs (read, wait, two MFMAs, read, wait, ...), and all operands of a slab up front would be 96 registers.
template <int NT>
__device__ __forceinline__ void tile_slab(const float* lane_base, const uint32_t (&sa)[3], const uint32_t (&sb)[3],
                                          f32x16 (&acc)[NT]) {
    constexpr int CK = NT == 3 ? 2 : 4, NC = 16 / CK;
    float av[2][NT][CK], bv[2][NT][CK];
    // operand addresses: one per-lane base plus wave-uniform offsets (formed here, per slab: six address registers kept
    // across the loop are six registers too many next to 48 accumulators)
    const float* ao[NT];
    const float* bo[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        ao[t] = lane_base + sa[t];
        bo[t] = lane_base + sb[t];
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < CK; ++j) {
            av[0][t][j] = ao[t][2 * j];
            bv[0][t][j] = bo[t][2 * j];
        }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c + 1 < NC) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < CK; ++j) {
                    av[(c + 1) & 1][t][j] = ao[t][2 * (CK * (c + 1) + j)];
                    bv[(c + 1) & 1][t][j] = bo[t][2 * (CK * (c + 1) + j)];
                }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < CK; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c & 1][t][j], bv[c & 1][t][j], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// rows of one LDS slab: the fill writes whole passes of 64 rows
__host__ __device__ inline uint32_t tile_slab_rows(uint32_t ng) { return (ng + 63u) & ~63u; }

// MAXNT: the most tiles a wave of this launch can own (ceil(T / 8) for the launch's ng x mg).  Launches of short lists
// (back-edge lists: 96 rows, six tiles) are compiled without the 32- and 48-accumulator paths and fit three workgroups
// per CU instead of two.
template <typename RT, int MAXNT>
__global__ __launch_bounds__(512, MAXNT == 1 ? 6 : 4) void gram_tiles_kernel(TileArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // two slabs of 32 columns (rows x 33 floats each): slab k + 1 is written while the MFMAs of slab k are in the pipe --
    // one barrier per slab, and the only phase of a workgroup without MFMA work is the first fill
    const uint32_t slab_floats = tile_slab_rows(a.ng) * 33u;
    float* const slab0 = reinterpret_cast<float*>(smem);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, item = a.order ? a.order[blockIdx.x] : blockIdx.x;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));  // wave-uniform for the compiler too
    // Wave w sits on SIMD w % 4, and two workgroups share a CU: odd workgroups count their waves in reverse, so that the
    // waves with one tile more than the others are not on the same SIMDs in both.
    const uint32_t wv = (blockIdx.x & 1u) ? (uint32_t)(kTileRowBlocks - 1) - wave : wave;
    const uint32_t N = a.sn[item];
    const uint32_t nrows = N < a.ng ? N : a.ng, ncols = N < a.mg ? N : a.mg;
    if (nrows == 0) return;
    const uint32_t TR = (nrows + 31u) >> 5, TC = (ncols + 31u) >> 5;
    const uint32_t* ids = a.sid + (uint64_t)item * a.pcap;
    const uint32_t dim = a.ix.dim;
    // slab fill: 8 threads x 4 elements per row, 64 rows per pass.  Every request of the steady state is unconditional
    // (a load under a per-lane condition gets its own s_waitcnt and the prefetch degenerates to one request in flight):
    // rows that do not exist or cannot be retrieved point at row 0 and are zeroed when the slab is written.
    const uint32_t lr = tid >> 3, c4 = (tid & 7u) << 2;
    // (register diet: 48 accumulators + 16 prefetch registers leave little under the 128 of two workgroups per CU -- a
    // row is kept as its id, kEmpty for a zero row, and its address is formed at the request)
    uint32_t rowid[kTilePasses];
    double nsq[kTilePasses];
#pragma unroll
    for (int p = 0; p < kTilePasses; ++p) {
        const uint32_t r = ((uint32_t)p << 6) + lr;
        const uint32_t id = r < nrows ? ids[r] : kEmpty;
        rowid[p] = id < a.ix.nslots ? id : kEmpty;  // ids the sweep excludes anyway (not retrievable) read as zero rows
        nsq[p] = 0.0;
    }
    const uint8_t* const rows_c4 = a.ix.rows + (size_t)c4 * sizeof(RT);
    auto row_ptr = [&](int p) -> const uint8_t* {
        return rows_c4 + (uint64_t)(rowid[p] != kEmpty ? rowid[p] : 0u) * a.ix.row_stride;
    };
    float4 nxt[kTilePasses];
    const uint32_t dim_full = dim & ~31u;  // slabs below this are complete: one 4-element request per thread and row
    // The tiles of the item -- row block b has min(b, TC - 1) + 1 of them, column blocks 0 .. min(b, TC - 1) -- in row-major
    // order q = 0 .. T - 1; wave wv takes q = wv, wv + 8, wv + 16: at most one tile more than any other wave (a wave per
    // row block, as until round 4, leaves 1 + 2 + 3 + 3 + 3 tiles on five waves of a 160-row item: the SIMD with two of
    // them sets the pace of every slab).
    const uint32_t l31 = lane & 31u, hi = lane >> 5;
    uint32_t T = 0;
    for (uint32_t b = 0; b < TR; ++b) T += (b < TC ? b : TC - 1u) + 1u;
    const uint32_t nt = T > wv ? (T - wv + 7u) >> 3 : 0u;  // <= 3: T <= 21
    uint32_t trb[3] = {0, 0, 0}, tcb[3] = {0, 0, 0};
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        uint32_t q = wv + 8u * (uint32_t)t, b = 0;
        if ((uint32_t)t < nt) {
            for (;;) {
                const uint32_t c = (b < TC ? b : TC - 1u) + 1u;
                if (q < c) break;
                q -= c;
                ++b;
            }
            trb[t] = b;
            tcb[t] = q;
        }
    }
    const float* const lane_base = slab0 + l31 * 33u + hi;
    const uint32_t fill_passes = (TR + 1u) >> 1;  // whole passes (wave-uniform): rows past the item's last are zero rows
    auto write_slab = [&](float* slab) {
#pragma unroll
        for (int p = 0; p < kTilePasses; ++p) {
            if ((uint32_t)p < fill_passes) {
                float4 q = nxt[p];
                if (rowid[p] == kEmpty) q = float4{0.f, 0.f, 0.f, 0.f};
                float* dst = slab + (((uint32_t)p << 6) + lr) * 33u + c4;
                dst[0] = q.x, dst[1] = q.y, dst[2] = q.z, dst[3] = q.w;
                nsq[p] += (double)q.x * q.x + (double)q.y * q.y + (double)q.z * q.z + (double)q.w * q.w;
            }
        }
    };
    float* const g = a.gram + (uint64_t)item * a.ng * a.mg;
    // The whole slab loop is instantiated per tile count of the wave (0 .. 3).  A branch on the count inside the loop makes
    // the accumulators a three-way merge in every iteration: the compiler then keeps two copies of all 48 of them (and
    // moves one into the other every slab), which is what drove this kernel into scratch until round 4.
    auto run = [&](auto ntc) {
        constexpr int NT = decltype(ntc)::value;
        f32x16 acc[NT > 0 ? NT : 1];
#pragma unroll
        for (int t = 0; t < (NT > 0 ? NT : 1); ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        uint32_t cur = 0;  // the slab the next MFMAs read: slab0 + cur * slab_floats
        if (dim_full) {
            // steady state: one basic block of requests per slab -- all kTilePasses of them, unconditionally (passes
            // beyond the item's rows re-read row 0 from cache): a request under a branch, even a uniform one, is fenced
            // by its own s_waitcnt.  The last iteration re-requests its own slab instead of branching around the prefetch.
#pragma unroll
            for (int p = 0; p < kTilePasses; ++p) nxt[p] = tile_load4<RT>(row_ptr(p));
            write_slab(slab0);
            __syncthreads();
            for (uint32_t k0 = 0; k0 < dim_full; k0 += 32u) {
                const bool more = k0 + 32u < dim_full;
                const uint32_t kn = more ? k0 + 32u : k0;
#pragma unroll
                for (int p = 0; p < kTilePasses; ++p) nxt[p] = tile_load4<RT>(row_ptr(p) + (size_t)kn * sizeof(RT));
                __builtin_amdgcn_sched_barrier(0);  // the requests go out before the MFMAs, not behind them
                if constexpr (NT > 0) {
                    const uint32_t so = cur * slab_floats;
                    const uint32_t sa[3] = {so + trb[0] * 1056u, so + trb[1] * 1056u, so + trb[2] * 1056u};
                    const uint32_t sb[3] = {so + tcb[0] * 1056u, so + tcb[1] * 1056u, so + tcb[2] * 1056u};
                    tile_slab<NT>(lane_base, sa, sb, acc);
                }
                cur ^= 1u;
                if (more) write_slab(slab0 + cur * slab_floats);  // nobody reads this one before the barrier
                __syncthreads();
            }
        }
        if (dim_full < dim) {  // the last, partial slab (dim % 32 != 0): element-wise requests, once per item
#pragma unroll
            for (int p = 0; p < kTilePasses; ++p)
                nxt[p] = tile_fetch4<RT>(row_ptr(p) - (size_t)c4 * sizeof(RT), dim_full + c4, dim);
            write_slab(slab0 + cur * slab_floats);  // every wave is past the barrier behind the last MFMAs
            __syncthreads();
            if constexpr (NT > 0) {
                const uint32_t so = cur * slab_floats;
                const uint32_t sa[3] = {so + trb[0] * 1056u, so + trb[1] * 1056u, so + trb[2] * 1056u};
                const uint32_t sb[3] = {so + tcb[0] * 1056u, so + tcb[1] * 1056u, so + tcb[2] * 1056u};
                tile_slab<NT>(lane_base, sa, sb, acc);
            }
        }
        // C layout of v_mfma_f32_32x32x2_f32: register r of lane l holds row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31
        if constexpr (NT > 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t i = (trb[t] << 5) + (uint32_t)((r & 3) + 8 * (r >> 2)) + 4u * hi;
                    g[(uint64_t)i * a.mg + (tcb[t] << 5) + l31] = acc[t][r];  // i < 32 TR <= ng, column < 32 TC <= mg
                }
            }
        }
    };
    if (MAXNT >= 3 && nt == 3u) run(std::integral_constant<int, 3>{});
    else if (MAXNT >= 2 && nt == 2u) run(std::integral_constant<int, 2>{});
    else if (nt >= 1u) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 0>{});
    // squared norms: the 8 threads of a row hold f64 partial sums
    float* nr = a.nrm + (uint64_t)item * a.ng;
#pragma unroll
    for (int p = 0; p < kTilePasses; ++p) {
        double v = nsq[p];
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        const uint32_t r = ((uint32_t)p << 6) + lr;
        if ((tid & 7u) == 0 && r < nrows) nr[r] = (float)v;
    }
    if (tid == 0 && a.counters) {
        atomicAdd(&stat_stripe(a.counters)[2], (unsigned long long)nrows);
        atomicAdd(&stat_stripe(a.counters)[3], (unsigned long long)T * 1024ull);
    }
}

// ---- f16 rows on the f16 matrix core (round 6) ---------------------------------------------------------------------------
// gram_tiles_kernel<__half> widens f16 rows and runs v_mfma_f32_32x32x2_f32 -- the f32 matrix core, 1/16 of the f16 rate,
// and 32 LDS operand reads per tile and 32-column slab.  Here the slabs hold the rows' halfs as they are (80-byte row
// stride: the 16-byte operand reads of 16 consecutive rows cover all 64 banks) and a tile advances 16 columns per
// v_mfma_f32_32x32x16_f16: lane l supplies row l & 31, halfs 8 (l >> 5) .. + 7 of the 16-column step, for BOTH operands, so
// whatever order the hardware walks those sixteen k in, row i's k meets row j's k (scratch/probe_mfma_f16.hip: exact on
// integer data, 1 - 9 u sum|x y| off the exact product on random rows of 16 .. 1536 columns -- a third of the f32 chain's
// error).  Products of two halfs are exact in f32; the sum is not the reference's FMA chain, and does not have to be: every
// decision of the sweep goes through the error interval E = c1 (|x|^2 + |y|^2) + c2 |d'| (c1 = the chain's bound, which
// covers any summation order of exactly rounded or truncated f32 additions with room to spare) and falls back to the
// bit-exact pair kernel where E does not decide.  Same tiling, same dealing of tiles to waves, same
// output as gram_tiles_kernel.
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
constexpr uint32_t kF16SlabRowBytes = 80;  // 32 halfs + 16 bytes: conflict-free 16-byte operand reads
template <int NT>
__device__ __forceinline__ void tile_slab_f16(const uint8_t* lane_base, const uint32_t (&sa)[3], const uint32_t (&sb)[3],
                                              f32x16 (&acc)[NT]) {
    f16x8_t av[NT][2], bv[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            av[t][m] = *reinterpret_cast<const f16x8_t*>(lane_base + sa[t] + 32u * m);
            bv[t][m] = *reinterpret_cast<const f16x8_t*>(lane_base + sb[t] + 32u * m);
        }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[t][m], bv[t][m], acc[t], 0, 0, 0);
}

template <int MAXNT>
__global__ __launch_bounds__(512, MAXNT == 1 ? 6 : 4) void gram_tiles_f16_kernel(TileArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t slab_bytes = tile_slab_rows(a.ng) * kF16SlabRowBytes;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, item = a.order ? a.order[blockIdx.x] : blockIdx.x;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t wv = (blockIdx.x & 1u) ? (uint32_t)(kTileRowBlocks - 1) - wave : wave;
    const uint32_t N = a.sn[item];
    const uint32_t nrows = N < a.ng ? N : a.ng, ncols = N < a.mg ? N : a.mg;
    if (nrows == 0) return;
    const uint32_t TR = (nrows + 31u) >> 5, TC = (ncols + 31u) >> 5;
    const uint32_t* ids = a.sid + (uint64_t)item * a.pcap;
    const uint32_t dim = a.ix.dim;
    // slab fill: 8 threads x 4 halfs per row, 64 rows per pass; unconditional requests (see gram_tiles_kernel)
    const uint32_t lr = tid >> 3, c4 = (tid & 7u) << 2;
    uint32_t rowid[kTilePasses];
    double nsq[kTilePasses];
#pragma unroll
    for (int p = 0; p < kTilePasses; ++p) {
        const uint32_t r = ((uint32_t)p << 6) + lr;
        const uint32_t id = r < nrows ? ids[r] : kEmpty;
        rowid[p] = id < a.ix.nslots ? id : kEmpty;
        nsq[p] = 0.0;
    }
    const uint8_t* const rows_c4 = a.ix.rows + (size_t)c4 * 2u;
    auto row_ptr = [&](int p) -> const uint8_t* {
        return rows_c4 + (uint64_t)(rowid[p] != kEmpty ? rowid[p] : 0u) * a.ix.row_stride;
    };
    uint2 nxt[kTilePasses];
    const uint32_t dim_full = dim & ~31u;
    const uint32_t l31 = lane & 31u, hi = lane >> 5;
    uint32_t T = 0;
    for (uint32_t b = 0; b < TR; ++b) T += (b < TC ? b : TC - 1u) + 1u;
    const uint32_t nt = T > wv ? (T - wv + 7u) >> 3 : 0u;
    uint32_t trb[3] = {0, 0, 0}, tcb[3] = {0, 0, 0};
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        uint32_t q = wv + 8u * (uint32_t)t, b = 0;
        if ((uint32_t)t < nt) {
            for (;;) {
                const uint32_t c = (b < TC ? b : TC - 1u) + 1u;
                if (q < c) break;
                q -= c;
                ++b;
            }
            trb[t] = b;
            tcb[t] = q;
        }
    }
    const uint8_t* const lane_base = smem + l31 * kF16SlabRowBytes + hi * 16u;
    const uint32_t fill_passes = (TR + 1u) >> 1;
    auto write_slab = [&](uint8_t* slab) {
#pragma unroll
        for (int p = 0; p < kTilePasses; ++p) {
            if ((uint32_t)p < fill_passes) {
                uint2 q = nxt[p];
                if (rowid[p] == kEmpty) q = make_uint2(0u, 0u);
                *reinterpret_cast<uint2*>(slab + (((uint32_t)p << 6) + lr) * kF16SlabRowBytes + c4 * 2u) = q;
                const float2 x = __half22float2(__builtin_bit_cast(__half2, q.x)), y = __half22float2(__builtin_bit_cast(__half2, q.y));
                nsq[p] += (double)x.x * x.x + (double)x.y * x.y + (double)y.x * y.x + (double)y.y * y.y;
            }
        }
    };
    auto fetch_tail = [&](int p) -> uint2 {  // the partial slab: elements beyond the row's end read as zero
        const __half* r = reinterpret_cast<const __half*>(row_ptr(p) - (size_t)c4 * 2u);
        const uint32_t k = dim_full + c4;
        __half h[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (k + (uint32_t)e < dim) ? r[k + (uint32_t)e] : __float2half(0.0f);
        uint2 q;
        q.x = (uint32_t)__builtin_bit_cast(uint16_t, h[0]) | ((uint32_t)__builtin_bit_cast(uint16_t, h[1]) << 16);
        q.y = (uint32_t)__builtin_bit_cast(uint16_t, h[2]) | ((uint32_t)__builtin_bit_cast(uint16_t, h[3]) << 16);
        return q;
    };
    float* const g = a.gram + (uint64_t)item * a.ng * a.mg;
    auto run = [&](auto ntc) {
        constexpr int NT = decltype(ntc)::value;
        f32x16 acc[NT > 0 ? NT : 1];
#pragma unroll
        for (int t = 0; t < (NT > 0 ? NT : 1); ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        uint32_t cur = 0;
        constexpr uint32_t kBlock = 32u * kF16SlabRowBytes;  // bytes of one 32-row block of a slab
        if (dim_full) {
#pragma unroll
            for (int p = 0; p < kTilePasses; ++p) nxt[p] = *reinterpret_cast<const uint2*>(row_ptr(p));
            write_slab(smem);
            __syncthreads();
            for (uint32_t k0 = 0; k0 < dim_full; k0 += 32u) {
                const bool more = k0 + 32u < dim_full;
                const uint32_t kn = more ? k0 + 32u : k0;
#pragma unroll
                for (int p = 0; p < kTilePasses; ++p) nxt[p] = *reinterpret_cast<const uint2*>(row_ptr(p) + (size_t)kn * 2u);
                __builtin_amdgcn_sched_barrier(0);  // the requests go out before the MFMAs, not behind them
                if constexpr (NT > 0) {
                    const uint32_t so = cur * slab_bytes;
                    const uint32_t sa[3] = {so + trb[0] * kBlock, so + trb[1] * kBlock, so + trb[2] * kBlock};
                    const uint32_t sb[3] = {so + tcb[0] * kBlock, so + tcb[1] * kBlock, so + tcb[2] * kBlock};
                    tile_slab_f16<NT>(lane_base, sa, sb, acc);
                }
                cur ^= 1u;
                if (more) write_slab(smem + cur * slab_bytes);
                __syncthreads();
            }
        }
        if (dim_full < dim) {
#pragma unroll
            for (int p = 0; p < kTilePasses; ++p) nxt[p] = fetch_tail(p);
            write_slab(smem + cur * slab_bytes);
            __syncthreads();
            if constexpr (NT > 0) {
                const uint32_t so = cur * slab_bytes;
                const uint32_t sa[3] = {so + trb[0] * kBlock, so + trb[1] * kBlock, so + trb[2] * kBlock};
                const uint32_t sb[3] = {so + tcb[0] * kBlock, so + tcb[1] * kBlock, so + tcb[2] * kBlock};
                tile_slab_f16<NT>(lane_base, sa, sb, acc);
            }
        }
        // C layout as v_mfma_f32_32x32x2_f32: register r of lane l holds row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31
        if constexpr (NT > 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t i = (trb[t] << 5) + (uint32_t)((r & 3) + 8 * (r >> 2)) + 4u * hi;
                    g[(uint64_t)i * a.mg + (tcb[t] << 5) + l31] = acc[t][r];
                }
            }
        }
    };
    if (MAXNT >= 3 && nt == 3u) run(std::integral_constant<int, 3>{});
    else if (MAXNT >= 2 && nt == 2u) run(std::integral_constant<int, 2>{});
    else if (nt >= 1u) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 0>{});
    float* nr = a.nrm + (uint64_t)item * a.ng;
#pragma unroll
    for (int p = 0; p < kTilePasses; ++p) {
        double v = nsq[p];
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        const uint32_t r = ((uint32_t)p << 6) + lr;
        if ((tid & 7u) == 0 && r < nrows) nr[r] = (float)v;
    }
    if (tid == 0 && a.counters) {
        atomicAdd(&stat_stripe(a.counters)[2], (unsigned long long)nrows);
        atomicAdd(&stat_stripe(a.counters)[3], (unsigned long long)T * 1024ull);
    }
}

// Longest lists first: a sweep is a serial walk over its list, and a launch in item order ends with a few long lists on an
// otherwise idle chip (1 M x 768, 16 384 pools per launch: 6 of 13 wavefront slots per CU occupied on average).  The
// workgroups of the Gram and sweep kernels take their items in descending order of list length instead -- a counting
// sort by length, one workgroup (the order among equal lengths is whatever the atomics give: no result depends on it).
constexpr uint32_t kOrderBins = kMaxPool + 1u;
__global__ __launch_bounds__(1024) void lpt_order_kernel(const uint32_t* sn, uint32_t m, uint32_t* order) {
    __shared__ uint32_t hist[kOrderBins];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < kOrderBins; i += 1024u) hist[i] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < m; i += 1024u) atomicAdd(&hist[sn[i] < kMaxPool ? sn[i] : kMaxPool], 1u);
    __syncthreads();
    if (tid < 64u) {  // exclusive scan from the longest bin down: 64 bins per step
        uint32_t acc = 0;
        for (int base = (int)kOrderBins - 1; base >= 0; base -= 64) {
            const int b = base - (int)tid;
            const uint32_t c = b >= 0 ? hist[b] : 0u;
            uint32_t inc = c;
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_up(inc, o);
                if ((int)tid >= o) inc += t;
            }
            if (b >= 0) hist[b] = acc + inc - c;
            acc += __shfl(inc, 63);
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < m; i += 1024u) order[atomicAdd(&hist[sn[i] < kMaxPool ? sn[i] : kMaxPool], 1u)] = i;
}

struct SweepArgs {
    PoolArgs p;
    const uint32_t* sid;
    const float* sd;
    const uint32_t* sn;
    const float* gram;
    const float* nrm;
    uint32_t ng, mg;
    float escale, c1, c2;
    // back-edge prunes (add_edge_and_prune): the location of item i is out_loc[i] and the result goes straight into its
    // adjacency row (nobody else reads that row during the back-edge phase); null = pool prune (p.locs, p.out)
    const uint32_t* out_loc = nullptr;
    uint32_t one_by_one = 0;         // development switch (DANN_DBG_SWEEP_ONE_BY_ONE): the candidate-at-a-time sweep
    const uint32_t* order = nullptr; // optional: workgroup b works on item order[b] (longest lists first)
    uint32_t compact_lds = 0;        // sweep_lds_layout instead of pool_lds_layout (the batched sweep only)
};

// the batched sweep (sweep_gram_batched) serves this launch: one lane per selected entry, at most 128 Gram columns
inline bool sweep_is_batched(const PruneCfg& cfg, uint32_t mg, uint32_t one_by_one) {
    return cfg.pruned_degree <= (uint32_t)kWave && mg <= 128u && !one_by_one;
}
inline size_t sweep_lds_bytes(const SweepArgs& sw) {
    const uint32_t pool = sw.compact_lds ? sweep_lds_layout(sw.p.pcap, sw.p.cfg.pruned_degree).total
                                         : pool_lds_layout(sw.p.pcap, sw.p.cfg.pruned_degree).total;
    return (((size_t)pool + 15u) & ~(size_t)15u) + kSweepRowsLds;
}

template <int DT, int OP, bool NORM>
__global__ __launch_bounds__(kWave) void pool_sweep_kernel(SweepArgs sa) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const PoolArgs& a = sa.p;
    const uint32_t lane = threadIdx.x, wi = sa.order ? sa.order[blockIdx.x] : blockIdx.x, item = a.pos0 + wi;
    const uint32_t N = sa.sn[wi];
    if (N == 0) {  // pool prune: overflow (reported by pool_sort_kernel) or an empty pool; back-edges: nothing to prune
        if (lane == 0 && !sa.out_loc) a.out[(uint64_t)wi * a.out_stride] = 0;
        return;
    }
    const PoolLds L = sa.compact_lds ? sweep_lds_layout(a.pcap, a.cfg.pruned_degree) : pool_lds_layout(a.pcap, a.cfg.pruned_degree);
    uint32_t* sid = reinterpret_cast<uint32_t*>(smem + L.sid_off);
    float* sd = reinterpret_cast<float*>(smem + L.sd_off);
    float* occ = reinterpret_cast<float*>(smem + L.occ_off);
    uint16_t* last = reinterpret_cast<uint16_t*>(smem + L.last_off);
    const uint32_t* gs = sa.sid + (uint64_t)wi * a.pcap;
    const float* gd = sa.sd + (uint64_t)wi * a.pcap;
    for (uint32_t i = lane; i < N; i += kWave) {
        sid[i] = gs[i];
        sd[i] = gd[i];
        occ[i] = 0.0f;
        last[i] = 0;
    }
    wave_sync();
    const uint32_t nr = N < sa.ng ? N : sa.ng, nc = N < sa.mg ? N : sa.mg;
    const uint32_t location = sa.out_loc ? sa.out_loc[wi] : a.locs[item];
    uint32_t* out = sa.out_loc ? a.ix.adj + (uint64_t)location * a.ix.adj_stride : a.out + (uint64_t)wi * a.out_stride;
    const GramCtx gc{sa.gram + (uint64_t)wi * sa.ng * sa.mg, sa.nrm + (uint64_t)wi * sa.ng, sa.mg,
                     nr, nc, true, sa.escale, sa.c1, sa.c2, false, true};
    if (sa.compact_lds)
        sweep_gram_batched<DT, OP, NORM>(a.ix, a.cfg, location, N, smem, L, a.force_saturate != 0, out, gc,
                                         reinterpret_cast<float*>(smem + ((L.total + 15u) & ~15u)));
    else
        sweep_sorted_pool_gram<DT, OP, NORM>(a.ix, a.cfg, location, N, smem, L, a.force_saturate != 0, out, gc);
    if (lane == 0 && sa.out_loc && a.cfg.counters) atomicAdd(&stat_stripe(a.cfg.counters)[6], 1ull);
}

// back-edges on the matrix cores, first of three kernels (the other two are gram_tiles_kernel and pool_sweep_kernel):
// one wave per target of the short worklist builds the list adj(target) ++ unique new sources exactly as
// backedge_kernel does, appends when it still fits, and otherwise evaluates d(target, c) for the list, sorts it
// (SortedNeighbors::new) and leaves the sorted (id, distance) list in global memory for the Gram and the sweep.
struct BackListArgs {
    BackArgs b;
    uint32_t* sid;   // n x pcap
    float* sd;
    uint32_t* sn;    // n: sorted list length, 0 = nothing to prune for this item
    uint32_t* loc;   // n: the target
};

template <int DT, int OP, bool NORM>
__global__ __launch_bounds__(kWave) void backedge_list_kernel(BackListArgs la) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const BackArgs& a = la.b;
    const uint32_t lane = threadIdx.x, wi = blockIdx.x, seg = a.work ? a.work[wi] : wi;
    const PoolLds L = pool_lds_layout(a.pcap, a.cfg.pruned_degree);
    uint32_t* pid = reinterpret_cast<uint32_t*>(smem + L.pid_off);
    float* pd = reinterpret_cast<float*>(smem + L.pd_off);
    const uint32_t* sid = reinterpret_cast<const uint32_t*>(smem + L.sid_off);
    const float* sd = reinterpret_cast<const float*>(smem + L.sd_off);
    const uint32_t start = a.seg_start[seg];
    const uint32_t src = (uint32_t)(a.keys[start] >> 32);
    uint32_t* arow = a.ix.adj + (uint64_t)src * a.ix.adj_stride;
    if (lane == 0) {
        la.sn[wi] = 0;
        la.loc[wi] = src;
    }
    uint32_t len = arow[0];
    len = len < a.ix.max_degree ? len : a.ix.max_degree;
    for (uint32_t i = lane; i < len; i += kWave) pid[i] = arow[1 + i];
    wave_sync();
    // extend_from_slice(sorted sources): unique append
    uint32_t cnt = len;
    const uint32_t end = start + a.seg_len[start];
    for (uint32_t k0 = start; k0 < end; k0 += kWave) {
        const uint32_t k = k0 + lane;
        const bool mine = k < end;
        const uint32_t id = mine ? (uint32_t)a.keys[k] : kEmpty;
        bool take = mine;
        if (mine)
            for (uint32_t e = 0; e < len; ++e) take &= (pid[e] != id);
        const uint64_t tm = ballot64(take);
        const uint32_t ntake = (uint32_t)__popcll(tm);
        if (cnt + ntake > a.pcap) {
            if (lane == 0) *a.err = 1;
            return;
        }
        if (take) pid[cnt + mbcnt(tm)] = id;
        cnt += ntake;
    }
    wave_sync();
    const uint32_t added = cnt - len;
    if (added == 0) return;
    if (cnt <= a.cfg.max_degree) {  // append_vector with the provider-capacity clamp (provider.rs:795-822)
        const uint32_t slack = a.ix.max_degree - len;
        const uint32_t take = added < slack ? added : slack;
        for (uint32_t i = lane; i < take; i += kWave) arow[1 + len + i] = pid[len + i];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            arow[0] = len + take;
        }
        return;
    }
    fill_list_distances<DT, OP, NORM>(a.ix, src, pid, pd, cnt);
    if (lane == 0 && a.cfg.counters) atomicAdd(&stat_stripe(a.cfg.counters)[1], (unsigned long long)cnt);
    wave_sync();
    const uint32_t N = sort_pool_wave(a.cfg, cnt, a.pcap, smem, L);
    uint32_t* gs = la.sid + (uint64_t)wi * a.pcap;
    float* gd = la.sd + (uint64_t)wi * a.pcap;
    for (uint32_t i = lane; i < N; i += kWave) {
        gs[i] = sid[i];
        gd[i] = sd[i];
    }
    if (lane == 0) la.sn[wi] = N;
}

// ---- small utility kernels -------------------------------------------------------------------
// `limit`: back-edges go to the first `limit` neighbours of a new point only (DiskANNIndex::insert takes
// max_backedges of them, index.rs:324-327; multi_insert all, index.rs:123-143)
__global__ void make_keys_kernel(const uint32_t* locs, const uint32_t* pending, uint32_t pend_stride, uint32_t n,
                                 uint32_t degree, uint64_t* keys, uint32_t limit) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * degree) return;
    const uint32_t item = t / degree, e = t % degree;
    const uint32_t* row = pending + (uint64_t)item * pend_stride;
    keys[t] = (e < row[0] && e < limit) ? (((uint64_t)row[1 + e] << 32) | locs[item]) : ~0ull;
}

__global__ void segment_kernel(const uint64_t* keys, uint32_t total, uint32_t* seg_start, uint32_t* meta) {
    // meta[0] = #segments, meta[1] = #valid keys (invalid keys sort to the end)
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const uint64_t k = keys[t];
    if (k == ~0ull) return;
    if (t + 1 == total || keys[t + 1] == ~0ull) meta[1] = t + 1;
    if (t == 0 || (uint32_t)(keys[t - 1] >> 32) != (uint32_t)(k >> 32)) seg_start[atomicAdd(&meta[0], 1u)] = t;
}

__global__ void seglen_kernel(const uint64_t* keys, const uint32_t* seg_start, uint32_t* seg_len, uint32_t* meta) {
    // meta[2] = max segment length
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= meta[0]) return;
    const uint32_t st = seg_start[s];
    const uint32_t tgt = (uint32_t)(keys[st] >> 32);
    uint32_t e = st + 1;
    while (e < meta[1] && (uint32_t)(keys[e] >> 32) == tgt) ++e;
    seg_len[st] = e - st;
    // (a maximum only grows: skip the atomic when a plain read already shows it -- one address, a million segments)
    if (e - st > __hip_atomic_load(&meta[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&meta[2], e - st);
}

__global__ void set_bulk_kernel(IndexView ix, const uint32_t* locs, const uint32_t* pending, uint32_t pend_stride,
                                uint32_t n) {
    const uint32_t item = blockIdx.x;
    if (item >= n) return;
    const uint32_t* row = pending + (uint64_t)item * pend_stride;
    uint32_t* arow = ix.adj + (uint64_t)locs[item] * ix.adj_stride;
    const uint32_t len = row[0];
    for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) arow[1 + i] = row[1 + i];
    if (threadIdx.x == 0) arow[0] = len;
}

// ---- dispatch helpers ------------------------------------------------------------------------
template <template <int, int, bool> class Launcher, class Args>
int32_t dispatch(const IndexView& ix, const Args& a, uint32_t grid, size_t lds, hipStream_t stream) {
    int op;
    bool norm;
    if (!resolve_metric(ix.dtype, ix.metric, &op, &norm)) {
        set_error("metric %d is not defined for dtype %d", ix.metric, ix.dtype);
        return DANN_EUNSUPPORTED;
    }
#define DANN_CASE(DT)                                                                                  \
    case DT:                                                                                           \
        if (op == OP_L2) {                                                                             \
            if constexpr (DT == DT_SQ8) {                                                              \
                if (norm) return Launcher<DT, OP_L2, true>::run(a, grid, lds, stream);                 \
            }                                                                                          \
            return Launcher<DT, OP_L2, false>::run(a, grid, lds, stream);                              \
        }                                                                                              \
        if (op == OP_IP) {                                                                             \
            if constexpr (DT == DT_F32 || DT == DT_F16) {                                              \
                if (norm) return Launcher<DT, OP_IP, true>::run(a, grid, lds, stream);                 \
            }                                                                                          \
            return Launcher<DT, OP_IP, false>::run(a, grid, lds, stream);                              \
        }                                                                                              \
        if constexpr (DT != DT_SQ8) return Launcher<DT, OP_COS, false>::run(a, grid, lds, stream);     \
        return DANN_EUNSUPPORTED;
    switch (ix.dtype) {
        DANN_CASE(DT_F32)
        DANN_CASE(DT_F16)
        DANN_CASE(DT_U8)
        DANN_CASE(DT_I8)
        DANN_CASE(DT_SQ8)
    }
#undef DANN_CASE
    set_error("bad dtype %d", ix.dtype);
    return DANN_EINVAL;
}

#define DANN_LAUNCHER(NAME, KERNEL, ARGS)                                                          \
    template <int DT, int OP, bool NORM>                                                           \
    struct NAME {                                                                                  \
        static int32_t run(const ARGS& a, uint32_t grid, size_t lds, hipStream_t stream) {         \
            auto kern = KERNEL<DT, OP, NORM>;                                                      \
            if (lds > 64 * 1024) {                                                                 \
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),            \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute");                    \
            }                                                                                      \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(kWave), lds, stream, a);                     \
            hipError_t e = hipGetLastError();                                                      \
            if (e != hipSuccess) return hip_fail(e, #KERNEL " launch");                            \
            return DANN_OK;                                                                        \
        }                                                                                          \
    };
DANN_LAUNCHER(PoolLauncher, pool_prune_kernel, PoolArgs)
DANN_LAUNCHER(BootLauncher, bootstrap_kernel, ListArgs)
DANN_LAUNCHER(BackLauncher, backedge_kernel, BackArgs)

// the three kernels of the MFMA pool prune: float rows (f32 / f16), L2 / inner product / cosine-normalized
template <template <int, int, bool> class K, class Args>
int32_t dispatch_float(const IndexView& ix, const Args& a, uint32_t grid, size_t lds, hipStream_t stream) {
    int op;
    bool norm;
    if (!resolve_metric(ix.dtype, ix.metric, &op, &norm) || op == OP_COS || (ix.dtype != DT_F32 && ix.dtype != DT_F16))
        return DANN_EUNSUPPORTED;
    if (ix.dtype == DT_F32) {
        if (op == OP_L2) return K<DT_F32, OP_L2, false>::run(a, grid, lds, stream);
        return norm ? K<DT_F32, OP_IP, true>::run(a, grid, lds, stream) : K<DT_F32, OP_IP, false>::run(a, grid, lds, stream);
    }
    if (op == OP_L2) return K<DT_F16, OP_L2, false>::run(a, grid, lds, stream);
    return norm ? K<DT_F16, OP_IP, true>::run(a, grid, lds, stream) : K<DT_F16, OP_IP, false>::run(a, grid, lds, stream);
}
DANN_LAUNCHER(SortLauncher, pool_sort_kernel, SortArgs)
DANN_LAUNCHER(SweepLauncher, pool_sweep_kernel, SweepArgs)
DANN_LAUNCHER(BackListLauncher, backedge_list_kernel, BackListArgs)

// f16_mfma: f16 rows through v_mfma_f32_32x32x16_f16 (gram_tiles_f16_kernel; the default) instead of widened through the
// f32 matrix core (DANN_DBG_GRAM_F16_WIDEN: the round-3 form, whose entries are the reference's FMA chain bit for bit)
int32_t launch_gram_tiles(const TileArgs& a, uint32_t grid, hipStream_t stream, bool f16_mfma) {
    if (a.ix.dtype != DT_F32 && a.ix.dtype != DT_F16) return DANN_EUNSUPPORTED;
    const bool f32 = a.ix.dtype == DT_F32;
    const bool h16 = !f32 && f16_mfma;
    const size_t lds = h16 ? 2u * (size_t)tile_slab_rows(a.ng) * kF16SlabRowBytes
                           : 2u * (size_t)tile_slab_rows(a.ng) * 33u * 4u;  // two slabs
    // the most tiles a wave of this launch can own: T tiles of a full item over eight waves
    const uint32_t tr = a.ng >> 5, tc = a.mg >> 5;
    uint32_t tmax = 0;
    for (uint32_t b2 = 0; b2 < tr; ++b2) tmax += (b2 < tc ? b2 : tc - 1u) + 1u;
    const bool narrow = tmax <= 8u;
    const void* fn = h16 ? (narrow ? (const void*)gram_tiles_f16_kernel<1> : (const void*)gram_tiles_f16_kernel<3>)
                   : narrow ? (f32 ? (const void*)gram_tiles_kernel<float, 1> : (const void*)gram_tiles_kernel<__half, 1>)
                            : (f32 ? (const void*)gram_tiles_kernel<float, 3> : (const void*)gram_tiles_kernel<__half, 3>);
    if (lds > 64u * 1024u) {  // beyond the default limit of dynamic LDS
        hipError_t ea = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (ea != hipSuccess) return hip_fail(ea, "hipFuncSetAttribute");
    }
    if (h16) {
        if (narrow) hipLaunchKernelGGL((gram_tiles_f16_kernel<1>), dim3(grid), dim3(512), lds, stream, a);
        else hipLaunchKernelGGL((gram_tiles_f16_kernel<3>), dim3(grid), dim3(512), lds, stream, a);
    } else if (narrow) {
        if (f32) hipLaunchKernelGGL((gram_tiles_kernel<float, 1>), dim3(grid), dim3(512), lds, stream, a);
        else hipLaunchKernelGGL((gram_tiles_kernel<__half, 1>), dim3(grid), dim3(512), lds, stream, a);
    } else {
        if (f32) hipLaunchKernelGGL((gram_tiles_kernel<float, 3>), dim3(grid), dim3(512), lds, stream, a);
        else hipLaunchKernelGGL((gram_tiles_kernel<__half, 3>), dim3(grid), dim3(512), lds, stream, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gram_tiles_kernel launch");
    return DANN_OK;
}

uint32_t next_pow2(uint32_t x) {
    uint32_t p = 64;
    while (p < x) p <<= 1;
    return p;
}

// development switches (dann_debug_set): DANN_DBG_SWEEP_ONE_BY_ONE; DANN_DBG_POOL_GRAM = 0 keeps the row kernel for the
// pool prune of large rows (A/B runs)
uint32_t sweep_one_by_one(const dann_index* idx) { return idx->dbg_u32(DANN_DBG_SWEEP_ONE_BY_ONE, 0u) ? 1u : 0u; }
bool pool_gram_default(const dann_index* idx) { return idx->dbg_u32(DANN_DBG_POOL_GRAM, 1u) != 0u; }

PruneCfg to_prune_cfg(const dann_index* idx, const dann_build_config& c) {
    PruneCfg p;
    p.tie_order = idx->prune_tie_order;
    p.pruned_degree = c.pruned_degree;
    p.max_degree = c.max_degree;
    p.max_occlusion = c.max_occlusion_size ? c.max_occlusion_size : 750;
    p.alpha = c.alpha;
    p.saturate_after_prune = c.saturate_after_prune;
    p.counters = nullptr;
    return p;
}

int32_t validate_cfg(const dann_index* idx, const dann_build_config* cfg) {
    if (!cfg) return DANN_EINVAL;
    if (idx->cfg.dtype == DT_PQ) {
        set_error("index build / prune are not defined on DANN_PQ rows (build on the full-precision index)");
        return DANN_EUNSUPPORTED;
    }
    if (cfg->pruned_degree == 0 || cfg->l_build == 0 || cfg->max_degree < cfg->pruned_degree ||
        cfg->max_degree > idx->cfg.max_degree || !(cfg->alpha >= 1.0f)) {
        set_error("invalid build config (pruned_degree %u, max_degree %u (provider %u), l_build %u, alpha %g)",
                  cfg->pruned_degree, cfg->max_degree, idx->cfg.max_degree, cfg->l_build, (double)cfg->alpha);
        return DANN_EINVAL;
    }
    if (cfg->max_occlusion_size > kMaxPool) {
        set_error("max_occlusion_size %u exceeds the supported %u", cfg->max_occlusion_size, kMaxPool);
        return DANN_EUNSUPPORTED;
    }
    return DANN_OK;
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    hipError_t alloc(size_t n) {
        release();
        return hipMalloc(&p, n ? n : 1);
    }
    template <class T>
    T* as() {
        return reinterpret_cast<T*>(p);
    }
};

struct DeviceGuard {
    int prev = -1, cur = -1;
    explicit DeviceGuard(int dev) : cur(dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceGuard() {
        if (prev >= 0 && prev != cur) (void)hipSetDevice(prev);  // hipSetDevice costs ~0.5 ms on ROCm 7.2
    }
};

// persistent per-index scratch for batched inserts
struct BuildScratch {
    uint32_t batch_cap = 0, rec_stride = 0, pend_stride = 0, degree = 0;
    bool bootstrap_too_big = false;  // set when a batch needed the bootstrap with more members than its pool holds
    DevBuf slots, rec_ids, rec_d, rec_n, stats, pending, pending2, keys_in, keys_out, seg_start, seg_len, meta, sort_tmp;
    DevBuf counters;  // kStatStripes x 8 x u64, see PruneCfg::counters (accumulate over the index's lifetime)
    size_t sort_tmp_bytes = 0;
    // MFMA pool prune: sorted pools, Gram blocks and norms of one batch slice (grow-only)
    DevBuf g_sid, g_sd, g_sn, g_loc, g_gram, g_nrm, g_order;
    size_t g_sorted_elems = 0, g_items = 0, g_gram_elems = 0, g_nrm_elems = 0;
    // HIP-event time of the gram_tiles_kernel launches (dann_kernel_time, which = 5): pairs of events around every
    // launch on the build stream, resolved without ever making the build wait (a pair is read once it has completed,
    // or when the ring comes round to it -- hundreds of launches later)
    static constexpr uint32_t kTileEvents = 64;
    hipEvent_t tile_ev[2 * kTileEvents] = {};
    uint32_t tile_head = 0, tile_tail = 0;  // pairs [tail, head) are pending
    double tile_ms = 0.0;
    uint64_t tile_launches = 0;
    void tile_drain(bool all) {
        while (tile_tail != tile_head) {
            hipEvent_t e0 = tile_ev[2 * (tile_tail % kTileEvents)], e1 = tile_ev[2 * (tile_tail % kTileEvents) + 1];
            if (!all && tile_head - tile_tail < kTileEvents && hipEventQuery(e1) != hipSuccess) break;
            float ms = 0.f;
            if (hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) {
                tile_ms += ms;
                tile_launches += 1;
            }
            ++tile_tail;
        }
    }
    int32_t tile_begin(hipStream_t st) {
        tile_drain(false);
        hipEvent_t& e0 = tile_ev[2 * (tile_head % kTileEvents)];
        hipEvent_t& e1 = tile_ev[2 * (tile_head % kTileEvents) + 1];
        if (!e0) DANN_HIP(hipEventCreate(&e0));
        if (!e1) DANN_HIP(hipEventCreate(&e1));
        DANN_HIP(hipEventRecord(e0, st));
        return DANN_OK;
    }
    int32_t tile_end(hipStream_t st) {
        DANN_HIP(hipEventRecord(tile_ev[2 * (tile_head % kTileEvents) + 1], st));
        ++tile_head;
        return DANN_OK;
    }
    // second stream of a commit: the few long back-edge lists of a batch (one wavefront each, a serial prune of hundreds
    // of candidates) run beside the short lists instead of holding the chip for themselves
    hipStream_t side = nullptr;
    int32_t side_stream(hipStream_t* out) {
        if (!side) DANN_HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
        *out = side;
        return DANN_OK;
    }
    ~BuildScratch() {
        for (hipEvent_t e : tile_ev)
            if (e) (void)hipEventDestroy(e);
        if (side) (void)hipStreamDestroy(side);
    }
};

int32_t ensure_scratch(BuildScratch& s, uint32_t batch, uint32_t rec_stride, uint32_t degree) {
    if (batch <= s.batch_cap && rec_stride == s.rec_stride && degree == s.degree) return DANN_OK;
    const uint32_t cap = std::max<uint32_t>(batch, 1024);
    s.batch_cap = 0;  // committed only after every allocation succeeded (a failed hipMalloc must not pass the cache test)
    s.pend_stride = degree + 1;
    const size_t nkeys = (size_t)cap * degree;
    DANN_HIP(s.slots.alloc((size_t)cap * 4));
    DANN_HIP(s.rec_ids.alloc((size_t)cap * rec_stride * 4));
    DANN_HIP(s.rec_d.alloc((size_t)cap * rec_stride * 4));
    DANN_HIP(s.rec_n.alloc((size_t)cap * 4));
    DANN_HIP(s.stats.alloc((size_t)cap * sizeof(dann_search_stats)));
    DANN_HIP(s.pending.alloc((size_t)cap * s.pend_stride * 4));
    DANN_HIP(s.pending2.alloc((size_t)cap * s.pend_stride * 4));
    DANN_HIP(s.keys_in.alloc(nkeys * 8));
    DANN_HIP(s.keys_out.alloc(nkeys * 8));
    DANN_HIP(s.seg_start.alloc(nkeys * 4));
    DANN_HIP(s.seg_len.alloc(nkeys * 4 * 3));  // segment lengths | short worklist | long worklist
    DANN_HIP(s.meta.alloc(128));  // 16 words of flags and counts + the insert searches' totals (stats_reduce_kernel)
    DANN_HIP(hipMemset(s.meta.p, 0, 64));  // the commit phase may run first on this handle (sharded build: empty slice)
    if (!s.counters.p) {
        DANN_HIP(s.counters.alloc((size_t)kStatStripes * 64 + 64));
        DANN_HIP(hipMemset(s.counters.p, 0, (size_t)kStatStripes * 64 + 64));
    }
    size_t tmp = 0;
    DANN_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp, s.keys_in.as<uint64_t>(), s.keys_out.as<uint64_t>(),
                                               (int)nkeys, 0, 64, nullptr));
    s.sort_tmp_bytes = tmp;
    DANN_HIP(s.sort_tmp.alloc(tmp));
    s.batch_cap = cap;
    s.rec_stride = rec_stride;
    s.degree = degree;
    return DANN_OK;
}

// grow-only buffers of the MFMA prunes: sorted lists, Gram blocks and norms of `items` lists
int32_t ensure_gram_scratch(BuildScratch& s, size_t items, uint32_t pcap, uint32_t ng, uint32_t mg) {
    const size_t sorted_elems = items * pcap, gram_elems = items * ng * mg, nrm_elems = items * ng;
    if (s.g_sorted_elems < sorted_elems) {
        s.g_sorted_elems = 0;
        DANN_HIP(s.g_sid.alloc(sorted_elems * 4));
        DANN_HIP(s.g_sd.alloc(sorted_elems * 4));
        s.g_sorted_elems = sorted_elems;
    }
    if (s.g_items < items) {
        s.g_items = 0;
        DANN_HIP(s.g_sn.alloc(items * 4));
        DANN_HIP(s.g_loc.alloc(items * 4));
        DANN_HIP(s.g_order.alloc(items * 4));
        s.g_items = items;
    }
    if (s.g_gram_elems < gram_elems) {
        s.g_gram_elems = 0;
        DANN_HIP(s.g_gram.alloc(gram_elems * 4));
        s.g_gram_elems = gram_elems;
    }
    if (s.g_nrm_elems < nrm_elems) {
        s.g_nrm_elems = 0;
        DANN_HIP(s.g_nrm.alloc(nrm_elems * 4));
        s.g_nrm_elems = nrm_elems;
    }
    return DANN_OK;
}

}  // namespace

// what the host wants from the m insert searches of a slice: comparisons, hops, the first search that failed -- three
// words instead of m records through a pageable copy (327 KB per 16 384-point batch)
__global__ __launch_bounds__(256) void stats_reduce_kernel(const dann_search_stats* st, uint32_t m, unsigned long long* sums,
                                                           uint32_t* first_failed) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long c = 0, h = 0;
    uint32_t f = 0;  // m - index of a failed search (0: none): the maximum names the first one
    if (i < m) {
        c = st[i].cmps;
        h = st[i].hops;
        f = st[i].status ? m - i : 0u;
    }
    for (int o = 32; o > 0; o >>= 1) {
        c += __shfl_xor(c, o);
        h += __shfl_xor(h, o);
        f = max(f, (uint32_t)__shfl_xor((int)f, o));
    }
    if ((threadIdx.x & 63u) == 0) {
        atomicAdd(sums, c);
        atomicAdd(sums + 1, h);
        if (f) atomicMax(first_failed, f);
    }
}

// the order of the Gram / sweep workgroups of one slice (lpt_order_kernel)
static const uint32_t* longest_first(BuildScratch& s, const uint32_t* sn, uint32_t m, hipStream_t st) {
    if (m < 4096u) return nullptr;  // (a launch that does not fill the chip twice has no tail to shorten)
    hipLaunchKernelGGL(lpt_order_kernel, dim3(1), dim3(1024), 0, st, sn, m, s.g_order.as<uint32_t>());
    return s.g_order.as<uint32_t>();
}

// ---- one multi_insert batch in two phases, everything on the index stream -------------------------
// Phase 1 (candidate generation, index.rs:349-434): insert-time search + RobustPrune for the batch
// positions [lo, hi); writes (hi - lo) rows [len, ids...] of `pend_stride` u32 to d_pending_out.
// Only reads the graph, so ranks holding identical replicas can each take a slice of the batch.
static int32_t batch_candidates(dann_index* idx, const dann_build_config& cfg, BuildScratch& s,
                                const uint32_t* d_slots, uint32_t n, uint32_t lo, uint32_t hi,
                                uint32_t* d_pending_out) {
    const IndexView ix = idx->view();
    PruneCfg pc = to_prune_cfg(idx, cfg);
    pc.counters = s.counters.as<unsigned long long>();
    hipStream_t st = idx->main.stream;
    const uint32_t m = hi - lo;
    if (m == 0) return DANN_OK;
    const uint32_t cand = cfg.intra_batch_candidates == 0xFFFFFFFFu ? n : std::min(cfg.intra_batch_candidates, n);
    const uint32_t nex = (cand != 0 && n > 1) ? std::min(cand, n - 1) : 0;
    SearchArgs sa;
    sa.ix = ix;
    sa.queries = nullptr;
    sa.qslots = d_slots + lo;
    sa.nq = m;
    sa.l_value = cfg.l_build;
    sa.beam_width = 1;
    sa.k = 0;
    sa.ht_entries = auto_visited_entries(idx, cfg.l_build, 1);
    sa.out_ids = nullptr;
    sa.out_dists = nullptr;
    sa.stats = s.stats.as<dann_search_stats>();
    sa.rec_ids = s.rec_ids.as<uint32_t>();
    sa.rec_dists = s.rec_d.as<float>();
    sa.rec_stride = s.rec_stride;
    sa.rec_n = s.rec_n.as<uint32_t>();
    sa.qmap = nullptr;
    sa.range_ids = nullptr;
    sa.range_d = nullptr;
    sa.range_second = nullptr;
    sa.range_cap = sa.range_max = sa.range_thresh = sa.has_inner = 0;
    sa.radius = sa.inner_radius = sa.range_slack = 0.0f;
    sa.fail_flag = nullptr;
    sa.spill = nullptr;
    sa.spill_next = nullptr;
    sa.spill_slices = sa.spill_bits = 0;
    uint32_t* meta = s.meta.as<uint32_t>();  // [0] nseg [1] nkeys [2] maxseg [3] err [4] appends [5] prunes [6] max record
    DANN_HIP(hipMemsetAsync(meta, 0, 128, st));
    sa.rec_max = meta + 6;
    int32_t rc = search_with_retry(idx, idx->main, sa);
    if (rc != DANN_OK) return rc;
    // the pool's LDS footprint follows the longest record of this batch, not the worst-case bound
    uint32_t h_recmax = 0;
    DANN_HIP(hipMemcpyAsync(&h_recmax, meta + 6, 4, hipMemcpyDeviceToHost, st));
    DANN_HIP(hipStreamSynchronize(st));
    h_recmax = std::min<uint32_t>(std::max<uint32_t>(h_recmax, 32), s.rec_stride);
    PoolArgs pa;
    pa.ix = ix;
    pa.cfg = pc;
    pa.locs = d_slots;
    pa.pool_ids = s.rec_ids.as<uint32_t>();
    pa.pool_d = s.rec_d.as<float>();
    pa.offsets = nullptr;
    pa.stride = s.rec_stride;
    pa.counts = s.rec_n.as<uint32_t>();
    pa.cand = cand;
    pa.n = n;
    pa.pos0 = lo;
    pa.pcap = next_pow2(h_recmax + nex);
    pa.force_saturate = 0;
    pa.out = d_pending_out;
    pa.out_stride = s.pend_stride;
    pa.err = meta + 3;
    if (pa.pcap > kMaxPool) {
        set_error("candidate pool bound %u exceeds %u: lower l_build or intra_batch_candidates", pa.pcap, kMaxPool);
        return DANN_EUNSUPPORTED;
    }
    const size_t lds = pool_lds_layout(pa.pcap, pc.pruned_degree).total;
    bool pool_gram = false;
    // matrix-core path (three kernels: sort, Gram tiles, sweep).  Default for float rows of 1 KiB and more unless
    // DANN_BUILD_ROW_KERNEL_ONLY is set; DANN_BUILD_MFMA_POOL forces it for any row size.
    const bool floatrows = (ix.dtype == DT_F32 || ix.dtype == DT_F16) && ix.metric != M_COSINE;
    const bool want_pool_gram = floatrows && ((idx->build_flags & DANN_BUILD_MFMA_POOL) ||
                                              (!(idx->build_flags & DANN_BUILD_ROW_KERNEL_ONLY) && ix.layer_bytes >= 1024u &&
                                               pool_gram_default(idx)));
    if (want_pool_gram) {
        uint32_t mg = idx->dbg_u32(DANN_DBG_GRAM_COLS, 96u);  // tuning hook: columns of the Gram block (32 / 64 / 96)
        mg = std::min<uint32_t>(std::max<uint32_t>((mg + 31u) & ~31u, 32u), 32u * kTileColBlocks);
        const uint32_t ng = std::max<uint32_t>(mg, std::min<uint32_t>(32u * kTileRowBlocks, (h_recmax + nex + 31u) & ~31u));
        rc = ensure_gram_scratch(s, m, pa.pcap, ng, mg);
        if (rc != DANN_OK) return rc;
        SortArgs so;
        so.p = pa;
        so.sid = s.g_sid.as<uint32_t>();
        so.sd = s.g_sd.as<float>();
        so.sn = s.g_sn.as<uint32_t>();
        rc = dispatch_float<SortLauncher>(ix, so, m, lds, st);
        if (rc != DANN_OK) return rc;
        TileArgs ta;
        ta.ix = ix;
        ta.sid = so.sid;
        ta.sn = so.sn;
        ta.pcap = pa.pcap;
        ta.ng = ng;
        ta.mg = mg;
        ta.gram = s.g_gram.as<float>();
        ta.nrm = s.g_nrm.as<float>();
        ta.counters = pc.counters;
        ta.order = longest_first(s, so.sn, m, st);
        if (int32_t trc = s.tile_begin(st)) return trc;
        rc = launch_gram_tiles(ta, m, st, idx->dbg_u32(DANN_DBG_GRAM_F16_WIDEN, 0u) == 0u);
        if (rc != DANN_OK) return rc;
        if (int32_t trc = s.tile_end(st)) return trc;
        SweepArgs sw;
        sw.p = pa;
        sw.sid = so.sid;
        sw.sd = so.sd;
        sw.sn = so.sn;
        sw.gram = ta.gram;
        sw.nrm = ta.nrm;
        sw.ng = ng;
        sw.mg = mg;
        sw.escale = (float)idx->dbg_value(DANN_DBG_GRAM_ESCALE, 1.0);  // test hook
        sw.c1 = gram_c1_chained(ix.dim);
        sw.c2 = gram_c2_for_dim(ix.dim);
        sw.one_by_one = sweep_one_by_one(idx);
        sw.order = ta.order;
        sw.compact_lds = sweep_is_batched(pc, mg, sw.one_by_one) ? 1u : 0u;
        rc = dispatch_float<SweepLauncher>(ix, sw, m, sweep_lds_bytes(sw), st);
        if (rc != DANN_OK) return rc;
        pool_gram = true;
    }
    if (!pool_gram) {
        rc = dispatch<PoolLauncher>(ix, pa, m, lds, st);
        if (rc != DANN_OK) return rc;
    }
    hipLaunchKernelGGL(stats_reduce_kernel, dim3((m + 255u) / 256u), dim3(256), 0, st, s.stats.as<dann_search_stats>(), m,
                       reinterpret_cast<unsigned long long*>(meta + 16), meta + 15);
    // meta[3 .. 19] in one copy: [0] the pool overflow flag (meta[3]), [12] the first failed search (meta[15]), [13 .. 16]
    // the two 64-bit totals (meta[16 .. 19])
    struct {
        uint32_t w[17];
    } h_tail = {};
    DANN_HIP(hipMemcpyAsync(h_tail.w, meta + 3, sizeof(h_tail.w), hipMemcpyDeviceToHost, st));
    DANN_HIP(hipStreamSynchronize(st));
    unsigned long long h_sums[2];
    memcpy(h_sums, h_tail.w + 13, sizeof(h_sums));
    if (h_tail.w[0]) {
        set_error("candidate pool overflow in prune (pool cap %u)", pa.pcap);
        return DANN_EOVERFLOW;
    }
    idx->build_counters[2] += h_sums[0];
    idx->build_counters[3] += h_sums[1];
    if (h_tail.w[12]) {
        set_error("insert search %u: visited table or record buffer exhausted", lo + (m - h_tail.w[12]));
        return DANN_EOVERFLOW;
    }
    return DANN_OK;
}

// Phase 2 (graph update, index.rs:911-1024): back-edge aggregation, bootstrap when the graph is
// nearly empty, set_neighbors_bulk, add_edge_and_prune per distinct target.  Deterministic given the
// pending rows of the *whole* batch, so identical replicas stay identical.
static int32_t batch_commit(dann_index* idx, const dann_build_config& cfg, BuildScratch& s, const uint32_t* d_slots,
                            uint32_t n, const uint32_t* d_pending, uint32_t rank = 0, uint32_t world = 1,
                            uint32_t* d_rows_out = nullptr, uint32_t rows_cap = 0, uint32_t* count_out = nullptr,
                            uint32_t backedge_limit = 0xFFFFFFFFu) {
    if (count_out) *count_out = 0;
    s.bootstrap_too_big = false;
    const IndexView ix = idx->view();
    PruneCfg pc = to_prune_cfg(idx, cfg);
    pc.counters = s.counters.as<unsigned long long>();
    hipStream_t st = idx->main.stream;
    const uint32_t cand = cfg.intra_batch_candidates == 0xFFFFFFFFu ? n : std::min(cfg.intra_batch_candidates, n);
    uint32_t* meta = s.meta.as<uint32_t>();
    const uint32_t* pending = d_pending;
    int32_t rc;
    auto aggregate = [&](uint32_t* h_meta) -> int32_t {
        const uint32_t total = n * s.degree;
        DANN_HIP(hipMemsetAsync(meta, 0, 12, st));  // nseg, nkeys, maxseg -- the error word meta[3] is sticky
        hipLaunchKernelGGL(make_keys_kernel, dim3((total + 255) / 256), dim3(256), 0, st, d_slots, pending,
                           s.pend_stride, n, s.degree, s.keys_in.as<uint64_t>(), backedge_limit);
        size_t tmp = s.sort_tmp_bytes;
        DANN_HIP(hipcub::DeviceRadixSort::SortKeys(s.sort_tmp.p, tmp, s.keys_in.as<uint64_t>(),
                                                   s.keys_out.as<uint64_t>(), (int)total, 0, 64, st));
        hipLaunchKernelGGL(segment_kernel, dim3((total + 255) / 256), dim3(256), 0, st, s.keys_out.as<uint64_t>(),
                           total, s.seg_start.as<uint32_t>(), meta);
        hipLaunchKernelGGL(seglen_kernel, dim3((total + 255) / 256), dim3(256), 0, st, s.keys_out.as<uint64_t>(),
                           s.seg_start.as<uint32_t>(), s.seg_len.as<uint32_t>(), meta);
        DANN_HIP(hipMemcpyAsync(h_meta, meta, 16, hipMemcpyDeviceToHost, st));
        DANN_HIP(hipStreamSynchronize(st));
        return DANN_OK;
    };
    uint32_t h_meta[4] = {0, 0, 0, 0};
    // err, appends, prunes, longest record, pad: a commit does not inherit the error word of an earlier call on this
    // handle (phase 1 reports its own errors before it returns); within the commit the word stays sticky
    DANN_HIP(hipMemsetAsync(meta + 3, 0, 20, st));
    rc = aggregate(h_meta);
    if (rc != DANN_OK) return rc;

    // ---- bootstrap (index.rs:926-938) -----------------------------------------------------------------
    const uint32_t resolved = std::max<uint32_t>(cand, 1);
    if (resolved < n && (h_meta[0] + 7) / 8 <= n) {
        ListArgs la;
        la.ix = ix;
        la.cfg = pc;
        la.locs = d_slots;
        la.pending = pending;
        la.pend_stride = s.pend_stride;
        la.n = n;
        la.pcap = next_pow2(pc.pruned_degree + n);
        la.out = s.pending2.as<uint32_t>();
        la.out_stride = s.pend_stride;
        la.err = meta + 3;
        if (la.pcap > kMaxPool) {
            // nothing has been written to the graph yet: the caller may insert the same points in smaller batches
            s.bootstrap_too_big = true;
            set_error("bootstrap of a %u-point batch into a nearly empty graph is not supported (cap %u); "
                      "use a geometric batch schedule (dann_build)", n, kMaxPool);
            return DANN_EUNSUPPORTED;
        }
        const size_t lds = pool_lds_layout(la.pcap, pc.pruned_degree).total;
        rc = dispatch<BootLauncher>(ix, la, n, lds, st);
        if (rc != DANN_OK) return rc;
        pending = s.pending2.as<uint32_t>();
        rc = aggregate(h_meta);
        if (rc != DANN_OK) return rc;
        if (h_meta[3]) {
            set_error("candidate pool overflow in bootstrap prune");
            return DANN_EOVERFLOW;
        }
    }

    // ---- graph update -------------------------------------------------------------------------------------
    hipLaunchKernelGGL(set_bulk_kernel, dim3(n), dim3(64), 0, st, ix, d_slots, pending, s.pend_stride, n);
    if (h_meta[0]) {
        BackArgs ba;
        ba.ix = ix;
        ba.cfg = pc;
        ba.keys = s.keys_out.as<uint64_t>();
        ba.seg_start = s.seg_start.as<uint32_t>();
        ba.seg_len = s.seg_len.as<uint32_t>();
        ba.nseg = h_meta[0];
        ba.nkeys = h_meta[1];
        ba.pcap = 0;  // set per launch below
        ba.err = meta + 3;
        ba.work = nullptr;
        DANN_HIP(hipMemsetAsync(meta + 10, 0, 12, st));  // worklist counts
        // (1) every target: append if the list still fits, else queue it.  Lists of up to `short_cap` entries go
        //     to the short worklist (pool of 128 slots: full occupancy, and the MFMA path when it applies), the
        //     rare long ones (hubs hit by many back-edges in one batch) to a launch of their own whose LDS pool
        //     is sized by the longest of them -- one hub no longer sets the occupancy of the whole batch.
        // default policy: rows of 1 KiB and more (where it is measured to win: 1 M x 768 build 3.50 -> 3.24 s) take the
        // MFMA path unless DANN_BUILD_ROW_KERNEL_ONLY is set; DANN_BUILD_MFMA_BACKEDGE forces it for any row size
        const bool want_gram = (ix.dtype == DT_F32 || ix.dtype == DT_F16) && ix.metric != M_COSINE &&
                               ((idx->build_flags & DANN_BUILD_MFMA_BACKEDGE) ||
                                (!(idx->build_flags & DANN_BUILD_ROW_KERNEL_ONLY) && ix.layer_bytes >= 1024u));
        // Gram rows per list: degree + 8 rounded up (96 at R = 64) covers the lists of a steady-state batch; the tiles kernel
        // could take up to 256, but sizing every list's LDS pool and Gram stride for the rare hub costs more than the hubs
        // save (1 M x 768: 3.18 s at 256 against 2.81 s; DANN_DBG_BACKEDGE_GRAM_ROWS raises it for experiments).  Longer
        // lists stay on the row kernel.
        uint32_t pg = std::min<uint32_t>(128u, (ix.max_degree + 8u + 31u) & ~31u);
        if (want_gram) {
            if (const uint32_t e = idx->dbg_u32(DANN_DBG_BACKEDGE_GRAM_ROWS, 0u))
                pg = std::min<uint32_t>(32u * kTileRowBlocks, std::max<uint32_t>(pg, (e + 31u) & ~31u));
        }
        const uint32_t short_pcap = next_pow2(std::max<uint32_t>(pg, ix.max_degree + 1u));
        const uint32_t short_cap = want_gram ? pg : short_pcap;
        // small rows without the MFMA path: the single-kernel form (list build + prune per target, its LDS pool sized by
        // the longest list of the batch) only while that pool is small.  One hub with a thousand back-edges in a batch
        // (the 65 536-point batches of a 10 M-point build) sizes EVERY workgroup's LDS for its list -- 2 - 5 lists per CU
        // instead of 20: the phase was 60 % of the 10 M x 128 build (6.2 of 10.3 s).  Beyond kSinglePool entries the split
        // form runs the short lists at full occupancy and the hubs in a launch of their own.  Measured, round 5
        // (scratch/r05_backedge_ab.sh; pool limit 8192 / 512 / 256 / 128 / 64): 10 M x 128 build 10.2 / 6.6 / 6.0 / 6.0 /
        // 6.0 s, 1 M x 128 build 0.64 / 0.64 / 0.62 / 0.61 / 0.61 s (round 2 had measured the split form slower at 1 M:
        // 0.78 against 0.68 s, before the scan kernel served four targets per wavefront).  Rows of 1 KiB and more always
        // take the split form (1 M x 768: 5.2 s -> 3.5 s, 3.25 s with the MFMA path).
        const uint32_t pcap_all = next_pow2(ix.max_degree + h_meta[2]);
        const uint32_t single_pool = idx->dbg_u32(DANN_DBG_BACKEDGE_SINGLE_POOL, 128u);
        if (!want_gram && ix.layer_bytes < 1024u && world <= 1u && pcap_all <= single_pool) {
            if (pcap_all > kMaxPool) {
                set_error("a node received %u back-edges in one batch (cap %u): lower max_batch", h_meta[2], kMaxPool);
                return DANN_EOVERFLOW;
            }
            ba.pcap = pcap_all;
            rc = dispatch<BackLauncher>(ix, ba, ba.nseg, pool_lds_layout(ba.pcap, pc.pruned_degree).total, st);
            if (rc != DANN_OK) return rc;
        } else {
        uint32_t* work = s.seg_len.as<uint32_t>() + (size_t)n * s.degree;  // 2 x nseg entries behind seg_len
        ScanArgs sa;
        sa.ix = ix;
        sa.cfg_max_degree = pc.max_degree;
        sa.keys = ba.keys;
        sa.seg_start = ba.seg_start;
        sa.seg_len = ba.seg_len;
        sa.nseg = ba.nseg;
        sa.short_cap = short_cap;
        sa.work_short = work;
        sa.work_long = work + ba.nseg;
        sa.counts = meta + 10;
        sa.rank = rank;
        sa.world = world;
        hipLaunchKernelGGL(backedge_scan_kernel, dim3((ba.nseg + 3u) / 4u), dim3(kWave), 0, st, sa);
        uint32_t h_counts[3] = {0, 0, 0};
        DANN_HIP(hipMemcpyAsync(h_counts, meta + 10, 12, hipMemcpyDeviceToHost, st));
        DANN_HIP(hipStreamSynchronize(st));
        // (2) long lists first, on the side stream: every list belongs to another target row and a prune reads vector rows
        //     and its own list only, so the two classes do not depend on each other.  A launch of these is as long as its
        //     longest list (1 M x 768: 1.7 ms on average, 170 ms of a 2.4 s build, with a handful of CUs busy: some fifty
        //     hubs per 16 384-point batch, the longest with ~2 000 back-edges -- far beyond the 256 rows of a Gram block).
        hipStream_t side = nullptr;
        bool side_busy = false;
        struct SideGuard {  // an error return below must not leave the side stream working on this scratch
            hipStream_t& s;
            bool& busy;
            ~SideGuard() {
                if (busy) (void)hipStreamSynchronize(s);
            }
        } side_guard{side, side_busy};
        if (h_counts[1]) {
            BackArgs bl = ba;
            bl.work = work + ba.nseg;
            bl.nseg = h_counts[1];
            bl.pcap = next_pow2(h_counts[2]);
            bl.count_long = want_gram ? 1u : 0u;
            if (bl.pcap > kMaxPool) {
                set_error("a node received back-edges for a list of %u entries in one batch (cap %u): lower max_batch",
                          h_counts[2], kMaxPool);
                return DANN_EOVERFLOW;
            }
            hipStream_t run_on = st;
            if (h_counts[0]) {  // (with no short lists there is nothing to run beside)
                if (int32_t src = s.side_stream(&side)) return src;
                run_on = side;
                side_busy = true;
            }
            rc = dispatch<BackLauncher>(ix, bl, bl.nseg, pool_lds_layout(bl.pcap, pc.pruned_degree).total, run_on);
            if (rc != DANN_OK) return rc;
        }
        // (3) short lists
        if (h_counts[0]) {
            BackArgs bs = ba;
            bs.work = work;
            bs.nseg = h_counts[0];
            bs.pcap = short_pcap;
            bool gram = false;
            if (want_gram && pg > ix.max_degree) {
                // three kernels: list + d(target, c) + sort; Gram tiles of the sorted list on the matrix cores; sweep
                const uint32_t nshort = h_counts[0], mg = std::min<uint32_t>(pg, 32u * kTileColBlocks);
                rc = ensure_gram_scratch(s, nshort, bs.pcap, pg, mg);
                if (rc != DANN_OK) return rc;
                const size_t lds_pool = pool_lds_layout(bs.pcap, pc.pruned_degree).total;
                BackListArgs la;
                la.b = bs;
                la.sid = s.g_sid.as<uint32_t>();
                la.sd = s.g_sd.as<float>();
                la.sn = s.g_sn.as<uint32_t>();
                la.loc = s.g_loc.as<uint32_t>();
                rc = dispatch_float<BackListLauncher>(ix, la, nshort, lds_pool, st);
                if (rc != DANN_OK) return rc;
                TileArgs ta;
                ta.ix = ix;
                ta.sid = la.sid;
                ta.sn = la.sn;
                ta.pcap = bs.pcap;
                ta.ng = pg;
                ta.mg = mg;
                ta.gram = s.g_gram.as<float>();
                ta.nrm = s.g_nrm.as<float>();
                ta.counters = pc.counters;
                ta.order = longest_first(s, la.sn, nshort, st);
                if (int32_t trc = s.tile_begin(st)) return trc;
                rc = launch_gram_tiles(ta, nshort, st, idx->dbg_u32(DANN_DBG_GRAM_F16_WIDEN, 0u) == 0u);
                if (rc != DANN_OK) return rc;
                if (int32_t trc = s.tile_end(st)) return trc;
                SweepArgs sw;
                sw.p = PoolArgs{};
                sw.p.ix = ix;
                sw.p.cfg = pc;
                sw.p.pcap = bs.pcap;
                sw.p.force_saturate = 0;
                sw.sid = la.sid;
                sw.sd = la.sd;
                sw.sn = la.sn;
                sw.gram = ta.gram;
                sw.nrm = ta.nrm;
                sw.ng = pg;
                sw.mg = mg;
                sw.escale = (float)idx->dbg_value(DANN_DBG_GRAM_ESCALE, 1.0);  // test hook
                sw.c1 = gram_c1_chained(ix.dim);
                sw.c2 = gram_c2_for_dim(ix.dim);
                sw.one_by_one = sweep_one_by_one(idx);
                sw.order = ta.order;
                sw.out_loc = la.loc;
                sw.compact_lds = sweep_is_batched(pc, mg, sw.one_by_one) ? 1u : 0u;
                rc = dispatch_float<SweepLauncher>(ix, sw, nshort, sweep_lds_bytes(sw), st);
                if (rc != DANN_OK) return rc;
                gram = true;
            }
            if (!gram) {
                rc = dispatch<BackLauncher>(ix, bs, bs.nseg, pool_lds_layout(bs.pcap, pc.pruned_degree).total, st);
                if (rc != DANN_OK) return rc;
            }
        }
        if (side_busy) {  // both classes of lists are written before the rows are exported / the counters are read
            side_busy = false;
            DANN_HIP(hipStreamSynchronize(side));
        }
        if (world > 1u) {  // the rows this rank owns and has just rewritten, for the other replicas
            const uint32_t cnt = h_counts[0] + h_counts[1];
            if (cnt > rows_cap || (cnt && !d_rows_out)) {
                set_error("partitioned commit: %u rewritten rows exceed the export buffer (%u)", cnt, rows_cap);
                return DANN_EOVERFLOW;
            }
            if (cnt)
                hipLaunchKernelGGL(export_rows_kernel, dim3(cnt), dim3(kWave), 0, st, ix, ba.keys, ba.seg_start, work,
                                   h_counts[0], work + ba.nseg, h_counts[1], d_rows_out);
            if (count_out) *count_out = cnt;
        }
        }
    }
    uint32_t h_tail[1] = {0};  // meta[3]: err
    DANN_HIP(hipMemcpyAsync(h_tail, meta + 3, sizeof(h_tail), hipMemcpyDeviceToHost, st));
    DANN_HIP(hipStreamSynchronize(st));
    if (h_tail[0]) {
        set_error("back-edge list overflow");
        return DANN_EOVERFLOW;
    }
    return DANN_OK;
}

static int32_t insert_batch_device(dann_index* idx, const dann_build_config& cfg, BuildScratch& s,
                                   const uint32_t* d_slots, uint32_t n, uint32_t backedge_limit = 0xFFFFFFFFu) {
    int32_t rc = batch_candidates(idx, cfg, s, d_slots, n, 0, n, s.pending.as<uint32_t>());
    if (rc != DANN_OK) return rc;
    return batch_commit(idx, cfg, s, d_slots, n, s.pending.as<uint32_t>(), 0, 1, nullptr, 0, nullptr, backedge_limit);
}

}  // namespace dann

using namespace dann;

namespace dann {
// the last commit on this handle stopped because its batch needs a bootstrap larger than one prune pool (nothing was
// written to the graph): the callers that own a batch schedule (dann_build, dann_build_sharded) halve the batch
bool build_bootstrap_too_big(dann_index* idx) {
    return idx->build_scratch && static_cast<BuildScratch*>(idx->build_scratch)->bootstrap_too_big;
}
}  // namespace dann

static BuildScratch& scratch_of(dann_index* idx) {
    if (!idx->build_scratch) {
        idx->build_scratch = new BuildScratch();
        idx->build_scratch_free = [](void* p) { delete static_cast<BuildScratch*>(p); };
    }
    return *static_cast<BuildScratch*>(idx->build_scratch);
}

extern "C" {

int32_t dann_insert_batch(dann_index* idx, const dann_build_config* cfg, const uint32_t* slots, uint32_t n) try {
    if (!idx) return DANN_EINVAL;
    ::dann::ExclusiveGuard lock(idx);
    DANN_MUTATION(idx);
    DeviceGuard guard(idx->device);
    int32_t rc = validate_cfg(idx, cfg);
    if (rc != DANN_OK) return rc;
    if (n == 0) return DANN_OK;
    if (!slots) return DANN_EINVAL;
    for (uint32_t i = 0; i < n; ++i)
        if (slots[i] >= idx->cfg.capacity) {
            set_error("slot %u out of bounds (capacity %u)", slots[i], idx->cfg.capacity);
            return DANN_EBOUNDS;
        }
    BuildScratch& s = scratch_of(idx);
    const uint32_t rec_stride = 4 * (cfg->l_build + idx->cfg.num_start_points) + 64;
    rc = ensure_scratch(s, n, rec_stride, cfg->pruned_degree);
    if (rc != DANN_OK) return rc;
    DANN_HIP(hipMemcpyAsync(s.slots.p, slots, (size_t)n * 4, hipMemcpyHostToDevice, idx->main.stream));
    return insert_batch_device(idx, *cfg, s, s.slots.as<uint32_t>(), n);
} DANN_CATCH_ALL

// DiskANNIndex::insert (index.rs:226-341): the insert search, the prune, set_neighbors, then add_edge_and_prune for the
// first max_backedges of the new neighbours (:324-327) -- a multi_insert of one point whose back-edges stop there.
int32_t dann_insert(dann_index* idx, const dann_build_config* cfg, uint32_t slot) try {
    if (!idx) return DANN_EINVAL;
    ::dann::ExclusiveGuard lock(idx);
    DANN_MUTATION(idx);
    DeviceGuard guard(idx->device);
    int32_t rc = validate_cfg(idx, cfg);
    if (rc != DANN_OK) return rc;
    if (cfg->max_backedges > cfg->pruned_degree) {  // config/mod.rs:308-311
        set_error("parameter \"max_backedges\" (%u) must not be greater than \"pruned_degree\" (%u)", cfg->max_backedges,
                  cfg->pruned_degree);
        return DANN_EINVAL;
    }
    if (slot >= idx->cfg.capacity) {
        set_error("slot %u out of bounds (capacity %u)", slot, idx->cfg.capacity);
        return DANN_EBOUNDS;
    }
    BuildScratch& s = scratch_of(idx);
    const uint32_t rec_stride = 4 * (cfg->l_build + idx->cfg.num_start_points) + 64;
    rc = ensure_scratch(s, 1, rec_stride, cfg->pruned_degree);
    if (rc != DANN_OK) return rc;
    DANN_HIP(hipMemcpyAsync(s.slots.p, &slot, 4, hipMemcpyHostToDevice, idx->main.stream));
    DANN_HIP(hipStreamSynchronize(idx->main.stream));  // (`slot` lives on this frame)
    return insert_batch_device(idx, *cfg, s, s.slots.as<uint32_t>(), 1, cfg->max_backedges ? cfg->max_backedges : cfg->pruned_degree);
} DANN_CATCH_ALL

// multi-GPU build: phase 1 on a slice of the batch, phase 2 with the all-gathered pending rows.
// d_pending_* are DEVICE pointers ((pruned_degree + 1) u32 per row) so that they can be the send /
// receive buffers of an RCCL all-gather.
int32_t dann_insert_batch_candidates(dann_index* idx, const dann_build_config* cfg, const uint32_t* slots, uint32_t n,
                                     uint32_t lo, uint32_t hi, uint32_t* d_pending_out) try {
    if (!idx) return DANN_EINVAL;
    ::dann::ExclusiveGuard lock(idx);
    DeviceGuard guard(idx->device);
    int32_t rc = validate_cfg(idx, cfg);
    if (rc != DANN_OK) return rc;
    if (lo > hi || hi > n) return DANN_EINVAL;
    if (n == 0 || lo == hi) return DANN_OK;
    if (!slots || !d_pending_out) return DANN_EINVAL;
    for (uint32_t i = 0; i < n; ++i)
        if (slots[i] >= idx->cfg.capacity) return DANN_EBOUNDS;
    BuildScratch& s = scratch_of(idx);
    const uint32_t rec_stride = 4 * (cfg->l_build + idx->cfg.num_start_points) + 64;
    rc = ensure_scratch(s, n, rec_stride, cfg->pruned_degree);
    if (rc != DANN_OK) return rc;
    DANN_HIP(hipMemcpyAsync(s.slots.p, slots, (size_t)n * 4, hipMemcpyHostToDevice, idx->main.stream));
    return batch_candidates(idx, *cfg, s, s.slots.as<uint32_t>(), n, lo, hi, d_pending_out);
} DANN_CATCH_ALL

int32_t dann_insert_batch_commit(dann_index* idx, const dann_build_config* cfg, const uint32_t* slots, uint32_t n,
                                 const uint32_t* d_pending_all) try {
    if (!idx) return DANN_EINVAL;
    ::dann::ExclusiveGuard lock(idx);
    DANN_MUTATION(idx);
    DeviceGuard guard(idx->device);
    int32_t rc = validate_cfg(idx, cfg);
    if (rc != DANN_OK) return rc;
    if (n == 0) return DANN_OK;
    if (!slots || !d_pending_all) return DANN_EINVAL;
    for (uint32_t i = 0; i < n; ++i)
        if (slots[i] >= idx->cfg.capacity) return DANN_EBOUNDS;
    BuildScratch& s = scratch_of(idx);
    const uint32_t rec_stride = 4 * (cfg->l_build + idx->cfg.num_start_points) + 64;
    rc = ensure_scratch(s, n, rec_stride, cfg->pruned_degree);
    if (rc != DANN_OK) return rc;
    DANN_HIP(hipMemcpyAsync(s.slots.p, slots, (size_t)n * 4, hipMemcpyHostToDevice, idx->main.stream));
    return batch_commit(idx, *cfg, s, s.slots.as<uint32_t>(), n, d_pending_all);
} DANN_CATCH_ALL

int32_t dann_insert_batch_commit_part(dann_index* idx, const dann_build_config* cfg, const uint32_t* slots, uint32_t n,
                                      const uint32_t* d_pending_all, uint32_t rank, uint32_t world, uint32_t* d_rows_out,
                                      uint32_t rows_cap, uint32_t* count_out) try {
    if (!idx || !count_out || world == 0 || rank >= world) return DANN_EINVAL;
    *count_out = 0;
    ::dann::ExclusiveGuard lock(idx);
    DANN_MUTATION(idx);
    DeviceGuard guard(idx->device);
    int32_t rc = validate_cfg(idx, cfg);
    if (rc != DANN_OK) return rc;
    if (n == 0) return DANN_OK;
    if (!slots || !d_pending_all) return DANN_EINVAL;
    for (uint32_t i = 0; i < n; ++i)
        if (slots[i] >= idx->cfg.capacity) return DANN_EBOUNDS;
    BuildScratch& s = scratch_of(idx);
    const uint32_t rec_stride = 4 * (cfg->l_build + idx->cfg.num_start_points) + 64;
    rc = ensure_scratch(s, n, rec_stride, cfg->pruned_degree);
    if (rc != DANN_OK) return rc;
    DANN_HIP(hipMemcpyAsync(s.slots.p, slots, (size_t)n * 4, hipMemcpyHostToDevice, idx->main.stream));
    rc = batch_commit(idx, *cfg, s, s.slots.as<uint32_t>(), n, d_pending_all, rank, world, d_rows_out, rows_cap, count_out);
    if (rc != DANN_OK) return rc;
    DANN_HIP(hipStreamSynchronize(idx->main.stream));  // the exported rows are read by the caller's collective next
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_apply_neighbor_rows_device(dann_index* idx, const uint32_t* d_rows, uint32_t count) try {
    if (!idx || (count && !d_rows)) return DANN_EINVAL;
    if (count == 0) return DANN_OK;
    ::dann::ExclusiveGuard lock(idx);
    DANN_MUTATION(idx);
    DeviceGuard guard(idx->device);
    hipLaunchKernelGGL(apply_rows_kernel, dim3(count), dim3(kWave), 0, idx->main.stream, idx->view(), d_rows, count);
    DANN_HIP(hipGetLastError());
    DANN_HIP(hipStreamSynchronize(idx->main.stream));
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_debug_gram_tiles(int32_t device, int32_t dtype, const void* rows, uint32_t n, uint32_t dim, uint32_t mg,
                              float* out_gram, float* out_nrm) try {
    const bool widen = (dtype & 0x100) != 0;  // DANN_F16 | 0x100: f16 rows widened through the f32 matrix core
    dtype &= 0xFF;
    if (!rows || !out_gram || !out_nrm || n == 0 || n > 32u * kTileRowBlocks || dim == 0 || (dtype != DT_F32 && dtype != DT_F16))
        return DANN_EINVAL;
    mg = std::min<uint32_t>(std::max<uint32_t>((mg + 31u) & ~31u, 32u), 32u * kTileColBlocks);
    DeviceGuard guard(device < 0 ? 0 : device);
    const size_t esz = dtype == DT_F32 ? 4 : 2;
    const size_t stride = ((size_t)dim * esz + 15) & ~(size_t)15;
    const uint32_t ng = (n + 31u) & ~31u;
    DevBuf dr, dids, dsn, dg, dn;
    DANN_HIP(dr.alloc(stride * n + 256));
    DANN_HIP(dids.alloc((size_t)ng * 4));
    DANN_HIP(dsn.alloc(4));
    DANN_HIP(dg.alloc((size_t)ng * mg * 4));
    DANN_HIP(dn.alloc((size_t)ng * 4));
    DANN_HIP(hipMemset(dr.p, 0, stride * n + 256));
    DANN_HIP(hipMemset(dg.p, 0, (size_t)ng * mg * 4));
    DANN_HIP(hipMemset(dn.p, 0, (size_t)ng * 4));
    DANN_HIP(hipMemcpy2D(dr.p, stride, rows, (size_t)dim * esz, (size_t)dim * esz, n, hipMemcpyHostToDevice));
    std::vector<uint32_t> ids(ng);
    for (uint32_t i = 0; i < ng; ++i) ids[i] = i;
    DANN_HIP(hipMemcpy(dids.p, ids.data(), (size_t)ng * 4, hipMemcpyHostToDevice));
    DANN_HIP(hipMemcpy(dsn.p, &n, 4, hipMemcpyHostToDevice));
    TileArgs ta{};
    ta.ix.rows = dr.as<uint8_t>();
    ta.ix.row_stride = stride;
    ta.ix.dim = dim;
    ta.ix.dtype = dtype;
    ta.ix.nslots = n;
    ta.sid = dids.as<uint32_t>();
    ta.sn = dsn.as<uint32_t>();
    ta.pcap = ng;
    ta.ng = ng;
    ta.mg = std::min(mg, ng);
    ta.gram = dg.as<float>();
    ta.nrm = dn.as<float>();
    ta.counters = nullptr;
    int32_t rc = launch_gram_tiles(ta, 1, 0, !widen);
    if (rc != DANN_OK) return rc;
    DANN_HIP(hipDeviceSynchronize());
    // out_gram: n x mg (row stride mg as passed, entries beyond the computed block are zero)
    std::vector<float> g((size_t)ng * ta.mg);
    DANN_HIP(hipMemcpy(g.data(), dg.p, g.size() * 4, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; ++i)
        for (uint32_t j = 0; j < mg; ++j) out_gram[(size_t)i * mg + j] = j < ta.mg ? g[(size_t)i * ta.mg + j] : 0.0f;
    DANN_HIP(hipMemcpy(out_nrm, dn.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_set_build_options(dann_index* idx, uint32_t flags) try {
    if (!idx) return DANN_EINVAL;
    ::dann::ExclusiveGuard lock(idx);
    idx->build_flags = flags;
    return DANN_OK;
} DANN_CATCH_ALL

}  // extern "C"
// dann_kernel_time(which = 5): the gram_tiles_kernel launches of this index's builds (all pending events resolved)
int32_t dann::build_tile_clock(const dann_index* idx, double* total_ms, uint64_t* launches) {
    if (total_ms) *total_ms = 0.0;
    if (launches) *launches = 0;
    if (!idx->build_scratch) return DANN_OK;
    BuildScratch& s = *static_cast<BuildScratch*>(idx->build_scratch);
    DeviceGuard guard(idx->device);
    s.tile_drain(true);
    if (total_ms) *total_ms = s.tile_ms;
    if (launches) *launches = s.tile_launches;
    return DANN_OK;
}
void dann::build_tile_clock_reset(const dann_index* idx) {
    if (!idx->build_scratch) return;
    BuildScratch& s = *static_cast<BuildScratch*>(idx->build_scratch);
    s.tile_drain(true);
    s.tile_ms = 0.0;
    s.tile_launches = 0;
}
extern "C" {

int32_t dann_build_counters(const dann_index* idx, uint64_t* out, uint32_t n) try {
    if (!idx || (n && !out)) return DANN_EINVAL;
    ::dann::ExclusiveGuard lock(idx);
    uint64_t dev[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sort_fallbacks = 0;
    if (idx->build_scratch) {
        BuildScratch& s = *static_cast<BuildScratch*>(idx->build_scratch);
        if (s.counters.p) {
            DeviceGuard guard(idx->device);
            DANN_HIP(hipStreamSynchronize(idx->main.stream));
            std::vector<uint64_t> stripes((size_t)kStatStripes * 8 + 8);
            DANN_HIP(hipMemcpy(stripes.data(), s.counters.p, stripes.size() * 8, hipMemcpyDeviceToHost));
            for (uint32_t t = 0; t < kStatStripes; ++t)
                for (uint32_t c = 0; c < 8; ++c) dev[c] += stripes[(size_t)t * 8 + c];
            sort_fallbacks = stripes[(size_t)kStatStripes * 8];
        }
    }
    const uint64_t all[11] = {dev[6], dev[7], idx->build_counters[2], idx->build_counters[3],
                              dev[0], dev[1], dev[2], dev[3], dev[4], dev[5], sort_fallbacks};
    for (uint32_t i = 0; i < n; ++i) out[i] = i < 11 ? all[i] : 0u;
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_build(dann_index* idx, const dann_build_config* cfg, uint32_t first, uint32_t n, float growth,
                   uint32_t max_batch) try {
    if (!idx) return DANN_EINVAL;
    ::dann::ExclusiveGuard lock(idx);
    DANN_MUTATION(idx);
    DeviceGuard guard(idx->device);
    int32_t rc = validate_cfg(idx, cfg);
    if (rc != DANN_OK) return rc;
    if ((uint64_t)first + n > idx->cfg.capacity) return DANN_EBOUNDS;
    if (!(growth > 0.0f) || max_batch == 0) return DANN_EINVAL;
    BuildScratch& s = scratch_of(idx);
    const uint32_t rec_stride = 4 * (cfg->l_build + idx->cfg.num_start_points) + 64;
    rc = ensure_scratch(s, std::min(max_batch, n ? n : 1u), rec_stride, cfg->pruned_degree);
    if (rc != DANN_OK) return rc;
    std::vector<uint32_t> ids;
    uint32_t done = 0;
    int32_t batches = 0;
    uint32_t limit = max_batch;  // shrinks when a batch turns out to need a bootstrap larger than the pool
    while (done < n) {
        // geometric schedule: batch = clamp(ceil(inserted * growth), 1, max_batch)
        uint32_t b = (uint32_t)std::ceil((double)(first + done) * (double)growth);
        b = std::max<uint32_t>(1, std::min(b, limit));
        b = std::min(b, n - done);
        ids.resize(b);
        for (uint32_t i = 0; i < b; ++i) ids[i] = first + done + i;
        DANN_HIP(hipMemcpyAsync(s.slots.p, ids.data(), (size_t)b * 4, hipMemcpyHostToDevice, idx->main.stream));
        DANN_HIP(hipStreamSynchronize(idx->main.stream));
        s.bootstrap_too_big = false;
        rc = insert_batch_device(idx, *cfg, s, s.slots.as<uint32_t>(), b);
        if (rc == DANN_EUNSUPPORTED && s.bootstrap_too_big && b > 1) {
            // multi_insert's bootstrap test (index.rs:926-931) fired for a batch whose members do not fit one prune
            // pool; the graph is untouched at that point, so the same points go in as two smaller batches
            limit = std::max<uint32_t>(1, b / 2);
            continue;
        }
        if (rc != DANN_OK) return rc;
        limit = std::min<uint32_t>(max_batch, limit * 2 > limit ? limit * 2 : limit);
        done += b;
        ++batches;
    }
    return batches;
} DANN_CATCH_ALL

int32_t dann_prune_batch(dann_index* idx, const dann_build_config* cfg, const uint32_t* locs, uint32_t n,
                         const uint32_t* pool_ids, const float* pool_dists, const uint64_t* offsets,
                         int32_t force_saturate, uint32_t* out_adj) try {
    if (!idx) return DANN_EINVAL;
    ::dann::ExclusiveGuard lock(idx);
    DeviceGuard guard(idx->device);
    int32_t rc = validate_cfg(idx, cfg);
    if (rc != DANN_OK) return rc;
    if (n == 0) return DANN_OK;
    if (!locs || !pool_ids || !pool_dists || !offsets || !out_adj) return DANN_EINVAL;
    const uint64_t total = offsets[n];
    uint64_t maxlen = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (offsets[i + 1] < offsets[i]) return DANN_EINVAL;
        maxlen = std::max(maxlen, offsets[i + 1] - offsets[i]);
    }
    for (uint64_t i = 0; i < total; ++i)
        if (pool_ids[i] >= idx->nslots) return DANN_EBOUNDS;
    if (maxlen > kMaxPool) {
        set_error("pool of %llu candidates exceeds the supported %u", (unsigned long long)maxlen, kMaxPool);
        return DANN_EUNSUPPORTED;
    }
    const uint32_t ostride = cfg->pruned_degree + 1;
    DevBuf dl, di, dd, doff, dout, derr;
    DANN_HIP(dl.alloc((size_t)n * 4));
    DANN_HIP(di.alloc(total * 4));
    DANN_HIP(dd.alloc(total * 4));
    DANN_HIP(doff.alloc((size_t)(n + 1) * 8));
    DANN_HIP(dout.alloc((size_t)n * ostride * 4));
    DANN_HIP(derr.alloc(4));
    hipStream_t st = idx->main.stream;
    DANN_HIP(hipMemcpyAsync(dl.p, locs, (size_t)n * 4, hipMemcpyHostToDevice, st));
    DANN_HIP(hipMemcpyAsync(di.p, pool_ids, total * 4, hipMemcpyHostToDevice, st));
    DANN_HIP(hipMemcpyAsync(dd.p, pool_dists, total * 4, hipMemcpyHostToDevice, st));
    DANN_HIP(hipMemcpyAsync(doff.p, offsets, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
    DANN_HIP(hipMemsetAsync(derr.p, 0, 4, st));
    PoolArgs pa;
    pa.ix = idx->view();
    pa.cfg = to_prune_cfg(idx, *cfg);
    pa.locs = dl.as<uint32_t>();
    pa.pool_ids = di.as<uint32_t>();
    pa.pool_d = dd.as<float>();
    pa.offsets = doff.as<uint64_t>();
    pa.stride = 0;
    pa.counts = nullptr;
    pa.cand = 0;
    pa.n = n;
    pa.pos0 = 0;
    pa.pcap = next_pow2((uint32_t)maxlen);
    pa.force_saturate = force_saturate;
    pa.out = dout.as<uint32_t>();
    pa.out_stride = ostride;
    pa.err = derr.as<uint32_t>();
    const size_t lds = pool_lds_layout(pa.pcap, pa.cfg.pruned_degree).total;
    rc = dispatch<PoolLauncher>(pa.ix, pa, n, lds, st);
    if (rc != DANN_OK) return rc;
    DANN_HIP(hipMemcpyAsync(out_adj, dout.p, (size_t)n * ostride * 4, hipMemcpyDeviceToHost, st));
    DANN_HIP(hipStreamSynchronize(st));
    return DANN_OK;
} DANN_CATCH_ALL

}  // extern "C"
