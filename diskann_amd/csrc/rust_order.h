// rust_order.h -- the order SortedNeighbors::new leaves a candidate pool in (diskann/src/graph/internal/
// sorted_neighbors.rs:26-44: `select_nth_unstable_by(position, fast_distance)` + `sort_unstable_by(fast_distance)` on the
// prefix), for callers that ask for the reference's own order of EQUAL-distance candidates (dann_set_prune_tie_order).
//
// The default prune sorts by (distance, pool position) with a wavefront-wide bitonic network; which of several
// equal-distance candidates comes first is unspecified in the reference's API, but it is a deterministic function of the
// pool's arrival order in the standard library the workspace pins (rust-toolchain.toml: 1.97.1; "ipnsort" and its
// selection since Rust 1.81: core::slice::sort::{unstable, shared}, core::slice::select).  This header walks that
// algorithm, as it applies to 8-byte `Copy` elements (Neighbor<u32>), over an array of pool POSITIONS compared through
// their distances -- sequentially: one lane does it while the rest of the wavefront waits.  It is the conformance mode
// (tie-heavy inputs: integer lattices, duplicated rows), not the fast path.
//
//   sort_unstable       len <= 20: insertion sort; a non-descending / strictly descending slice is kept / reversed; else
//                       quicksort with 2 * floor(log2(len | 1)) levels before heapsort
//   quicksort           <= 32 elements: small sort (optimal 9- / 13-input networks + insertion; from 18 elements two
//                       halves and a merge from both ends); pivot = median of v[0], v[4 (n / 8)], v[7 (n / 8)], from 64
//                       elements on recursively; cyclic Lomuto partition; a pivot equal to the ancestor's partitions by <=
//   select_nth_unstable index len - 1 / 0: first maximum / first minimum swapped in; else the same pivots and partitions
//                       down to 16 elements (insertion sort); after 16 rounds a fallback (here: a sort of the range)
//
// No recursion and no private arrays: the explicit stacks and the merge buffer live in a caller-supplied work area
// (kWorkBytes, LDS on the device).  Plain C++: tests/test_rust_order_host.py compiles it for the host and compares it
// with the checker's independent restatement on random tied pools.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define DANN_RO_HD __host__ __device__
#else
#define DANN_RO_HD
#endif

namespace dann {
namespace rust_order {

constexpr uint32_t kMaxLen = 4096;   // pool positions are 16-bit, the pivot recursion is sized for this
constexpr uint32_t kWorkBytes = 512; // (16-bit form) merge buffer 64 + ancestors 52 + quicksort stack 26 x 6 + pivot stack 6 x 12, rounded up

// E: the element type that is permuted (16-bit pool positions, or whole 8-byte (distance, payload) keys); W: the word of
// the explicit stacks (holds an index into the array); Less: is_less(a, b) on two elements; QD / MD: depth of the
// quicksort / pivot stacks (2 log2(len) + 2 and log8(len) - 1 for the longest array).
template <typename E, typename W, typename Less, int QD, int MD>
struct SorterT {
    E* v;        // the elements, permuted in place
    Less lt;
    E* tmp;      // 32 entries
    E* anc_st;   // ancestor pivot of each quicksort frame
    W* qs;       // quicksort stack: 3 words per frame
    W* ms;       // pivot stack: 6 words per frame
    bool fallback;   // the selection's median-of-medians fallback was reached

    static constexpr uint32_t work_bytes() { return (uint32_t)((32 + QD) * sizeof(E) + (QD * 3 + MD * 6) * sizeof(W)); }
    DANN_RO_HD SorterT(E* elements, Less l, void* work)
        : v(elements), lt(l), tmp(static_cast<E*>(work)), anc_st(static_cast<E*>(work) + 32),
          qs(reinterpret_cast<W*>(static_cast<E*>(work) + 32 + QD)), ms(reinterpret_cast<W*>(static_cast<E*>(work) + 32 + QD) + QD * 3),
          fallback(false) {}

    // is_less = |a, b| fast_distance(a, b) == Less; partial_cmp: an unordered pair is Equal (neighbor/mod.rs:150-154)
    DANN_RO_HD bool less(const E& a, const E& b) const { return lt(a, b); }
    DANN_RO_HD void swap(uint32_t i, uint32_t j) {
        const E t = v[i];
        v[i] = v[j];
        v[j] = t;
    }

    // insertion_sort_shift_left(v[s .. s + len), offset)
    DANN_RO_HD void insertion(uint32_t s, uint32_t len, uint32_t offset) {
        for (uint32_t i = offset; i < len; ++i) {
            const E x = v[s + i];
            if (!less(x, v[s + i - 1])) continue;
            uint32_t j = i;
            do {
                v[s + j] = v[s + j - 1];
                --j;
            } while (j > 0 && less(x, v[s + j - 1]));
            v[s + j] = x;
        }
    }
    DANN_RO_HD void cswap(uint32_t s, uint32_t a, uint32_t b) {  // swap_if_less: exchanged when v[b] < v[a]
        if (less(v[s + b], v[s + a])) swap(s + a, s + b);
    }
    DANN_RO_HD void sort9(uint32_t s) {
        // 25 comparators, (a << 4) | b
        const uint8_t net[25] = {0x03, 0x17, 0x25, 0x48, 0x07, 0x24, 0x38, 0x56, 0x02, 0x13, 0x45, 0x78, 0x14,
                                 0x36, 0x57, 0x01, 0x24, 0x35, 0x68, 0x23, 0x45, 0x67, 0x12, 0x34, 0x56};
        for (uint32_t i = 0; i < 25; ++i) cswap(s, net[i] >> 4, net[i] & 15);
    }
    DANN_RO_HD void sort13(uint32_t s) {
        const uint8_t net[45] = {0x0C, 0x1A, 0x29, 0x37, 0x5B, 0x68, 0x16, 0x23, 0x4B, 0x79, 0x8A, 0x04, 0x12, 0x36, 0x78,
                                 0x9A, 0xBC, 0x46, 0x59, 0x8B, 0xAC, 0x05, 0x38, 0x47, 0x6B, 0x9A, 0x01, 0x25, 0x69, 0x78,
                                 0xAB, 0x13, 0x24, 0x56, 0x9A, 0x12, 0x34, 0x57, 0x68, 0x23, 0x45, 0x67, 0x89, 0x34, 0x56};
        for (uint32_t i = 0; i < 45; ++i) cswap(s, net[i] >> 4, net[i] & 15);
    }
    // small_sort_network over v[s .. s + len), len <= 32
    DANN_RO_HD void small_sort(uint32_t s, uint32_t len) {
        if (len < 2) return;
        const uint32_t half = len / 2;
        const bool no_merge = len < 18;
        uint32_t rs = s, rlen = no_merge ? len : half;
        for (;;) {
            uint32_t presorted = 1;
            if (rlen >= 13) {
                sort13(rs);
                presorted = 13;
            } else if (rlen >= 9) {
                sort9(rs);
                presorted = 9;
            }
            insertion(rs, rlen, presorted);
            if (no_merge) return;
            if (rs != s) break;
            rs = s + half;
            rlen = len - half;
        }
        // bidirectional_merge of the two sorted halves into tmp: front takes the left element unless the right one is
        // less, back takes the right element unless it is less than the left one
        uint32_t l = s, r = s + half, lr = s + half - 1, rr = s + len - 1, o = 0, orv = len - 1;
        for (uint32_t i = 0; i < half; ++i) {
            const bool tl = !less(v[r], v[l]);
            tmp[o++] = tl ? v[l] : v[r];
            l += tl ? 1u : 0u;
            r += tl ? 0u : 1u;
            const bool tr = !less(v[rr], v[lr]);
            tmp[orv--] = tr ? v[rr] : v[lr];
            rr -= tr ? 1u : 0u;
            lr -= tr ? 0u : 1u;
        }
        if (len & 1u) tmp[o] = (l < lr + 1) ? v[l] : v[r];
        for (uint32_t i = 0; i < len; ++i) v[s + i] = tmp[i];
    }

    // pivot.rs: median3 / median3_rec / choose_pivot; arguments are indices into v
    DANN_RO_HD uint32_t median3(uint32_t a, uint32_t b, uint32_t c) const {
        const bool x = less(v[a], v[b]);
        const bool y = less(v[a], v[c]);
        if (x == y) {
            const bool z = less(v[b], v[c]);
            return (z != x) ? c : b;
        }
        return a;
    }
    DANN_RO_HD uint32_t choose_pivot(uint32_t s, uint32_t len) {  // len >= 8; returns an offset into the slice
        const uint32_t n8 = len / 8;
        if (len < 64) return median3(s, s + n8 * 4, s + n8 * 7) - s;
        // median3_rec(a, b, c, n): while n * 8 >= 64 each of the three is replaced by the pseudo-median of its own
        // (p, p + 4 (n / 8), p + 7 (n / 8)); frame = [p0, p1, p2, n, k, -]
        uint32_t sp = 0;
        ms[0] = (W)s, ms[1] = (W)(s + n8 * 4), ms[2] = (W)(s + n8 * 7), ms[3] = (W)n8, ms[4] = 0;
        for (;;) {
            W* f = ms + sp * 6;
            if ((uint32_t)f[3] * 8u < 64u || f[4] == 3) {
                const uint32_t r = median3(f[0], f[1], f[2]);
                if (sp == 0) return r - s;
                --sp;
                W* p = ms + sp * 6;
                p[p[4]] = (W)r;
                ++p[4];
                continue;
            }
            const uint32_t m8 = f[3] / 8u, base = f[f[4]];
            W* c = ms + (sp + 1) * 6;
            c[0] = (W)base, c[1] = (W)(base + m8 * 4), c[2] = (W)(base + m8 * 7), c[3] = (W)m8, c[4] = 0;
            ++sp;
        }
    }

    // partition (quicksort.rs): pivot to the front, partition_lomuto_branchless_cyclic over the rest, pivot to its
    // place.  le: the predicate is "element <= pivot" (!less(pivot, element)).  Returns the pivot's final offset.
    DANN_RO_HD uint32_t partition(uint32_t s, uint32_t len, uint32_t pivot_off, bool le) {
        swap(s, s + pivot_off);
        const E pivot = v[s];
        const uint32_t b = s + 1, n = len - 1;
        uint32_t num_lt = 0;
        if (n != 0) {
            const E gap_value = v[b];
            uint32_t gap = 0;
            for (uint32_t right = 1; right < n; ++right) {
                const E e = v[b + right];
                const bool r = le ? !less(pivot, e) : less(e, pivot);
                v[b + gap] = v[b + num_lt];
                v[b + num_lt] = e;
                gap = right;
                num_lt += r ? 1u : 0u;
            }
            const bool r = le ? !less(pivot, gap_value) : less(gap_value, pivot);
            v[b + gap] = v[b + num_lt];
            v[b + num_lt] = gap_value;
            num_lt += r ? 1u : 0u;
        }
        swap(s, s + num_lt);
        return num_lt;
    }

    DANN_RO_HD void sift_down(uint32_t s, uint32_t len, uint32_t node) {
        for (;;) {
            uint32_t child = 2 * node + 1;
            if (child >= len) break;
            if (child + 1 < len && less(v[s + child], v[s + child + 1])) ++child;
            if (!less(v[s + node], v[s + child])) break;
            swap(s + node, s + child);
            node = child;
        }
    }
    DANN_RO_HD void heapsort(uint32_t s, uint32_t len) {
        for (uint32_t i = len + len / 2; i-- > 0;) {
            uint32_t sift_idx;
            if (i >= len) {
                sift_idx = i - len;
            } else {
                swap(s, s + i);
                sift_idx = 0;
            }
            sift_down(s, i < len ? i : len, sift_idx);
        }
    }

    // quicksort(v[s .. s + len), ancestor_pivot, limit): the left part is sorted first (recursion in the original), the
    // right part is pushed with the pivot as its ancestor.  frame = [start, len, limit | has << 15] + its ancestor value
    DANN_RO_HD void quicksort(uint32_t s, uint32_t len, uint32_t limit) {
        uint32_t sp = 0;
        bool has_anc = false;
        E anc = E();
        for (;;) {
            bool done = false;
            if (len <= 32) {
                small_sort(s, len);
                done = true;
            } else if (limit == 0) {
                heapsort(s, len);
                done = true;
            }
            if (done) {
                if (sp == 0) return;
                --sp;
                const W* f = qs + sp * 3;
                s = f[0], len = f[1], anc = anc_st[sp], limit = f[2] & 0x7FFFu, has_anc = ((f[2] >> 15) & 1u) != 0;
                continue;
            }
            --limit;
            const uint32_t pp = choose_pivot(s, len);
            if (has_anc && !less(anc, v[s + pp])) {
                const uint32_t num_le = partition(s, len, pp, true);
                s += num_le + 1;
                len -= num_le + 1;
                has_anc = false;
                continue;
            }
            const uint32_t num_lt = partition(s, len, pp, false);
            W* f = qs + sp * 3;  // the right part, for later
            f[0] = (W)(s + num_lt + 1), f[1] = (W)(len - num_lt - 1), anc_st[sp] = v[s + num_lt];
            f[2] = (W)(limit | 0x8000u);
            ++sp;
            len = num_lt;  // the left part now: same ancestor, same limit
        }
    }

    DANN_RO_HD void sort_unstable(uint32_t s, uint32_t len) {
        if (len < 2) return;
        if (len <= 20) {
            insertion(s, len, 1);
            return;
        }
        uint32_t run = 2;
        const bool desc = less(v[s + 1], v[s]);
        if (desc) {
            while (run < len && less(v[s + run], v[s + run - 1])) ++run;
        } else {
            while (run < len && !less(v[s + run], v[s + run - 1])) ++run;
        }
        if (run == len) {
            if (desc)
                for (uint32_t i = 0, j = len - 1; i < j; ++i, --j) swap(s + i, s + j);
            return;
        }
        uint32_t lg = 0;
        for (uint32_t x = len | 1u; x > 1; x >>= 1) ++lg;
        quicksort(s, len, 2 * lg);
    }

    DANN_RO_HD void select_nth(uint32_t len, uint32_t index) {  // over v[0 .. len), index < len
        if (index == len - 1) {
            uint32_t mx = 0;
            for (uint32_t i = 1; i < len; ++i)
                if (less(v[mx], v[i])) mx = i;
            swap(mx, index);
            return;
        }
        if (index == 0) {
            uint32_t mn = 0;
            for (uint32_t i = 1; i < len; ++i)
                if (less(v[i], v[mn])) mn = i;
            swap(mn, 0);
            return;
        }
        uint32_t s = 0, limit = 16;
        bool has_anc = false;
        E anc = E();
        for (;;) {
            if (len <= 16) {
                if (len >= 2) insertion(s, len, 1);
                return;
            }
            if (limit == 0) {  // median_of_medians in the original: any arrangement with the index-th element in place
                fallback = true;
                sort_unstable(s, len);
                return;
            }
            --limit;
            const uint32_t pp = choose_pivot(s, len);
            if (has_anc && !less(anc, v[s + pp])) {
                const uint32_t mid = partition(s, len, pp, true) + 1;
                if (mid > index) return;
                s += mid;
                len -= mid;
                index -= mid;
                has_anc = false;
                continue;
            }
            const uint32_t mid = partition(s, len, pp, false);
            if (mid < index) {
                anc = v[s + mid];
                has_anc = true;
                s += mid + 1;
                len -= mid + 1;
                index -= mid + 1;
            } else if (mid > index) {
                len = mid;
            } else {
                return;
            }
        }
    }
};

// the prune's form: 16-bit pool positions compared through their distances
struct DistLess {
    const float* d;
    DANN_RO_HD bool operator()(uint16_t a, uint16_t b) const { return d[a] < d[b]; }
};
using Sorter = SorterT<uint16_t, uint16_t, DistLess, 26, 6>;
static_assert(Sorter::work_bytes() <= kWorkBytes, "work area of the 16-bit form");

// positions[0 .. P) = 0 .. P-1 on entry; on return positions[0 .. min(P, max)) is the pool SortedNeighbors::new(pool, max)
// holds.  P <= kMaxLen; `work`: kWorkBytes, 4-byte aligned.  Returns whether the selection's fallback was reached.
DANN_RO_HD inline bool sorted_neighbors(uint16_t* positions, const float* dist, uint32_t P, uint32_t max, void* work) {
    Sorter s(positions, DistLess{dist}, work);
    const uint32_t keep = max < P ? max : P;
    if (keep >= 1) {
        s.select_nth(P, keep - 1);
        s.sort_unstable(0, keep - 1);
    }
    return s.fallback;
}

// `slice.sort_unstable_by(fast_distance)` over whole 8-byte elements -- Neighbor<u32> is (id, distance) -- as the
// post-processing of the filtered searches does it (inline_filter_search.rs:274, multihop_filter_search.rs:207): the
// elements are keys (order-preserving distance bits << 32 | payload), compared by their distance alone and in the
// reference's sense (partial_cmp: -0.0 == +0.0, an unordered pair is Equal).  n < 2^31; `work`: kKeyWorkBytes, 8-byte
// aligned.
struct KeyLess {
    // f32 `a < b` on the order-preserving bits o (ordered_bits: u | 0x80000000 for u >= 0, ~u below), in integer
    // arithmetic: the two zeros (0x7FFFFFFF / 0x80000000) are one value, a NaN (beyond +-infinity) is less than nothing
    // and nothing is less than it.  (The float form of this comparison -- bits back to f32, then `<` -- takes hipcc 7.2's
    // instruction selection down inside the quicksort: scratch/ notes in DESIGN.md.)
    DANN_RO_HD static uint32_t canon(unsigned long long k) {
        const uint32_t o = (uint32_t)(k >> 32);
        return o == 0x7FFFFFFFu ? 0x80000000u : o;
    }
    DANN_RO_HD static bool nan(uint32_t o) { return (o > 0xFF800000u) | (o < 0x007FFFFFu); }
    DANN_RO_HD bool operator()(unsigned long long a, unsigned long long b) const {
        const uint32_t x = canon(a), y = canon(b);
        return (int)!nan(x) & (int)!nan(y) & (int)(x < y);
    }
};
using KeySorter = SorterT<unsigned long long, uint32_t, KeyLess, 66, 12>;
constexpr uint32_t kKeyWorkBytes = 2048;
static_assert(KeySorter::work_bytes() <= kKeyWorkBytes, "work area of the key form");
DANN_RO_HD inline void sort_keys_unstable(unsigned long long* keys, uint32_t n, void* work) {
    KeySorter s(keys, KeyLess{}, work);
    s.sort_unstable(0, n);
}

}  // namespace rust_order
}  // namespace dann
