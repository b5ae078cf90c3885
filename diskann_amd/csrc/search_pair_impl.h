// Two queries per wavefront: the beam search of 128-byte integer rows (u8 / i8 / SQ-8 codes) in the throughput regime.
//
// Why.  With one wavefront per query these rows are bound by instruction issue, not by memory: 585 wave-instructions per
// hop, half of them scalar or branch, for seven memory instructions (profiles/r03_final_u8_insts_1.csv), and more
// resident queries per CU buy nothing (profiles/r04a_visited16_sgpr_ab_int.log: 21 -> 32 per CU, -3 %).  A query with
// L + start points <= 32 and max_degree <= 32 needs 32 lanes for everything except the row gather: the queue is one
// entry per lane, an adjacency row one id per lane.  So a wavefront carries two queries, one per half (lanes 0-31 /
// 32-63), and every instruction of the hop's bookkeeping -- pop, adjacency row, visited filter, compaction, merge --
// is issued once for both.  The row gather keeps its shape (8 lanes per 128-byte row, 4 rows per lane group in
// flight): 16 candidates per half and pass, both halves in every pass.
//
// What stays exactly as in beam_search_one (search_kernel_impl.h), which remains the statement of the algorithm and
// serves every other configuration: the queue rule (rank merge == sequential NeighborPriorityQueue::insert,
// queue.rs:130-171), pop order (queue.rs:297-313), adjacency clamp (neighbors.rs:146-148), the exact visited set
// (glue.rs:542-549; here always the 16-bit table, frozen at 75 % load and continued in a spill table in global
// memory), counters and the result rule (provider.rs:933-944).  Values that are wave-uniform there are uniform per
// half here: they live in pairs of scalars (x0 for lanes 0-31, x1 for lanes 32-63) and are turned into a vector
// operand by one select where a lane needs "its" value.
//
// Round 5: the same hop for L + start points <= 96 (up to three queue entries per lane: entry p of a half in lane
// p & 31, register p >> 5) and max_degree <= 64 (two adjacency ids per lane; the hop's candidates are evaluated and merged in
// two passes of 32 -- merging a hop's candidates in two batches is the same sequence of queue inserts) -- SURVEY 8(a)'s
// C-int8 sizing (L = 64) and config 5's degree.
//
// Eligibility (pair_shape): plain Knn search (beam width 1, no filter, no tags, no record), integer rows of 128
// bytes, L + start points <= 96, at most 32 start points, max_degree <= 64, a 16-bit table geometry for the index, a
// launch beyond the latency regime.  Everything else takes beam_search_kernel.  A query that exhausts table and spill pool reports
// DANN_EOVERFLOW and is re-run by search_with_retry through beam_search_kernel with a larger table.
#pragma once
#include "search_kernel_impl.h"

namespace dann {
namespace {

constexpr uint32_t kPairHalf = 32;
// LDS of one half: candidates (ids, distances), the scatter buffer of the merge ((id, distance) pairs), the queue's
// distances in order (what the lower-bound searches read), the survivors' distances of one merge, a sink for the
// stores of lanes that have nothing to store, the visited table.  QE = queue entries per lane (1: L + start points <=
// 32, 2: <= 64), RE = adjacency ids per lane (1: degree <= 32, 2: <= 64).
struct PairLds {
    uint32_t cand_id_off, cand_d_off, stage_off, qimg_off, qpiv_off, sd_off, sink_off, ht_off, ov_off, half_bytes;
};
// keys of the queue image: a power of two beyond the queue's entries (the lower-bound search needs no bound check)
__host__ __device__ inline uint32_t pair_qimg_keys(uint32_t qe) { return qe == 1u ? 64u : 128u; }
__host__ __device__ inline PairLds pair_lds_layout(uint32_t qe, uint32_t re, uint32_t ht_words, uint32_t ov_words) {
    PairLds l;
    l.cand_id_off = 0;
    l.cand_d_off = 128u * re;
    uint32_t off = 256u * re;
    if (re == 1u) {
        // one pass per hop: the scatter buffer of the merge takes the candidates' place -- they are in registers by then
        l.stage_off = 0;
        off = 256u * qe > off ? 256u * qe : off;
    } else {
        // two passes per hop: the second pass's candidates are still in their buffer when the first pass is merged
        l.stage_off = off;
        off += 256u * qe;
    }
    l.qimg_off = off;  // the entries beyond the queue stay "empty" = larger than every distance
    off += 4u * pair_qimg_keys(qe);
    l.qpiv_off = off;  // one queue entry per lane: the last key of each eighth of the queue image (first level of the
    off += 32u;        // lower-bound search); 32 bytes
    l.sd_off = off;
    off += 128u;
    l.sink_off = off;  // one dword per lane (stores of many lanes to ONE address serialise like a bank conflict); the
    off += 128u;       // 8-byte stores of the merge's scatter sink into [sd, sink + 128): the survivors' keys are dead by then
    l.ht_off = off;
    l.ov_off = l.ht_off + ht_words * 4u;  // the overflow table of the 16-bit table (ov_insert; 0 words: none)
    l.half_bytes = l.ov_off + ov_words * 4u;
    return l;
}
// the instantiation a launch takes: queue entries per lane = ceil((L + start points) / 32) (1 .. 3), two adjacency ids per
// lane beyond degree 32 (which comes with at least two queue entries per lane: five instantiations per metric, not six)
__host__ __device__ inline uint32_t pair_re(const SearchArgs& a) { return a.ix.max_degree > kPairHalf ? 2u : 1u; }
__host__ __device__ inline uint32_t pair_qe(const SearchArgs& a) {
    const uint32_t q = (a.l_value + a.ix.nstart + kPairHalf - 1u) / kPairHalf;
    return (q < 2u && pair_re(a) == 2u) ? 2u : (q ? q : 1u);
}

__device__ __forceinline__ uint32_t rl_u32(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }

// Insert into the open 16-bit table, written without a per-lane branch: the hop is bound by instruction issue (every
// `if` on a lane condition costs an exec-mask save / branch / restore), so the probe is one straight body that all lanes
// run until the last one is done.  Same probe sequence, same table contents as ht16_insert_open.  Returns 1 = inserted
// (the id was new), 2 = no room among its probes (the caller freezes the table), 0 = already present / inactive.
// `own`: a dword of LDS that belongs to this lane alone -- a lane that has nothing (more) to insert swaps THAT word for
// itself: compare-and-swaps of many lanes on one address (the lanes beyond a list's length all carry the same id) are
// served one after the other, like a bank conflict.
__device__ __forceinline__ uint32_t ht16_insert_flat(uint32_t* htw, const Ht16& t, uint32_t id, bool active, uint32_t* own) {
    const uint32_t tagmask = (1u << t.tb) - 1u;
    uint32_t x = id * kHt16A;
    const uint32_t step = id * kHt16B2;
    uint32_t k = 0, res = 0;
    bool pending = active;
    do {
        const uint32_t val = ht16_tag(t, x, tagmask) | (k << t.tb);
        uint32_t* const wp = pending ? htw + ht16_bucket(t, x) : own;
        const uint32_t w = *wp;
        const uint32_t lo = w & 0xFFFFu, hi = w >> 16;
        const bool present = (lo == val) | (hi == val);
        const bool e0 = lo == 0xFFFFu, empty = e0 | (hi == 0xFFFFu);
        const bool tryins = pending & !present & empty;
        const uint32_t put = (0xFFFFu ^ val) << (e0 ? 0u : 16u);  // the first empty half: low, then high
        const uint32_t neww = tryins ? (w ^ put) : w;
        const uint32_t old = atomicCAS(wp, w, neww);  // (a lost swap -- the other half changed -- reads the bucket again)
        const bool inserted = tryins & (old == w);
        res = inserted ? 1u : res;
        const bool next = pending & !present & !empty;  // both halves hold other ids: next probe
        k += next ? 1u : 0u;
        x += next ? step : 0u;
        const bool exh = next & (k >= t.kmax);
        res = exh ? 2u : res;
        pending = pending & !present & !inserted & !exh;
    } while (ballot64(pending));
    return res;
}

// The overflow table of a 16-bit visited table (SearchArgs::ht_ov words, a power of two, 32-bit ids, linear probing):
// it takes the ids whose kmax probes of the 16-bit table are all taken.  With 2^21 index slots and more a 16-bit entry
// has 14 tag bits and room for three probes (six places per id): at the tables' load some 4 % of a search's ids find all
// six taken -- freezing the table at the first of them (rounds 4-5) sent nearly every search of a 10 M-point index
// through the spill tables in global memory (57 ms per 100 000 u8 queries at L = 64, profiles/r06b_*).  A full bucket
// stays full, so "all probes taken" is permanent for an id: insert and every later lookup of it end here alike.  The
// caller keeps one slot free (the probing ends).
__device__ __forceinline__ bool ov_insert(uint32_t* ov, uint32_t mask, uint32_t id) {
    uint32_t h = ((id * 0x9E3779B1u) >> 16) & mask;
    for (;;) {
        const uint32_t old = atomicCAS(&ov[h], kEmpty, id);
        if (old == kEmpty) return true;
        if (old == id) return false;
        h = (h + 1u) & mask;
    }
}
__device__ __forceinline__ bool ov_contains(const uint32_t* ov, uint32_t mask, uint32_t id) {
    uint32_t h = ((id * 0x9E3779B1u) >> 16) & mask;
    for (;;) {
        const uint32_t old = ov[h];
        if (old == kEmpty) return false;
        if (old == id) return true;
        h = (h + 1u) & mask;
    }
}

template <int DT, int OP, bool NORM, int QE, int RE>
__global__ __launch_bounds__(kWave) void pair_search_kernel(SearchArgs a) {
    static_assert(DT == DT_U8 || DT == DT_I8 || DT == DT_SQ8, "integer rows");
    static_assert(QE >= 1 && QE <= 3 && (RE == 1 || (RE == 2 && QE >= 2)), "queue entries / adjacency ids per lane");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr bool SIGNED = DT == DT_I8;
    constexpr uint32_t QP = QE == 1 ? 64u : 128u;  // keys of the queue image (pair_qimg_keys)
    const IndexView& ix = a.ix;
    const uint32_t lane = threadIdx.x, li = lane & 31u;
    const bool up = lane >= kPairHalf;     // this lane serves the second query
    const uint32_t g4 = (lane >> 3) & 3u;  // lane group within the half
    const int v = (int)(lane & 7u);
    const uint32_t R = ix.max_degree, ns = ix.nstart, qcap = a.l_value + ns;  // qcap <= 32 QE, R <= 32 RE, ns <= 32
    const SqParams sqp{ix.sq_k, ix.sq_shift_norm_sq};
    const uint32_t row_stride32 = (uint32_t)ix.row_stride;  // (dann_config::row_stride is 32 bits wide)
    constexpr uint32_t kOverflow = (uint32_t)(-DANN_EOVERFLOW);

    // ---- the two queries of this wavefront (the upper half of the last wavefront of an odd batch idles) --------------
    const uint32_t slot = 2u * blockIdx.x + (up ? 1u : 0u);
    const bool exists = slot < a.nq;
    const uint32_t slot_c = exists ? slot : a.nq - 1u;
    const uint32_t qi = a.qmap ? a.qmap[slot_c] : slot_c;

    const PairLds L = pair_lds_layout(QE, RE, a.ht_entries, a.ht_ov);
    uint8_t* const hbase = smem + (up ? L.half_bytes : 0u);
    uint32_t* const cand_id = reinterpret_cast<uint32_t*>(hbase + L.cand_id_off);
    float* const cand_d = reinterpret_cast<float*>(hbase + L.cand_d_off);
    uint2* const stage = reinterpret_cast<uint2*>(hbase + L.stage_off);
    // distances in these two are order-preserving integer keys (ordered_bits): the rank arithmetic of the merge is
    // integer compares feeding add-with-carry, no lane-mask logic on the scalar unit
    uint32_t* const qimg = reinterpret_cast<uint32_t*>(hbase + L.qimg_off);
    uint32_t* const qpiv = reinterpret_cast<uint32_t*>(hbase + L.qpiv_off);
    uint32_t* const sd = reinterpret_cast<uint32_t*>(hbase + L.sd_off);
    uint32_t* const sink = reinterpret_cast<uint32_t*>(hbase + L.sink_off) + li;  // this lane's own sink
    uint2* const sink2 = reinterpret_cast<uint2*>(hbase + L.sd_off) + li;
    uint32_t* const ht = reinterpret_cast<uint32_t*>(hbase + L.ht_off);
    uint32_t* const ov = reinterpret_cast<uint32_t*>(hbase + L.ov_off);  // (behind the table: wiped with it)
    const Ht16 h16 = ht16_of(a);
    const uint32_t ht_limit = a.ht_open;  // ids the open table takes (75 % of its entries by default)
    // the overflow table takes the ids that find their few probes taken (indexes of 2^21 slots and more: 14 tag bits
    // leave three probes); it always keeps a slot free (linear probing ends), so it is open while ovcv + a hop's ids fit
    const uint32_t ov_mask = a.ht_ov - 1u, ov_limit = a.ht_ov ? a.ht_ov - 1u : 0u;
    {
        const u32x4 e4 = {kEmpty, kEmpty, kEmpty, kEmpty};
        for (uint32_t i = li * 4u; i < a.ht_entries + a.ht_ov; i += kPairHalf * 4u) *reinterpret_cast<u32x4*>(ht + i) = e4;
#pragma unroll
        for (int e = 0; e < RE; ++e) cand_id[e * kPairHalf + li] = 0u;
#pragma unroll
        for (uint32_t e = 0; e < QP / kPairHalf; ++e) qimg[e * kPairHalf + li] = kEmpty;
        if (li < 8u) qpiv[li] = kEmpty;
    }
    // the lane's 16 query bytes and the query's squared norm stay in registers (as in the fixed-length integer path)
    const uint8_t* const qsrc = reinterpret_cast<const uint8_t*>(a.queries) + (uint64_t)qi * ix.layer_bytes;
    uint4 xqi;
    {
        const uint32_t* qw = reinterpret_cast<const uint32_t*>(qsrc + 16 * v);  // (SQ-8 queries are 4-byte aligned only)
        xqi = make_uint4(qw[0], qw[1], qw[2], qw[3]);
    }
    const int xx_pre = group_norm_int_pre<SIGNED>(xqi);

    // ---- state.  Queue: entry p of a half in lane p & 31, register p >> 5.  What is wave-uniform in beam_search_one is
    // uniform per half here and kept in vector registers (every lane holds the value of its half): no scalar selects.
    uint32_t qid[QE];
    float qd[QE];
#pragma unroll
    for (int e = 0; e < QE; ++e) {
        qid[e] = kEmpty;
        qd[e] = 0.0f;
    }
    uint32_t sizev = 0, cmpsv = 0, hopsv = 0, htcv = ns, ovcv = 0, spcv = 0, stv = 0, ncv = 0;
    bool alivev = exists, openv = true;
    uint32_t* spv = nullptr;  // the half's spill table once its LDS table is frozen
    const uint32_t spill_size = 1u << a.spill_bits, spill_mask = spill_size - 1u, spill_shift = 32u - a.spill_bits;
    const uint32_t spill_limit = spill_size - (spill_size >> 2);
    // (cold) claims a table of the spill pool for the half whose first lane is `first_lane`; null if none is free
    auto claim_spill = [&](uint32_t first_lane) -> uint32_t* {
        uint32_t slice = kEmpty;
        if (a.spill) {
            if (lane == first_lane) {
                uint32_t* busy = a.spill_next + 16;
                uint32_t s = atomicAdd(a.spill_next, 1u) % a.spill_slices;
                for (uint32_t t = 0; t < 2u * a.spill_slices; ++t) {
                    if (atomicCAS(&busy[s], 0u, 1u) == 0u) {
                        slice = s;
                        break;
                    }
                    s = (s + 1 == a.spill_slices) ? 0u : s + 1;
                }
            }
            slice = rl_u32(slice, (int)first_lane);
        }
        return slice < a.spill_slices ? a.spill + ((uint64_t)slice << a.spill_bits) : nullptr;
    };
    // (cold) freezes the table of every half with a lane in `want`: new ids go to a spill table from here on; a half
    // that gets none gives up (DANN_EOVERFLOW: the host re-runs the query with one wavefront and a larger table)
    auto freeze = [&](bool want) {
        const uint64_t fm = ballot64(want);
        uint32_t* p0 = nullptr;
        uint32_t* p1 = nullptr;
        if ((uint32_t)fm) p0 = claim_spill(0u);
        if ((uint32_t)(fm >> 32)) p1 = claim_spill(kPairHalf);
        const bool mine = up ? ((uint32_t)(fm >> 32) != 0u) : ((uint32_t)fm != 0u);
        if (mine) {
            openv = false;
            if (!spv) spv = up ? p1 : p0;
            if (!spv) stv = kOverflow;
        }
    };

    // ---- start points: frozen slots [capacity, capacity + nstart) (index.rs:1950-1958), the candidates of "hop 0" ----
    {
        const bool on = (li < ns) & alivev;
        const uint32_t id = ix.capacity + li;
        if (li < ns) cand_id[li] = id;
        __syncthreads();  // the tables are wiped
        if (ballot64(ht16_insert_flat(ht, h16, id, on, sink) == 2u)) stv = kOverflow;  // (a table of >= 64 entries: never)
        ncv = alivev ? ns : 0u;
    }
    __syncthreads();

    uint32_t pfnv = kEmpty;  // the node whose adjacency row was requested ahead (pf_len / pf_val), per half
    uint32_t pf_len = 0, pf_val[RE];
#pragma unroll
    for (int e = 0; e < RE; ++e) pf_val[e] = kEmpty;
    for (;;) {
        // ---- distances of the candidates and their merge into the queues, 32 candidates of each half per pass (one pass
        // per hop up to degree 32, two beyond).  Merging a hop's candidates in two batches equals merging them at once:
        // either way it is the sequence of NeighborPriorityQueue::insert calls in adjacency order (queue.rs:130-171).
        const uint32_t nc0 = rl_u32(ncv, 0), nc1 = rl_u32(ncv, (int)kPairHalf);
        const uint32_t ncboth = nc0 > nc1 ? nc0 : nc1;
        cmpsv += ncv;
#pragma nounroll
        for (uint32_t c0 = 0; c0 < (RE == 1 ? 1u : ncboth); c0 += kPairHalf) {
            // 4 lane groups per half, every row of the pass requested before the first one is evaluated (one memory round
            // trip per pass: 4, 6 or 8 rows per lane group, by the larger of the halves' candidate counts).  The sums end
            // up in lanes 0-3 of a group: lane u stores the result of row u.
            {
                const uint32_t ncmax = ncboth - c0;
                auto gather = [&](auto tag) {
                    constexpr int U = decltype(tag)::value;
                    const uint8_t* rows[U];
                    float out[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const uint32_t ci = c0 + (uint32_t)u * 4u + g4;
                        const uint32_t raw = cand_id[ci];            // (read first, select after: no load under a branch)
                        const uint32_t id = ci < ncv ? raw : 0u;     // unused slots evaluate row 0
                        rows[u] = ix.rows + (uint64_t)id * row_stride32;  // (one 32 x 32 -> 64-bit multiply-add per row)
                    }
                    group_distance_int_pre<OP, SIGNED, U>(xqi, xx_pre, rows, v, out);
#pragma unroll
                    for (int h = 0; h < (U + 3) / 4; ++h) {
                        float val = out[4 * h];
                        const uint8_t* row = rows[4 * h];
#pragma unroll
                        for (int u = 1; u < 4 && 4 * h + u < U; ++u) {
                            val = v == u ? out[4 * h + u] : val;
                            if constexpr (DT == DT_SQ8 && OP != OP_L2) row = v == u ? rows[4 * h + u] : row;
                        }
                        const uint32_t ci = c0 + (uint32_t)(4 * h + v) * 4u + g4;
                        const bool ok = (v < 4) & (4 * h + v < U) & (ci < ncv);
                        float* dst = ok ? cand_d + (ci & (kPairHalf * RE - 1u)) : reinterpret_cast<float*>(sink);
                        *dst = finish_distance<DT, OP, NORM>(val, qsrc, row, ix.dim, sqp);
                    }
                };
                if (ncmax > 24u) gather(std::integral_constant<int, 8>());
                else if (ncmax > 16u) gather(std::integral_constant<int, 6>());
                else if (ncmax) gather(std::integral_constant<int, 4>());
            }
            __syncthreads();

            // ---- merge (see merge_regs in beam_search_one for the rule and its proof): a surviving candidate j lands at
            // #{old e: d_e < d_j} + #{surviving i: d_i < d_j or (d_i == d_j, i > j)}, an old entry e moves up by
            // #{surviving j: d_j <= d_e}.  The survivors' distances go to LDS in emission order; every lane walks them
            // (broadcast reads), the lower bounds come from a binary search in the queue's distances.
            {
                const bool has = c0 + li < ncv;
                const float nd = cand_d[c0 + li];
                const uint32_t nid = cand_id[c0 + li];
                asm volatile("" ::: "memory");  // (one pass per hop: the scatter below is written over the candidates' buffers)
                const uint32_t oknd = ordered_bits(nd);
                uint32_t okq[QE];
#pragma unroll
                for (int e = 0; e < QE; ++e) okq[e] = ordered_bits(qd[e]);
                // a full queue rejects what is worse than its last entry (queue.rs:142-146)
                const uint32_t okw = qimg[(sizev - 1u) & (QP - 1u)];
                const bool nvalid = has & (nd == nd) & !((sizev == qcap) & (okw < oknd));
                const uint64_t km = ballot64(nvalid);
                if (km) {
                    const uint32_t k0 = (uint32_t)km, k1 = (uint32_t)(km >> 32);
                    const uint32_t nvv = (uint32_t)__popc(up ? k1 : k0);
                    const uint32_t nvmax = max((uint32_t)__popc(k0), (uint32_t)__popc(k1));
                    const uint32_t cj = __builtin_amdgcn_mbcnt_hi(k1, up ? 0u : __builtin_amdgcn_mbcnt_lo(k0, 0u));
                    sd[li] = kEmpty;  // (keys beyond a half's survivors: larger than every distance)
                    *(nvalid ? sd + cj : sink) = oknd;
                    __syncthreads();
                    uint32_t before = 0, shift[QE];
#pragma unroll
                    for (int e = 0; e < QE; ++e) shift[e] = 0;
                    // the survivors' keys, four per LDS round trip (the buffer is padded with keys larger than every distance)
#pragma nounroll
                    for (uint32_t t = 0; t < nvmax; t += 4u) {
                        const u32x4 k4 = *reinterpret_cast<const u32x4*>(sd + t);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const uint32_t okj = k4[i];
                            before += okj < oknd + (t + (uint32_t)i > cj ? 1u : 0u) ? 1u : 0u;  // d_j < d, or equal and emitted later
#pragma unroll
                            for (int e = 0; e < QE; ++e) shift[e] += okj <= okq[e] ? 1u : 0u;  // (entries at or beyond the size are never scattered)
                        }
                    }
                    // #{old e: d_e < d}.  The hop of this kernel is a chain of LDS round trips (round 6: requesting the
                    // gather's rows a hop ahead changes nothing, profiles/README.md), and a binary search in the queue
                    // image is six or seven DEPENDENT ones.  One queue entry per lane (64 keys): the eighth of the image
                    // the key falls into (eight pivots, one broadcast read), then that eighth's eight keys -- two round
                    // trips, 33 vector instructions instead of 24: 1.82 -> 1.76 ms at L = 26 (u8; SQ-8 1.85 -> 1.77) on
                    // the same box.  Longer queues keep the binary search: the 16-key eighths of a 128-key image cost more
                    // vector work than the five round trips they save (L = 64: 4.88 -> 4.90 ms; scratch/r06_lib_ab2.sh).
                    uint32_t lb = 0;
                    if constexpr (QE == 1) {
                        constexpr uint32_t B = QP / 8u;
                        const u32x4 p0 = *reinterpret_cast<const u32x4*>(qpiv), p1 = *reinterpret_cast<const u32x4*>(qpiv + 4);
#pragma unroll
                        for (int i = 0; i < 4; ++i) lb += (p0[i] < oknd ? 1u : 0u) + (p1[i] < oknd ? 1u : 0u);
                        lb *= B;
                        const uint32_t* blk = qimg + lb;
#pragma unroll
                        for (uint32_t q4 = 0; q4 < B / 4u; ++q4) {
                            const u32x4 kk = *reinterpret_cast<const u32x4*>(blk + 4u * q4);
#pragma unroll
                            for (int i = 0; i < 4; ++i) lb += kk[i] < oknd ? 1u : 0u;
                        }
                    } else {
#pragma unroll
                        for (uint32_t step = QP / 2u; step > 0; step >>= 1) lb = qimg[lb + step - 1u] < oknd ? lb + step : lb;
                    }
                    const uint32_t pos_new = lb + before;
#pragma unroll
                    for (int e = 0; e < QE; ++e) {
                        const uint32_t p = (uint32_t)e * kPairHalf + li, np = p + shift[e];
                        *(((p < sizev) & (np < qcap)) ? stage + np : sink2) = make_uint2(qid[e], __builtin_bit_cast(uint32_t, qd[e]));
                    }
                    *((nvalid & (pos_new < qcap)) ? stage + pos_new : sink2) = make_uint2(nid, __builtin_bit_cast(uint32_t, nd));
                    const uint32_t total = sizev + nvv;
                    sizev = total < qcap ? total : qcap;
                    __syncthreads();
#pragma unroll
                    for (int e = 0; e < QE; ++e) {
                        const uint32_t p = (uint32_t)e * kPairHalf + li;
                        const uint2 en = stage[p];
                        const bool in = p < sizev;
                        qid[e] = in ? en.x : qid[e];
                        qd[e] = in ? __builtin_bit_cast(float, en.y) : qd[e];
                        const uint32_t key = in ? ordered_bits(qd[e]) : kEmpty;
                        qimg[p] = key;
                        if constexpr (QE == 1) *(((p & (QP / 8u - 1u)) == QP / 8u - 1u) ? qpiv + p / (QP / 8u) : sink) = key;
                    }
                }
            }
            if constexpr (RE > 1) __syncthreads();  // (the next pass reads the queue image this one wrote)
        }

        // ---- pop: the closest unexpanded entry of each queue (queue.rs:297-313); a half without one has finished -------
        uint32_t nextv;  // the next unexpanded entry after it: the node the following hop expands unless a new candidate
                         // gets in front of it (kEmpty: none)
        uint32_t nodev;
        {
            uint64_t um[QE];
            uint64_t any = 0;
#pragma unroll
            for (int e = 0; e < QE; ++e) {
                um[e] = ballot64(((uint32_t)e * kPairHalf + li < sizev) & !(qid[e] & kVisitedBit) & alivev & (stv == 0u));
                any |= um[e];
            }
            if (!any) break;
            const uint32_t hb = up ? kPairHalf : 0u;
            if constexpr (QE == 1) {
                const uint32_t mybits = up ? (uint32_t)(um[0] >> 32) : (uint32_t)um[0];
                alivev = mybits != 0u;
                const uint32_t l1 = (uint32_t)__builtin_ctz(mybits | 0x80000000u);
                const uint32_t rest = mybits & (mybits - 1u);
                const uint32_t l2 = (uint32_t)__builtin_ctz(rest | 0x80000000u);
                const uint32_t n1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((hb + l1) << 2), (int)qid[0]);
                const uint32_t n2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((hb + l2) << 2), (int)qid[0]);
                nodev = alivev ? n1 : 0u;
                nextv = (alivev & (rest != 0u)) ? n2 : kEmpty;
                qid[0] |= (alivev & (li == l1)) ? kVisitedBit : 0u;
            } else {
                // entry p of a half sits in lane p & 31 of register p >> 5: the first two set positions over the QE masks,
                // from the last register down (what an earlier register holds precedes what has been found so far)
                uint32_t p1 = kEmpty, p2 = kEmpty;
#pragma unroll
                for (int e = QE - 1; e >= 0; --e) {
                    const uint32_t m = up ? (uint32_t)(um[e] >> 32) : (uint32_t)um[e];
                    const uint32_t rest = m & (m - 1u);
                    const uint32_t f1 = (uint32_t)e * kPairHalf + (uint32_t)__builtin_ctz(m | 0x80000000u);
                    const uint32_t f2 = (uint32_t)e * kPairHalf + (uint32_t)__builtin_ctz(rest | 0x80000000u);
                    p2 = m ? (rest ? f2 : p1) : p2;
                    p1 = m ? f1 : p1;
                }
                alivev = p1 != kEmpty;
                const int a1 = (int)((hb + (p1 & 31u)) << 2), a2 = (int)((hb + (p2 & 31u)) << 2);
                uint32_t n1 = 0, n2 = 0;
#pragma unroll
                for (int e = 0; e < QE; ++e) {
                    const uint32_t t1 = (uint32_t)__builtin_amdgcn_ds_bpermute(a1, (int)qid[e]);
                    const uint32_t t2 = (uint32_t)__builtin_amdgcn_ds_bpermute(a2, (int)qid[e]);
                    n1 = (p1 >> 5) == (uint32_t)e ? t1 : n1;
                    n2 = (p2 >> 5) == (uint32_t)e ? t2 : n2;
                }
                nodev = alivev ? n1 : 0u;
                nextv = (alivev & (p2 != kEmpty)) ? n2 : kEmpty;
                const bool mine = alivev & (li == (p1 & 31u));
#pragma unroll
                for (int e = 0; e < QE; ++e) qid[e] |= (mine & ((p1 >> 5) == (uint32_t)e)) ? kVisitedBit : 0u;
            }
            hopsv += alivev ? 1u : 0u;
        }

        // ---- expand: adjacency row (requested a hop ahead when the prediction held), visited filter, compaction ---------
        {
            const bool miss = alivev & (nodev != pfnv);
            uint32_t lenl = pf_len, vall[RE];
#pragma unroll
            for (int e = 0; e < RE; ++e) vall[e] = pf_val[e];
            if (ballot64(miss)) {  // some half has to read its row now (both do: a half that had it reads the same again)
                const uint32_t* arow = ix.adj + (uint64_t)nodev * ix.adj_stride;
                lenl = arow[0];
#pragma unroll
                for (int e = 0; e < RE; ++e) {
                    const uint32_t j = (uint32_t)e * kPairHalf + li;
                    vall[e] = arow[1u + (j < R ? j : R - 1u)];
                }
            }
            const uint32_t len = lenl < R ? lenl : R;  // Neighbors::get clamps (neighbors.rs:146-148)
            // the open table takes ids up to 75 % of its entries; then it is frozen and new ids go to a spill table
            const bool live = alivev & (stv == 0u);
            const bool tofreeze = (htcv + len > ht_limit) | ((a.ht_ov != 0u) & (ovcv + len > ov_limit));
            if (ballot64(live & (openv ? tofreeze : (spcv + len > spill_limit)))) {
                freeze(live & openv & tofreeze);
                if (alivev & !openv & (!spv | (spcv + len > spill_limit))) stv = kOverflow;
            }
            uint32_t total_new = 0;
#pragma unroll
            for (int e = 0; e < RE; ++e) {
                if (e > 0 && !ballot64(alivev & (stv == 0u) & (len > kPairHalf))) break;  // (no list of this hop is that long)
                const uint32_t j = (uint32_t)e * kPairHalf + li;
                const bool inb = alivev & (stv == 0u) & (j < len);
                const uint32_t id = inb ? vall[e] : kEmpty;
                // (the 16-bit table holds ids below 2^m only; an id beyond the index is never a candidate anyway)
                const bool act = inb & (id != kEmpty) & (id < ix.nslots);
                const uint32_t r = ht16_insert_flat(ht, h16, id, act & openv, sink);
                bool isnew = r == 1u;
                const bool exh = r == 2u;
                if (ballot64(exh | (act & !openv))) {  // (rare) no room among an id's probes, or a frozen table
                    if (ballot64(exh)) {
                        if (a.ht_ov) {  // the id goes to the half's overflow table (room for this hop: checked above)
                            if (exh) isnew = ov_insert(ov, ov_mask, id);
                            const uint64_t om = ballot64(exh & isnew);
                            ovcv += (uint32_t)__popc(up ? (uint32_t)(om >> 32) : (uint32_t)om);
                        } else {
                            freeze(exh);  // that half's table is frozen; the id goes to its spill table
                            if (exh && spv) isnew = spill_insert(spv, spill_mask, spill_shift, id);
                        }
                    }
                    if (act && !openv && !exh && spv && !stv && !isnew)
                        isnew = !ht16_contains(ht, h16, id) && !(a.ht_ov && ov_contains(ov, ov_mask, id)) &&
                                spill_insert(spv, spill_mask, spill_shift, id);
                }
                const uint64_t km = ballot64(isnew);
                const uint32_t k0 = (uint32_t)km, k1 = (uint32_t)(km >> 32);
                const uint32_t rank = total_new + __builtin_amdgcn_mbcnt_hi(k1, up ? 0u : __builtin_amdgcn_mbcnt_lo(k0, 0u));
                *(isnew ? cand_id + (rank & (kPairHalf * RE - 1u)) : sink) = id;
                total_new += (uint32_t)__popc(up ? k1 : k0);
            }
            ncv = stv ? 0u : total_new;
            htcv += openv ? ncv : 0u;
            spcv += openv ? 0u : ncv;
        }
        // ---- request the adjacency row of the predicted next node; it lands while the rows are evaluated --------------
        {
            pfnv = nextv;
            const uint32_t* prow = ix.adj + (uint64_t)(pfnv != kEmpty ? pfnv : 0u) * ix.adj_stride;
            pf_len = prow[0];
#pragma unroll
            for (int e = 0; e < RE; ++e) {
                const uint32_t j = (uint32_t)e * kPairHalf + li;
                pf_val[e] = prow[1u + (j < R ? j : R - 1u)];
            }
        }
        __syncthreads();
    }

    // ---- spill tables go back clean ------------------------------------------------------------------------------------
    if (ballot64(spv != nullptr)) {
        __syncthreads();
        if (spv) {
            const u32x4 e = {kEmpty, kEmpty, kEmpty, kEmpty};
            for (uint32_t i = li * 4u; i < spill_size; i += kPairHalf * 4u) {
                uint32_t* p = spv + i;
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(e) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (spv && li == 0u) atomicExch(a.spill_next + 16 + (uint32_t)((spv - a.spill) >> a.spill_bits), 0u);
    }
    // ---- results: best entries in order, start points dropped (provider.rs:933-944) -----------------------------------
    {
        uint32_t wv = 0;
        uint32_t* oi = a.out_ids + (uint64_t)qi * a.k;
        float* od = a.out_dists + (uint64_t)qi * a.k;
#pragma unroll
        for (int e = 0; e < QE; ++e) {
            const uint32_t id = qid[e] & ~kVisitedBit;
            const bool res = exists & ((uint32_t)e * kPairHalf + li < sizev) & (id < ix.capacity);
            const uint64_t rm = ballot64(res);
            const uint32_t r0 = (uint32_t)rm, r1 = (uint32_t)(rm >> 32);
            const uint32_t r = wv + __builtin_amdgcn_mbcnt_hi(r1, up ? 0u : __builtin_amdgcn_mbcnt_lo(r0, 0u));
            if (a.out_ids && res && r < a.k) {
                oi[r] = id;
                od[r] = qd[e];
            }
            wv += (uint32_t)__popc(up ? r1 : r0);
        }
        const uint32_t written = wv < a.k ? wv : a.k;
        if (a.out_ids && exists) {
            for (uint32_t t = written + li; t < a.k; t += kPairHalf) {
                oi[t] = kEmpty;
                od[t] = __builtin_inff();
            }
        }
        if (li == 0u && exists) {
            if (a.stats) {
                dann_search_stats st;
                st.cmps = cmpsv;
                st.hops = hopsv;
                // Translate::post_process counts a push only while the buffer still has room afterwards
                // (provider.rs:933-944, search_output_buffer.rs:107-124): k - 1 when the buffer of length k fills
                st.result_count = (a.k && written == a.k) ? a.k - 1u : written;
                st.written = written;
                st.status = stv;
                a.stats[qi] = st;
            }
            if (stv && a.fail_flag) __hip_atomic_store(a.fail_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// what the pair kernel serves (host side; the table geometry is checked by the caller)
inline bool pair_shape(const SearchArgs& a) {
    const int dt = a.ix.dtype;
    return plain_mode(a) && !a.team && !a.grid && !a.srv.ring && !a.rec_ids && !a.range_ids && !a.qslots && a.out_ids &&
           (dt == DT_U8 || dt == DT_I8 || dt == DT_SQ8) && a.ix.dim == 128u && a.l_value + a.ix.nstart <= 3u * kPairHalf &&
           a.ix.max_degree <= 2u * kPairHalf && a.ix.nstart >= 1u && a.ix.nstart <= kPairHalf && a.ix.row_stride % 16u == 0u;
}

template <int DT>
int32_t launch_pair_dt(const SearchArgs& a, size_t lds, hipStream_t stream) {
    int op;
    bool norm;
    if (!resolve_metric(a.ix.dtype, a.ix.metric, &op, &norm)) {
        set_error("metric %d is not defined for dtype %d", a.ix.metric, a.ix.dtype);
        return DANN_EUNSUPPORTED;
    }
    const uint32_t grid = (a.nq + 1u) / 2u;
    auto go = [&](auto kern) -> int32_t {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               160 * 1024);
            if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kWave), lds, stream, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "pair_search_kernel launch");
        return DANN_OK;
    };
    const uint32_t qe = pair_qe(a), re = pair_re(a);
#define DANN_PAIR_GO(OPV, NORMV)                                                                   \
    return re == 2u ? (qe == 3u ? go(pair_search_kernel<DT, OPV, NORMV, 3, 2>)                     \
                                : go(pair_search_kernel<DT, OPV, NORMV, 2, 2>))                    \
         : qe == 3u ? go(pair_search_kernel<DT, OPV, NORMV, 3, 1>)                                 \
         : qe == 2u ? go(pair_search_kernel<DT, OPV, NORMV, 2, 1>)                                 \
                    : go(pair_search_kernel<DT, OPV, NORMV, 1, 1>)
    if (op == OP_L2) {
        if constexpr (DT == DT_SQ8) {
            if (norm) DANN_PAIR_GO(OP_L2, true);
        }
        DANN_PAIR_GO(OP_L2, false);
    }
    if (op == OP_IP) DANN_PAIR_GO(OP_IP, false);
    if constexpr (DT != DT_SQ8) DANN_PAIR_GO(OP_COS, false);
#undef DANN_PAIR_GO
    return DANN_EUNSUPPORTED;
}

}  // namespace
}  // namespace dann
