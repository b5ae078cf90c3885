// search_kernels.hip -- batched Vamana beam search, one wavefront per query.
//
// Replaces, for a batch of independent queries, the reference call chain
//   DiskANNIndex::search_internal          diskann/src/graph/index.rs:1933-2000
//     NeighborPriorityQueue                diskann/src/neighbor/queue.rs:130-318
//     SearchAccessor::expand_beam          diskann-inmem/src/provider.rs:436-480
//       Neighbors::get                     diskann-inmem/src/neighbors.rs:124-163
//       NotInMut (visited set)             diskann/src/graph/glue.rs:524-561
//       expand_beam_inner + QueryDistance  diskann-inmem/src/provider.rs:620-690, layers/full.rs:317-336
//   Translate::post_process                diskann-inmem/src/provider.rs:899-950
// with results identical to the CPU path (ids, distances, cmps, hops).
//
// Design (MI355X): the whole beam loop of one query runs inside one 64-lane wavefront
// (workgroup = 1 wave, so the only barriers are wave-local).  Per hop:
//   1. pop the W closest unexpanded queue entries (ballot + readlane; the sorted L-queue
//      lives in registers, entry p in lane p%64 slot p/64);
//   2. read their adjacency rows (one coalesced 4*(R+1)-byte read each), test-and-insert
//      every neighbour id into an exact open-addressing visited table in LDS
//      (ds_cmpst), compact the survivors in adjacency order (ballot + mbcnt);
//   3. gather: G lanes per surviving candidate row, 16-byte loads, 64/G rows per
//      wave-instruction, U rows in flight per lane group -- random 512-byte rows are
//      read as whole 128-byte lines; FMA chains in the reference's association order;
//   4. merge the (id, dist) batch into the queue by rank: the sequential
//      `insert` calls of index.rs:1986-1988 keep the best `capacity` elements under the
//      total order (distance asc, insertion time desc) -- queue.rs:142-170: lower-bound
//      insertion puts a new element *before* equal-distance ones, a full queue drops its
//      last element, and an element worse than the last is rejected -- so inserting a
//      batch one by one equals taking the top-`capacity` of old ∪ new under that order.
//      Ranks are computed with wave-uniform readlane broadcasts, the permutation goes
//      through an LDS staging buffer.
// HBM traffic per query = cmps * row bytes + hops * adjacency row; everything else stays
// in registers/LDS.
#include "search_kernel_impl.h"
#include "search_pair_impl.h"
#include "search_pq_impl.h"

namespace dann {
#ifdef DANN_PHASE_CYCLES
unsigned long long* dann_phase_buffer();
#endif
namespace {

// collect the indices of queries whose status is non-zero
__global__ void collect_failed_kernel(const dann_search_stats* stats, const uint32_t* qmap, uint32_t n, uint32_t* count,
                                      uint32_t* list) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t q = qmap ? qmap[t] : t;
    if (stats[q].status) list[atomicAdd(count, 1u)] = q;
}

}  // namespace

size_t search_lds_bytes(const SearchArgs& a) {
    if (a.pair) return 2u * (size_t)pair_lds_layout(pair_qe(a), pair_re(a), a.ht_entries, a.ht_ov).half_bytes;
    if (a.pqlut) return pq_lds_layout(pq_lut_qs(a), a.ht_entries, a.ht_ov).total;
    return search_lds_layout(a.ht_entries, cmax_of(a), lds_queue_entries(a), query_lds_bytes(a.ix), a.team != 0).total;
}

// ---- sizing of the LDS visited table ---------------------------------------------------------
// The table trades occupancy (LDS per query) against probe length and the spill rate; results
// never depend on it.  Measured on MI355X (1M x 128 f32, R = 32, L = 10..250): LDS is allocated in
// 1280-byte granules (128 per CU), occupancy is capped by the kernel's VGPRs anyway (16 queries per CU
// for the 128-d f32 kernel, 24 for the integer kernels) so LDS up to that point is free, and the best
// size sits at the top of the occupancy step that holds about the 90th percentile of comparisons
// per query at 75 % load.
constexpr uint32_t kLdsGranule = 1280, kLdsGranules = 128, kHistBins = 512;

uint32_t snap_visited_entries(SearchArgs a, uint32_t cap_ids, uint32_t useful_waves) {
    a.ht_entries = 0;
    const int64_t other = (int64_t)search_lds_bytes(a);
    uint64_t need = ((uint64_t)((double)cap_ids / 0.75) + 63) / 64 * 64;
    need = std::min<uint64_t>(std::max<uint64_t>(need, 256), 32768);
    const uint64_t granules = ((uint64_t)other + need * 4 + kLdsGranule - 1) / kLdsGranule;
    if (granules > kLdsGranules) return (uint32_t)need;
    const uint32_t waves = std::min<uint32_t>(kLdsGranules / (uint32_t)granules, useful_waves);
    int64_t top = ((int64_t)(kLdsGranules / waves) * kLdsGranule - other) / 4 / 64 * 64;
    // beyond ~8 slots per id the probe chains are already one step long; a larger table only costs its wipe
    top = std::min<int64_t>(top, ((int64_t)cap_ids * 8 + 63) / 64 * 64);
    return (uint32_t)std::min<int64_t>(std::max<int64_t>(top, (int64_t)need), 32768);
}

uint32_t largest_prime_leq(uint32_t n);

// ---- 16-bit table entries (SearchArgs::ht16; device side: ht16_insert_open) -----------------------------------------
// Geometry of a table of `words` dwords = `words` buckets of two 16-bit entries (any count) for ids below the index's
// slot count: m id bits; the ids of one bucket are at most ceil(2^m / words) consecutive values, told apart by tb tag
// bits; 16 - tb bits are left for the probe number (at least two: three probes = six places per id).
struct Ht16Geom {
    bool ok = false;
    uint32_t shift = 0, tb = 0, kmax = 0, slots = 0;  // slots = 2 * words: the entries the table holds
};
Ht16Geom ht16_geometry(uint32_t words, uint32_t nslots, uint32_t kcap = 64u) {
    Ht16Geom g;
    if (words < 32u || words > 65536u) return g;
    uint32_t m = 1;
    while (m < 32u && (1ull << m) < (uint64_t)nslots) ++m;
    if (m >= 32u) return g;
    const uint64_t per_bucket = ((1ull << m) + words - 1) / words;  // ids of one bucket: at most this many consecutive values
    uint32_t tb = 0;
    while ((1ull << tb) < per_bucket) ++tb;
    if (tb > 14u) return g;  // fewer than 2 bits for the probe number: too few probes per id
    g.tb = tb;
    g.shift = 32u - m;
    g.kmax = std::min<uint32_t>((1u << (16u - tb)) - 1u, std::max<uint32_t>(kcap, 1u));  // (kcap: DANN_DBG_HT16_MAX_PROBES)
    g.slots = words * 2u;
    g.ok = true;
    return g;
}
// Overflow table of the pair / PQ-table kernels (SearchArgs::ht_ov, ov_insert): where a 16-bit entry leaves fewer than
// eight probes per id (indexes of 2^18 slots and more at these table sizes) some percent of a search's ids find all of
// them taken (simulated at 75 % load: 75 of 2 064 ids with three probes, 5 with seven); a small table of 32-bit ids
// takes those instead of freezing the whole table at the first of them.  Words per query (a power of two).
uint32_t ht16_overflow_words(const Ht16Geom& g, bool pair) { return !g.ok || g.kmax >= 8u ? 0u : pair ? 128u : 256u; }
// may this launch use 16-bit entries at all?  (plain-mode kernels, one wave per query)
bool ht16_eligible(const SearchArgs& a) { return plain_mode(a) && !a.team; }

uint32_t ht16_kcap(const dann_index* idx) { return std::min(64u, std::max(1u, idx->dbg_u32(DANN_DBG_HT16_MAX_PROBES, 64u))); }
uint32_t ht16_open_eighths(const dann_index* idx) { return std::min(7u, std::max(4u, idx->dbg_u32(DANN_DBG_HT16_OPEN_EIGHTHS, 6u))); }

// probing modulus / slot count, the 16-bit geometry and the open-table limit of the table `a` has been given
int32_t finish_visited_table(SearchArgs& a, uint32_t open_eighths, uint32_t kcap) {
    if (a.ht16) {
        const Ht16Geom g = ht16_geometry(a.ht_entries, a.ix.nslots, kcap);
        if (!g.ok || !ht16_eligible(a)) {
            set_error("internal: no 16-bit visited table of %u words for %u slots", a.ht_entries, a.ix.nslots);
            return DANN_EINTERNAL;
        }
        a.ht_prime = g.slots;
        a.ht_shift = g.shift;
        a.ht_tb = g.tb;
        a.ht_kmax = g.kmax;
        a.ht_open = (uint32_t)((uint64_t)g.slots * open_eighths / 8u);
    } else {
        a.ht_prime = largest_prime_leq(a.ht_entries);
        a.ht_open = a.ht_prime - (a.ht_prime >> 2);
    }
    return DANN_OK;
}
// ids the open table takes before it is frozen
uint64_t visited_open_capacity(const SearchArgs& a, uint32_t open_eighths) {
    return a.ht16 ? (uint64_t)a.ht_entries * 2u * open_eighths / 8u : (uint64_t)largest_prime_leq(a.ht_entries) * 3u / 4u;
}

// sizes the table of an automatically sized launch: the 32-bit table at the top of its occupancy step, or -- where the
// kernel has them and they buy a higher step -- 16-bit entries, also at the top of their step
void choose_visited_table(SearchArgs& a, uint32_t cap_ids, uint32_t useful_waves, uint32_t format, uint32_t open_eighths) {
    a.ht16 = 0;
    a.ht_entries = snap_visited_entries(a, cap_ids, useful_waves);
    if (format == 32u || !ht16_eligible(a)) return;
    auto waves_of = [&](uint32_t words) -> uint32_t {
        SearchArgs t = a;
        t.ht_entries = words;
        const uint64_t granules = (search_lds_bytes(t) + kLdsGranule - 1) / kLdsGranule;
        return granules > kLdsGranules ? 0u : std::min<uint32_t>(kLdsGranules / (uint32_t)granules, useful_waves);
    };
    // 16-bit entries the table needs so that cap_ids of them are below its open limit (open_eighths / 8 of the slots)
    const uint64_t need = std::max<uint64_t>(((uint64_t)cap_ids * 8u + open_eighths - 1u) / open_eighths, 512);
    uint32_t words = (uint32_t)std::min<uint64_t>(((need + 1) / 2 + 63) / 64 * 64, 32768);  // multiples of 64 words
    while (words < 32768u && !ht16_geometry(words, a.ix.nslots).ok) words = std::min<uint32_t>(words * 2u, 32768u);
    if (!ht16_geometry(words, a.ix.nslots).ok) return;
    const uint32_t w16 = waves_of(words), w32 = waves_of(a.ht_entries);
    // Measured (profiles/r04a_visited16_sgpr_ab_*.log): where the 32-bit table already lets a dozen and more queries
    // share a CU the search is bound by instruction issue, not by latency -- u8 rows at L = 26 went from 21 to 32
    // queries per CU for -3 % (and +4 % where the SGPR count capped the gain at 24: the 16-bit probe is a few
    // instructions longer); with few queries per CU (10 M x 128 f32 at L = 56: 11 -> 16) the extra residents pay.
    if (format != 16u && (w16 <= w32 || w32 > 12u)) return;
    if (w16 == 0 && format != 16u) return;
    // a sparser table on the same step costs nothing but its wipe (cf. snap_visited_entries)
    while (words + 64u <= 32768u && (uint64_t)(words + 64u) * 2u <= (uint64_t)cap_ids * 8u && waves_of(words + 64u) == w16 &&
           ht16_geometry(words + 64u, a.ix.nslots).ok)
        words += 64u;
    a.ht16 = 1;
    a.ht_entries = words;
}

// prior for a (L, beam) never seen on this index: comparisons per query ~= 4.3 R (L + W)^0.55 on
// Vamana graphs (about half of an expanded node's neighbours were seen before), 90th pct ~= 1.3x
uint32_t prior_visited_cap(const SearchArgs& a) {
    const double l = (double)(a.range_ids ? std::max<uint32_t>(a.l_value, 64) : a.l_value) + a.beam_width;
    return (uint32_t)(1.3 * 4.3 * (double)a.ix.max_degree * pow(l, 0.55)) + a.ix.nstart;
}

uint32_t largest_prime_leq(uint32_t n) {
    for (uint32_t c = n | 1u; c >= 3; c -= 2) {
        if (c > n) continue;
        bool prime = true;
        for (uint32_t d = 3; d * d <= c; d += 2)
            if (c % d == 0) {
                prime = false;
                break;
            }
        if (prime) return c;
    }
    return 2;
}

__global__ void cmps_hist_kernel(const dann_search_stats* stats, uint32_t n, uint32_t* hist) {
    __shared__ uint32_t h[kHistBins];
    for (uint32_t i = threadIdx.x; i < kHistBins; i += blockDim.x) h[i] = 0;
    __syncthreads();
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x)
        if (!stats[t].status) atomicAdd(&h[min(stats[t].cmps / 64u, kHistBins - 1u)], 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kHistBins; i += blockDim.x)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}

// which kernel family a launch with these arguments runs (include/dann_debug.h)
int search_family(const SearchArgs& a) {
    if (a.srv.ring) return DANN_FAMILY_SERVER;
    if (a.pair) return DANN_FAMILY_PAIR;
    if (a.pqlut) return DANN_FAMILY_PQ_LUT;
    if (a.team) return DANN_FAMILY_TEAM;
    if (a.grid) return DANN_FAMILY_PERSISTENT;
    return DANN_FAMILY_ONE_WAVE;
}

int32_t launch_search(const SearchArgs& a, hipStream_t stream, int* regs_out) {
    if (a.nq == 0) return DANN_OK;
    if (a.l_value == 0 || a.beam_width == 0) {
        set_error("l_value and beam_width must be non-zero (KnnSearchError, knn_search.rs:27-33)");
        return DANN_EINVAL;
    }
    if (a.beam_width > (uint32_t)kMaxBeam) {
        set_error("beam_width %u exceeds the supported maximum of %d", a.beam_width, kMaxBeam);
        return DANN_EUNSUPPORTED;
    }
    const uint32_t qcap = std::max(a.l_value + a.ix.nstart, a.qcap_max);  // the queue registers cover AdaptiveL's resize
    if (a.filter_mode == DANN_FILTER_MULTIHOP && cmax_of(a) > (uint32_t)kWave) {
        set_error("multihop filter search supports beam_width * max_degree <= 64 (got %u x %u)", a.beam_width,
                  a.ix.max_degree);
        return DANN_EUNSUPPORTED;
    }
    const size_t lds = search_lds_bytes(a);
    if (lds > 160 * 1024 && !regs_out) {
        set_error("per-query LDS footprint %zu B exceeds 160 KiB (visited table %u entries)", lds, a.ht_entries);
        return DANN_EOVERFLOW;
    }
    if (a.pqlut && !regs_out) return launch_search_pqlut(a, lds, stream);
    if (a.pair && !regs_out) return launch_search_pair(a, lds, stream);
    switch (a.ix.dtype) {
        case DT_F32: return launch_search_f32(a, qcap, lds, stream, regs_out);
        case DT_F16: return launch_search_f16(a, qcap, lds, stream, regs_out);
        case DT_U8: return launch_search_u8(a, qcap, lds, stream, regs_out);
        case DT_I8: return launch_search_i8(a, qcap, lds, stream, regs_out);
        case DT_SQ8: return launch_search_sq8(a, qcap, lds, stream, regs_out);
        case DT_PQ: return launch_search_pq(a, qcap, lds, stream, regs_out);
    }
    set_error("bad dtype %d", a.ix.dtype);
    return DANN_EINVAL;
}

#ifdef DANN_PHASE_CYCLES
static unsigned long long* g_phase_buf = nullptr;  // debug builds: per-phase cycle sums (SearchArgs::phase_cycles)
unsigned long long* dann_phase_buffer() {
    if (!g_phase_buf) {
        if (hipMalloc((void**)&g_phase_buf, 128) != hipSuccess) return nullptr;
        (void)hipMemset(g_phase_buf, 0, 128);
    }
    return g_phase_buf;
}
extern "C" int32_t dann_debug_phase_cycles(unsigned long long* out, int reset) try {
    unsigned long long* b = dann_phase_buffer();
    if (!b) return DANN_EHIP;
    if (out) (void)hipMemcpy(out, b, 128, hipMemcpyDeviceToHost);
    if (reset) (void)hipMemset(b, 0, 128);
    return 0;
} DANN_CATCH_ALL
#endif

static uint64_t calib_key(const SearchArgs& a) {
    return ((uint64_t)a.l_value << 32) | ((uint64_t)a.beam_width << 8) | (a.rec_ids ? 1u : 0u) | (a.range_ids ? 2u : 0u) |
           (a.filter_mode << 2);
}

// everything a beam-search launch needs besides its arguments: the context's spill pool (zeroed counters), the size of
// the LDS visited table (calibrated per (L, beam, mode), never affects results) and the tuning bits.  `inflight` is the
// number of wavefronts the launch keeps resident.
static int32_t prepare_launch(dann_index* idx, SearchCtx& ctx, SearchArgs& a, uint32_t inflight) {
    hipStream_t st = ctx.stream;
    // failure flag in pinned host memory: written over the fabric only by a query that
    // overflows (rare), read by the host after the stream sync -- no memset / D2H copy
    if (!ctx.d_spill) {  // 512 spill tables of 2^14 ids (32 MiB), cleaned and released by their users
        const uint32_t slices = 512, sbits = 14;
        // tables | counter (+pad) | busy flags | cmps histogram
        const size_t words = ((size_t)slices << sbits) + 16 + slices + kHistBins;
        DANN_HIP(hipMalloc((void**)&ctx.d_spill, words * 4));
        DANN_HIP(hipMemsetAsync(ctx.d_spill, 0xFF, words * 4, st));
        ctx.spill_slices = slices;
        ctx.spill_bits = sbits;
    }
    a.spill = ctx.d_spill;
    a.spill_slices = ctx.spill_slices;
    a.spill_bits = ctx.spill_bits;
    a.spill_next = ctx.d_spill + ((size_t)ctx.spill_slices << ctx.spill_bits);
    // latency regime with at most one query per SIMD: a team of five wavefronts per query -- queue, control, visited
    // filter, two for the row gather (search_kernel_impl.h, team_control_wave).  Knn searches and the build's insert-time searches (the
    // queue wave's pop records the visited node) only (the launch falls back to one wave per query where no team
    // instantiation exists).  Decided before the table is sized: teams carry more LDS.
    // DANN_DBG_TUNE_OFF bit 4 (teams) / bit 8 (speculation) / DANN_DBG_TEAM_MAX_QUERIES: development switches
    // (dann_debug_set; read on every call).
    // (dann_set_max_concurrency: the launch will be `max_concurrency` persistent waves over the batch -- search_with_retry
    // sets a.grid after this function -- never teams, pairs or the PQ table kernel: those launch one block per query
    // (pair) and never read `grid`; the persistent waves draw their queries from a counter in the spill pool's pad)
    const bool will_grid = idx->max_concurrency && a.nq > idx->max_concurrency && plain_mode(a);
    {
        const uint32_t limit = idx->dbg_u32(DANN_DBG_TEAM_MAX_QUERIES, 4u * idx->num_cus);
        a.team = (inflight <= limit && !a.grid && !will_grid && !a.srv.ring && !a.range_ids && !a.qmap && plain_mode(a) &&
                  a.ix.max_degree <= 63u /* an adjacency row fits one 64-lane request */ && !idx->tune_off(4) &&
                  team_shape(a)) ? 1u : 0u;
        if (idx->tune_off(8)) a.tune |= kTuneNoSpeculation;
        if (idx->tune_off(64)) a.tune |= kTuneNoSelfStart;
    }
    // throughput regime of 128-byte integer rows: two queries per wavefront (search_pair_impl.h).  A pair-hop is longer
    // than a hop of one query, so the pairing pays once the chip is full: measured on 1 M u8 rows at L = 26
    // (scratch/pair_latency.py, kernel us, pair / one wave per query): 4 096 queries 292 / 260, 6 144: 301 / 346,
    // 16 384: 497 / 585, 65 536: 1 423 / 1 801.  DANN_DBG_TUNE_OFF bit 16 / DANN_DBG_PAIR_MIN_QUERIES: development
    // switches (dann_debug_set; read on every call).
    a.pair = 0;
    a.ht_ov = 0;
    {
        const uint32_t floor_q = idx->dbg_u32(DANN_DBG_PAIR_MIN_QUERIES, 20u * idx->num_cus);
        SearchArgs t = a;
        t.team = 0;
        if (a.nq >= floor_q && inflight >= floor_q && !will_grid && idx->visited_format != 32u && pair_shape(t) && !idx->tune_off(16)) {
            a.pair = 1;
            a.team = 0;
        }
    }
    // the pool's allocation counter and busy flags start every launch at zero -- except a team launch, which never
    // touches the pool (a team gives a query that outgrows its table back to the host): one device operation less on
    // the single-query path
    if (!a.team) DANN_HIP(hipMemsetAsync(a.spill_next, 0, (16 + (size_t)ctx.spill_slices) * 4, st));
    // PQ rows of at most 64 chunks, plain Knn search: the lookup table in registers (search_pq_impl.h).
    // DANN_DBG_TUNE_OFF bit 32: development switch.
    a.pqlut = (!will_grid && pq_lut_shape(a) && idx->visited_format != 32u && !idx->tune_off(32)) ? 1u : 0u;
    const bool autosize = a.ht_entries == 0;
    const uint64_t key = calib_key(a);
    // calibration state of this (L, beam, mode) -- shared by concurrent callers: read and written under stat_mu
    VisitedCalib cal;
    if (autosize) {
        {
            std::lock_guard<std::mutex> lk(idx->stat_mu);
            cal = idx->calib[key];
        }
        if (a.pqlut) cal.waves = pq_lut_waves_per_cu(a.ix.pq_chunks);  // (what pq_search_kernel's table size is compiled for)
        if (!cal.waves) {  // queries per CU the registers of this instantiation allow (512 VGPRs per SIMD lane)
            int regs = 0;
            a.ht_entries = 256;
            int32_t qrc = launch_search(a, st, &regs);
            if (qrc != DANN_OK) return qrc;
            const uint32_t per_simd = regs > 0 ? 512u / (((uint32_t)regs + 7u) & ~7u) : 4u;
            cal.waves = 4u * std::min<uint32_t>(std::max<uint32_t>(per_simd, 1u), 8u);
            {
                std::lock_guard<std::mutex> lk(idx->stat_mu);
                idx->calib[key].waves = cal.waves;
            }
            if (idx->verbose()) fprintf(stderr, "[dann] search kernel: %d VGPRs -> %u queries per CU\n", regs, cal.waves);
        }
        // a launch with fewer queries than the chip has wave slots leaves LDS idle: give each query the share of a CU
        // it will actually have (a sparse table keeps the slowest lane's probe chain short -- the latency regime)
        const uint32_t per_cu = std::max<uint32_t>(1u, (inflight + idx->num_cus - 1) / idx->num_cus);
        const uint32_t waves = idx->tune_off(2) ? cal.waves : std::min<uint32_t>(cal.waves, per_cu);
        if (a.pair) {
            // one 16-bit table per query: the largest table of the first LDS step (1 280-byte granules per wavefront = two
            // queries) whose open capacity -- 75 % of its slots -- holds the 90th percentile of the comparisons with a
            // tenth to spare: the step decides how many wavefronts share a CU, and the pair kernel lives on that
            // (profiles/r04m: 16 / 8 / 4 wavefronts per CU -> 2.09 / 2.94 / 5.27 ms)
            const uint32_t cap = cal.cap_ids ? cal.cap_ids : prior_visited_cap(a);
            const uint32_t fixed = pair_lds_layout(pair_qe(a), pair_re(a), 0, 0).half_bytes;
            uint32_t words = 0, ovw = 0;
            for (uint32_t g = 2; g <= kLdsGranules && !words; ++g) {
                const uint32_t half = g * kLdsGranule / 2u;
                if (half <= fixed) continue;
                uint32_t w = std::min<uint32_t>((half - fixed) / 4u / 4u * 4u, 16384u);
                const uint32_t o = ht16_overflow_words(ht16_geometry(w, a.ix.nslots, ht16_kcap(idx)), true);
                if (w <= o + 32u) continue;
                w -= o;
                if ((uint64_t)w * 2u * ht16_open_eighths(idx) / 8u >= (uint64_t)cap + cap / 10u && ht16_geometry(w, a.ix.nslots).ok &&
                    ht16_overflow_words(ht16_geometry(w, a.ix.nslots, ht16_kcap(idx)), true) <= o) {
                    words = w;
                    ovw = o;
                }
            }
            if (words) {
                a.ht16 = 1;
                a.ht_entries = words;
                a.ht_ov = ovw;
            } else {
                a.pair = 0;
            }
        }
        if (a.pqlut) {
            // one 16-bit table: registers cap the CU at 16 queries = 8 LDS granules each (8 / 4 queries beyond 16 / 48
            // chunks); the table takes what is left of them (a sparse table costs nothing but its wipe), or -- a larger
            // 90th percentile of comparisons -- the first step whose open capacity holds it with a tenth to spare
            const uint32_t cap = cal.cap_ids ? cal.cap_ids : prior_visited_cap(a);
            const uint32_t fixed = pq_lds_layout(pq_lut_qs(a), 0, 0).total;
            uint32_t words = 0, ovw = 0;
            for (uint32_t g = kLdsGranules / pq_lut_waves_per_cu(a.ix.pq_chunks); g <= kLdsGranules && !words; ++g) {
                if (g * kLdsGranule <= fixed) continue;
                uint32_t w = std::min<uint32_t>((g * kLdsGranule - fixed) / 4u / 64u * 64u, 32768u);
                const uint32_t o = ht16_overflow_words(ht16_geometry(w, a.ix.nslots, ht16_kcap(idx)), false);
                if (w <= o + 64u) continue;
                w -= o;
                if ((uint64_t)w * 2u * ht16_open_eighths(idx) / 8u >= (uint64_t)cap + cap / 10u && ht16_geometry(w, a.ix.nslots).ok &&
                    ht16_overflow_words(ht16_geometry(w, a.ix.nslots, ht16_kcap(idx)), false) <= o) {
                    words = w;
                    ovw = o;
                }
            }
            if (words) {
                a.ht16 = 1;
                a.ht_entries = words;
                a.ht_ov = ovw;
            } else {
                a.pqlut = 0;
            }
        }
        if (!a.pair && !a.pqlut) choose_visited_table(a, cal.cap_ids ? cal.cap_ids : prior_visited_cap(a), waves, idx->visited_format, ht16_open_eighths(idx));
        if (idx->verbose() && (cal.calls & (cal.calls - 1)) == 0)
            fprintf(stderr, "[dann] L=%u W=%u: visited cap %u (%s) -> %u %s, %zu B LDS\n", a.l_value, a.beam_width,
                    cal.cap_ids ? cal.cap_ids : prior_visited_cap(a), cal.cap_ids ? "p90" : "prior",
                    a.ht16 ? a.ht_entries * 2u : a.ht_entries,
                    a.pair ? "16-bit slots per query, two queries per wavefront" : a.pqlut ? "16-bit slots, PQ table in registers" : a.ht16 ? "16-bit slots" : "entries",
                    search_lds_bytes(a));
    } else {
        // explicit size (dann_set_visited_bits): 16-bit entries only on request (dann_set_visited_format)
        a.ht16 = 0;
        if (a.pair && idx->visited_format != 16u) a.pair = 0;  // (the pair kernel has 16-bit tables only)
        if ((idx->visited_format == 16u || a.pqlut) && ht16_eligible(a)) {
            uint32_t words = std::max<uint32_t>((a.ht_entries + 63u) / 64u * 64u, 64u);
            while (words < 32768u && !ht16_geometry(words, a.ix.nslots).ok) words = std::min<uint32_t>(words * 2u, 32768u);
            if (ht16_geometry(words, a.ix.nslots).ok) {
                a.ht16 = 1;
                a.ht_entries = words;
                if (a.pair || a.pqlut) a.ht_ov = ht16_overflow_words(ht16_geometry(words, a.ix.nslots, ht16_kcap(idx)), a.pair != 0);
            }
        }
        if (a.pair && !a.ht16) a.pair = 0;
        if (a.pqlut && !a.ht16) a.pqlut = 0;
        // (the two special kernels carry their own LDS layout: a table they cannot hold goes to beam_search_kernel)
        if ((a.pair || a.pqlut) && search_lds_bytes(a) > 160 * 1024) a.pair = a.pqlut = 0;
        if (!a.pair && !a.pqlut) a.ht_ov = 0;
    }
    // the start points are inserted unconditionally and the first hop needs room before the freeze test can
    // trigger: the open table must hold nstart + W * R ids below its 75 % load limit, or ht_visit could probe a
    // full table forever (explicit dann_set_visited_bits sizes and small calibrated sizes are grown, never results)
    {
        const uint64_t floor_ids = (uint64_t)a.ix.nstart + (uint64_t)a.beam_width * a.ix.max_degree + 1;
        while (a.ht_entries < 32768 && visited_open_capacity(a, ht16_open_eighths(idx)) <= floor_ids) a.ht_entries *= 2;
        if (visited_open_capacity(a, ht16_open_eighths(idx)) <= floor_ids) {
            set_error("visited table: %u start points + beam %u x degree %u do not fit the largest LDS table", a.ix.nstart,
                      a.beam_width, a.ix.max_degree);
            return DANN_EINVAL;
        }
    }
    // latency mode: the launch is bound by per-hop latency, not bandwidth -- rows (and, in teams, adjacency rows) of the
    // predicted next hop are requested a hop ahead.  Measured on 1 M x 128 f32, L = 26 (scratch/prefetch_ab.py, kernel
    // time with / without): 64 queries 131 / 144 us, 256: 146 / 161, 512: 164 / 175, 1024: 200 / 189, 2048: 304 / 246 --
    // from about three queries per CU on, the requests of mispredicted hops cost more than the early ones gain.
    if (inflight <= 3u * idx->num_cus && !idx->tune_off(1)) a.tune |= kTuneRowPrefetch;
    // development switch DANN_DBG_TUNE_ON bit 1: the row prefetch in the throughput regime too (A/B on large indexes)
    if (idx->tune_on(1)) a.tune |= kTuneRowPrefetch;
    return DANN_OK;
}

// the persistent server of dann_server_start: sized like a launch that keeps `workers` searches in flight, enqueued on
// the context's stream, not waited for
int32_t launch_search_server(dann_index* idx, SearchCtx& ctx, SearchArgs a) {
    if (!plain_mode(a)) {
        set_error("the search server runs the plain Knn search (beam width 1, no inline tags, degree <= 64)");
        return DANN_EUNSUPPORTED;
    }
    a.nq = a.srv.workers;
    int32_t rc = prepare_launch(idx, ctx, a, a.srv.workers);
    if (rc != DANN_OK) return rc;
    a.fail_flag = nullptr;
    if (int32_t frc = finish_visited_table(a, ht16_open_eighths(idx), ht16_kcap(idx))) return frc;
    rc = launch_search(a, ctx.stream);
    if (rc == DANN_OK) {
        std::lock_guard<std::mutex> lk(idx->stat_mu);
        idx->families[DANN_FAMILY_SERVER].launches += 1;
    }
    return rc;
}

int32_t search_with_retry(dann_index* idx, SearchCtx& ctx, SearchArgs a) {
    hipStream_t st = ctx.stream;
    if (a.nq == 0) return DANN_OK;
    if (ctx.fail_cap < a.nq || !ctx.d_fail) {
        if (ctx.d_fail) (void)hipFree(ctx.d_fail);
        ctx.d_fail = nullptr;
        ctx.fail_cap = 0;
        DANN_HIP(hipMalloc((void**)&ctx.d_fail, (2 * (size_t)a.nq + 4) * 4));
        ctx.fail_cap = a.nq;
    }
    uint32_t* count = ctx.d_fail;
    uint32_t* lists[2] = {ctx.d_fail + 4, ctx.d_fail + 4 + ctx.fail_cap};
    // dann_set_max_concurrency: `grid` persistent waves share the queries (counter in the zeroed pad words above)
    auto cap_grid = [&](SearchArgs& x) {
        const bool capped = idx->max_concurrency && x.nq > idx->max_concurrency && plain_mode(x);
        x.grid = capped ? idx->max_concurrency : 0u;
        x.work_next = capped ? x.spill_next + 8 : nullptr;
    };
    const bool capped0 = idx->max_concurrency && a.nq > idx->max_concurrency && plain_mode(a);
    const uint32_t inflight = capped0 ? idx->max_concurrency : a.nq;
    const bool autosize = a.ht_entries == 0;
    const uint64_t key = calib_key(a);
    if (int32_t prc = prepare_launch(idx, ctx, a, inflight)) return prc;
    cap_grid(a);  // (the counter lives behind the spill pool prepare_launch has just attached)
    if (a.grid) a.team = 0;  // persistent waves over a block: one wave per query
    volatile uint32_t* hflag = ctx.h_flag;
    *hflag = 0;
    a.fail_flag = ctx.h_flag;
    // HIP events bracket exactly the beam-search launches, on the stream they run on
    float last_ms = 0.f;
    auto timed_launch = [&](SearchArgs& args) -> int32_t {
        if (int32_t frc = finish_visited_table(args, ht16_open_eighths(idx), ht16_kcap(idx))) return frc;
#ifdef DANN_PHASE_CYCLES
        args.phase_cycles = dann_phase_buffer();
#endif
        float ms = 0.f;
        constexpr uint32_t kUntimedLaunchQueries = 2048;  // (a launch of that many queries lasts 150-300 us)
        // a Knn call of a few queries is a latency measurement of its caller's: the two event records and the elapsed-time
        // query around it are 4-5 us of ~90 (16 callers sharing launches: 103 k -> 107 k calls/s; a 1 024-query batch: 3 % of its 170 us) -- it is waited for with a
        // plain stream synchronisation and counted with 0 ms unless DANN_DBG_TIME_SMALL_LAUNCHES asks for the events
        if (args.nq <= kUntimedLaunchQueries && !args.rec_ids && !args.qslots && !args.range_ids &&
            idx->dbg_u32(DANN_DBG_TIME_SMALL_LAUNCHES, 0u) == 0u) {
            int32_t r = launch_search(args, st);
            if (r != DANN_OK) return r;
            DANN_HIP(hipStreamSynchronize(st));
        } else {
        DANN_HIP(hipEventRecord(ctx.ev0, st));
        int32_t r = launch_search(args, st);
        if (r != DANN_OK) return r;
        DANN_HIP(hipEventRecord(ctx.ev1, st));
        DANN_HIP(hipEventSynchronize(ctx.ev1));
        DANN_HIP(hipEventElapsedTime(&ms, ctx.ev0, ctx.ev1));
        }
        last_ms = ms;
        std::lock_guard<std::mutex> lk(idx->stat_mu);
        idx->clocks[0].total_ms += ms;
        KernelClock& fam = idx->families[search_family(args)];
        fam.total_ms += ms;
        fam.launches += 1;
        return DANN_OK;
    };
    int32_t rc = timed_launch(a);
    if (rc != DANN_OK) return rc;
    {
        std::lock_guard<std::mutex> lk(idx->stat_mu);
        idx->clocks[0].launches += 1;  // one logical search = one "launch" (+ rare retry launches, time included)
    }
    // recalibrate on calls 1, 2, 4, 8, ... of this key (a 512-bin histogram of cmps, 2 KiB D2H)
    if (autosize && a.stats && !a.qmap && !a.range_ids && a.nq >= 256) {
        uint64_t calls;
        {
            std::lock_guard<std::mutex> lk(idx->stat_mu);
            calls = ++idx->calib[key].calls;
        }
        // insert-time searches run on a growing graph (comparisons grow with it): recalibrate on every batch
        if ((calls & (calls - 1)) == 0 || a.rec_ids) {
            uint32_t* d_hist = a.spill_next + 16 + ctx.spill_slices;
            DANN_HIP(hipMemsetAsync(d_hist, 0, kHistBins * 4, st));
            hipLaunchKernelGGL(cmps_hist_kernel, dim3(std::min<uint32_t>((a.nq + 255) / 256, 256)), dim3(256), 0, st,
                               a.stats, a.nq, d_hist);
            uint32_t hist[kHistBins];
            DANN_HIP(hipMemcpyAsync(hist, d_hist, sizeof(hist), hipMemcpyDeviceToHost, st));
            DANN_HIP(hipStreamSynchronize(st));
            uint64_t total = 0, acc = 0;
            for (uint32_t b = 0; b < kHistBins; ++b) total += hist[b];
            for (uint32_t b = 0; b < kHistBins && total; ++b) {
                acc += hist[b];
                if (acc * 10 >= total * 9) {
                    std::lock_guard<std::mutex> lk(idx->stat_mu);
                    idx->calib[key].cap_ids = (b + 1) * 64;
                    break;
                }
            }
        }
    }
    if (!*hflag || !a.stats) return DANN_OK;
    // rare path: queries that exhausted LDS table + spill pool are re-run with a larger LDS table
    uint32_t n = a.nq;
    const uint32_t* qmap = a.qmap;
    for (int round = 0;; ++round) {
        DANN_HIP(hipMemsetAsync(count, 0, 4, st));
        hipLaunchKernelGGL(collect_failed_kernel, dim3((n + 255) / 256), dim3(256), 0, st, a.stats, qmap, n, count,
                           lists[round & 1]);
        uint32_t h = 0;
        DANN_HIP(hipMemcpyAsync(&h, count, 4, hipMemcpyDeviceToHost, st));
        DANN_HIP(hipStreamSynchronize(st));
        if (h == 0) return DANN_OK;
        if (a.team) {
            a.team = 0;  // a team never spills its visited table: the same table, one wave per query (which does)
        } else if (a.pair || a.pqlut) {
            a.pair = a.pqlut = 0;  // re-runs go through beam_search_kernel: the table of one query doubled
            a.ht_ov = 0;
            a.ht_entries = std::min<uint32_t>(a.ht_entries * 2, 32768);
        } else {
            if (a.ht_entries >= 32768) return DANN_OK;  // callers see the per-query status
            a.ht_entries = std::min<uint32_t>(a.ht_entries * 2, 32768);
        }
        a.qmap = qmap = lists[round & 1];
        a.nq = n = h;
        cap_grid(a);
        if (search_lds_bytes(a) > 160 * 1024) return DANN_OK;
        DANN_HIP(hipMemsetAsync(a.spill_next, 0, (16 + (size_t)ctx.spill_slices) * 4, st));
        *hflag = 0;
        rc = timed_launch(a);
        if (rc != DANN_OK) return rc;
        std::lock_guard<std::mutex> lk(idx->stat_mu);
        idx->clocks[4].total_ms += last_ms;
        idx->clocks[4].launches += h;
    }
}

}  // namespace dann
