// dann_internal.h -- host-side structures shared by the translation units of libdann_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <exception>
#include <mutex>
#include <new>
#include <shared_mutex>
#include <string>
#include <thread>
#include <functional>
#include <unordered_map>
#include <vector>

#include "../../include/dann.h"
#include "../../include/dann_debug.h"
#include "small_calls.h"

struct dann_index;

namespace dann {

void set_error(const char* fmt, ...);
int32_t hip_fail(hipError_t e, const char* what);

#define DANN_HIP(call)                                        \
    do {                                                      \
        hipError_t _e = (call);                               \
        if (_e != hipSuccess) return ::dann::hip_fail(_e, #call); \
    } while (0)

// every extern "C" entry point is a function-try-block closed by this: nothing unwinds across the C boundary
#define DANN_CATCH_ALL                                          \
    catch (const std::bad_alloc&) {                             \
        ::dann::set_error("out of host memory");                \
        return DANN_ENOMEM;                                     \
    }                                                           \
    catch (const std::exception& e) {                           \
        ::dann::set_error("internal error: %s", e.what());      \
        return DANN_EINVAL;                                     \
    }                                                           \
    catch (...) {                                               \
        ::dann::set_error("internal error");                    \
        return DANN_EINVAL;                                     \
    }

struct KernelClock {
    double total_ms = 0.0;
    uint64_t launches = 0;
};

// Device view of the index, passed by value to kernels.
struct IndexView {
    const uint8_t* rows;   // (capacity + nstart) rows, row_stride bytes apart
    uint32_t* adj;         // (capacity + nstart) x (max_degree + 1) u32: [len, ids...]
    uint64_t row_stride;
    uint32_t adj_stride;   // max_degree + 1
    uint32_t dim;
    uint32_t capacity;
    uint32_t nslots;       // capacity + nstart
    uint32_t max_degree;
    uint32_t nstart;
    int32_t dtype;
    int32_t metric;
    uint32_t layer_bytes;  // bytes of one row's payload (dim * sizeof(T); SQ-8: dim + 4)
    float sq_k;            // SQ-8: (1/255)^2 * scale^2
    float sq_shift_norm_sq;
    // PQ rows (DT_PQ): codes of pq_chunks bytes; pivots 256 x dim f32; chunk offsets pq_chunks + 1
    const float* pq_pivots;
    const uint32_t* pq_offsets;
    uint32_t pq_chunks;
    // PQ rows, packed search layout (dann_pq_pack_neighbors; null = none): node i's row at pq_pack + i * pq_pack_stride =
    // [u32 len][u32 x max_degree neighbour ids][pad to 16][16-byte code row of each neighbour], 64-byte aligned
    const uint8_t* pq_pack;
    uint32_t pq_pack_stride;
    uint32_t pq_pack_codes;  // byte offset of the code rows within a packed row
    // inline concurrency tags (dann_config::inline_tags): byte offset of a row's tag (== layer_bytes), 0 = none.
    // A slot is readable iff tag >= 254 (Tag::can_read, diskann-inmem/src/tag.rs:86-133).
    uint32_t tag_off;
};

// Persistent search server (dann_server_start): device view of the submission ring.  The ring lives in host-mapped,
// fine-grained memory: a caller copies its query into slot (ticket % ring) and publishes it by storing the slot's lap
// number; wave 0 of the kernel (the dispatcher) polls the publication words over PCIe, 64 at a time, and advances
// `avail` in device memory; the worker waves draw tickets from `head`, wait for avail > ticket, stage the query into
// device memory, run the ordinary beam search and write the result plus a completion word back to the host ring.
// lap tag of a submission-ring entry: never 0 (the ring starts zeroed), consecutive laps differ
__host__ __device__ inline uint32_t server_lap_tag(unsigned long long ticket, uint32_t ring_shift) {
    return (uint32_t)((ticket >> ring_shift) % 4095ull) + 1u;
}
struct ServerView {
    const uint8_t* h_queries = nullptr;  // host ring: ring x qstride bytes
    const uint32_t* h_pub = nullptr;     // host: submission ring, entry of ticket t at t % ring: lap tag << 20 | result slot
    uint32_t* h_ack = nullptr;           // host: lap tag of the last entry a worker has taken from each ring position
    uint32_t* h_res_ids = nullptr;       // host: ring x k
    float* h_res_d = nullptr;            // host: ring x k
    dann_search_stats* h_res_stats = nullptr;  // host: ring
    uint32_t* h_done = nullptr;          // host: per result slot, low 32 bits of (ticket + 1) once the result is written
    uint32_t* h_ctl = nullptr;           // host: [0] stop request (host -> GPU), [1] the dispatcher has decided to exit
    unsigned long long* d_head = nullptr;   // device: next ticket a worker draws
    unsigned long long* d_avail = nullptr;  // device: tickets below this are published
    uint32_t* d_stop = nullptr;          // device: workers leave when they see it
    uint8_t* d_q = nullptr;              // device: workers x qstride (staged queries)
    uint32_t ring = 0;                   // entries, a power of two
    uint32_t ring_shift = 0;             // log2(ring)
    uint32_t qstride = 0;                // bytes per query slot (multiple of 16)
    uint32_t qbytes = 0;                 // bytes of one query
    uint32_t workers = 0;
    uint32_t ticks_per_us = 100;         // wall_clock64 rate (hipDeviceAttributeWallClockRate)
    uint32_t idle_timeout_us = 100000;   // the kernel leaves after this long without a new submission (a later submit
                                         // relaunches it): a device-wide synchronisation elsewhere in the process must
                                         // not wait for ever on an idle server
    uint32_t max_resident_us = 200000;   // ... nor on a busy one: the kernel also leaves (drains and is relaunched by the
                                         // next submit / wait / poll) once it has been resident this long.  hipFree is a
                                         // device-wide synchronisation: without the bound, destroying another index while
                                         // callers keep this server busy would block until they pause
};

struct SearchArgs {
    IndexView ix;
    const void* queries = nullptr;     // nq rows of layer bytes, or nullptr when `qslots` is used
    const uint32_t* qslots = nullptr;  // insert-time search: query i = stored row qslots[i]
    uint32_t nq = 0;
    uint32_t l_value = 0;
    uint32_t beam_width = 0;
    uint32_t k = 0;
    uint32_t ht_entries = 0;     // per-query LDS visited-table entries (multiple of 64)
    uint32_t ht_prime = 0;       // probing modulus, set by search_with_retry: largest prime <= ht_entries
    // 16-bit table entries (plain-mode kernels, chosen per launch by the host; search_kernel_impl.h, ht16_insert_open):
    // the table holds 2 * ht_entries slots, ht_prime is that slot count
    uint32_t ht16 = 0;
    uint32_t ht_shift = 0;       // 32 - m, m = bits of the index's slot count
    uint32_t ht_tb = 0;          // tag bits: 2^tb >= ceil(2^m / slots)
    uint32_t ht_kmax = 0;        // probes per id
    uint32_t ht_ov = 0;          // pair / PQ-table kernels: words of the overflow table behind the 16-bit table (a power
                                 // of two; 0 = none): it takes the ids whose ht_kmax probes are all taken (ov_insert)
    uint32_t ht_open = 0;        // ids the open table takes before it is frozen (set with ht_prime: 75 % of the 32-bit
                                 // table's prime, 75 % -- DANN_DBG_HT16_OPEN_EIGHTHS -- of the 16-bit table's entries)
    uint32_t* out_ids = nullptr; // nq x k (may be null in record mode)
    float* out_dists = nullptr;
    dann_search_stats* stats = nullptr;
    uint32_t* rec_ids = nullptr; // nq x rec_stride (record mode) or null
    float* rec_dists = nullptr;
    uint32_t rec_stride = 0;
    uint32_t* rec_n = nullptr;
    uint32_t* rec_max = nullptr; // optional: atomicMax of the record lengths of this launch
    // graph::search::Range (null range_ids = plain Knn): scratch list of in-range (id, dist) per query
    uint32_t* range_ids = nullptr;
    float* range_d = nullptr;
    uint32_t* range_second = nullptr;  // per query: did the second round run
    uint32_t range_cap = 0;      // entries per query in range_ids/range_d
    uint32_t range_max = 0;      // max_returned (0xFFFFFFFF = unlimited)
    uint32_t range_thresh = 0;   // (starting_l as f32 * initial_slack) as usize
    uint32_t has_inner = 0;
    float radius = 0.f, inner_radius = 0.f, range_slack = 1.f;
    uint32_t* spill = nullptr;       // pool of global-memory visited tables (all kEmpty between launches)
    uint32_t* spill_next = nullptr;  // pool allocation counter (zeroed before each launch)
    uint32_t spill_slices = 0;
    uint32_t spill_bits = 0;         // log2 entries per slice
    uint32_t* fail_flag = nullptr;   // set non-zero by any query that exhausts its scratch
    const uint32_t* qmap = nullptr;  // optional: process queries qmap[0..nq) (retry of overflowed queries)
    // filtered searches (graph/ext/labeled.rs): QueryLabelProvider == bitmap over slot ids
    uint32_t filter_mode = 0;        // 0 none, DANN_FILTER_INLINE, DANN_FILTER_MULTIHOP
    const uint32_t* filter = nullptr;
    uint64_t filter_stride = 0;      // words between the bitmaps of consecutive queries (0 = shared)
    uint32_t* m_ids = nullptr;       // inline: matched_results per query in push order (nq x m_cap)
    float* m_d = nullptr;
    uint32_t m_cap = 0;
    unsigned long long* m_keys = nullptr;  // sort scratch, nq x key_cap (key_cap a power of two)
    uint32_t key_cap = 0;
    // DANN_TIE_RUST: lists with equal distances are ordered as Rust's sort_unstable_by leaves them (rust_order.h); per
    // query kTieWorkBytes of scratch: 64 keys of a multihop hop + the sorter's work area
    uint8_t* tie_work = nullptr;
    uint32_t ad_samples = 0;         // AdaptiveL::sample_count (0 = none)
    const uint32_t* ad_table = nullptr;  // new L for (visited - ad_samples, matched): row stride ad_stride
    uint32_t ad_stride = 0;
    unsigned long long* phase_cycles = nullptr;  // -DDANN_PHASE_CYCLES builds only
    uint32_t qcap_max = 0;           // largest queue capacity an adaptive resize can ask for (0 = l_value + nstart)
    uint32_t tune = 0;               // kTune* bits, chosen per launch by search_with_retry (never affect results)
    uint32_t grid = 0;               // 0: one wave per query; else `grid` persistent waves share the nq queries through
    uint32_t* work_next = nullptr;   //    this counter (zeroed before the launch): dann_set_max_concurrency
    ServerView srv;                  // srv.ring != 0: the launch is the persistent server (grid = workers + 1 waves)
    uint32_t team = 0;               // 1: several wavefronts per query (latency regime; plain fixed-length searches only)
    uint32_t pqlut = 0;              // 1: PQ rows through pq_search_kernel (search_pq_impl.h: lookup table in registers, 16-bit
                                     //    visited table; plain Knn, <= 64 chunks, L + start points <= 256)
    uint32_t pair = 0;               // 1: two queries per wavefront (search_pair_impl.h; 128-byte integer rows, L + start
                                     //    points <= 96, degree <= 64); ht_entries = table words of ONE query then
};

// Everything one in-flight search call needs besides the (read-only) index: its own stream and events, the retry
// scratch, a pool of global-memory visited tables, the failure flag and staging buffers.  The index owns one
// (`main`, also the stream of every mutation) plus a pool of further ones handed to concurrent searches -- the
// reference's model is N workers calling `search` on one shared `&DiskANNIndex`
// (diskann-benchmark-core/src/search/api.rs:409-425); here N callers' launches run side by side on N streams.
struct SearchCtx {
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    uint32_t* d_fail = nullptr;  // retry scratch: [count, pad, list A (cap), list B (cap)]
    size_t fail_cap = 0;
    uint32_t* d_spill = nullptr;  // spill tables | counter (+pad) | busy flags | cmps histogram
    uint32_t spill_slices = 0, spill_bits = 0;
    uint32_t* h_flag = nullptr;  // pinned, device-visible: set by a query that exhausts its scratch
    // grow-only device staging for the host-pointer search entry (no hipMalloc / hipFree per call)
    // [0] queries, [1] outputs, [2] stats, [3] second query buffer of the chunked pipeline, [4] the scratch arena of the
    // range / filtered searches (one block, carved per call)
    void* stage[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t stage_bytes[5] = {0, 0, 0, 0, 0};
    void* h_stage = nullptr;     // pinned host staging (small batches: one H2D + one D2H per call; large: chunk ring)
    size_t h_stage_bytes = 0;
    hipStream_t copy_stream = nullptr;  // second stream of the chunked host-pointer pipeline
    hipEvent_t chunk_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int32_t init();   // stream, events, failure flag (device already current)
    void destroy();
};

int32_t launch_search(const SearchArgs& a, hipStream_t stream, int* regs_out = nullptr);
// one translation unit per row type (search_<type>.hip) holds the kernel instantiations
#define DANN_DECL_LAUNCH(name) \
    int32_t launch_search_##name(const SearchArgs& a, uint32_t qcap, size_t lds, hipStream_t stream, int* regs_out)
DANN_DECL_LAUNCH(f32);
DANN_DECL_LAUNCH(f16);
DANN_DECL_LAUNCH(u8);
DANN_DECL_LAUNCH(i8);
DANN_DECL_LAUNCH(sq8);
DANN_DECL_LAUNCH(pq);
#undef DANN_DECL_LAUNCH
int32_t launch_search_pqlut(const SearchArgs& a, size_t lds, hipStream_t stream);  // search_pqlut.hip (SearchArgs::pqlut)
int32_t launch_search_pqlut_g1(const SearchArgs& a, size_t lds, hipStream_t stream);
int32_t launch_search_pqlut_g2(const SearchArgs& a, size_t lds, hipStream_t stream);  // search_pqlut2.hip .. 4: 17 .. 64 chunks
int32_t launch_search_pqlut_g3(const SearchArgs& a, size_t lds, hipStream_t stream);
int32_t launch_search_pqlut_g4(const SearchArgs& a, size_t lds, hipStream_t stream);
int32_t launch_search_pair(const SearchArgs& a, size_t lds, hipStream_t stream);   // search_pair.hip (SearchArgs::pair)
// launch + re-run queries whose visited table overflowed with a table twice as large (up to 2^15)
int32_t search_with_retry(dann_index* idx, SearchCtx& ctx, SearchArgs a);
// load limit of the 16-bit visited tables of this index, in eighths of their entries (DANN_DBG_HT16_OPEN_EIGHTHS)
uint32_t ht16_open_eighths(const dann_index* idx);
// enqueue the persistent server kernel (a.srv filled in) on ctx.stream; returns without waiting
int32_t launch_search_server(dann_index* idx, SearchCtx& ctx, SearchArgs a);  // also feeds clocks[0] with the main launch's HIP-event time
size_t search_lds_bytes(const SearchArgs& a);
// explicit table size set with dann_set_visited_bits, or 0 = let search_with_retry size it
uint32_t auto_visited_entries(const dann_index* idx, uint32_t l_value, uint32_t beam);
// per (L, beam, mode) sizing state of the LDS visited table: cap_ids = 90th percentile of the
// comparisons per query seen in earlier launches (0 = none yet, use the prior)
struct VisitedCalib {
    uint32_t cap_ids = 0;
    uint64_t calls = 0;
    uint32_t waves = 0;  // occupancy the kernel's VGPRs allow (queries per CU); the table never costs more than that
};

// HIP-event time of the MFMA Gram-tile launches of the build path (build_kernels.hip; dann_kernel_time which = 5)
int32_t build_tile_clock(const dann_index* idx, double* total_ms, uint64_t* launches);
void build_tile_clock_reset(const dann_index* idx);

int32_t launch_expand_beam(const IndexView& ix, const void* d_queries, uint32_t nq, const uint32_t* d_ids,
                           const uint64_t* d_offsets, uint64_t max_len, float* d_out, hipStream_t stream);
int32_t launch_rerank(const IndexView& ix, const void* d_queries, uint32_t nq, const uint32_t* d_cand, uint32_t stride,
                      uint32_t k, uint32_t* d_out_ids, float* d_out_d, hipStream_t stream);
int32_t launch_distance_pairs(const IndexView& ix, const uint32_t* d_a, const uint32_t* d_b, uint32_t n, float* d_out,
                              hipStream_t stream);
// dann_pq_pack_neighbors: writes the packed rows (adjacency + neighbours' code rows) of every slot of the index
int32_t launch_pq_pack(const IndexView& ix, uint8_t* d_pack, uint32_t stride, uint32_t codes_off, hipStream_t stream);
// raw rows x[i] vs y[i] (pair kernel numerics), n pairs of `bytes` each
int32_t launch_distance_raw(const IndexView& ix, const void* d_x, const void* d_y, uint64_t stride, uint32_t n,
                            float* d_out, hipStream_t stream);

}  // namespace dann

struct dann_server;  // persistent search server (server.hip)

namespace dann {
// A counter many threads bump at millions of operations per second (the per-query path of the search server): one cell
// per cache line, a thread uses "its" cell; only the rare readers (dann_server_stop, a mutation's busy test) sum them.
// With sixteen callers on ONE atomic the server's throughput halved (4.5 -> 2.2 M queries/s at 64 tickets in flight per
// thread).  All operations are sequentially consistent: the publish-then-look handshakes built on it rely on that.
struct ShardedCounter {
    static constexpr uint32_t kCells = 32;
    struct alignas(64) Cell {
        std::atomic<int64_t> v{0};
    };
    Cell cell[kCells];
    static uint32_t home() {
        static thread_local const uint32_t h = (uint32_t)(std::hash<std::thread::id>()(std::this_thread::get_id()) *
                                                          0x9E3779B97F4A7C15ull >> 59);
        return h & (kCells - 1u);
    }
    void add(int64_t d) { cell[home()].v.fetch_add(d, std::memory_order_seq_cst); }
    int64_t sum() const {
        int64_t s = 0;
        for (const Cell& c : cell) s += c.v.load(std::memory_order_seq_cst);
        return s;
    }
    void reset() {
        for (Cell& c : cell) c.v.store(0, std::memory_order_seq_cst);
    }
};
}  // namespace dann

struct dann_index {
    dann_config cfg;
    int device = 0;
    dann::SearchCtx main;  // stream of every mutation and of the non-concurrent entry points
    uint8_t* d_rows = nullptr;
    uint32_t* d_adj = nullptr;
    uint32_t layer_bytes = 0;
    uint32_t nslots = 0;
    uint32_t visited_bits = 0;
    uint32_t visited_format = 0;   // dann_set_visited_format: 0 = automatic, 32 / 16 = entry width of the LDS visited table
    uint32_t max_concurrency = 0;  // dann_set_max_concurrency: queries in flight per search call (0 = all of them)
    uint32_t prune_tie_order = DANN_TIE_RUST;  // dann_set_prune_tie_order: DANN_TIE_RUST (default) / DANN_TIE_POSITION
    // dann_search_batch on pageable host buffers: the last eight (queries, ids, distances, nq) it was called with (stat_mu)
    // -- a call seen before page-locks its buffers for its duration -- and whether doing so is cheap on this system
    struct HostCall {
        const void* q = nullptr;
        const void* i = nullptr;
        const void* d = nullptr;
        uint32_t nq = 0;
        uint32_t registered = 0;  // calls that page-locked these buffers so far
        bool operator==(const HostCall& o) const { return q == o.q && i == o.i && d == o.d && nq == o.nq; }
    };
    HostCall host_calls[8];
    uint32_t host_calls_next = 0;
    std::atomic<bool> host_register_pays{true};
    // dann_search_batch, small calls: several threads calling side by side are served by one launch (small_calls.h)
    dann::SmallCallQueue comb;
    uint32_t num_cus = 256;      // compute units of the device (hipDeviceProp_t::multiProcessorCount)
    uint32_t build_flags = 0;    // DANN_BUILD_* (dann_set_build_options)
    // [0] back-edge prunes through the MFMA path, [1] ... on the lazy path inside it, [2] / [3] comparisons / hops of
    // the insert-time searches (host side; the device-side counters live in the build scratch)
    uint64_t build_counters[4] = {0, 0, 0, 0};  // [2] comparisons, [3] hops of the insert searches ([0], [1]: unused)
    float* d_pq_pivots = nullptr;
    uint32_t* d_pq_offsets = nullptr;
    // dann_pq_pack_neighbors: adjacency + neighbours' code rows per node; dropped (valid = false) by every mutation
    uint8_t* d_pq_pack = nullptr;
    size_t pq_pack_bytes = 0;
    uint32_t pq_pack_stride = 0, pq_pack_codes = 0;
    bool pq_pack_valid = false;
    std::unordered_map<uint64_t, dann::VisitedCalib> calib;  // guarded by stat_mu
    void* build_scratch = nullptr;            // owned by build_kernels.hip
    void (*build_scratch_free)(void*) = nullptr;
    dann::KernelClock clocks[5];  // 4 = beam-search retry launches (ms already in [0]; launches = re-run queries); stat_mu
    dann::KernelClock families[DANN_FAMILY_COUNT];  // beam-search launches per kernel family (dann_debug.h); stat_mu
    // development switches (dann_debug_set; NaN = default).  Plain doubles written by the test / bench thread before
    // the calls they are meant for: relaxed atomics keep concurrent searches well-defined.
    std::atomic<double> dbg[DANN_DBG_COUNT];
    double dbg_value(int key, double dflt) const {
        const double v = dbg[key].load(std::memory_order_relaxed);
        return v == v ? v : dflt;
    }
    uint32_t dbg_u32(int key, uint32_t dflt) const {
        const double v = dbg[key].load(std::memory_order_relaxed);
        return v == v ? (v <= 0.0 ? 0u : v >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)v) : dflt;
    }
    bool tune_off(uint32_t bit) const { return (dbg_u32(DANN_DBG_TUNE_OFF, 0u) & bit) != 0; }
    bool tune_on(uint32_t bit) const { return (dbg_u32(DANN_DBG_TUNE_ON, 0u) & bit) != 0; }
    bool verbose() const { return dbg_u32(DANN_DBG_VERBOSE, 0u) != 0; }
    std::vector<uint64_t> ext_ids;  // slot -> external id (empty = identity for dynamic slots)
    std::vector<uint8_t> h_tags;    // inline_tags: host mirror of the tag bytes (the reference's Store::tags, store.rs:150)
    // Locking.  `rw` is the index: shared by the Knn search entry points (dann_search_batch(_device), the server),
    // exclusive for everything that mutates the index or uses the `main` context.  Exclusive calls may nest
    // (dann_append_neighbors -> dann_get_neighbors): `mu` serialises them among themselves and `excl_depth` takes /
    // drops `rw` at the outermost level only.  std::shared_mutex does not promise writer priority: a mutation waits
    // for the searches in flight (the reference's writers go through EBR / tags instead; out of scope).
    mutable std::recursive_mutex mu;
    mutable std::shared_mutex rw;
    mutable uint32_t excl_depth = 0;
    mutable std::mutex stat_mu;   // calib, clocks
    // pool of further search contexts for concurrent callers
    std::mutex ctx_mu;
    std::condition_variable ctx_cv;
    std::vector<dann::SearchCtx*> ctx_free;
    uint32_t ctx_created = 0;
    // dann_server_start / dann_search_submit.  submit / wait / poll take no lock: they pin the server (srv_users) for the
    // duration of the call and dann_server_stop unpublishes the pointer, then waits for the pins to drain before it
    // frees anything.  srv_outstanding = tickets submitted and not yet collected; `mutating` = mutations in progress:
    // the two sides of the "no mutation while tickets are outstanding" rule (MutationScope, DANN_EBUSY).
    std::atomic<dann_server*> server{nullptr};
    dann::ShardedCounter srv_users;
    dann::ShardedCounter srv_outstanding;
    std::atomic<uint32_t> mutating{0};
    dann::IndexView view() const;
};

namespace dann {
// exclusive access (mutations, every entry point that runs on idx->main)
struct ExclusiveGuard {
    const dann_index* i;
    explicit ExclusiveGuard(const dann_index* idx) : i(idx) {
        i->mu.lock();
        if (i->excl_depth++ == 0) i->rw.lock();
    }
    ~ExclusiveGuard() {
        if (--i->excl_depth == 0) i->rw.unlock();
        i->mu.unlock();
    }
    ExclusiveGuard(const ExclusiveGuard&) = delete;
    ExclusiveGuard& operator=(const ExclusiveGuard&) = delete;
};
// a mutation of the index (rows, tags, adjacency, build): refused with DANN_EBUSY while server tickets are outstanding,
// and while it runs dann_search_submit refuses new tickets.  Both sides publish first and look second (sequentially
// consistent): at least one of a racing pair sees the other.
struct MutationScope {
    dann_index* i;
    bool ok;
    explicit MutationScope(const dann_index* idx) : i(const_cast<dann_index*>(idx)) {
        if (i->srv_outstanding.sum() != 0) {  // refused without ever raising `mutating`: no submit bounces on our account
            ok = false;
            return;
        }
        i->mutating.fetch_add(1, std::memory_order_seq_cst);
        ok = i->srv_outstanding.sum() == 0;
        if (!ok) i->mutating.fetch_sub(1, std::memory_order_seq_cst);
        else i->pq_pack_valid = false;  // (the caller holds the index exclusively) derived layouts die with the mutation
    }
    ~MutationScope() {
        if (ok) i->mutating.fetch_sub(1, std::memory_order_seq_cst);
    }
    MutationScope(const MutationScope&) = delete;
    MutationScope& operator=(const MutationScope&) = delete;
};
bool server_quiesce(dann_index* idx);  // server.hip: the resident kernel leaves and is waited for (relaunched by the next
                                       // submit); false: the server is poisoned and was not waited for
#define DANN_MUTATION(idx)                                                                                            \
    ::dann::MutationScope _mut(idx);                                                                                  \
    if (!_mut.ok) {                                                                                                   \
        ::dann::set_error("the index has search-server tickets outstanding: collect them (dann_search_wait) or stop " \
                          "the server before mutating the index");                                                   \
        return DANN_EBUSY;                                                                                            \
    }                                                                                                                 \
    if (!::dann::server_quiesce(const_cast<dann_index*>(static_cast<const dann_index*>(idx)))) {                     \
        ::dann::set_error("the index's search server stopped answering (a wait ran into its limit): dann_server_stop " \
                          "before mutating the index");                                                              \
        return DANN_EHIP;                                                                                             \
    }                                                                                                                 \
    (void)0
constexpr uint32_t kTieWorkBytes = 512 + 2048;
constexpr uint32_t kMaxSearchCtx = 16;
// a search context for one concurrent call: from the pool, created on demand (at most kMaxSearchCtx), else waits
struct CtxLease {
    dann_index* idx;
    SearchCtx* ctx = nullptr;
    int32_t status = DANN_OK;
    explicit CtxLease(dann_index* idx, bool try_only = false);
    ~CtxLease();
    CtxLease(const CtxLease&) = delete;
    CtxLease& operator=(const CtxLease&) = delete;
};
}  // namespace dann

struct dann_query {
    const dann_index* idx;
    void* d_query = nullptr;
};
