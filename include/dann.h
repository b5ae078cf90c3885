/*
 * dann.h -- C ABI of the MI355X-native DiskANN hot path (libdann_hip.so).
 *
 * This is the drop-in boundary: the batched distance-evaluation path inside Vamana beam
 * search and RobustPrune index build, behind the surface of the reference's
 * `diskann-inmem` provider.  Each entry point names the reference interface it
 * replaces (paths relative to the microsoft/DiskANN Rust workspace, v0.56).  The
 * reference-side binding a maintainer would add (Rust `extern "C"` block + adapter) is
 * shown in INTEGRATION.md.
 *
 * Conventions (those of the reference's only C surface, diskann-garnet/src/lib.rs:261-372,
 * 682-756): opaque handles, caller-owned buffers passed as pointer + length, `int32_t`
 * status (>= 0 ok, < 0 one of DANN_E*), no exceptions or aborts cross the boundary,
 * a thread-local message is available from dann_last_error().  Entry points are
 * thread-safe.  Threading model (the reference: N workers on one shared `&DiskANNIndex`, for every search kind):
 * dann_search_batch(_device), dann_range_search_batch, dann_filtered_search_batch, dann_filtered_range_search_batch,
 * dann_paged_begin / dann_paged_next take the index shared -- calls from different threads run side by side on the
 * device, each on its own stream and scratch (a pool of search contexts) -- and the server entry points
 * (dann_search_submit / wait / poll) take no lock at all; everything that mutates the index, and the fine-grained
 * seam and the record / rerank entry points, take it exclusively and wait for the searches in flight (the GPU index
 * is an immutable snapshot between mutations; the reference's EBR / tag machinery stays on the host).
 *
 * All "host" pointers are plain host memory; `_device` variants take HIP device
 * pointers that live on the index's device.
 */
#ifndef DANN_H
#define DANN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element type of the stored rows: layers::Full<T>, diskann-inmem/src/layers/full.rs:351-504 */
typedef enum {
    DANN_F32 = 0,
    DANN_F16 = 1,
    DANN_U8 = 2,
    DANN_I8 = 3,
    /* 8-bit scalar-quantised rows: `dim` code bytes followed by the f32 compensation
     * (CompensatedVector<8>, diskann-quantization/src/scalar/vectors.rs:150-175); distances are
     * CompensatedSquaredL2 / CompensatedIP / CompensatedCosineNormalized (:171-465).  Queries are
     * compressed the same way (symmetric, diskann-providers/.../inmem/scalar.rs:261-320). */
    DANN_SQ8 = 4,
    /* product-quantised rows: `pq_chunks` code bytes per row; queries are full-precision f32 vectors,
     * distances are lookup-table sums (diskann-providers/src/model/pq/fixed_chunk_pq_table.rs:82-192).
     * The pivot table is attached with dann_set_pq_table(); search entry points only -- the fine-grained
     * seam for PQ rows is dann_pq_build_lut / dann_pq_scan, dann_expand_beam returns DANN_EINVAL -- (re-rank the
     * candidates on a full-precision index with dann_rerank_batch). */
    DANN_PQ = 5
} dann_dtype;

/* == `#[repr(C)] enum Metric`, diskann-vector/src/distance/metric.rs:8-20 */
typedef enum {
    DANN_COSINE = 0,
    DANN_INNER_PRODUCT = 1,
    DANN_L2 = 2,
    DANN_COSINE_NORMALIZED = 3
} dann_metric;

enum {
    DANN_OK = 0,
    DANN_EINVAL = -1,    /* null pointer / zero L / zero beam width / bad enum         */
    DANN_ELENGTH = -2,   /* byte length does not match the layer (full.rs:228-241)    */
    DANN_EBOUNDS = -3,   /* slot id out of bounds (neighbors.rs:OutOfBounds)          */
    DANN_ETOOLONG = -4,  /* adjacency list longer than max_degree (neighbors.rs:TooLong) */
    DANN_EHIP = -5,      /* a HIP runtime call failed; see dann_last_error()          */
    DANN_ENOMEM = -6,
    DANN_EOVERFLOW = -7, /* per-query scratch (visited table / record) exhausted      */
    DANN_EUNSUPPORTED = -8,
    DANN_EINTERNAL = -9, /* a consistency check inside a kernel failed (a bug, never an input)  */
    DANN_EBUSY = -10     /* a mutation while search-server tickets are outstanding, or a submit during a mutation */
} /* dann_status */;

typedef struct dann_index dann_index; /* == diskann_inmem::Provider<Full<T>, _> + DiskANNIndex */
typedef struct dann_query dann_query; /* == layers::QueryDistance / ExpandBeam object   */

/* provider::Config + Full::new (diskann-inmem/src/provider.rs:160-217, 96-131) */
typedef struct {
    int32_t dtype;             /* dann_dtype  */
    int32_t metric;            /* dann_metric */
    uint32_t dim;
    uint32_t capacity;         /* dynamic slots [0, capacity)                              */
    uint32_t max_degree;       /* adjacency capacity per slot                              */
    uint32_t num_start_points; /* frozen slots [capacity, capacity+n) (store.rs:259-262)   */
    uint32_t row_stride;       /* bytes between rows; 0 = packed (dim*sizeof(T) rounded up
                                  to 16).  dann_inmem2_row_stride() gives the reference's
                                  own stride so a Store buffer can be uploaded verbatim    */
    int32_t device;            /* HIP device ordinal, -1 = current                         */
    float sq_scale;            /* DANN_SQ8 only: ScalarQuantizer::scale()                   */
    float sq_shift_norm_sq;    /* DANN_SQ8 only: ScalarQuantizer::shift_square_norm()       */
    uint32_t pq_chunks;        /* DANN_PQ only: code bytes per row (number of PQ chunks)    */
    uint32_t inline_tags;      /* 1: every row carries the reference's 1-byte concurrency tag right after its
                                  payload (Store layout, store.rs:133-158; needs row_stride > payload bytes, e.g.
                                  dann_inmem2_row_stride()).  A slot whose tag is below Tag::PUBLISHED (254,
                                  tag.rs:86-133) is not readable: searches skip it after the visited insert and
                                  do not count it (provider.rs:448-473, 681-686).  dann_upload_store() then copies
                                  the tag bytes verbatim, dann_set_element(s) publishes the slots it writes,
                                  start points are FROZEN (255).  0: no tags, every slot readable.            */
} dann_config;

/* graph::config::Builder (diskann/src/graph/config/mod.rs:261-338, defaults.rs:14-41) */
typedef struct {
    uint32_t pruned_degree;
    uint32_t max_degree;             /* "max_degree_with_slack"; <= dann_config.max_degree  */
    uint32_t l_build;
    float alpha;                     /* default 1.2                                         */
    uint32_t max_occlusion_size;     /* default 750                                         */
    uint32_t max_backedges;          /* single-insert only; default pruned_degree           */
    uint32_t intra_batch_candidates; /* 0 = None, n = Max(n), 0xFFFFFFFF = All              */
    uint32_t saturate_after_prune;   /* default 0                                           */
} dann_build_config;

/* per-query search statistics: SearchStats (diskann/src/graph/index.rs:90-102) */
typedef struct {
    uint32_t cmps;
    uint32_t hops;
    uint32_t result_count; /* SearchStats::result_count as the inmem2 provider reports it: its Translate
                              post-processor counts a push only while the buffer still has room afterwards
                              (provider.rs:933-944, search_output_buffer.rs:107-124), i.e. k-1 when the output
                              buffer of length k fills, else the number written.  Range searches (unbounded
                              output Vec) report the number written.                                  */
    uint32_t status;       /* 0 ok, else DANN_E* negated (per-query overflow reporting)      */
    uint32_t written;      /* entries actually written to out_ids / out_dists (what the reference's other
                              post-processors, e.g. the test provider's, return as the count)         */
} dann_search_stats;

/* ---- layer / lifetime ---------------------------------------------------------- */
/* Layer::bytes (layers/mod.rs:37-42): dim * sizeof(T), or DANN_EINVAL */
int32_t dann_layer_bytes(int32_t dtype, uint32_t dim);
/* Store row stride of the reference: round_up(bytes + 1 tag byte, 32) (store.rs:198-211) */
int32_t dann_inmem2_row_stride(int32_t dtype, uint32_t dim);
/* Provider::new + DiskANNIndex::new (provider.rs:96-131).  start_rows: num_start_points
 * rows of dann_layer_bytes() each. */
int32_t dann_index_create(const dann_config* cfg, const void* start_rows, uint64_t start_len,
                          dann_index** out);
int32_t dann_index_destroy(dann_index* idx);
int32_t dann_index_max_degree(const dann_index* idx);   /* Provider::max_degree */
int32_t dann_index_get_config(const dann_index* idx, dann_config* out);

/* ---- vectors: Set::set / SetElement::set_element (full.rs:110-130, provider.rs:334-373) */
int32_t dann_set_element(dann_index* idx, uint32_t slot, const void* bytes, uint64_t len);
int32_t dann_set_elements(dann_index* idx, uint32_t first_slot, uint32_t n, const void* rows, uint64_t len);
int32_t dann_get_element(const dann_index* idx, uint32_t slot, void* bytes, uint64_t len);
/* the same from DEVICE memory on the index's device: n rows, `src_stride` bytes apart (>= layer bytes), copied device to
 * device -- a data set that is produced or already resident on the GPU never crosses PCIe */
int32_t dann_set_elements_device(dann_index* idx, uint32_t first_slot, uint32_t n, const void* d_rows, uint64_t src_stride);
/* read-only views of the index's device buffers for zero-copy interop (rows: (capacity + start points) x row_stride bytes;
 * adjacency: the Neighbors layout, (capacity + start points) x (max_degree + 1) u32).  Valid until the index is
 * destroyed; contents change under mutations. */
int32_t dann_index_device_pointers(const dann_index* idx, const void** d_rows, const uint32_t** d_adjacency);
/* upload a whole diskann-inmem Store buffer verbatim (rows at `stride`; with dann_config::inline_tags the tag
 * byte after each payload is uploaded too and decides readability, otherwise tag bytes are ignored) */
int32_t dann_upload_store(dann_index* idx, const void* base, uint64_t stride, uint32_t nrows);
/* inline concurrency tags of slots [first_slot, first_slot + n) (inline_tags indexes only; tag.rs:86-133:
 * 0 AVAILABLE, 1 OWNED, 2 RETIRING, 254 PUBLISHED, 255 FROZEN; readable iff >= 254).  The host keeps the
 * reference's mirror of the tags (store.rs:150) for the fine-grained dann_expand_beam seam. */
int32_t dann_set_tags(dann_index* idx, uint32_t first_slot, uint32_t n, const uint8_t* tags);
int32_t dann_get_tags(const dann_index* idx, uint32_t first_slot, uint32_t n, uint8_t* tags);

/* ---- external ids: IdMap / Translate post-processing (diskann-inmem/src/ids.rs:18-107, provider.rs:899-950).
 * Search results are slot ids; these helpers keep the slot -> external id table and translate result
 * buffers (start points and unmapped slots have no external id: to_external reports them as
 * 0xFFFFFFFFFFFFFFFF, exactly the entries Translate drops). */
int32_t dann_set_external_ids(dann_index* idx, uint32_t first_slot, uint32_t n, const uint64_t* ext_ids);
int32_t dann_to_external(const dann_index* idx, const uint32_t* slot_ids, uint64_t n, uint64_t* out_ext);

/* ---- adjacency: NeighborAccessor(Mut) (provider.rs:768-823, neighbors.rs:124-224) -- */
int32_t dann_get_neighbors(const dann_index* idx, uint32_t slot, uint32_t* out, uint32_t cap, uint32_t* out_len);
int32_t dann_set_neighbors(dann_index* idx, uint32_t slot, const uint32_t* ids, uint32_t n);
/* append with the reference's clamp-on-overflow (provider.rs:804-816) */
int32_t dann_append_neighbors(dann_index* idx, uint32_t slot, const uint32_t* ids, uint32_t n);
/* set_neighbors_bulk: lists[i] = [len, ids...] rows of (max_degree+1) u32 */
int32_t dann_set_neighbors_bulk(dann_index* idx, const uint32_t* slots, uint32_t n, const uint32_t* lists);
/* whole Neighbors buffer, (capacity+num_start_points) x (max_degree+1) u32 (neighbors.rs:60-101) */
int32_t dann_upload_graph(dann_index* idx, const uint32_t* adj, uint64_t nrows);
int32_t dann_download_graph(const dann_index* idx, uint32_t* adj, uint64_t nrows);

/* ---- distances (parity seam) ----------------------------------------------------- */
/* layers::Distance::evaluate (full.rs:224-242): two raw rows, T x T kernel */
int32_t dann_distance(const dann_index* idx, const void* x, uint64_t xlen, const void* y, uint64_t ylen,
                      float* out);
/* the same between stored rows, n pairs in one launch (RobustPrune's primitive, prune.rs:212-215) */
int32_t dann_distance_pairs(const dann_index* idx, const uint32_t* a, const uint32_t* b, uint32_t n, float* out);
/* Search::query_distance + QueryDistance::evaluate (full.rs:165-171, 317-336) */
int32_t dann_query_create(const dann_index* idx, const void* query, uint64_t len, dann_query** out);
int32_t dann_query_destroy(dann_query* q);
int32_t dann_query_distance(const dann_query* q, const void* row, uint64_t len, float* out);
/* ExpandBeam::expand_beam (provider.rs:492-497, 620-690): pre-filtered ids -> (id, dist) */
int32_t dann_expand_beam(const dann_query* q, const uint32_t* ids, uint32_t n, uint32_t* out_ids,
                         float* out_dists, uint32_t* out_n);
/* batched form: nq queries, ragged id lists (offsets has nq+1 entries); out_dists is
 * aligned with `ids` */
int32_t dann_expand_beam_batch(const dann_index* idx, const void* queries, uint32_t nq, const uint32_t* ids,
                               const uint64_t* offsets, float* out_dists);

/* ---- search: DiskANNIndex::search(Knn{l_value, beam_width}) for nq independent queries
 * (index.rs:1933-2000,2029-2065; knn_search.rs:155-193; provider.rs:408-480,899-950).
 * out_ids/out_dists: nq x k (unwritten entries 0xFFFFFFFF / +inf); ids are slot ids. ---- */
int32_t dann_search_batch(dann_index* idx, const void* queries, uint32_t nq, uint32_t l_value,
                          uint32_t beam_width, uint32_t k, uint32_t* out_ids, float* out_dists,
                          dann_search_stats* out_stats);
/* device-resident form: every pointer is a device pointer on the index's device; the
 * launch is enqueued on the index stream and the call returns after it completes. */
int32_t dann_search_batch_device(dann_index* idx, const void* d_queries, uint32_t nq, uint32_t l_value,
                                 uint32_t beam_width, uint32_t k, uint32_t* d_out_ids, float* d_out_dists,
                                 dann_search_stats* d_out_stats);
/* graph::search::Range for nq queries (diskann/src/graph/search/range_search.rs:51-470): Knn-style
 * search with L = starting_l, then -- if at least starting_l * initial_slack of that list lies within
 * `radius` and fewer than max_returned -- breadth-first expansion of every point within
 * radius * range_slack.  Results (slot ids, start points dropped, `inner_radius < d <= radius`) go to
 * out_ids/out_dists (nq x out_cap, unwritten = 0xFFFFFFFF / +inf).  max_returned == 0: unlimited
 * (the GPU scratch list is then sized 4 * out_cap + 1024 per query; a longer list is DANN_EOVERFLOW).
 * Stats follow the reference's accounting (cmps of the first phase only, hops = initial + cumulative,
 * :308-314); out_second_round[q] = range_search_second_round.  Parameter errors == RangeSearchError. */
int32_t dann_range_search_batch(dann_index* idx, const void* queries, uint32_t nq, uint32_t starting_l,
                                uint32_t beam_width, float radius, int32_t has_inner_radius, float inner_radius,
                                float initial_slack, float range_slack, uint32_t max_returned, uint32_t out_cap,
                                uint32_t* out_ids, float* out_dists, dann_search_stats* out_stats,
                                uint32_t* out_second_round);
/* ---- filtered searches (diskann/src/graph/ext/labeled.rs: a SearchAccessor wrapped with a QueryLabelProvider).
 * The label provider crosses the boundary as a bitmap over slot ids: is_match(i) = bit (i & 31) of bits[i >> 5]
 * for i in [0, capacity + num_start_points) -- start points are classified too (labeled.rs:166-175). */
enum { DANN_FILTER_INLINE = 1, DANN_FILTER_MULTIHOP = 2 };
typedef struct {
    uint32_t mode;              /* DANN_FILTER_INLINE: InlineFilterSearch (search/inline_filter_search.rs:69-301);
                                   DANN_FILTER_MULTIHOP: MultihopFilterSearch (search/multihop_filter_search.rs:46-244) */
    const uint32_t* bits;       /* host memory */
    uint64_t stride_words;      /* words between the bitmaps of consecutive queries; 0 = one bitmap for all queries */
    uint32_t adaptive_samples;  /* inline only: AdaptiveL::sample_count, 0 = no AdaptiveL (:40-62) */
    double adaptive_scale;      /* AdaptiveL::scale_factor, must be >= 1.0 */
    uint32_t matched_cap;       /* inline only: capacity of the per-query matched list (every accepted id that was
                                   compared); 0 = min(capacity + starts, 8192). Exceeding it is DANN_EOVERFLOW. */
} dann_filter;
/* index.search(InlineFilterSearch | MultihopFilterSearch, Filtered<Strategy>, ..) for nq queries; outputs as
 * dann_search_batch.  Stats follow the reference: start points are not counted in cmps.  Ties: the reference sorts
 * matched results / rejected candidates with sort_unstable_by(distance); equal distances keep arrival order here. */
int32_t dann_filtered_search_batch(dann_index* idx, const void* queries, uint32_t nq, uint32_t l_value,
                                   uint32_t beam_width, uint32_t k, const dann_filter* filter, uint32_t* out_ids,
                                   float* out_dists, dann_search_stats* out_stats);
/* index.search(FilteredRange, ..) (search/filtered_range_search.rs:111-330); parameters and outputs as
 * dann_range_search_batch, filter->mode must be DANN_FILTER_INLINE, AdaptiveL is not used (:148-156).
 * cmps / hops accumulate over both rounds. */
int32_t dann_filtered_range_search_batch(dann_index* idx, const void* queries, uint32_t nq, uint32_t starting_l,
                                         uint32_t beam_width, float radius, int32_t has_inner_radius,
                                         float inner_radius, float initial_slack, float range_slack,
                                         uint32_t max_returned, uint32_t out_cap, const dann_filter* filter,
                                         uint32_t* out_ids, float* out_dists, dann_search_stats* out_stats,
                                         uint32_t* out_second_round);

/* ---- paged search: DiskANNIndex::paged_search + PagedSearch::next_page (diskann/src/graph/index.rs:2075-2155,
 * diskann/src/graph/search/paged.rs:53-149) for nq queries at once.  The session owns the reference's per-search
 * scratch on the device: the auto-resizable candidate list (nothing is ever dropped, queue.rs:95-121), the visited
 * set and the tail of the last computed page.  list_cap bounds the list per query (0 = min(capacity + starts,
 * max(16384, 64 * l_value))); exceeding it is DANN_EOVERFLOW.  next_page: k results per query (k <= l_value, errors
 * as paged.rs:57-64), out_ids/out_dists nq x k (unwritten 0xFFFFFFFF / +inf), out_counts[q] = results of the page
 * (0 = exhausted).  Pages never overlap; within a page distances are non-decreasing.  Not for DANN_PQ. */
typedef struct dann_paged dann_paged;
int32_t dann_paged_begin(dann_index* idx, const void* queries, uint32_t nq, uint32_t l_value, uint32_t list_cap,
                         dann_paged** out);
int32_t dann_paged_next(dann_paged* session, uint32_t k, uint32_t* out_ids, float* out_dists, uint32_t* out_counts);
int32_t dann_paged_end(dann_paged* session);

/* Rerank post-processor (diskann-providers/src/model/graph/provider/async_/inmem/full_precision.rs:348-397):
 * full-precision distances query x stored row for every candidate id of a quantised search, sorted
 * ascending (equal distances keep candidate order; the reference's sort is unstable), first k returned.
 * `idx` is the full-precision index; cand_ids: nq x cand_stride, entries equal to 0xFFFFFFFF are skipped. */
int32_t dann_rerank_batch(dann_index* idx, const void* queries, uint32_t nq, const uint32_t* cand_ids,
                          uint32_t cand_stride, uint32_t k, uint32_t* out_ids, float* out_dists);
/* device-pointer form (queries, cand_ids, outputs on the index's device) */
int32_t dann_rerank_batch_device(dann_index* idx, const void* d_queries, uint32_t nq, const uint32_t* d_cand_ids,
                                 uint32_t cand_stride, uint32_t k, uint32_t* d_out_ids, float* d_out_dists);
/* insert-time search: also returns the VisitedSearchRecord (record.rs:86-93) per query:
 * rec_ids/rec_dists: nq x rec_stride, rec_n: nq */
int32_t dann_search_record_batch(dann_index* idx, const uint32_t* slots, uint32_t nq, uint32_t l_value,
                                 uint32_t* rec_ids, float* rec_dists, uint32_t rec_stride, uint32_t* rec_n,
                                 dann_search_stats* out_stats);

/* the same for external queries (host pointer, layer bytes each): the VisitedSearchRecord of an ordinary Knn search
 * with beam width 1 -- every node the search expanded, in expansion order.  Diagnostics / replay checks. */
int32_t dann_search_record_queries(dann_index* idx, const void* queries, uint32_t nq, uint32_t l_value, uint32_t* rec_ids,
                                   float* rec_dists, uint32_t rec_stride, uint32_t* rec_n, dann_search_stats* out_stats);

/* ---- build ------------------------------------------------------------------------ */
/* occlude_list / robust_prune over caller pools (index.rs:2565-2650, prune.rs:106-259):
 * pool i = pool_ids/pool_dists[offsets[i] .. offsets[i+1]) for location locs[i];
 * out_adj: n rows of (pruned_degree+1) u32 [len, ids...] */
int32_t dann_prune_batch(dann_index* idx, const dann_build_config* cfg, const uint32_t* locs, uint32_t n,
                         const uint32_t* pool_ids, const float* pool_dists, const uint64_t* offsets,
                         int32_t force_saturate, uint32_t* out_adj);
/* DiskANNIndex::multi_insert (index.rs:815-1030) for rows already stored at `slots` */
int32_t dann_insert_batch(dann_index* idx, const dann_build_config* cfg, const uint32_t* slots, uint32_t n);
/* DiskANNIndex::insert (diskann/src/graph/index.rs:226-341) for a row already stored at `slot`: insert search (beam 1,
 * L = l_build), RobustPrune of the visited record, set_neighbors, then add_edge_and_prune towards the first
 * cfg->max_backedges of the new neighbours (:324-327; 0 = pruned_degree, the reference's default; more than
 * pruned_degree is DANN_EINVAL as in config/mod.rs:308-311).  With the default it equals dann_insert_batch of one
 * point; multi_insert itself sends back-edges to every new neighbour (index.rs:123-143). */
int32_t dann_insert(dann_index* idx, const dann_build_config* cfg, uint32_t slot);
/* multi-GPU build (replicated index, batch partitioned across ranks): multi_insert split at its
 * only exchange point.  Phase 1 = candidate generation (search + RobustPrune, index.rs:349-434) for the
 * batch positions [lo, hi), reading the graph only; phase 2 = graph update (index.rs:911-1024) from the
 * pending rows of the whole batch.  d_pending_* are DEVICE pointers, rows of (pruned_degree + 1) u32
 * [len, ids...] in batch order, i.e. directly usable as RCCL all-gather buffers.  Running phase 1 over
 * [0, n) and then phase 2 equals dann_insert_batch. */
int32_t dann_insert_batch_candidates(dann_index* idx, const dann_build_config* cfg, const uint32_t* slots, uint32_t n,
                                     uint32_t lo, uint32_t hi, uint32_t* d_pending_out);
int32_t dann_insert_batch_commit(dann_index* idx, const dann_build_config* cfg, const uint32_t* slots, uint32_t n,
                                 const uint32_t* d_pending_all);
/* phase 2 with the expensive part partitioned (multi-GPU build): every replica applies the new rows and the
 * back-edges whose target list still fits; a target whose list must be pruned (robust_prune of add_edge_and_prune,
 * index.rs:2264-2341) is handled only by the rank that owns it (id % world == rank).  The rows rewritten by this rank
 * are exported to d_rows_out (DEVICE, rows of max_degree + 2 u32: [target id, len, ids...], at most rows_cap of
 * them, *count_out on return) -- all-gather them and hand the other ranks' rows to dann_apply_neighbor_rows_device;
 * after that every replica equals the result of dann_insert_batch_commit.  world = 1 is dann_insert_batch_commit. */
int32_t dann_insert_batch_commit_part(dann_index* idx, const dann_build_config* cfg, const uint32_t* slots, uint32_t n,
                                      const uint32_t* d_pending_all, uint32_t rank, uint32_t world,
                                      uint32_t* d_rows_out, uint32_t rows_cap, uint32_t* count_out);
int32_t dann_apply_neighbor_rows_device(dann_index* idx, const uint32_t* d_rows, uint32_t count);
/* insert every slot in [first, first+n) in id order with a geometric batch schedule
 * (batch = clamp(ceil(inserted * growth), 1, max_batch)); returns the number of batches */
int32_t dann_build(dann_index* idx, const dann_build_config* cfg, uint32_t first, uint32_t n, float growth,
                   uint32_t max_batch);

/* ---- product quantisation: lookup-table build + scan -------------------------------------
 * FixedChunkPQTable::populate_chunk_distances_impl (L2 / inner product) and pq_dist_lookup_single
 * (diskann-providers/src/model/pq/fixed_chunk_pq_table.rs:152-192, 82-100), batched:
 *   pivots: 256 x dim f32 row-major (:105-128), chunk_offsets: nchunks+1 (usize -> u32),
 *   queries: nq x dim f32 (already centred/rotated by the caller), lut: nq x nchunks x 256 f32.
 * metric: DANN_L2 or DANN_INNER_PRODUCT.  Host pointers. */
int32_t dann_pq_build_lut(int32_t device, int32_t metric, const float* pivots, const uint32_t* chunk_offsets,
                          uint32_t nchunks, uint32_t dim, const float* queries, uint32_t nq, float* lut);
/* scan: out[q][i] = sum over chunks, in chunk order, of lut[q][c][codes[ids[q][i]][c]];
 * codes: npoints x nchunks u8; ids: ragged lists (offsets has nq+1 entries), out aligned with ids */
int32_t dann_pq_scan(int32_t device, const float* lut, uint32_t nq, uint32_t nchunks, const uint8_t* codes,
                     uint64_t npoints, const uint32_t* ids, const uint64_t* offsets, float* out);

/* TransposedTable::compress_into for a batch (diskann-quantization/src/product/tables/transposed/table.rs:382-403,
 * 465-530; Chunk::find_closest, pivots.rs:253-345): codes[r][c] = index of the pivot of chunk c closest (L2) to
 * row r's chunk, with the reference's arithmetic (|p|^2 - 2 x.p, fma chain in dimension order) and its lane-wise
 * tie rule.  pivots: ncenters (<= 256) x dim f32; rows: n x dim f32; codes: n x nchunks u8.  Host pointers.
 * DANN_EINVAL when a chunk is infinitely far from / NaN against every pivot (InfinityOrNaN). */
int32_t dann_pq_compress(int32_t device, const float* pivots, uint32_t ncenters, const uint32_t* chunk_offsets,
                         uint32_t nchunks, uint32_t dim, const float* rows, uint64_t n, uint8_t* codes);

/* PQ training minus the seeding: the Lloyd iterations of LightPQTrainingParameters::train
 * (diskann-quantization/src/product/train.rs:96-226 -> algorithms/kmeans/lloyds.rs:23-438) for every chunk, with the
 * reference's arithmetic (assignment scores ((n_c - ip) - ip) + |x|^2, first strictly smaller centre; f64 centroid
 * sums in row order; empty clusters become the zero vector).  `centers` (ncenters x dim f32) carries the initial
 * centres in -- k-means++ (kmeans/plusplus.rs, seeded per chunk from rand's StdRng) stays with the caller -- and the
 * trained pivots out.  data: n x dim f32.  Optional outputs: assignments nchunks x n (those of the last assignment
 * step, as lloyds_inner returns them), residuals nchunks.  Host pointers. */
int32_t dann_pq_lloyds(int32_t device, const float* data, uint64_t n, uint32_t dim, const uint32_t* chunk_offsets,
                       uint32_t nchunks, uint32_t ncenters, float* centers, uint32_t max_reps, uint32_t* assignments,
                       float* residuals);

/* the caller's random generator for the two draws of k-means++: on the Rust side closures over the
 * `rng_builder.build_boxed_rng(chunk)` generators of LightPQTrainingParameters::train (train.rs:164-165,
 * diskann-quantization/src/random.rs:33-44), so that the GPU consumes exactly the reference's stream:
 *   uniform_index(ctx, chunk, n)   == Uniform::new(0, n).unwrap().sample(rng_chunk)              (plusplus.rs:417)
 *   uniform_f64(ctx, chunk, high)  == Uniform::<f64>::new(0.0, high).unwrap().sample(rng_chunk)  (plusplus.rs:440-444)
 * Called from the calling thread only.  Draws of different chunks are interleaved (centre by centre across the
 * chunks); within one chunk they come in the reference's order -- which is all a per-chunk generator can observe. */
typedef struct {
    void* ctx;
    uint64_t (*uniform_index)(void* ctx, uint32_t chunk, uint64_t n);
    double (*uniform_f64)(void* ctx, uint32_t chunk, double high);
} dann_rng;
/* kmeans::plusplus::kmeans_plusplus_into_inner (algorithms/kmeans/plusplus.rs:366-497) for every chunk.  centers:
 * ncenters x dim out (the chunk columns of centres that could not be selected stay zero); selected: nchunks out or NULL
 * (fewer than ncenters = the reference's recoverable DatasetTooSmall / InsufficientDiversity).  DANN_EINVAL when a
 * non-finite distance total appears (SawInfinity). */
int32_t dann_pq_kmeanspp(int32_t device, const float* data, uint64_t n, uint32_t dim, const uint32_t* chunk_offsets,
                         uint32_t nchunks, uint32_t ncenters, const dann_rng* rng, float* centers, uint32_t* selected);
/* LightPQTrainingParameters::train (product/train.rs:96-226): dann_pq_kmeanspp + dann_pq_lloyds; pivots: ncenters x dim */
int32_t dann_pq_train(int32_t device, const float* data, uint64_t n, uint32_t dim, const uint32_t* chunk_offsets,
                      uint32_t nchunks, uint32_t ncenters, uint32_t lloyds_reps, const dann_rng* rng, float* pivots);

/* ---- on-disk formats of the reference (so a GPU-built index loads in the reference and vice versa)
 * graph: diskann-providers/src/storage/bin.rs:234-380 -- 24-byte header {u64 file_size, u32 max_degree,
 *        u32 start_point, u64 num_start_points} then per node {u32 len, len x u32}, nodes in slot order
 *        (dynamic slots, then the frozen start points).
 * vectors: diskann-utils/src/io.rs:24 `.bin` -- {u32 npts, u32 dim} then row-major payload. */
int32_t dann_save_graph(const dann_index* idx, const char* path);
/* loads adjacency for min(file points, index slots) nodes; out_* may be NULL */
int32_t dann_load_graph(dann_index* idx, const char* path, uint32_t* out_start, uint64_t* out_num_start,
                        uint64_t* out_num_points);
int32_t dann_save_vectors_bin(const dann_index* idx, const char* path, uint32_t first_slot, uint32_t n);
int32_t dann_load_vectors_bin(dann_index* idx, const char* path, uint32_t first_slot, uint32_t* out_n);

/* attach the PQ schema of a DANN_PQ index: pivots 256 x dim f32 row-major, chunk_offsets pq_chunks + 1
 * (FixedChunkPQTable::new, fixed_chunk_pq_table.rs:105-140) */
int32_t dann_set_pq_table(dann_index* idx, const float* pivots, const uint32_t* chunk_offsets);
/* Optional GPU-private search layout of a DANN_PQ index (at most 64 chunks, no inline tags): for every slot one
 * 64-byte-aligned row holding its adjacency list AND the code rows of its neighbours, so that a hop of the beam search
 * (expand_beam over PQ rows: diskann-providers/src/model/pq/fixed_chunk_pq_table.rs:82-100 per neighbour) is one
 * contiguous read instead of one adjacency row plus one code-row gather per neighbour.  Built from the index's current
 * rows and adjacency; costs (capacity + start points) x round_up(16-rounded 4 (R + 1) + C R, 64) bytes of device
 * memory, C = the chunk count rounded up to 16.  Search results never depend on it.  Any later mutation of the index (rows, adjacency, build, load) drops
 * the layout -- searches fall back to the plain rows -- until this is called again.  DANN_EUNSUPPORTED for other
 * indexes. */
int32_t dann_pq_pack_neighbors(dann_index* idx);

/* ScalarQuantizationParameters::train (diskann-quantization/src/scalar/train.rs:33-52): shift[d] = mean_d - p,
 * scale = 2p with p = standard_deviations * sqrt(max_d variance_d) (f64 statistics in row order, utils.rs:109-199);
 * mean_norm (optional) = mean L2 norm of the rows.  data: n x dim f32, host pointers.  The reference's default is
 * standard_deviations = 2. */
int32_t dann_sq8_train(int32_t device, const float* data, uint64_t n, uint32_t dim, double standard_deviations,
                       float* shift, float* scale, float* mean_norm);

/* ---- scalar quantisation: ScalarQuantizer::compress_into for 8 bits
 * (diskann-quantization/src/scalar/quantizer.rs:189-236, 395-430): code = round(clamp((x - shift) *
 * 255/scale, 0, 255)), compensation = scale/255 * sum(code * shift).  x: n x dim f32, shift: dim f32,
 * out: n rows of dim + 4 bytes.  Host pointers. */
int32_t dann_sq8_compress(int32_t device, const float* x, uint32_t n, uint32_t dim, const float* shift, float scale,
                          void* out);

/* build-path options (never change the resulting graph).  The matrix-core path evaluates the pair similarities a
 * RobustPrune asks for (prune.rs:196-232) as the lower triangle of one Gram matrix per candidate list
 * (v_mfma_f32_32x32x2_f32; three kernels: list / sort, Gram tiles, sweep), with a bit-exact re-evaluation of every
 * comparison the rounding-error interval of the Gram value does not decide.  f32 and f16 rows, L2 / inner product /
 * cosine-normalized; other configurations ignore the flags.  Default: rows of 1 KiB and more use it. */
enum { /* back-edge prunes (add_edge_and_prune -> robust_prune_list, diskann/src/graph/index.rs:2264-2341) of any row size */
       DANN_BUILD_MFMA_BACKEDGE = 1,
       /* the pool prune of every inserted point (robust_prune_with, index.rs:2476-2532) of any row size: Gram of the
        * first <= 256 sorted candidates against the first 96; pairs outside that block use the row kernel */
       DANN_BUILD_MFMA_POOL = 2,
       /* never use the matrix-core path */
       DANN_BUILD_ROW_KERNEL_ONLY = 4 };
int32_t dann_set_build_options(dann_index* idx, uint32_t flags);
/* work counters of the build path since index creation (the algorithmic-bytes model of profiles/): out[0] back-edge
 * prunes through the MFMA path, [1] back-edge prunes of lists too long for it, [2] comparisons and [3] hops of the
 * insert-time searches, [4] pair distances d(c_i, c_j) evaluated by the row kernel in the prune sweeps, [5] list /
 * extra distances d(location, c), [6] candidate rows that went through an MFMA Gram, [7] Gram entries computed
 * (x dim x 2 = MFMA flop), [8] pair distances the lazy scans of the MFMA sweeps asked for, [9] those of [8] that needed
 * an exact re-evaluation by the row kernel (the rest were answered from a Gram), [10] tied candidate pools whose
 * DANN_TIE_RUST walk ran into the selection's last resort: after sixteen unlucky partitions core's select_nth_unstable_by
 * calls median_of_medians, which csrc/rust_order.h replaces by a sort of the range -- the index-th element is in place
 * either way, equal keys may be arranged differently, so such a pool is not guaranteed to match the reference's order
 * (never reached by the reference's goldens or by any build measured so far; 0 = the promise of
 * dann_set_prune_tie_order holds for every pool of this index's builds).  n <= 11 entries are written. */
int32_t dann_build_counters(const dann_index* idx, uint64_t* out, uint32_t n);

/* ABI revision of this header; bumped on any incompatible change of a signature or struct layout */
#define DANN_ABI_VERSION 4
int32_t dann_abi_version(void);

/* ---- diagnostics ------------------------------------------------------------------- */
/* thread-local message of the last failing call on this thread; returns its length */
int32_t dann_last_error(char* buf, uint64_t len);
/* HIP-event time (ms) and launch count of the named kernel since the last reset.
 * which: 0 = beam search, 1 = gather distance, 2 = prune, 3 = back-edge,
 * 4 = beam-search retry launches (their time is also part of 0; `launches` counts re-run queries),
 * 5 = gram_tiles_kernel, the matrix-core kernel of the build's prunes (with dann_build_counters()[7] x dim x 2 flop:
 *     the MFMA rate of the build) */
int32_t dann_kernel_time(const dann_index* idx, int32_t which, double* total_ms, uint64_t* launches);
int32_t dann_kernel_time_reset(dann_index* idx);
/* tuning knob: per-query LDS visited-table size. 0 = auto from L and degree; 6..15 = log2(entries);
 * 64..32768 = explicit entry count (rounded up to a multiple of 64). Never affects results. */
int32_t dann_set_visited_bits(dann_index* idx, uint32_t bits);
/* tuning knob: width of a visited-table entry.  0 (default) = automatic: 16-bit entries -- an exact set all the same:
 * slot and entry together determine the id -- where they let more queries share a compute unit, 32-bit entries
 * otherwise; 32 / 16 = always that width (16 applies to the plain searches and falls back to 32 where the index has
 * too many slots for the table).  Never affects results. */
int32_t dann_set_visited_format(dann_index* idx, uint32_t entry_bits);
/* queries in flight per search call.  0 (default) = every query of the call gets its own wavefront at once;
 * N > 0 = N persistent wavefronts take the call's queries one after the other from a shared counter -- what a
 * server does that keeps N searches in flight (the reference: N tokio workers calling DiskANNIndex::search on a
 * shared index, SURVEY 8(b) "Threading"): a finished search is replaced at once instead of waiting for the slowest
 * of its batch.  Applies to the plain searches (beam width 1, no filter, no inline tags, degree <= 64: Knn, Range,
 * the insert search); the other modes keep one wavefront per query.  Never affects results. */
int32_t dann_set_max_concurrency(dann_index* idx, uint32_t max_queries_in_flight);
/* Order of EQUAL-distance candidates in RobustPrune's candidate pool (every prune of dann_prune_batch,
 * dann_insert_batch*, dann_build*).  SortedNeighbors::new (diskann/src/graph/internal/sorted_neighbors.rs:26-44) sorts
 * with select_nth_unstable_by + sort_unstable_by, whose order of equal keys is unspecified in the API and a
 * deterministic function of the pool's arrival order in the implementation.
 *   DANN_TIE_RUST (default): the order the reference's own toolchain leaves (rust-toolchain.toml: 1.97.1; core's
 *     "ipnsort" and its selection, Rust >= 1.81), walked sequentially by one lane per pool (csrc/rust_order.h), and the
 *     bootstrap's candidate list in AdjacencyList::from_iter_untrusted's ascending id order (adjacencylist.rs:181-190).
 *     With it the GPU build reproduces the reference's tie-heavy grid_insert goldens (tests/test_gpu_tie_order.py):
 *     the graph is the reference's graph on integer lattices, byte data and duplicated rows as well.
 *     Only pools that hold equal distances are walked (the sorting network's output is inspected first): builds of
 *     float rows cost what they cost under DANN_TIE_POSITION, a 200 k x 128 u8 build 12 % more.
 *   DANN_TIE_POSITION: equal distances keep their pool order (a stable sort; one wavefront-wide sorting network per
 *     pool, no walk).  Identical to DANN_TIE_RUST wherever the distances of a pool are distinct; on tied pools one of the
 *     orders the reference's API permits, not the one its implementation produces.
 * The same setting orders equal distances in the filtered searches' post-processing (dann_filtered_search_batch): the
 * matched list of the inline-filter search (`sort_unstable_by`, inline_filter_search.rs:274) and the rejected candidates
 * a multihop hop expands (multihop_filter_search.rs:207-210) -- Rust's order under DANN_TIE_RUST (lists with ties are
 * sorted again by one lane, in global memory: a conformance mode, slow on integer data), push order under
 * DANN_TIE_POSITION; tests/test_gpu_filtered.py::test_equal_distances_follow_the_references_unstable_sort.
 * Replicas of one sharded build (dann_build_sharded; dann_multi_build: dann_multi_replica) must use the same order. */
enum { DANN_TIE_POSITION = 0, DANN_TIE_RUST = 1 };
int32_t dann_set_prune_tie_order(dann_index* idx, uint32_t order);

/* ---- multi-GPU (replicated index per device).  Search shards with no data-path collective; the build splits every
 * multi_insert batch at its only exchange point (diskann/src/graph/index.rs:815-1030, :911-1024): candidates per rank,
 * all-gather of the pending adjacency rows, the same graph update on every replica with the prunes of overflowing
 * back-edge targets done by their owner (id % world) and a second all-gather of the rewritten rows.  The collectives
 * are issued by the library on device buffers: RCCL (ncclAllGather over xGMI, librccl bound at run time), an in-process
 * form for one process driving several devices, or the caller's own callback. ---- */
typedef struct dann_comm dann_comm;
typedef struct { char internal[128]; } dann_rccl_unique_id; /* == ncclUniqueId */
typedef struct {
    void* ctx;
    uint32_t rank, world;
    /* all-gather `bytes` bytes per rank, device buffers (d_recv holds world * bytes), enqueued on `hip_stream` (a
     * hipStream_t) or complete on return; 0 = ok */
    int32_t (*all_gather)(void* ctx, const void* d_send, void* d_recv, uint64_t bytes, void* hip_stream);
} dann_comm_ops;
int32_t dann_comm_create_callbacks(const dann_comm_ops* ops, dann_comm** out);
/* one process per GPU: rank 0 draws the id, the host distributes it (MPI / torch.distributed / a socket), every rank
 * creates its communicator on its device (-1 = current).  DANN_EUNSUPPORTED when librccl cannot be loaded. */
int32_t dann_comm_rccl_unique_id(dann_rccl_unique_id* out);
int32_t dann_comm_create_rccl(const dann_rccl_unique_id* id, uint32_t rank, uint32_t world, int32_t device, dann_comm** out);
/* one process, `world` devices (ranks = host threads): out receives `world` communicators */
int32_t dann_comm_create_local(const int32_t* devices, uint32_t world, dann_comm** out);
int32_t dann_comm_destroy(dann_comm* comm);
int32_t dann_comm_rank(const dann_comm* comm);
int32_t dann_comm_world(const dann_comm* comm);
/* the communicator's all-gather on caller buffers (device pointers on `device`); the build's primitive, exported for
 * pre-flight checks of a deployment */
int32_t dann_comm_all_gather_device(dann_comm* comm, int32_t device, const void* d_send, void* d_recv, uint64_t bytes);
/* dann_build over `world` identical replicas (rows already stored on every rank): same batch schedule, same graph on
 * every replica as a single-GPU dann_build.  Collective: every rank calls it with the same arguments.  Returns the
 * number of batches; stats (optional, 4 entries): exchange rounds, bytes of pending rows gathered, rows rewritten by
 * their owners, bytes of rewritten rows gathered. */
int32_t dann_build_sharded(dann_index* idx, dann_comm* comm, const dann_build_config* cfg, uint32_t first, uint32_t n,
                           float growth, uint32_t max_batch, uint64_t* stats);
/* every rank holds the same nq queries (host) and receives all nq results: rank r searches partition r of the block
 * (diskann/src/utils/async_tools.rs:289-365), the k results per query are all-gathered.  Collective. */
int32_t dann_search_sharded(dann_index* idx, dann_comm* comm, const void* queries, uint32_t nq, uint32_t l_value,
                            uint32_t beam_width, uint32_t k, uint32_t* out_ids, float* out_dists);
/* plain copies for hosts without HIP bindings (kind 0 host->device, 1 device->host, 2 device->device; synchronous) */
int32_t dann_memcpy_device(int32_t device, void* dst, const void* src, uint64_t bytes, int32_t kind);

/* one process driving several devices ("device mask"): one replica per entry of `devices` (an ordinal may repeat).
 * set_elements replicates the rows, build runs dann_build_sharded on one host thread per replica over the in-process
 * communicator, search_batch partitions the query block over the replicas (host buffers; out_stats optional). */
typedef struct dann_multi dann_multi;
int32_t dann_multi_create(const dann_config* cfg, const void* start_rows, uint64_t start_len, const int32_t* devices,
                          uint32_t ndev, dann_multi** out);
int32_t dann_multi_destroy(dann_multi* m);
int32_t dann_multi_size(const dann_multi* m);
dann_index* dann_multi_replica(dann_multi* m, uint32_t i);
int32_t dann_multi_set_elements(dann_multi* m, uint32_t first_slot, uint32_t n, const void* rows, uint64_t len);
int32_t dann_multi_build(dann_multi* m, const dann_build_config* cfg, uint32_t first, uint32_t n, float growth,
                         uint32_t max_batch, uint64_t* stats);
int32_t dann_multi_search_batch(dann_multi* m, const void* queries, uint32_t nq, uint32_t l_value, uint32_t beam_width,
                                uint32_t k, uint32_t* out_ids, float* out_dists, dann_search_stats* out_stats);

/* ---- search server: the reference's serving model -- N workers calling DiskANNIndex::search on one shared index,
 * one query per call (diskann-benchmark-core/src/search/api.rs:399-436, tokio.rs:10-14) -- without a kernel launch
 * per call.  dann_server_start keeps `workers` wavefronts resident on the device (Knn search, beam width 1, L =
 * l_value, k results); dann_search_submit copies one query (host pointer, layer bytes) into a ring in host-mapped
 * memory and returns a ticket; dann_search_wait blocks until that query's result has arrived and copies it out
 * (ids are slot ids, unwritten entries 0xFFFFFFFF / +inf, as dann_search_batch).  Every ticket must be waited for
 * exactly once, in any order and by any thread; at most `ring` tickets can be outstanding (a further submit waits for a
 * dann_search_wait to return a result slot; collecting tickets late never holds up another caller's submission).
 * Submit / wait / poll may be called from any number of threads concurrently and take no lock on the index; results are
 * identical to dann_search_batch.  Mutations of the index (set / insert / build / load) are refused with DANN_EBUSY
 * while tickets are outstanding (submitted and not yet collected by dann_search_wait), and a submit during a mutation
 * is refused with DANN_EBUSY.  dann_server_stop may be called while other threads are inside submit / wait / poll:
 * it unpublishes the server, those calls return DANN_EINVAL ("the server is being stopped"), and nothing is freed
 * before the last of them has left; uncollected tickets die with the server.  A submit whose ring position no worker
 * takes within 30 s means the resident kernel is dead: from then on every submit / wait on this server fails at once
 * with DANN_EHIP until dann_server_stop + dann_server_start.  The resident kernel leaves after idle_timeout_us
 * without a submission (default 100 ms), and after 200 ms of residence under a steady stream of submissions, and is
 * relaunched by the next submit, wait or poll: a device-wide synchronisation elsewhere in the process (every hipFree is
 * one -- dann_index_destroy of another index, for instance) waits at most that long on this server.  Row lengths must be a multiple
 * of 16 bytes; L + start points <= 256. */
typedef struct {
    uint32_t l_value;
    uint32_t k;
    uint32_t workers;          /* searches in flight on the device (wavefronts), e.g. 1024 */
    uint32_t ring;             /* ring entries (rounded up to a power of two); 0 = 4 * workers */
    uint32_t idle_timeout_us;  /* 0 = 100000 */
} dann_server_config;
int32_t dann_server_start(dann_index* idx, const dann_server_config* cfg);
int32_t dann_server_stop(dann_index* idx);  /* also called by dann_index_destroy */
int32_t dann_search_submit(dann_index* idx, const void* query, uint64_t* ticket);
/* 1 = the result of `ticket` has arrived (dann_search_wait will not block), 0 = not yet */
int32_t dann_search_poll(dann_index* idx, uint64_t ticket);
int32_t dann_search_wait(dann_index* idx, uint64_t ticket, uint32_t* out_ids, float* out_dists,
                         dann_search_stats* out_stats);
/* tickets issued so far, relaunches of the resident kernel after an idle exit */
int32_t dann_server_stats(dann_index* idx, uint64_t* submitted, uint64_t* relaunches);

#ifdef __cplusplus
}
#endif
#endif /* DANN_H */
