/*
 * oracle/dann_oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is a CPU restatement of the reference
 * (microsoft/DiskANN, Rust workspace v0.56) algorithm for the hot path named in
 * BASELINE.json.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load it -- and only as the checker / the timed CPU
 * baseline, never as part of the shipped product path (diskann_amd/).
 *
 * Parity pins (tests/golden/, tests/test_oracle_*.py): the reference's own golden JSONs --
 * grid_search (18 cases: ids, distances, comparisons, hops), range_search (5), inline (12,
 * incl. AdaptiveL), multihop (2), filtered_range_search (7), paged_search (3, page by page),
 * grid_insert (all 15 cases exact -- the 12 tie-heavy lattices under tie rule 6, Rust's own sort order
 * restated in rust_unstable_sort.h) -- the exhaustive f16
 * conversion table, the in-source provider `smoke` expectations, compute_adaptive_l's unit
 * tests, the PQ lookup KAT and the Chunk::find_closest test pattern, the SQ training
 * contract.  The reference itself (Rust) cannot be compiled in this image (no cargo/rustc),
 * so there is no oracle/_ref.
 *
 * Enum values equal include/dann.h (and the reference's `#[repr(C)] Metric`,
 * diskann-vector/src/distance/metric.rs:8-20).
 */
#ifndef DANN_ORACLE_H
#define DANN_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    ORC_F32 = 0,
    ORC_F16 = 1,
    ORC_U8 = 2,
    ORC_I8 = 3,
    ORC_SQ8 = 4, /* dim code bytes + f32 compensation */
    ORC_PQ = 5   /* pq_chunks code bytes; f32 queries; lookup-table distances */
};
enum { ORC_COSINE = 0, ORC_INNER_PRODUCT = 1, ORC_L2 = 2, ORC_COSINE_NORMALIZED = 3 };
enum { ORC_IBC_NONE = 0, ORC_IBC_ALL = 0xFFFFFFFFu }; /* else Max(n) */

/* An index over caller-owned arrays, laid out as diskann-inmem lays them out:
 * rows:  (capacity + nstart) rows of `row_stride` bytes (store.rs:198-241), the
 *        frozen start points occupy slots [capacity, capacity+nstart)
 *        (store.rs:259-262);
 * adj:   (capacity + nstart) rows of (max_degree+1) u32: [len, ids...]
 *        (neighbors.rs:60-101). */
typedef struct {
    int32_t dtype;
    int32_t metric;
    uint32_t dim;
    uint32_t capacity;
    uint32_t nstart;
    uint32_t max_degree;
    uint64_t row_stride;
    uint8_t* rows;
    uint32_t* adj;
    float sq_scale;          /* ORC_SQ8: ScalarQuantizer::scale() */
    float sq_shift_norm_sq;  /* ORC_SQ8: shift_square_norm()      */
    const float* pq_pivots;      /* ORC_PQ: 256 x dim */
    const uint32_t* pq_offsets;  /* ORC_PQ: pq_chunks + 1 */
    uint32_t pq_chunks;
    /* inline concurrency tags (store.rs:133-158, tag.rs): byte offset of the tag inside a row (the reference puts
     * it right after the payload: dim * sizeof(T)), 0 = no tags / every slot readable.  A slot whose tag is below
     * Tag::PUBLISHED (254) is skipped by expand_beam after the visited insert and is not counted (provider.rs:448-473,
     * 681-686). */
    uint32_t tag_offset;
} orc_index;

/* graph::config::Builder (diskann/src/graph/config/mod.rs:261-338, defaults.rs) */
typedef struct {
    uint32_t pruned_degree;
    uint32_t max_degree;      /* "max_degree_with_slack" */
    uint32_t l_build;
    float alpha;
    uint32_t max_occlusion_size;
    uint32_t max_backedges;
    uint32_t intra_batch_candidates; /* ORC_IBC_NONE, n, or ORC_IBC_ALL */
    uint32_t saturate_after_prune;
} orc_build_config;

/* ---- numerics -------------------------------------------------------- */
/* f16 bit pattern -> f32 (pinned by diskann-wide/test_data/float16_conversion.txt) */
float orc_f16_to_f32(uint16_t h);
uint16_t orc_f32_to_f16(float f);

/* layers::Distance::evaluate (T x T, prune path), V3 association.
 * diskann-inmem/src/layers/full.rs:224-242 -> distance_provider.rs -> simd.rs */
float orc_distance(int32_t dtype, int32_t metric, const void* x, const void* y, size_t dim);
/* layers::QueryDistance::evaluate (query x row, search path; f16 rows use the
 * f32 x f16 kernels with the query widened once). full.rs:317-336,351-504 */
float orc_query_distance(int32_t dtype, int32_t metric, const void* query, const void* row, size_t dim);
/* scalar reference (diskann-vector/src/distance/reference.rs) for tolerance tests */
float orc_distance_scalar_ref(int32_t dtype, int32_t metric, const void* x, const void* y, size_t dim);
/* AVX2 intrinsics twin of orc_query_distance (bitwise identical; used for the timed
 * CPU baseline).  Falls back to the emulation for combinations without a twin. */
float orc_query_distance_fast(int32_t dtype, int32_t metric, const void* query, const void* row, size_t dim);

/* ---- search ---------------------------------------------------------- */
/* DiskANNIndex::search(Knn{l,beam}) through the inmem2 accessor:
 * index.rs:1933-2000, knn_search.rs:155-193, provider.rs:408-480,899-950, queue.rs.
 * stats = {cmps, hops, result_count(reference quirk: k-1 when the buffer fills)}.
 * out_ids are slot ids (< capacity); unwritten entries are 0xFFFFFFFF / +inf.
 * rec_* (optional) receives the VisitedSearchRecord (record.rs:86-93).
 * Returns number of results written (>=0) or <0 on error. */
int32_t orc_search(const orc_index* ix, const void* query, uint32_t l_value, uint32_t beam_width,
                   uint32_t k, uint32_t* out_ids, float* out_dists, uint32_t* stats,
                   uint32_t* rec_ids, float* rec_dists, uint32_t rec_cap, uint32_t* rec_n);

/* nq independent queries over `threads` host threads, static block partition
 * (diskann-benchmark-core/src/search/api.rs:399-436). per_query_ns optional. */
int32_t orc_search_batch(const orc_index* ix, const void* queries, uint32_t nq, uint32_t l_value,
                         uint32_t beam_width, uint32_t k, uint32_t* out_ids, float* out_dists,
                         uint32_t* out_counts, uint32_t* stats, uint32_t threads, int32_t fast,
                         uint64_t* per_query_ns);

/* graph::search::Range (diskann/src/graph/search/range_search.rs:246-470): initial Knn-style search with
 * L = starting_l, then (if enough of the list is in range) breadth-first expansion of everything within
 * radius * range_slack.  max_returned == 0 means unlimited; has_inner != 0 enables inner_radius.
 * stats = {cmps, hops, result_count, range_search_second_round} with the reference's accounting
 * (cmps of the initial phase only; hops = initial + cumulative, :308-314).  Returns results written. */
int32_t orc_range_search(const orc_index* ix, const void* query, uint32_t starting_l, uint32_t beam_width,
                         float radius, int32_t has_inner, float inner_radius, float initial_slack,
                         float range_slack, uint64_t max_returned, uint32_t* out_ids, float* out_dists,
                         uint64_t out_cap, uint32_t* stats);

/* Filtered searches through graph/ext/labeled.rs (QueryLabelProvider == bitmap over slot ids, start points
 * included; bit i of filter_bits[i >> 5]).
 * InlineFilterSearch (search/inline_filter_search.rs:69-301): adaptive_samples == 0 means no AdaptiveL.
 * MultihopFilterSearch (search/multihop_filter_search.rs:46-244).
 * FilteredRange (search/filtered_range_search.rs:111-330): stats as orc_range_search but cmps/hops cumulative.
 * ORACLE TIE RULE: the reference's sort_unstable_by(distance) calls are restated as stable sorts. */
int32_t orc_adaptive_l(uint32_t base_l, uint32_t visited, uint32_t matched, double max_multiplier);
int32_t orc_inline_filter_search(const orc_index* ix, const void* query, uint32_t l_value, uint32_t beam_width,
                                 uint32_t k, const uint32_t* filter_bits, uint32_t adaptive_samples,
                                 double adaptive_scale, uint32_t* out_ids, float* out_dists, uint32_t* stats);
int32_t orc_multihop_search(const orc_index* ix, const void* query, uint32_t l_value, uint32_t beam_width, uint32_t k,
                            const uint32_t* filter_bits, uint32_t* out_ids, float* out_dists, uint32_t* stats);
int32_t orc_filtered_range_search(const orc_index* ix, const void* query, uint32_t starting_l, uint32_t beam_width,
                                  float radius, int32_t has_inner, float inner_radius, float initial_slack,
                                  float range_slack, uint64_t max_returned, const uint32_t* filter_bits,
                                  uint32_t* out_ids, float* out_dists, uint64_t out_cap, uint32_t* stats);

/* Paged search: DiskANNIndex::paged_search (diskann/src/graph/index.rs:2075-2155) + PagedSearch::next_page
 * (search/paged.rs:53-149).  The queue is the auto-resizable variant (queue.rs:95-121): nothing is ever dropped.
 * next returns the number of results of the page (0 = exhausted) or < 0 (k == 0 or k > l_value). */
typedef struct orc_paged orc_paged;
orc_paged* orc_paged_begin(const orc_index* ix, const void* query, uint32_t l_value);
int32_t orc_paged_next(orc_paged* s, uint32_t k, uint32_t* out_ids, float* out_dists);
void orc_paged_end(orc_paged* s);

/* ExpandBeam::expand_beam (provider.rs:620-690) for a pre-filtered id list. */
int32_t orc_expand_beam(const orc_index* ix, const void* query, const uint32_t* ids, uint32_t n,
                        uint32_t* out_ids, float* out_dists);

/* checker for the GPU build path's MFMA Gram (dann_debug_gram_tiles): one f32 fmaf chain per entry */
void orc_gram_chain(const float* rows, uint32_t n, uint32_t dim, float* out);

/* ---- build ----------------------------------------------------------- */
/* prune::robust_prune over a sorted pool (internal/prune.rs:106-259) + occlude_list
 * (index.rs:2565-2650).  pool_* is sorted here with the oracle's tie rule
 * (distance, then original position).  Returns the number of neighbours. */
int32_t orc_prune_pool(const orc_index* ix, const orc_build_config* cfg, uint32_t location,
                       uint32_t* pool_ids, float* pool_dists, uint32_t pool_n, int32_t force_saturate,
                       uint32_t* out_neighbors, uint64_t* pair_evals);
/* DiskANNIndex::insert for a row already stored at `slot` (index.rs:226-341). */
/* tie order of RobustPrune's candidate sort (see sort_pool): 6 (default) = Rust's own select_nth_unstable_by +
 * sort_unstable_by order (rust_unstable_sort.h), under which every counter and every search result of all fifteen
 * grid_insert goldens is reproduced exactly -- the product's DANN_TIE_RUST; 0 = pool position, the product's
 * DANN_TIE_POSITION; 1..5 = alternative orders used only to measure the tie envelope of the reference's grid_insert
 * goldens.  Process-global, not thread-safe. */
void orc_set_tie_rule(int32_t rule, uint64_t seed);
/* the restated Rust sort on its own, in place over (ids, dists): mode 0 SortedNeighbors::new(v, max) (returns the new
 * length), 1 sort_unstable_by, 2 the small sort (n <= 32), 3 select_nth_unstable_by(max) */
int64_t orc_rust_sort(int32_t mode, uint32_t* ids, float* dists, uint64_t n, uint64_t max);
uint64_t orc_rust_sort_fallbacks(void);
/* how often each part of the restated sort ran since the library was loaded (rust_sort::Path order) */
uint32_t orc_rust_sort_paths(uint64_t* out, uint32_t n);
/* CPU distance micro-benchmark in the shape of diskann-benchmark-simd (see dann_oracle.cpp); distances per second */
double orc_bench_distance(int32_t dtype, int32_t metric, uint32_t dim, uint64_t nrows, uint32_t loops, int32_t random_order,
                          uint32_t threads, uint64_t seed, double* checksum);
/* counters: NULL or five words, added to: query distances, pair (prune) distances, set_neighbors, append_neighbors,
 * get_neighbors (the last three as the reference's test provider counts them, graph/test/provider.rs:196-220) */
int32_t orc_insert(orc_index* ix, const orc_build_config* cfg, uint32_t slot, uint64_t* counters);
/* DiskANNIndex::multi_insert for rows already stored at slots[0..n)
 * (index.rs:815-1030, max_minibatch_par = 1). */
int32_t orc_multi_insert(orc_index* ix, const orc_build_config* cfg, const uint32_t* slots, uint32_t n,
                         uint64_t* counters);
/* medoid rule (diskann-utils/src/sampling/medoid.rs:15-48), f32 rows. Returns row index. */
int64_t orc_medoid_f32(const float* data, uint64_t nrows, uint32_t dim, float* out_mean);

/* ---- quantised variants ---------------------------------------------- */
/* PQ: FixedChunkPQTable::populate_chunk_distances_impl + pq_dist_lookup_single
 * (diskann-providers/src/model/pq/fixed_chunk_pq_table.rs:152-192,82-100). */
void orc_pq_build_lut(int32_t metric, const float* pivots /*256 x dim*/, const float* centroid /*dim or NULL*/,
                      const uint32_t* chunk_offsets /*nchunks+1*/, uint32_t nchunks, uint32_t dim,
                      const float* query, float* lut /*nchunks x 256*/);
float orc_pq_lookup(const float* lut, const uint8_t* code, uint32_t nchunks);
/* PQ compression, TransposedTable::compress_into (diskann-quantization/src/product/tables/transposed/table.rs:382-403,
 * pivots.rs:253-345): code[r][c] = the pivot of chunk c closest to row r's chunk (lane-wise tie rule, see .cpp).
 * pivots: ncenters (<= 256) x dim.  Returns 0, -1 on bad arguments, or -(2 + r * nchunks + c) when every score of
 * (row r, chunk c) is infinite / NaN (TableCompressionError::InfinityOrNaN). */
int32_t orc_pq_square_norms(const float* pivots, uint32_t ncenters, const uint32_t* chunk_offsets, uint32_t nchunks,
                            uint32_t dim, float* norms);
int64_t orc_pq_compress(const float* pivots, uint32_t ncenters, const uint32_t* chunk_offsets, uint32_t nchunks,
                        uint32_t dim, const float* rows, uint64_t n, uint8_t* codes);
/* k-means++ seeding (kmeans_plusplus_into_inner, algorithms/kmeans/plusplus.rs:366-497) per chunk with the caller's
 * random draws; centers: ncenters x dim (chunk columns of unselected centres are zero); selected: nchunks. */
typedef struct {
    void* ctx;
    uint64_t (*uniform_index)(void* ctx, uint32_t chunk, uint64_t n);
    double (*uniform_f64)(void* ctx, uint32_t chunk, double high);
} orc_rng;
int32_t orc_pq_kmeanspp(const float* data, uint64_t n, uint32_t dim, const uint32_t* chunk_offsets, uint32_t nchunks,
                        uint32_t ncenters, const orc_rng* rng, float* centers, uint32_t* selected);
/* PQ training minus the seeding: the Lloyd iterations of LightPQTrainingParameters::train
 * (product/train.rs:96-226, algorithms/kmeans/lloyds.rs:23-438) for every chunk; `centers` (ncenters x dim)
 * carries the initial centres in and the trained pivots out.  assignments: nchunks x n (optional), residuals:
 * nchunks (optional). */
int32_t orc_pq_lloyds(const float* data, uint64_t n, uint32_t dim, const uint32_t* chunk_offsets, uint32_t nchunks,
                      uint32_t ncenters, float* centers, uint32_t max_reps, uint32_t* assignments, float* residuals);
/* SQ training: ScalarQuantizationParameters::train (diskann-quantization/src/scalar/train.rs:33-52, utils.rs:109-199):
 * shift[d] = mean_d - p, scale = 2p, p = standard_deviations * sqrt(max_d variance_d); f64 sums in row order. */
void orc_sq8_train(const float* data, uint64_t n, uint32_t dim, double standard_deviations, float* shift,
                   float* scale, float* mean_norm);
/* SQ-8: ScalarQuantizer::compress + compensated distances
 * (diskann-quantization/src/scalar/quantizer.rs:189-236,407-430, vectors.rs:171-338). */
void orc_sq8_compress(const float* x, uint32_t dim, const float* shift, float scale, uint8_t* code,
                      float* compensation);
float orc_sq8_distance(int32_t metric, const uint8_t* x, float cx, const uint8_t* y, float cy,
                       uint32_t dim, float scale, float shift_norm_sq);

#ifdef __cplusplus
}
#endif
#endif
