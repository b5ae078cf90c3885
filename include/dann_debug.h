/* dann_debug.h -- measurement and diagnostic hooks of libdann_hip.so.
 *
 * Not part of the drop-in boundary (include/dann.h): nothing here replaces an interface of the reference.  The
 * benchmark harness (bench.py) and the tests use these to measure the device and to check one kernel in isolation;
 * an adapter for diskann-inmem never needs them, and bindings/rust/dann_sys.rs does not declare them. */
#ifndef DANN_DEBUG_H
#define DANN_DEBUG_H

#include "dann.h"

#ifdef __cplusplus
extern "C" {
#endif

/* diagnostic: bandwidth (GB/s) of a plain streaming read of `bytes` bytes of HBM on `device` (16-byte loads, four in
 * flight per lane), averaged over `reps` launches -- the achievable line to hold next to the 8 TB/s peak */
int32_t dann_debug_stream_read_gbps(int32_t device, uint64_t bytes, uint32_t reps, double* gbps);

/* diagnostic: the Gram block of the MFMA prunes (gram_tiles_kernel) for n <= 256 rows of dtype
 * DANN_F32 / DANN_F16: out_gram is n x mg (mg rounded up to 32, at most 96): entry (i, j), j <= i, is the f32 FMA chain
 * over k = 0 .. dim-1 of row_i[k] * row_j[k] for DANN_F32 rows and for DANN_F16 | 0x100 (f16 rows widened exactly, the
 * f32 matrix core); plain DANN_F16 is the builds' default for f16 rows, v_mfma_f32_32x32x16_f16: exact products, f32
 * sums in the hardware's order -- within the sweep's error interval of the chain; out_nrm[i] = |row_i|^2 accumulated in
 * f64 */
int32_t dann_debug_gram_tiles(int32_t device, int32_t dtype, const void* rows, uint32_t n, uint32_t dim, uint32_t mg,
                              float* out_gram, float* out_nrm);

/* measurement harness: `threads` host threads issue single-query calls on the shared index, thread t serving queries
 * t, t + threads, ...: mode 0 = dann_search_batch(nq = 1) per call, mode 1 = submit / wait with up to `depth` tickets
 * outstanding per thread (1 = synchronous).  out_latency_us (nq, optional): submit -> result, host clock. */
int32_t dann_debug_concurrent_callers(dann_index* idx, const void* queries, uint32_t nq, uint32_t l_value, uint32_t k,
                                      uint32_t threads, uint32_t mode, uint32_t depth, uint32_t* out_ids,
                                      float* out_dists, float* out_latency_us, double* out_seconds);

/* ---- development switches (per index; none of them ever changes a result) ------------------------------------------
 * The library used to read these from DANN_* environment variables, some of them cached at the first call of the
 * process; they are per-index settings now, read on every call.  value = NaN restores the default. */
enum {
    DANN_DBG_TUNE_OFF = 0,               /* bit mask: 1 row prefetch in latency mode, 2 latency-mode table sizing, 4 teams of
                                            wavefronts, 8 the teams' speculative expansion, 16 two queries per wavefront,
                                            32 the lookup-table kernel of PQ rows, 64 the self-start of the teams' visited wave
                                            (default 0) */
    DANN_DBG_TUNE_ON = 1,                /* bit mask: 1 row prefetch in the throughput regime too (default 0) */
    DANN_DBG_PAIR_MIN_QUERIES = 2,       /* launches of at least this many queries take two queries per wavefront
                                            (default 20 x compute units) */
    DANN_DBG_TEAM_MAX_QUERIES = 3,       /* launches of at most this many queries take a team per query (default 4 x CUs) */
    DANN_DBG_HOST_PIPELINE = 4,          /* 0: dann_search_batch never chunks its host buffers; 2 .. 8: that many lanes and
                                            no temporary page-locking of buffers seen before (default 1 = three lanes,
                                            page-locking on) */
    DANN_DBG_SWEEP_ONE_BY_ONE = 5,       /* 1: the MFMA prune's sweep decides one candidate at a time (default 0) */
    DANN_DBG_POOL_GRAM = 6,              /* 0: the pool prune of rows >= 1 KiB stays on the row kernel (default 1) */
    DANN_DBG_GRAM_COLS = 7,              /* columns of the Gram block: 32 / 64 / 96 (default 96) */
    DANN_DBG_GRAM_ESCALE = 8,            /* test hook: widens the error interval of the Gram distances (default 1.0) */
    DANN_DBG_BACKEDGE_GRAM_ROWS = 9,     /* Gram rows per back-edge list (default degree + 8 rounded up to 32) */
    DANN_DBG_SERVER_MAX_RESIDENT_US = 10,/* residency bound of the server kernel, read by dann_server_start (default 200000) */
    DANN_DBG_VERBOSE = 11,               /* 1: launch sizing decisions on stderr (default 0) */
    DANN_DBG_HT16_OPEN_EIGHTHS = 12,     /* load (in eighths of its entries, 4 .. 7) up to which a 16-bit visited table takes
                                            new ids before it is frozen (default 6 = 75 %) */
    DANN_DBG_BACKEDGE_SINGLE_POOL = 13,  /* small rows: the back-edge phase runs as ONE kernel (list build + prune per target, LDS
                                            pool sized by the batch's longest list) while that pool has at most this many
                                            entries; beyond it, scan + short / long worklists (default 128) */
    DANN_DBG_HT16_MAX_PROBES = 14,       /* test hook: cap on the probes per id of a 16-bit visited table (default 64; a large
                                            index leaves three): small test indexes reach the overflow table with it */
    DANN_DBG_HOST_CHUNK = 15,            /* queries per chunk of the host-pointer pipeline (default 16384) */
    DANN_DBG_GRAM_F16_WIDEN = 16,        /* 1: the MFMA prunes widen f16 rows and use the f32 matrix core (rounds 3-5; default
                                            0: v_mfma_f32_32x32x16_f16) */
    DANN_DBG_TIME_SMALL_LAUNCHES = 17,   /* 1: HIP events also around Knn search launches of at most 2 048 queries (they
                                            cost such a call 4-5 us of its 80 ... 300; default 0: those launches count in
                                            dann_kernel_time and dann_debug_search_families with 0 ms) */
    DANN_DBG_COUNT = 18
};
int32_t dann_debug_set(dann_index* idx, int32_t key, double value);
int32_t dann_debug_get(const dann_index* idx, int32_t key, double* value);

/* which beam-search kernel family served the launches of this index since the last dann_kernel_time_reset: launches
 * and HIP-event milliseconds per family (tests assert that the kernel they claim to test is the one that ran; bench.py
 * names the family of every timed leg).  out_launches / out_ms: DANN_FAMILY_COUNT entries each (either may be null). */
enum {
    DANN_FAMILY_ONE_WAVE = 0,    /* beam_search_kernel, one wavefront per query */
    DANN_FAMILY_TEAM = 1,        /* beam_search_kernel, a team of five wavefronts per query (latency regime) */
    DANN_FAMILY_PAIR = 2,        /* pair_search_kernel, two queries per wavefront (128-byte integer rows) */
    DANN_FAMILY_PERSISTENT = 3,  /* beam_search_kernel, persistent wavefronts sharing a batch (dann_set_max_concurrency) */
    DANN_FAMILY_SERVER = 4,      /* the resident server kernel */
    DANN_FAMILY_PQ_LUT = 5,      /* pq_search_kernel, lookup table in registers (PQ rows) */
    DANN_FAMILY_COUNT = 6
};
int32_t dann_debug_search_families(const dann_index* idx, uint64_t* out_launches, double* out_ms);
/* family name for logs ("one_wave", "team", "pair", "persistent", "server", "pq_lut"); null for an unknown family */
const char* dann_debug_family_name(int32_t family);

/* small dann_search_batch calls (host pointers, at most 16 queries) of several threads share launches: out2 = {launches,
 * calls served by them} since the index was created -- calls / launches is the mean number of calls per launch */
int32_t dann_debug_small_call_stats(dann_index* idx, uint64_t* out2);

/* how the PQ trainer's rolling f64 sums (the D^2 draw and its totals, plusplus.rs:446-462) were evaluated on `device`
 * since the last reset: out4 = {wavefront-level ranges and thread-level ranges summed in parallel because no addition
 * in them can round, ranges walked element by element, elements walked}.  Same bits either way; the tests assert that
 * their hard inputs reach the walk. */
int32_t dann_debug_pq_rolling_sum_stats(int32_t device, uint64_t* out4, int32_t reset);

#ifdef __cplusplus
}
#endif

#endif /* DANN_DEBUG_H */
