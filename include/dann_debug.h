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
 * over k = 0 .. dim-1 of row_i[k] * row_j[k] (f16 rows widened exactly); out_nrm[i] = |row_i|^2 accumulated in f64 */
int32_t dann_debug_gram_tiles(int32_t device, int32_t dtype, const void* rows, uint32_t n, uint32_t dim, uint32_t mg,
                              float* out_gram, float* out_nrm);

/* measurement harness: `threads` host threads issue single-query calls on the shared index, thread t serving queries
 * t, t + threads, ...: mode 0 = dann_search_batch(nq = 1) per call, mode 1 = submit / wait with up to `depth` tickets
 * outstanding per thread (1 = synchronous).  out_latency_us (nq, optional): submit -> result, host clock. */
int32_t dann_debug_concurrent_callers(dann_index* idx, const void* queries, uint32_t nq, uint32_t l_value, uint32_t k,
                                      uint32_t threads, uint32_t mode, uint32_t depth, uint32_t* out_ids,
                                      float* out_dists, float* out_latency_us, double* out_seconds);

#ifdef __cplusplus
}
#endif

#endif /* DANN_DEBUG_H */
