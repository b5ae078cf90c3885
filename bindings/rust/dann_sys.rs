//! dann_sys.rs -- `extern "C"` declarations of include/dann.h for a `diskann-amd-sys` crate.
//!
//! NOT COMPILED HERE: the build image has no Rust toolchain.  The block is mechanical (one item per C export of
//! libdann_hip.so; struct layouts are checked against the header by tests/test_abi.py on the ctypes side) and is
//! the same text INTEGRATION.md section 1 shows.  Link with `-ldann_hip` (ROCm's libamdhip64 is its only dependency).
#![allow(non_camel_case_types, dead_code)]
use std::ffi::{c_char, c_void};

#[repr(C)] pub struct DannConfig {            // include/dann.h: dann_config
    pub dtype: i32,                           // 0 f32, 1 f16, 2 u8, 3 i8, 4 SQ-8 (codes + f32 compensation), 5 PQ codes
    pub metric: i32,                          // == diskann_vector::distance::Metric as i32 (metric.rs:8-20, already #[repr(C)])
    pub dim: u32, pub capacity: u32, pub max_degree: u32, pub num_start_points: u32,
    pub row_stride: u32,                      // 0 = packed; dann_inmem2_row_stride() = Store stride (store.rs:198-211)
    pub device: i32,
    pub sq_scale: f32, pub sq_shift_norm_sq: f32,   // SQ-8: ScalarQuantizer::scale(), shift_square_norm()
    pub pq_chunks: u32,                       // PQ: code bytes per row
    pub inline_tags: u32,                     // 1: rows carry the Store's tag byte (store.rs:133-158); unreadable slots are skipped
}
#[repr(C)] pub struct DannBuildConfig {       // graph::config::Config (config/mod.rs:178-216)
    pub pruned_degree: u32, pub max_degree: u32, pub l_build: u32, pub alpha: f32,
    pub max_occlusion_size: u32, pub max_backedges: u32, pub intra_batch_candidates: u32, pub saturate_after_prune: u32,
}
#[repr(C)] pub struct DannSearchStats {       // SearchStats (index.rs:90-102) + per-query status
    pub cmps: u32, pub hops: u32,
    pub result_count: u32,                    // what Translate::post_process returns (provider.rs:933-944): k - 1 when the buffer fills
    pub status: u32,
    pub written: u32,                         // entries actually written to the output buffers
}

#[repr(C)] pub struct DannFilter {            // include/dann.h: dann_filter == labeled::QueryLabelProvider as a bitmap
    pub mode: u32, pub bits: *const u32, pub stride_words: u64,
    pub adaptive_samples: u32, pub adaptive_scale: f64, pub matched_cap: u32 }

#[repr(C)] pub struct DannRng {              // include/dann.h: dann_rng -- closures over the per-chunk StdRng (train.rs:164-165)
    pub ctx: *mut c_void,
    pub uniform_index: extern "C" fn(ctx: *mut c_void, chunk: u32, n: u64) -> u64,     // Uniform::new(0, n).sample(rng)
    pub uniform_f64: extern "C" fn(ctx: *mut c_void, chunk: u32, high: f64) -> f64,    // Uniform::<f64>::new(0.0, high).sample(rng)
}

// Every export of include/dann.h (generated from diskann_amd/_ffi.py::SYMBOLS, which tests/test_abi.py checks against
// the header and the built library; argument meaning and const-ness: see the header).
#[link(name = "dann_hip")]
extern "C" {
    pub fn dann_layer_bytes(a0: i32, a1: u32) -> i32;
    pub fn dann_inmem2_row_stride(a0: i32, a1: u32) -> i32;
    pub fn dann_index_create(a0: *const DannConfig, a1: *mut c_void, a2: u64, a3: *mut *mut c_void) -> i32;
    pub fn dann_index_destroy(a0: *mut c_void) -> i32;
    pub fn dann_index_max_degree(a0: *mut c_void) -> i32;
    pub fn dann_index_get_config(a0: *mut c_void, a1: *const DannConfig) -> i32;
    pub fn dann_set_element(a0: *mut c_void, a1: u32, a2: *mut c_void, a3: u64) -> i32;
    pub fn dann_set_elements(a0: *mut c_void, a1: u32, a2: u32, a3: *mut c_void, a4: u64) -> i32;
    pub fn dann_get_element(a0: *mut c_void, a1: u32, a2: *mut c_void, a3: u64) -> i32;
    pub fn dann_upload_store(a0: *mut c_void, a1: *mut c_void, a2: u64, a3: u32) -> i32;
    pub fn dann_set_tags(a0: *mut c_void, a1: u32, a2: u32, a3: *mut c_void) -> i32;
    pub fn dann_get_tags(a0: *mut c_void, a1: u32, a2: u32, a3: *mut c_void) -> i32;
    pub fn dann_set_external_ids(a0: *mut c_void, a1: u32, a2: u32, a3: *mut c_void) -> i32;
    pub fn dann_to_external(a0: *mut c_void, a1: *mut c_void, a2: u64, a3: *mut c_void) -> i32;
    pub fn dann_get_neighbors(a0: *mut c_void, a1: u32, a2: *mut c_void, a3: u32, a4: *mut u32) -> i32;
    pub fn dann_set_neighbors(a0: *mut c_void, a1: u32, a2: *mut c_void, a3: u32) -> i32;
    pub fn dann_append_neighbors(a0: *mut c_void, a1: u32, a2: *mut c_void, a3: u32) -> i32;
    pub fn dann_set_neighbors_bulk(a0: *mut c_void, a1: *mut c_void, a2: u32, a3: *mut c_void) -> i32;
    pub fn dann_upload_graph(a0: *mut c_void, a1: *mut c_void, a2: u64) -> i32;
    pub fn dann_download_graph(a0: *mut c_void, a1: *mut c_void, a2: u64) -> i32;
    pub fn dann_distance(a0: *mut c_void, a1: *mut c_void, a2: u64, a3: *mut c_void, a4: u64, a5: *mut f32) -> i32;
    pub fn dann_distance_pairs(a0: *mut c_void, a1: *mut c_void, a2: *mut c_void, a3: u32, a4: *mut c_void) -> i32;
    pub fn dann_query_create(a0: *mut c_void, a1: *mut c_void, a2: u64, a3: *mut *mut c_void) -> i32;
    pub fn dann_query_destroy(a0: *mut c_void) -> i32;
    pub fn dann_query_distance(a0: *mut c_void, a1: *mut c_void, a2: u64, a3: *mut f32) -> i32;
    pub fn dann_expand_beam(a0: *mut c_void, a1: *mut c_void, a2: u32, a3: *mut c_void, a4: *mut c_void, a5: *mut u32) -> i32;
    pub fn dann_expand_beam_batch(a0: *mut c_void, a1: *mut c_void, a2: u32, a3: *mut c_void, a4: *mut c_void, a5: *mut c_void) -> i32;
    pub fn dann_search_batch(a0: *mut c_void, a1: *mut c_void, a2: u32, a3: u32, a4: u32, a5: u32, a6: *mut c_void, a7: *mut c_void, a8: *mut c_void) -> i32;
    pub fn dann_search_batch_device(a0: *mut c_void, a1: *mut c_void, a2: u32, a3: u32, a4: u32, a5: u32, a6: *mut c_void, a7: *mut c_void, a8: *mut c_void) -> i32;
    pub fn dann_range_search_batch(a0: *mut c_void, a1: *mut c_void, a2: u32, a3: u32, a4: u32, a5: f32, a6: i32, a7: f32, a8: f32, a9: f32, a10: u32, a11: u32, a12: *mut c_void, a13: *mut c_void, a14: *mut c_void, a15: *mut c_void) -> i32;
    pub fn dann_filtered_search_batch(a0: *mut c_void, a1: *mut c_void, a2: u32, a3: u32, a4: u32, a5: u32, a6: *const DannFilter, a7: *mut c_void, a8: *mut c_void, a9: *mut c_void) -> i32;
    pub fn dann_filtered_range_search_batch(a0: *mut c_void, a1: *mut c_void, a2: u32, a3: u32, a4: u32, a5: f32, a6: i32, a7: f32, a8: f32, a9: f32, a10: u32, a11: u32, a12: *const DannFilter, a13: *mut c_void, a14: *mut c_void, a15: *mut c_void, a16: *mut c_void) -> i32;
    pub fn dann_paged_begin(a0: *mut c_void, a1: *mut c_void, a2: u32, a3: u32, a4: u32, a5: *mut *mut c_void) -> i32;
    pub fn dann_paged_next(a0: *mut c_void, a1: u32, a2: *mut c_void, a3: *mut c_void, a4: *mut c_void) -> i32;
    pub fn dann_paged_end(a0: *mut c_void) -> i32;
    pub fn dann_rerank_batch(a0: *mut c_void, a1: *mut c_void, a2: u32, a3: *mut c_void, a4: u32, a5: u32, a6: *mut c_void, a7: *mut c_void) -> i32;
    pub fn dann_rerank_batch_device(a0: *mut c_void, a1: *mut c_void, a2: u32, a3: *mut c_void, a4: u32, a5: u32, a6: *mut c_void, a7: *mut c_void) -> i32;
    pub fn dann_search_record_batch(a0: *mut c_void, a1: *mut c_void, a2: u32, a3: u32, a4: *mut c_void, a5: *mut c_void, a6: u32, a7: *mut c_void, a8: *mut c_void) -> i32;
    pub fn dann_prune_batch(a0: *mut c_void, a1: *const DannBuildConfig, a2: *mut c_void, a3: u32, a4: *mut c_void, a5: *mut c_void, a6: *mut c_void, a7: i32, a8: *mut c_void) -> i32;
    pub fn dann_insert_batch(a0: *mut c_void, a1: *const DannBuildConfig, a2: *mut c_void, a3: u32) -> i32;
    pub fn dann_insert_batch_candidates(a0: *mut c_void, a1: *const DannBuildConfig, a2: *mut c_void, a3: u32, a4: u32, a5: u32, a6: *mut c_void) -> i32;
    pub fn dann_insert_batch_commit(a0: *mut c_void, a1: *const DannBuildConfig, a2: *mut c_void, a3: u32, a4: *mut c_void) -> i32;
    pub fn dann_insert_batch_commit_part(a0: *mut c_void, a1: *const DannBuildConfig, a2: *const c_void, a3: u32, a4: *const c_void, a5: u32, a6: u32, a7: *mut c_void, a8: u32, a9: *mut u32) -> i32;
    pub fn dann_apply_neighbor_rows_device(a0: *mut c_void, a1: *const c_void, a2: u32) -> i32;
    pub fn dann_build(a0: *mut c_void, a1: *const DannBuildConfig, a2: u32, a3: u32, a4: f32, a5: u32) -> i32;
    pub fn dann_set_build_options(a0: *mut c_void, a1: u32) -> i32;
    pub fn dann_build_counters(a0: *mut c_void, a1: *mut c_void, a2: u32) -> i32;
    pub fn dann_debug_gram(a0: i32, a1: *mut c_void, a2: u32, a3: u32, a4: *mut c_void) -> i32;
    pub fn dann_save_graph(a0: *mut c_void, a1: *const c_char) -> i32;
    pub fn dann_load_graph(a0: *mut c_void, a1: *const c_char, a2: *mut u32, a3: *mut u64, a4: *mut u64) -> i32;
    pub fn dann_save_vectors_bin(a0: *mut c_void, a1: *const c_char, a2: u32, a3: u32) -> i32;
    pub fn dann_load_vectors_bin(a0: *mut c_void, a1: *const c_char, a2: u32, a3: *mut u32) -> i32;
    pub fn dann_set_pq_table(a0: *mut c_void, a1: *mut c_void, a2: *mut c_void) -> i32;
    pub fn dann_sq8_train(a0: i32, a1: *mut c_void, a2: u64, a3: u32, a4: f64, a5: *mut c_void, a6: *mut c_void, a7: *mut c_void) -> i32;
    pub fn dann_sq8_compress(a0: i32, a1: *mut c_void, a2: u32, a3: u32, a4: *mut c_void, a5: f32, a6: *mut c_void) -> i32;
    pub fn dann_pq_build_lut(a0: i32, a1: i32, a2: *mut c_void, a3: *mut c_void, a4: u32, a5: u32, a6: *mut c_void, a7: u32, a8: *mut c_void) -> i32;
    pub fn dann_pq_compress(a0: i32, a1: *mut c_void, a2: u32, a3: *mut c_void, a4: u32, a5: u32, a6: *mut c_void, a7: u64, a8: *mut c_void) -> i32;
    pub fn dann_pq_lloyds(a0: i32, a1: *mut c_void, a2: u64, a3: u32, a4: *mut c_void, a5: u32, a6: u32, a7: *mut c_void, a8: u32, a9: *mut c_void, a10: *mut c_void) -> i32;
    pub fn dann_pq_scan(a0: i32, a1: *mut c_void, a2: u32, a3: u32, a4: *mut c_void, a5: u64, a6: *mut c_void, a7: *mut c_void, a8: *mut c_void) -> i32;
    pub fn dann_pq_kmeanspp(a0: i32, a1: *mut c_void, a2: u64, a3: u32, a4: *mut c_void, a5: u32, a6: u32, a7: *const DannRng, a8: *mut c_void, a9: *mut c_void) -> i32;
    pub fn dann_pq_train(a0: i32, a1: *mut c_void, a2: u64, a3: u32, a4: *mut c_void, a5: u32, a6: u32, a7: u32, a8: *const DannRng, a9: *mut c_void) -> i32;
    pub fn dann_abi_version() -> i32;
    pub fn dann_debug_stream_read_gbps(a0: i32, a1: u64, a2: u32, a3: *mut f64) -> i32;
    pub fn dann_last_error(a0: *mut c_char, a1: u64) -> i32;
    pub fn dann_kernel_time(a0: *mut c_void, a1: i32, a2: *mut f64, a3: *mut u64) -> i32;
    pub fn dann_kernel_time_reset(a0: *mut c_void) -> i32;
    pub fn dann_set_visited_bits(a0: *mut c_void, a1: u32) -> i32;
    pub fn dann_set_max_concurrency(a0: *mut c_void, a1: u32) -> i32;
}

fn check(status: i32) -> diskann::ANNResult<i32> {
    if status >= 0 { return Ok(status); }
    let mut buf = [0u8; 512];
    unsafe { dann_last_error(buf.as_mut_ptr().cast(), 512) };
    Err(diskann::ANNError::message(String::from_utf8_lossy(&buf).trim_end_matches('\0').to_string()))
}
