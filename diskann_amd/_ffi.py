"""ctypes binding of include/dann.h (libdann_hip.so).

The product path has no CPU fallback: if the HIP library is missing or fails to load,
importing this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DANN_LIB_PATH: developer hook for A/B builds of the library (scratch/); the default is the in-tree build
LIB_PATH = os.environ.get("DANN_LIB_PATH") or os.path.join(_HERE, "libdann_hip.so")

F32, F16, U8, I8, SQ8, PQ = 0, 1, 2, 3, 4, 5
COSINE, INNER_PRODUCT, L2, COSINE_NORMALIZED = 0, 1, 2, 3
OK, EINVAL, ELENGTH, EBOUNDS, ETOOLONG, EHIP, ENOMEM, EOVERFLOW, EUNSUPPORTED, EINTERNAL, EBUSY = (
    0, -1, -2, -3, -4, -5, -6, -7, -8, -9, -10)
IBC_NONE, IBC_ALL = 0, 0xFFFFFFFF
TIE_POSITION, TIE_RUST = 0, 1  # dann_set_prune_tie_order
BUILD_MFMA_BACKEDGE, BUILD_MFMA_POOL, BUILD_ROW_KERNEL_ONLY = 1, 2, 4


class Config(C.Structure):
    """dann_config == provider::Config + Full::new (diskann-inmem/src/provider.rs:160-217)."""
    _fields_ = [("dtype", C.c_int32), ("metric", C.c_int32), ("dim", C.c_uint32), ("capacity", C.c_uint32),
                ("max_degree", C.c_uint32), ("num_start_points", C.c_uint32), ("row_stride", C.c_uint32),
                ("device", C.c_int32), ("sq_scale", C.c_float), ("sq_shift_norm_sq", C.c_float),
                ("pq_chunks", C.c_uint32), ("inline_tags", C.c_uint32)]


class BuildConfig(C.Structure):
    """dann_build_config == graph::config::Builder (diskann/src/graph/config/mod.rs:261-338)."""
    _fields_ = [("pruned_degree", C.c_uint32), ("max_degree", C.c_uint32), ("l_build", C.c_uint32),
                ("alpha", C.c_float), ("max_occlusion_size", C.c_uint32), ("max_backedges", C.c_uint32),
                ("intra_batch_candidates", C.c_uint32), ("saturate_after_prune", C.c_uint32)]


class Filter(C.Structure):
    """dann_filter == a QueryLabelProvider (diskann/src/graph/ext/labeled.rs:44-68) as a bitmap over slots."""
    _fields_ = [("mode", C.c_uint32), ("bits", C.c_void_p), ("stride_words", C.c_uint64),
                ("adaptive_samples", C.c_uint32), ("adaptive_scale", C.c_double), ("matched_cap", C.c_uint32)]


RNG_INDEX_FN = C.CFUNCTYPE(C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint64)
RNG_F64_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_uint32, C.c_double)


class Rng(C.Structure):
    """dann_rng: the caller's generator for the two random draws of k-means++ (plusplus.rs:417, 440-444)."""
    _fields_ = [("ctx", C.c_void_p), ("uniform_index", RNG_INDEX_FN), ("uniform_f64", RNG_F64_FN)]


COMM_ALL_GATHER_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)


class CommOps(C.Structure):
    """dann_comm_ops: a caller-supplied all-gather over device buffers (tests: gloo through host memory)."""
    _fields_ = [("ctx", C.c_void_p), ("rank", C.c_uint32), ("world", C.c_uint32), ("all_gather", COMM_ALL_GATHER_FN)]


class ServerConfig(C.Structure):
    """dann_server_config: the resident search server (dann_server_start)."""
    _fields_ = [("l_value", C.c_uint32), ("k", C.c_uint32), ("workers", C.c_uint32), ("ring", C.c_uint32),
                ("idle_timeout_us", C.c_uint32)]


class SearchStats(C.Structure):
    _fields_ = [("cmps", C.c_uint32), ("hops", C.c_uint32), ("result_count", C.c_uint32), ("status", C.c_uint32),
                ("written", C.c_uint32)]


# every symbol include/dann.h declares: name -> (restype, argtypes)
_vp, _u32, _i32, _u64, _f32 = C.c_void_p, C.c_uint32, C.c_int32, C.c_uint64, C.c_float
_P = C.POINTER
SYMBOLS = {
    "dann_layer_bytes": (_i32, [_i32, _u32]),
    "dann_inmem2_row_stride": (_i32, [_i32, _u32]),
    "dann_index_create": (_i32, [_P(Config), _vp, _u64, _P(_vp)]),
    "dann_index_destroy": (_i32, [_vp]),
    "dann_index_max_degree": (_i32, [_vp]),
    "dann_index_get_config": (_i32, [_vp, _P(Config)]),
    "dann_set_element": (_i32, [_vp, _u32, _vp, _u64]),
    "dann_set_elements": (_i32, [_vp, _u32, _u32, _vp, _u64]),
    "dann_get_element": (_i32, [_vp, _u32, _vp, _u64]),
    "dann_set_elements_device": (_i32, [_vp, _u32, _u32, _vp, _u64]),
    "dann_index_device_pointers": (_i32, [_vp, _P(_vp), _P(_vp)]),
    "dann_search_record_queries": (_i32, [_vp, _vp, _u32, _u32, _vp, _vp, _u32, _vp, _vp]),
    "dann_upload_store": (_i32, [_vp, _vp, _u64, _u32]),
    "dann_set_tags": (_i32, [_vp, _u32, _u32, _vp]),
    "dann_get_tags": (_i32, [_vp, _u32, _u32, _vp]),
    "dann_set_external_ids": (_i32, [_vp, _u32, _u32, _vp]),
    "dann_to_external": (_i32, [_vp, _vp, _u64, _vp]),
    "dann_get_neighbors": (_i32, [_vp, _u32, _vp, _u32, _P(_u32)]),
    "dann_set_neighbors": (_i32, [_vp, _u32, _vp, _u32]),
    "dann_append_neighbors": (_i32, [_vp, _u32, _vp, _u32]),
    "dann_set_neighbors_bulk": (_i32, [_vp, _vp, _u32, _vp]),
    "dann_upload_graph": (_i32, [_vp, _vp, _u64]),
    "dann_download_graph": (_i32, [_vp, _vp, _u64]),
    "dann_distance": (_i32, [_vp, _vp, _u64, _vp, _u64, _P(_f32)]),
    "dann_distance_pairs": (_i32, [_vp, _vp, _vp, _u32, _vp]),
    "dann_query_create": (_i32, [_vp, _vp, _u64, _P(_vp)]),
    "dann_query_destroy": (_i32, [_vp]),
    "dann_query_distance": (_i32, [_vp, _vp, _u64, _P(_f32)]),
    "dann_expand_beam": (_i32, [_vp, _vp, _u32, _vp, _vp, _P(_u32)]),
    "dann_expand_beam_batch": (_i32, [_vp, _vp, _u32, _vp, _vp, _vp]),
    "dann_search_batch": (_i32, [_vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp, _vp]),
    "dann_search_batch_device": (_i32, [_vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp, _vp]),
    "dann_range_search_batch": (_i32, [_vp, _vp, _u32, _u32, _u32, _f32, _i32, _f32, _f32, _f32, _u32, _u32, _vp, _vp,
                                       _vp, _vp]),
    "dann_filtered_search_batch": (_i32, [_vp, _vp, _u32, _u32, _u32, _u32, _P(Filter), _vp, _vp, _vp]),
    "dann_filtered_range_search_batch": (_i32, [_vp, _vp, _u32, _u32, _u32, _f32, _i32, _f32, _f32, _f32, _u32, _u32,
                                                _P(Filter), _vp, _vp, _vp, _vp]),
    "dann_paged_begin": (_i32, [_vp, _vp, _u32, _u32, _u32, _P(_vp)]),
    "dann_paged_next": (_i32, [_vp, _u32, _vp, _vp, _vp]),
    "dann_paged_end": (_i32, [_vp]),
    "dann_rerank_batch": (_i32, [_vp, _vp, _u32, _vp, _u32, _u32, _vp, _vp]),
    "dann_rerank_batch_device": (_i32, [_vp, _vp, _u32, _vp, _u32, _u32, _vp, _vp]),
    "dann_search_record_batch": (_i32, [_vp, _vp, _u32, _u32, _vp, _vp, _u32, _vp, _vp]),
    "dann_prune_batch": (_i32, [_vp, _P(BuildConfig), _vp, _u32, _vp, _vp, _vp, _i32, _vp]),
    "dann_insert_batch": (_i32, [_vp, _P(BuildConfig), _vp, _u32]),
    "dann_insert": (_i32, [_vp, _P(BuildConfig), _u32]),
    "dann_insert_batch_candidates": (_i32, [_vp, _P(BuildConfig), _vp, _u32, _u32, _u32, _vp]),
    "dann_insert_batch_commit": (_i32, [_vp, _P(BuildConfig), _vp, _u32, _vp]),
    "dann_insert_batch_commit_part": (_i32, [_vp, _P(BuildConfig), _vp, _u32, _vp, _u32, _u32, _vp, _u32, _P(_u32)]),
    "dann_apply_neighbor_rows_device": (_i32, [_vp, _vp, _u32]),
    "dann_build": (_i32, [_vp, _P(BuildConfig), _u32, _u32, _f32, _u32]),
    "dann_set_build_options": (_i32, [_vp, _u32]),
    "dann_build_counters": (_i32, [_vp, _vp, _u32]),
    "dann_debug_gram_tiles": (_i32, [_i32, _i32, _vp, _u32, _u32, _u32, _vp, _vp]),
    "dann_save_graph": (_i32, [_vp, C.c_char_p]),
    "dann_load_graph": (_i32, [_vp, C.c_char_p, _P(_u32), _P(_u64), _P(_u64)]),
    "dann_save_vectors_bin": (_i32, [_vp, C.c_char_p, _u32, _u32]),
    "dann_load_vectors_bin": (_i32, [_vp, C.c_char_p, _u32, _P(_u32)]),
    "dann_set_pq_table": (_i32, [_vp, _vp, _vp]),
    "dann_pq_pack_neighbors": (_i32, [_vp]),
    "dann_sq8_train": (_i32, [_i32, _vp, _u64, _u32, C.c_double, _vp, _vp, _vp]),
    "dann_sq8_compress": (_i32, [_i32, _vp, _u32, _u32, _vp, _f32, _vp]),
    "dann_pq_build_lut": (_i32, [_i32, _i32, _vp, _vp, _u32, _u32, _vp, _u32, _vp]),
    "dann_pq_compress": (_i32, [_i32, _vp, _u32, _vp, _u32, _u32, _vp, _u64, _vp]),
    "dann_pq_lloyds": (_i32, [_i32, _vp, _u64, _u32, _vp, _u32, _u32, _vp, _u32, _vp, _vp]),
    "dann_pq_scan": (_i32, [_i32, _vp, _u32, _u32, _vp, _u64, _vp, _vp, _vp]),
    "dann_pq_kmeanspp": (_i32, [_i32, _vp, _u64, _u32, _vp, _u32, _u32, _P(Rng), _vp, _vp]),
    "dann_pq_train": (_i32, [_i32, _vp, _u64, _u32, _vp, _u32, _u32, _u32, _P(Rng), _vp]),
    "dann_abi_version": (_i32, []),
    "dann_debug_stream_read_gbps": (_i32, [_i32, _u64, _u32, _P(C.c_double)]),
    "dann_last_error": (_i32, [C.c_char_p, _u64]),
    "dann_kernel_time": (_i32, [_vp, _i32, _P(C.c_double), _P(_u64)]),
    "dann_kernel_time_reset": (_i32, [_vp]),
    "dann_set_visited_bits": (_i32, [_vp, _u32]),
    "dann_set_visited_format": (_i32, [_vp, _u32]),
    "dann_set_max_concurrency": (_i32, [_vp, _u32]),
    "dann_set_prune_tie_order": (_i32, [_vp, _u32]),
    "dann_comm_create_callbacks": (_i32, [_vp, _P(_vp)]),
    "dann_comm_rccl_unique_id": (_i32, [_vp]),
    "dann_comm_create_rccl": (_i32, [_vp, _u32, _u32, _i32, _P(_vp)]),
    "dann_comm_create_local": (_i32, [_vp, _u32, _vp]),
    "dann_comm_destroy": (_i32, [_vp]),
    "dann_comm_rank": (_i32, [_vp]),
    "dann_comm_world": (_i32, [_vp]),
    "dann_comm_all_gather_device": (_i32, [_vp, _i32, _vp, _vp, _u64]),
    "dann_build_sharded": (_i32, [_vp, _vp, _P(BuildConfig), _u32, _u32, _f32, _u32, _vp]),
    "dann_search_sharded": (_i32, [_vp, _vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp]),
    "dann_memcpy_device": (_i32, [_i32, _vp, _vp, _u64, _i32]),
    "dann_multi_create": (_i32, [_P(Config), _vp, _u64, _vp, _u32, _P(_vp)]),
    "dann_multi_destroy": (_i32, [_vp]),
    "dann_multi_size": (_i32, [_vp]),
    "dann_multi_replica": (_vp, [_vp, _u32]),
    "dann_multi_set_elements": (_i32, [_vp, _u32, _u32, _vp, _u64]),
    "dann_multi_build": (_i32, [_vp, _P(BuildConfig), _u32, _u32, _f32, _u32, _vp]),
    "dann_multi_search_batch": (_i32, [_vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp, _vp]),
    "dann_server_start": (_i32, [_vp, _vp]),
    "dann_server_stop": (_i32, [_vp]),
    "dann_search_submit": (_i32, [_vp, _vp, _P(_u64)]),
    "dann_search_poll": (_i32, [_vp, _u64]),
    "dann_search_wait": (_i32, [_vp, _u64, _vp, _vp, _vp]),
    "dann_server_stats": (_i32, [_vp, _P(_u64), _P(_u64)]),
    "dann_debug_concurrent_callers": (_i32, [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp,
                                             _P(C.c_double)]),
    "dann_debug_set": (_i32, [_vp, _i32, C.c_double]),
    "dann_debug_get": (_i32, [_vp, _i32, _P(C.c_double)]),
    "dann_debug_search_families": (_i32, [_vp, _P(_u64), _P(C.c_double)]),
    "dann_debug_small_call_stats": (_i32, [_vp, _P(_u64)]),
    "dann_debug_family_name": (C.c_char_p, [_i32]),
    "dann_debug_pq_rolling_sum_stats": (_i32, [_i32, _P(_u64), _i32]),
}

# dann_debug.h: development switches (dann_debug_set) and kernel families (dann_debug_search_families)
DBG_KEYS = {"tune_off": 0, "tune_on": 1, "pair_min_queries": 2, "team_max_queries": 3, "host_pipeline": 4,
            "sweep_one_by_one": 5, "pool_gram": 6, "gram_cols": 7, "gram_escale": 8, "backedge_gram_rows": 9,
            "server_max_resident_us": 10, "verbose": 11, "ht16_open_eighths": 12, "backedge_single_pool": 13,
            "ht16_max_probes": 14, "host_chunk": 15, "gram_f16_widen": 16, "time_small_launches": 17}
FAMILIES = ("one_wave", "team", "pair", "persistent", "server", "pq_lut")

_lib = None


def lib():
    """Load libdann_hip.so (raises if it is absent: there is no fallback path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m diskann_amd.build` "
                "(or __graft_entry__.build()).  diskann_amd has no CPU fallback.")
        # PyTorch-ROCm wheels bundle their own libamdhip64; two HIP runtimes in one process cannot
        # both own the GPU ("No HIP GPUs are available" for whichever initialises second).  If torch
        # is installed, load it first so that libdann_hip.so binds to the runtime torch uses.
        try:
            import torch  # noqa: F401
        except Exception:  # torch is optional for the library itself
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


class DannError(RuntimeError):
    """ANNError analogue: carries the DANN_E* status and the library's message."""

    def __init__(self, status, where):
        buf = C.create_string_buffer(512)
        lib().dann_last_error(buf, 512)
        self.status = status
        super().__init__(f"{where} failed with status {status}: {buf.value.decode(errors='replace')}")


def check(status, where):
    if status < 0:
        raise DannError(status, where)
    return status
