"""ctypes loader for the CPU oracle (oracle/dann_oracle.cpp).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (diskann_amd/) must never import it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libdann_oracle.so")

F32, F16, U8, I8, SQ8, PQ = 0, 1, 2, 3, 4, 5
COSINE, INNER_PRODUCT, L2, COSINE_NORMALIZED = 0, 1, 2, 3
IBC_NONE, IBC_ALL = 0, 0xFFFFFFFF

NP_DTYPE = {F32: np.float32, F16: np.float16, U8: np.uint8, I8: np.int8, SQ8: np.uint8, PQ: np.uint8}


def build(force=False):
    src = os.path.join(_HERE, "dann_oracle.cpp")
    hdrs = [os.path.join(_HERE, h) for h in ("dann_oracle.h", "rust_unstable_sort.h")]
    stale = (not os.path.exists(_LIB)) or any(
        os.path.getmtime(p) > os.path.getmtime(_LIB) for p in [src] + hdrs if os.path.exists(p)
    )
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B" if force else "all"])
    return _LIB


class OrcIndex(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32),
        ("metric", C.c_int32),
        ("dim", C.c_uint32),
        ("capacity", C.c_uint32),
        ("nstart", C.c_uint32),
        ("max_degree", C.c_uint32),
        ("row_stride", C.c_uint64),
        ("rows", C.c_void_p),
        ("adj", C.c_void_p),
        ("sq_scale", C.c_float),
        ("sq_shift_norm_sq", C.c_float),
        ("pq_pivots", C.c_void_p),
        ("pq_offsets", C.c_void_p),
        ("pq_chunks", C.c_uint32),
        ("tag_offset", C.c_uint32),
    ]


class OrcBuildConfig(C.Structure):
    _fields_ = [
        ("pruned_degree", C.c_uint32),
        ("max_degree", C.c_uint32),
        ("l_build", C.c_uint32),
        ("alpha", C.c_float),
        ("max_occlusion_size", C.c_uint32),
        ("max_backedges", C.c_uint32),
        ("intra_batch_candidates", C.c_uint32),
        ("saturate_after_prune", C.c_uint32),
    ]


RNG_INDEX_FN = C.CFUNCTYPE(C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint64)
RNG_F64_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_uint32, C.c_double)


class OrcRng(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("uniform_index", RNG_INDEX_FN), ("uniform_f64", RNG_F64_FN)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB)
    vp, u32, i32, f32, u64, sz = C.c_void_p, C.c_uint32, C.c_int32, C.c_float, C.c_uint64, C.c_size_t
    P = C.POINTER
    L.orc_f16_to_f32.restype = f32
    L.orc_f16_to_f32.argtypes = [C.c_uint16]
    L.orc_f32_to_f16.restype = C.c_uint16
    L.orc_f32_to_f16.argtypes = [f32]
    for name in ("orc_distance", "orc_query_distance", "orc_distance_scalar_ref", "orc_query_distance_fast"):
        fn = getattr(L, name)
        fn.restype = f32
        fn.argtypes = [i32, i32, vp, vp, sz]
    L.orc_search.restype = i32
    L.orc_search.argtypes = [P(OrcIndex), vp, u32, u32, u32, vp, vp, vp, vp, vp, u32, vp]
    L.orc_search_batch.restype = i32
    L.orc_search_batch.argtypes = [P(OrcIndex), vp, u32, u32, u32, u32, vp, vp, vp, vp, u32, i32, vp]
    f64 = C.c_double
    L.orc_adaptive_l.restype = i32
    L.orc_adaptive_l.argtypes = [u32, u32, u32, f64]
    L.orc_inline_filter_search.restype = i32
    L.orc_inline_filter_search.argtypes = [P(OrcIndex), vp, u32, u32, u32, vp, u32, f64, vp, vp, vp]
    L.orc_multihop_search.restype = i32
    L.orc_multihop_search.argtypes = [P(OrcIndex), vp, u32, u32, u32, vp, vp, vp, vp]
    L.orc_filtered_range_search.restype = i32
    L.orc_filtered_range_search.argtypes = [P(OrcIndex), vp, u32, u32, f32, i32, f32, f32, f32, u64, vp, vp, vp, u64, vp]
    L.orc_paged_begin.restype = vp
    L.orc_paged_begin.argtypes = [P(OrcIndex), vp, u32]
    L.orc_paged_next.restype = i32
    L.orc_paged_next.argtypes = [vp, u32, vp, vp]
    L.orc_paged_end.restype = None
    L.orc_paged_end.argtypes = [vp]
    L.orc_range_search.restype = i32
    L.orc_range_search.argtypes = [P(OrcIndex), vp, u32, u32, f32, i32, f32, f32, f32, u64, vp, vp, u64, vp]
    L.orc_expand_beam.restype = i32
    L.orc_expand_beam.argtypes = [P(OrcIndex), vp, vp, u32, vp, vp]
    L.orc_gram_chain.restype = None
    L.orc_gram_chain.argtypes = [vp, u32, u32, vp]
    L.orc_prune_pool.restype = i32
    L.orc_prune_pool.argtypes = [P(OrcIndex), P(OrcBuildConfig), u32, vp, vp, u32, i32, vp, vp]
    L.orc_bench_distance.restype = f64
    L.orc_bench_distance.argtypes = [i32, i32, u32, u64, u32, i32, u32, u64, vp]
    L.orc_set_tie_rule.restype = None
    L.orc_set_tie_rule.argtypes = [i32, u64]
    L.orc_rust_sort.restype = C.c_int64
    L.orc_rust_sort.argtypes = [i32, vp, vp, u64, u64]
    L.orc_rust_sort_fallbacks.restype = u64
    L.orc_rust_sort_fallbacks.argtypes = []
    L.orc_rust_sort_paths.restype = u32
    L.orc_rust_sort_paths.argtypes = [vp, u32]
    L.orc_insert.restype = i32
    L.orc_insert.argtypes = [P(OrcIndex), P(OrcBuildConfig), u32, vp]
    L.orc_multi_insert.restype = i32
    L.orc_multi_insert.argtypes = [P(OrcIndex), P(OrcBuildConfig), vp, u32, vp]
    L.orc_medoid_f32.restype = C.c_int64
    L.orc_medoid_f32.argtypes = [vp, u64, u32, vp]
    L.orc_pq_build_lut.restype = None
    L.orc_pq_build_lut.argtypes = [i32, vp, vp, vp, u32, u32, vp, vp]
    L.orc_pq_compress.restype = C.c_int64
    L.orc_pq_compress.argtypes = [vp, u32, vp, u32, u32, vp, u64, vp]
    L.orc_pq_square_norms.restype = i32
    L.orc_pq_square_norms.argtypes = [vp, u32, vp, u32, u32, vp]
    L.orc_pq_kmeanspp.restype = i32
    L.orc_pq_kmeanspp.argtypes = [vp, u64, u32, vp, u32, u32, P(OrcRng), vp, vp]
    L.orc_pq_lloyds.restype = i32
    L.orc_pq_lloyds.argtypes = [vp, u64, u32, vp, u32, u32, vp, u32, vp, vp]
    L.orc_sq8_train.restype = None
    L.orc_sq8_train.argtypes = [vp, u64, u32, C.c_double, vp, vp, vp]
    L.orc_pq_lookup.restype = f32
    L.orc_pq_lookup.argtypes = [vp, vp, u32]
    L.orc_sq8_compress.restype = None
    L.orc_sq8_compress.argtypes = [vp, u32, vp, f32, vp, vp]
    L.orc_sq8_distance.restype = f32
    L.orc_sq8_distance.argtypes = [i32, vp, f32, vp, f32, u32, f32, f32]
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def layer_bytes(dtype, dim):
    return int(dim) * np.dtype(NP_DTYPE[dtype]).itemsize + (4 if dtype == SQ8 else 0)


def inmem2_stride(dtype, dim):
    """diskann-inmem row stride: round_up(bytes + 1 tag byte, 32) (store.rs:198-211)."""
    b = layer_bytes(dtype, dim) + 1
    return (b + 31) // 32 * 32


class Index:
    """Host-side arrays in the diskann-inmem layout + the oracle's algorithms over them."""

    def __init__(self, dtype, metric, dim, capacity, max_degree, start_rows, row_stride=None, sq_scale=0.0,
                 sq_shift_norm_sq=0.0, pq_pivots=None, pq_offsets=None, tags=False):
        self.dtype, self.metric, self.dim = dtype, metric, int(dim)
        self.capacity, self.max_degree = int(capacity), int(max_degree)
        self.row_elems = self.dim + 4 if dtype == SQ8 else self.dim
        self.query_dtype = NP_DTYPE[dtype]
        self.query_elems = self.row_elems
        self.pq_pivots = self.pq_offsets = None
        pq_chunks = 0
        if dtype == PQ:
            self.pq_pivots = np.ascontiguousarray(pq_pivots, dtype=np.float32)
            self.pq_offsets = np.ascontiguousarray(pq_offsets, dtype=np.uint32)
            pq_chunks = self.pq_offsets.size - 1
            self.row_elems = pq_chunks
            self.query_dtype, self.query_elems = np.float32, self.dim
        start_rows = np.ascontiguousarray(start_rows, dtype=NP_DTYPE[dtype]).reshape(-1, self.row_elems)
        self.nstart = start_rows.shape[0]
        self.row_bytes = pq_chunks if dtype == PQ else layer_bytes(dtype, dim)
        self.row_stride = int(row_stride) if row_stride else self.row_bytes
        n = self.capacity + self.nstart
        self.rows = np.zeros((n, self.row_stride), dtype=np.uint8)
        self.adj = np.zeros((n, self.max_degree + 1), dtype=np.uint32)
        # inline tags of the diskann-inmem Store (store.rs:133-158): one byte right after the payload; needs the
        # reference stride (payload + 1 rounded up).  Start points are FROZEN (255), other slots AVAILABLE (0)
        # until set_row / set_rows publishes them (254).
        self.tag_offset = 0
        if tags:
            assert self.row_stride > self.row_bytes, "inline tags need row_stride > payload bytes"
            self.tag_offset = self.row_bytes
        for i in range(self.nstart):
            self.set_row(self.capacity + i, start_rows[i])
            if tags:
                self.rows[self.capacity + i, self.tag_offset] = 255
        self._c = OrcIndex(dtype, metric, self.dim, self.capacity, self.nstart, self.max_degree,
                           self.row_stride, self.rows.ctypes.data, self.adj.ctypes.data, sq_scale, sq_shift_norm_sq,
                           self.pq_pivots.ctypes.data if dtype == PQ else None,
                           self.pq_offsets.ctypes.data if dtype == PQ else None, pq_chunks, self.tag_offset)

    # -- storage ------------------------------------------------------------
    def set_row(self, slot, vec):
        vec = np.ascontiguousarray(vec, dtype=NP_DTYPE[self.dtype]).reshape(-1)
        assert vec.size == self.row_elems
        self.rows[slot, : self.row_bytes] = vec.view(np.uint8)
        if self.tag_offset:
            self.rows[slot, self.tag_offset] = 254  # Slot::publish (store.rs:776-782)

    def set_rows(self, first, mat):
        mat = np.ascontiguousarray(mat, dtype=NP_DTYPE[self.dtype]).reshape(-1, self.row_elems)
        self.rows[first: first + mat.shape[0], : self.row_bytes] = mat.view(np.uint8).reshape(mat.shape[0], -1)
        if self.tag_offset:
            self.rows[first: first + mat.shape[0], self.tag_offset] = 254

    def set_tags(self, first, tags):
        """raw tag bytes of slots [first, first + len(tags)) (0 AVAILABLE, 1 OWNED, 2 RETIRING, 254 PUBLISHED, 255 FROZEN)"""
        assert self.tag_offset, "index was created without inline tags"
        t = np.asarray(tags, dtype=np.uint8)
        self.rows[first: first + t.size, self.tag_offset] = t

    def row(self, slot):
        return self.rows[slot, : self.row_bytes].view(NP_DTYPE[self.dtype])

    def set_neighbors(self, slot, ids):
        ids = np.asarray(ids, dtype=np.uint32)
        assert ids.size <= self.max_degree
        self.adj[slot, 0] = ids.size
        self.adj[slot, 1: 1 + ids.size] = ids

    def neighbors(self, slot):
        n = min(int(self.adj[slot, 0]), self.max_degree)
        return self.adj[slot, 1: 1 + n].copy()

    # -- algorithms -----------------------------------------------------------
    def search(self, query, l_value, beam_width=1, k=10, record=False):
        q = np.ascontiguousarray(query, dtype=NP_DTYPE[self.dtype])
        ids = np.empty(k, np.uint32)
        dists = np.empty(k, np.float32)
        stats = np.zeros(3, np.uint32)
        if record:
            cap = 64 * (l_value + self.nstart) + 1024
            rid = np.empty(cap, np.uint32)
            rd = np.empty(cap, np.float32)
            rn = np.zeros(1, np.uint32)
            n = lib().orc_search(C.byref(self._c), _p(q), l_value, beam_width, k, _p(ids), _p(dists), _p(stats),
                                 _p(rid), _p(rd), cap, _p(rn))
            assert rn[0] <= cap
            return n, ids, dists, stats, rid[: rn[0]].copy(), rd[: rn[0]].copy()
        n = lib().orc_search(C.byref(self._c), _p(q), l_value, beam_width, k, _p(ids), _p(dists), _p(stats),
                             None, None, 0, None)
        if n < 0:
            raise RuntimeError(f"orc_search failed: {n}")
        return n, ids, dists, stats

    def search_batch(self, queries, l_value, beam_width=1, k=10, threads=1, fast=False, timing=False):
        q = np.ascontiguousarray(queries, dtype=self.query_dtype).reshape(-1, self.query_elems)
        nq = q.shape[0]
        ids = np.empty((nq, k), np.uint32)
        dists = np.empty((nq, k), np.float32)
        counts = np.zeros(nq, np.uint32)
        stats = np.zeros((nq, 3), np.uint32)
        ns = np.zeros(nq, np.uint64) if timing else None
        rc = lib().orc_search_batch(C.byref(self._c), _p(q), nq, l_value, beam_width, k, _p(ids), _p(dists),
                                    _p(counts), _p(stats), threads, int(fast), _p(ns))
        if rc < 0:
            raise RuntimeError(f"orc_search_batch failed: {rc}")
        return (ids, dists, counts, stats, ns) if timing else (ids, dists, counts, stats)

    def range_search(self, query, starting_l, radius, beam_width=1, inner_radius=None, initial_slack=1.0,
                     range_slack=1.0, max_returned=0, out_cap=None):
        q = np.ascontiguousarray(query, dtype=NP_DTYPE[self.dtype])
        cap = out_cap or (max_returned if max_returned else self.capacity + self.nstart)
        ids = np.empty(cap, np.uint32)
        dists = np.empty(cap, np.float32)
        stats = np.zeros(4, np.uint32)
        n = lib().orc_range_search(C.byref(self._c), _p(q), starting_l, beam_width, radius,
                                   int(inner_radius is not None), inner_radius or 0.0, initial_slack, range_slack,
                                   max_returned, _p(ids), _p(dists), cap, _p(stats))
        if n < 0:
            raise RuntimeError(f"orc_range_search failed: {n}")
        return ids[:n].copy(), dists[:n].copy(), stats

    def filter_bits(self, match):
        """bitmap over slots [0, capacity + nstart) from a boolean array / iterable of matching slot ids"""
        nslots = self.capacity + self.nstart
        m = np.asarray(match)
        if m.dtype != np.bool_:
            b = np.zeros(nslots, bool)
            b[m.astype(np.int64)] = True
            m = b
        assert m.size == nslots
        return np.packbits(m, bitorder="little").view(np.uint8).tobytes().ljust((nslots + 31) // 32 * 4, b"\0")

    def _bits(self, match):
        return np.frombuffer(self.filter_bits(match), np.uint32).copy()

    def inline_filter_search(self, query, l_value, k, match, beam_width=1, adaptive=None):
        q = np.ascontiguousarray(query, dtype=NP_DTYPE[self.dtype])
        bits = self._bits(match)
        ids = np.empty(k, np.uint32)
        dists = np.empty(k, np.float32)
        stats = np.zeros(3, np.uint32)
        samples, scale = adaptive if adaptive else (0, 1.0)
        n = lib().orc_inline_filter_search(C.byref(self._c), _p(q), l_value, beam_width, k, _p(bits), samples, scale,
                                           _p(ids), _p(dists), _p(stats))
        if n < 0:
            raise RuntimeError(f"orc_inline_filter_search failed: {n}")
        return n, ids, dists, stats

    def multihop_search(self, query, l_value, k, match, beam_width=1):
        q = np.ascontiguousarray(query, dtype=NP_DTYPE[self.dtype])
        bits = self._bits(match)
        ids = np.empty(k, np.uint32)
        dists = np.empty(k, np.float32)
        stats = np.zeros(3, np.uint32)
        n = lib().orc_multihop_search(C.byref(self._c), _p(q), l_value, beam_width, k, _p(bits), _p(ids), _p(dists),
                                      _p(stats))
        if n < 0:
            raise RuntimeError(f"orc_multihop_search failed: {n}")
        return n, ids, dists, stats

    def filtered_range_search(self, query, starting_l, radius, match, beam_width=1, inner_radius=None,
                              initial_slack=1.0, range_slack=1.0, max_returned=0, out_cap=None):
        q = np.ascontiguousarray(query, dtype=NP_DTYPE[self.dtype])
        bits = self._bits(match)
        cap = out_cap or (max_returned if max_returned else self.capacity + self.nstart)
        ids = np.empty(cap, np.uint32)
        dists = np.empty(cap, np.float32)
        stats = np.zeros(4, np.uint32)
        n = lib().orc_filtered_range_search(C.byref(self._c), _p(q), starting_l, beam_width, radius,
                                            int(inner_radius is not None), inner_radius or 0.0, initial_slack,
                                            range_slack, max_returned, _p(bits), _p(ids), _p(dists), cap, _p(stats))
        if n < 0:
            raise RuntimeError(f"orc_filtered_range_search failed: {n}")
        return ids[:n].copy(), dists[:n].copy(), stats

    def paged_search(self, query, l_value, page_size, max_pages=None):
        """all pages of index.paged_search(query, l_value) with next_page(page_size) until an empty page"""
        q = np.ascontiguousarray(query, dtype=NP_DTYPE[self.dtype])
        h = lib().orc_paged_begin(C.byref(self._c), _p(q), l_value)
        if not h:
            raise RuntimeError("orc_paged_begin failed")
        pages = []
        try:
            while max_pages is None or len(pages) < max_pages:
                ids = np.empty(page_size, np.uint32)
                dists = np.empty(page_size, np.float32)
                n = lib().orc_paged_next(h, page_size, _p(ids), _p(dists))
                if n < 0:
                    raise RuntimeError(f"orc_paged_next failed: {n}")
                if n == 0:
                    break
                pages.append((ids[:n].copy(), dists[:n].copy()))
        finally:
            lib().orc_paged_end(h)
        return pages

    def expand_beam(self, query, ids):
        q = np.ascontiguousarray(query, dtype=NP_DTYPE[self.dtype])
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        oi = np.empty(ids.size, np.uint32)
        od = np.empty(ids.size, np.float32)
        n = lib().orc_expand_beam(C.byref(self._c), _p(q), _p(ids), ids.size, _p(oi), _p(od))
        if n < 0:
            raise RuntimeError(f"orc_expand_beam failed: {n}")
        return oi[:n], od[:n]

    def prune_pool(self, cfg, location, pool_ids, pool_dists, force_saturate=False):
        pid = np.ascontiguousarray(pool_ids, dtype=np.uint32).copy()
        pd = np.ascontiguousarray(pool_dists, dtype=np.float32).copy()
        out = np.empty(max(cfg.pruned_degree, 1), np.uint32)
        evals = np.zeros(1, np.uint64)
        n = lib().orc_prune_pool(C.byref(self._c), C.byref(cfg), location, _p(pid), _p(pd), pid.size,
                                 int(force_saturate), _p(out), _p(evals))
        if n < 0:
            raise RuntimeError(f"orc_prune_pool failed: {n}")
        return out[:n].copy(), int(evals[0])

    def insert(self, cfg, slot, counters=None):
        """counters: None or five uint64 words (query distances, pair distances, set_neighbors, appends, get_neighbors)"""
        _check_counters(counters)
        rc = lib().orc_insert(C.byref(self._c), C.byref(cfg), slot, _p(counters))
        if rc < 0:
            raise RuntimeError(f"orc_insert failed: {rc}")
        return rc

    def multi_insert(self, cfg, slots, counters=None):
        s = np.ascontiguousarray(slots, dtype=np.uint32)
        _check_counters(counters)
        rc = lib().orc_multi_insert(C.byref(self._c), C.byref(cfg), _p(s), s.size, _p(counters))
        if rc < 0:
            raise RuntimeError(f"orc_multi_insert failed: {rc}")
        return rc


def build_config(pruned_degree, max_degree, l_build, alpha=1.2, max_occlusion_size=750, max_backedges=None,
                 intra_batch_candidates=IBC_ALL, saturate_after_prune=False):
    """graph::config::Builder defaults (diskann/src/graph/config/defaults.rs:14-41)."""
    return OrcBuildConfig(pruned_degree, max_degree, l_build, alpha, max_occlusion_size,
                          pruned_degree if max_backedges is None else max_backedges,
                          intra_batch_candidates, int(saturate_after_prune))


def bench_distance(dtype, metric, dim, nrows, loops, random_order=False, threads=1, seed=1):
    """distances per second of the CPU kernels, diskann-benchmark-simd shape (see dann_oracle.cpp orc_bench_distance)"""
    cs = C.c_double(0.0)
    r = lib().orc_bench_distance(dtype, metric, dim, nrows, loops, int(bool(random_order)), threads, seed, C.byref(cs))
    if r < 0:
        raise ValueError("orc_bench_distance: unsupported dtype / metric")
    return float(r)


def _check_counters(counters):
    if counters is not None and (counters.dtype != np.uint64 or counters.size < 5 or not counters.flags.c_contiguous):
        raise ValueError("counters: five contiguous uint64 words")


DEFAULT_TIE_RULE, POSITION_TIE_RULE = 6, 0


def set_tie_rule(rule=DEFAULT_TIE_RULE, seed=0):
    """order of equal-distance candidates in RobustPrune's sort: 6 (default) = Rust's own order
    (oracle/rust_unstable_sort.h), which reproduces the reference's tie-heavy grid_insert goldens exactly -- the
    product's DANN_TIE_RUST; 0 = pool position, the product's DANN_TIE_POSITION; 1..5 see dann_oracle.cpp sort_pool
    (tie-envelope measurement).  set_tie_rule() restores the default."""
    lib().orc_set_tie_rule(rule, seed)


RUST_SORTED_NEIGHBORS, RUST_SORT_UNSTABLE, RUST_SMALL_SORT, RUST_SELECT_NTH = 0, 1, 2, 3


def rust_sort(mode, ids, dists, max_or_index=0):
    """the restated Rust unstable sort / selection (oracle/rust_unstable_sort.h) over (ids, dists) pairs compared by
    distance alone; returns the reordered (ids, dists) -- truncated to `max_or_index` entries for RUST_SORTED_NEIGHBORS"""
    i = np.ascontiguousarray(ids, dtype=np.uint32).copy()
    d = np.ascontiguousarray(dists, dtype=np.float32).copy()
    n = lib().orc_rust_sort(mode, _p(i), _p(d), i.size, int(max_or_index))
    if n < 0:
        raise ValueError("orc_rust_sort: bad arguments")
    return i[:n], d[:n]


RUST_SORT_PATHS = ("insertion_20", "run_kept", "run_reversed", "quicksort", "small_network", "sort9", "sort13", "merge",
                   "partition_lt", "partition_le", "median3", "median3_rec", "heapsort", "select_max", "select_min",
                   "select_loop", "select_insertion_16", "select_partition_lt", "select_partition_le", "select_fallback")


def rust_sort_paths():
    """how often each part of the restated Rust sort ran since the library was loaded (name -> count)"""
    out = np.zeros(len(RUST_SORT_PATHS), np.uint64)
    n = lib().orc_rust_sort_paths(_p(out), out.size)
    assert n == len(RUST_SORT_PATHS)
    return dict(zip(RUST_SORT_PATHS, (int(x) for x in out)))


def rust_sort_fallbacks():
    return int(lib().orc_rust_sort_fallbacks())


def gram_chain(rows):
    """checker of dann_debug_gram_tiles: one f32 fmaf chain over the whole row per entry"""
    r = np.ascontiguousarray(rows, dtype=np.float32)
    out = np.empty((r.shape[0], r.shape[0]), np.float32)
    lib().orc_gram_chain(_p(r), r.shape[0], r.shape[1], _p(out))
    return out


def distance(dtype, metric, x, y):
    x = np.ascontiguousarray(x, dtype=NP_DTYPE[dtype])
    y = np.ascontiguousarray(y, dtype=NP_DTYPE[dtype])
    return float(lib().orc_distance(dtype, metric, _p(x), _p(y), x.size))


def query_distance(dtype, metric, q, row, fast=False):
    q = np.ascontiguousarray(q, dtype=NP_DTYPE[dtype])
    row = np.ascontiguousarray(row, dtype=NP_DTYPE[dtype])
    fn = lib().orc_query_distance_fast if fast else lib().orc_query_distance
    return float(fn(dtype, metric, _p(q), _p(row), q.size))


def distance_scalar_ref(dtype, metric, x, y):
    x = np.ascontiguousarray(x, dtype=NP_DTYPE[dtype])
    y = np.ascontiguousarray(y, dtype=NP_DTYPE[dtype])
    return float(lib().orc_distance_scalar_ref(dtype, metric, _p(x), _p(y), x.size))


def medoid_f32(data):
    data = np.ascontiguousarray(data, dtype=np.float32)
    mean = np.empty(data.shape[1], np.float32)
    r = lib().orc_medoid_f32(_p(data), data.shape[0], data.shape[1], _p(mean))
    return int(r), mean


def pq_compress(pivots, chunk_offsets, rows):
    """TransposedTable::compress_into for a batch: returns (status, codes[n, nchunks])"""
    piv = np.ascontiguousarray(pivots, dtype=np.float32)
    off = np.ascontiguousarray(chunk_offsets, dtype=np.uint32)
    x = np.ascontiguousarray(rows, dtype=np.float32).reshape(-1, piv.shape[1])
    codes = np.zeros((x.shape[0], off.size - 1), np.uint8)
    rc = lib().orc_pq_compress(_p(piv), piv.shape[0], _p(off), off.size - 1, piv.shape[1], _p(x), x.shape[0], _p(codes))
    return int(rc), codes


def pq_kmeanspp(data, chunk_offsets, ncenters, uniform_index, uniform_f64):
    """kmeans_plusplus_into_inner per chunk; uniform_index(chunk, n) -> int, uniform_f64(chunk, high) -> float are the
    caller's random draws.  Returns (status, centers[ncenters, dim], selected[nchunks])."""
    x = np.ascontiguousarray(data, dtype=np.float32)
    off = np.ascontiguousarray(chunk_offsets, dtype=np.uint32)
    cen = np.zeros((ncenters, x.shape[1]), np.float32)
    sel = np.zeros(off.size - 1, np.uint32)
    rng = OrcRng(None, RNG_INDEX_FN(lambda ctx, c, n: int(uniform_index(c, n))),
                 RNG_F64_FN(lambda ctx, c, h: float(uniform_f64(c, h))))
    rc = lib().orc_pq_kmeanspp(_p(x), x.shape[0], x.shape[1], _p(off), off.size - 1, ncenters, C.byref(rng), _p(cen), _p(sel))
    return int(rc), cen, sel


def pq_lloyds(data, chunk_offsets, centers, max_reps):
    """Lloyd iterations of the PQ trainer for every chunk; returns (centers, assignments[nchunks, n], residuals)"""
    x = np.ascontiguousarray(data, dtype=np.float32)
    off = np.ascontiguousarray(chunk_offsets, dtype=np.uint32)
    cen = np.ascontiguousarray(centers, dtype=np.float32).copy()
    assign = np.zeros((off.size - 1, x.shape[0]), np.uint32)
    res = np.zeros(off.size - 1, np.float32)
    rc = lib().orc_pq_lloyds(_p(x), x.shape[0], x.shape[1], _p(off), off.size - 1, cen.shape[0], _p(cen), max_reps,
                             _p(assign), _p(res))
    if rc < 0:
        raise RuntimeError(f"orc_pq_lloyds failed: {rc}")
    return cen, assign, res


def sq8_train(data, standard_deviations=2.0):
    """ScalarQuantizationParameters::train: returns (shift[dim] f32, scale f32, mean_norm f32)"""
    x = np.ascontiguousarray(data, dtype=np.float32)
    shift = np.empty(x.shape[1], np.float32)
    scale = np.zeros(1, np.float32)
    mn = np.zeros(1, np.float32)
    lib().orc_sq8_train(_p(x), x.shape[0], x.shape[1], float(standard_deviations), _p(shift), _p(scale), _p(mn))
    return shift, scale[0], mn[0]
