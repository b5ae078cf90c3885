"""Host-side mirror of the reference's provider/index surface for the hot path.

Names follow the reference: `Provider` (diskann_inmem::Provider), `Config`
(provider::Config), `Knn` (graph::search::Knn), `Index.search` (DiskANNIndex::search),
`set_element`, `get_neighbors` / `set_neighbors` / `append_vector`
(NeighborAccessor(Mut)), `expand_beam` (ExpandBeam), `distance` (layers::Distance).
All compute happens in libdann_hip.so (HIP kernels); this module only marshals numpy
buffers through the C ABI of include/dann.h.
"""
import ctypes as C

import os

import numpy as np

from . import _ffi
from ._ffi import (BuildConfig, Config, DannError, SearchStats, check, F32, F16, U8, I8, SQ8, PQ, COSINE, INNER_PRODUCT, L2,
                   COSINE_NORMALIZED, IBC_ALL, IBC_NONE)

NP_DTYPE = {F32: np.float32, F16: np.float16, U8: np.uint8, I8: np.int8, SQ8: np.uint8, PQ: np.uint8}
FILTER_INLINE, FILTER_MULTIHOP = 1, 2  # dann.h DANN_FILTER_*
STATS_DTYPE = np.dtype([("cmps", np.uint32), ("hops", np.uint32), ("result_count", np.uint32), ("status", np.uint32),
                        ("written", np.uint32)])


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Knn:
    """graph::search::Knn (diskann/src/graph/search/knn_search.rs:65-118)."""

    def __init__(self, l_value, beam_width=None):
        if l_value == 0:
            raise ValueError("l_value cannot be zero")
        if beam_width == 0:
            raise ValueError("beam width cannot be zero")
        self.l_value = int(l_value)
        self.beam_width = 1 if beam_width is None else int(beam_width)


def build_config(pruned_degree, max_degree, l_build, alpha=1.2, max_occlusion_size=750, max_backedges=None,
                 intra_batch_candidates=IBC_ALL, saturate_after_prune=False):
    """graph::config::Builder with the reference defaults (config/defaults.rs:14-41)."""
    return BuildConfig(pruned_degree, max_degree, l_build, alpha, max_occlusion_size,
                       pruned_degree if max_backedges is None else max_backedges, intra_batch_candidates,
                       int(saturate_after_prune))


class PagedSearch:
    """graph::search::PagedSearch (diskann/src/graph/search/paged.rs) for nq queries."""

    def __init__(self, provider, queries, l_value, list_cap=0):
        q = np.ascontiguousarray(queries, dtype=provider.query_dtype).reshape(-1, provider.query_elems)
        self.nq = q.shape[0]
        self._h = C.c_void_p()
        self._provider = provider  # keeps the index alive
        check(_ffi.lib().dann_paged_begin(provider._h, _p(q), self.nq, int(l_value), int(list_cap), C.byref(self._h)),
              "dann_paged_begin")

    def next_page(self, k):
        """(ids[nq, k], dists[nq, k], counts[nq]); counts == 0 where the search is exhausted"""
        ids = np.empty((self.nq, k), np.uint32)
        dists = np.empty((self.nq, k), np.float32)
        counts = np.zeros(self.nq, np.uint32)
        check(_ffi.lib().dann_paged_next(self._h, int(k), _p(ids), _p(dists), _p(counts)), "dann_paged_next")
        return ids, dists, counts

    def close(self):
        if self._h:
            _ffi.lib().dann_paged_end(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Provider:
    """diskann_inmem::Provider<Full<T>, u32> + DiskANNIndex, resident in one GPU's HBM."""

    def __init__(self, dtype, metric, dim, capacity, max_degree, start_points, row_stride=0, device=-1,
                 sq_scale=0.0, sq_shift_norm_sq=0.0, pq_pivots=None, pq_offsets=None, inline_tags=False):
        self.dtype, self.metric, self.dim = dtype, metric, int(dim)
        self.capacity, self.max_degree = int(capacity), int(max_degree)
        self.row_elems = self.dim + 4 if dtype == SQ8 else self.dim  # SQ-8 rows carry a trailing f32 compensation
        self.query_dtype, self.query_elems = NP_DTYPE[dtype], self.row_elems
        pq_chunks = 0
        if dtype == PQ:  # rows are PQ codes, queries stay full-precision f32
            pq_offsets = np.ascontiguousarray(pq_offsets, dtype=np.uint32)
            pq_chunks = pq_offsets.size - 1
            self.row_elems = pq_chunks
            self.query_dtype, self.query_elems = np.float32, self.dim
        sp = np.ascontiguousarray(start_points, dtype=NP_DTYPE[dtype]).reshape(-1, self.row_elems)
        self.num_start_points = sp.shape[0]
        cfg = Config(dtype, metric, self.dim, self.capacity, self.max_degree, self.num_start_points, row_stride,
                     device, sq_scale, sq_shift_norm_sq, pq_chunks, int(bool(inline_tags)))
        self.inline_tags = bool(inline_tags)
        h = C.c_void_p()
        check(_ffi.lib().dann_index_create(C.byref(cfg), _p(sp), sp.nbytes, C.byref(h)), "dann_index_create")
        self._h = h
        got = Config()
        check(_ffi.lib().dann_index_get_config(self._h, C.byref(got)), "dann_index_get_config")
        self.row_stride, self.device = got.row_stride, got.device
        self.layer_bytes = pq_chunks if dtype == PQ else _ffi.lib().dann_layer_bytes(dtype, self.dim)
        if dtype == PQ:
            piv = np.ascontiguousarray(pq_pivots, dtype=np.float32)
            assert piv.shape == (256, self.dim)
            check(_ffi.lib().dann_set_pq_table(self._h, _p(piv), _p(pq_offsets)), "dann_set_pq_table")
        # developer convenience for A/B scripts (scratch/, profiles/): DANN_<SWITCH>=<value> in the environment of the
        # *Python* process is applied to every Provider it creates; the library itself reads no environment variable
        for name in _ffi.DBG_KEYS:
            env = os.environ.get("DANN_" + name.upper())
            if env not in (None, ""):
                self.debug_set(**{name: float(int(env, 0)) if name.startswith("tune") else float(env)})

    def close(self):
        if getattr(self, "_h", None):
            _ffi.lib().dann_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- SetElement ---------------------------------------------------------
    def set_element(self, slot, vector):
        v = np.ascontiguousarray(vector, dtype=NP_DTYPE[self.dtype])
        check(_ffi.lib().dann_set_element(self._h, slot, _p(v), v.nbytes), "dann_set_element")

    def set_elements(self, first_slot, rows):
        r = np.ascontiguousarray(rows, dtype=NP_DTYPE[self.dtype])
        n = r.shape[0] if r.ndim == 2 else r.size // self.row_elems
        check(_ffi.lib().dann_set_elements(self._h, first_slot, n, _p(r), r.nbytes), "dann_set_elements")

    def get_element(self, slot):
        out = np.empty(self.row_elems, NP_DTYPE[self.dtype])
        check(_ffi.lib().dann_get_element(self._h, slot, _p(out), out.nbytes), "dann_get_element")
        return out

    def upload_store(self, raw_rows):
        raw = np.ascontiguousarray(raw_rows, dtype=np.uint8)
        check(_ffi.lib().dann_upload_store(self._h, _p(raw), raw.shape[1], raw.shape[0]), "dann_upload_store")

    def set_tags(self, first_slot, tags):
        """raw inline tag bytes (tag.rs: 0 AVAILABLE, 1 OWNED, 2 RETIRING, 254 PUBLISHED, 255 FROZEN)"""
        t = np.ascontiguousarray(tags, dtype=np.uint8)
        check(_ffi.lib().dann_set_tags(self._h, first_slot, t.size, _p(t)), "dann_set_tags")

    def get_tags(self, first_slot, n):
        t = np.empty(n, np.uint8)
        check(_ffi.lib().dann_get_tags(self._h, first_slot, n, _p(t)), "dann_get_tags")
        return t

    # -- IdMap / Translate ------------------------------------------------------
    def set_external_ids(self, first_slot, ext_ids):
        e = np.ascontiguousarray(ext_ids, dtype=np.uint64)
        check(_ffi.lib().dann_set_external_ids(self._h, first_slot, e.size, _p(e)), "dann_set_external_ids")

    def to_external(self, slot_ids):
        s = np.ascontiguousarray(slot_ids, dtype=np.uint32)
        out = np.empty(s.shape, np.uint64)
        check(_ffi.lib().dann_to_external(self._h, _p(s), s.size, _p(out)), "dann_to_external")
        return out

    # -- NeighborAccessor(Mut) ------------------------------------------------
    def get_neighbors(self, slot):
        out = np.empty(self.max_degree, np.uint32)
        n = C.c_uint32()
        check(_ffi.lib().dann_get_neighbors(self._h, slot, _p(out), self.max_degree, C.byref(n)), "dann_get_neighbors")
        return out[: n.value].copy()

    def set_neighbors(self, slot, ids):
        a = np.ascontiguousarray(ids, dtype=np.uint32)
        check(_ffi.lib().dann_set_neighbors(self._h, slot, _p(a), a.size), "dann_set_neighbors")

    def append_vector(self, slot, ids):
        a = np.ascontiguousarray(ids, dtype=np.uint32)
        check(_ffi.lib().dann_append_neighbors(self._h, slot, _p(a), a.size), "dann_append_neighbors")

    def upload_graph(self, adj):
        a = np.ascontiguousarray(adj, dtype=np.uint32)
        assert a.shape[1] == self.max_degree + 1
        check(_ffi.lib().dann_upload_graph(self._h, _p(a), a.shape[0]), "dann_upload_graph")

    def download_graph(self):
        a = np.empty((self.capacity + self.num_start_points, self.max_degree + 1), np.uint32)
        check(_ffi.lib().dann_download_graph(self._h, _p(a), a.shape[0]), "dann_download_graph")
        return a

    # -- layers::Distance / QueryDistance / ExpandBeam ------------------------
    def distance(self, x, y):
        x = np.ascontiguousarray(x, dtype=NP_DTYPE[self.dtype])
        y = np.ascontiguousarray(y, dtype=NP_DTYPE[self.dtype])
        out = C.c_float()
        check(_ffi.lib().dann_distance(self._h, _p(x), x.nbytes, _p(y), y.nbytes, C.byref(out)), "dann_distance")
        return out.value

    def distance_pairs(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint32)
        b = np.ascontiguousarray(b, dtype=np.uint32)
        out = np.empty(a.size, np.float32)
        check(_ffi.lib().dann_distance_pairs(self._h, _p(a), _p(b), a.size, _p(out)), "dann_distance_pairs")
        return out

    def query_distance(self, query, row):
        q = np.ascontiguousarray(query, dtype=NP_DTYPE[self.dtype])
        r = np.ascontiguousarray(row, dtype=NP_DTYPE[self.dtype])
        h = C.c_void_p()
        check(_ffi.lib().dann_query_create(self._h, _p(q), q.nbytes, C.byref(h)), "dann_query_create")
        try:
            out = C.c_float()
            check(_ffi.lib().dann_query_distance(h, _p(r), r.nbytes, C.byref(out)), "dann_query_distance")
            return out.value
        finally:
            _ffi.lib().dann_query_destroy(h)

    def expand_beam(self, query, ids):
        q = np.ascontiguousarray(query, dtype=NP_DTYPE[self.dtype])
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        h = C.c_void_p()
        check(_ffi.lib().dann_query_create(self._h, _p(q), q.nbytes, C.byref(h)), "dann_query_create")
        try:
            oi = np.empty(ids.size, np.uint32)
            od = np.empty(ids.size, np.float32)
            n = C.c_uint32()
            check(_ffi.lib().dann_expand_beam(h, _p(ids), ids.size, _p(oi), _p(od), C.byref(n)), "dann_expand_beam")
            return oi[: n.value], od[: n.value]
        finally:
            _ffi.lib().dann_query_destroy(h)

    def expand_beam_batch(self, queries, ids, offsets):
        q = np.ascontiguousarray(queries, dtype=self.query_dtype).reshape(-1, self.query_elems)
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        out = np.empty(ids.size, np.float32)
        check(_ffi.lib().dann_expand_beam_batch(self._h, _p(q), q.shape[0], _p(ids), _p(off), _p(out)),
              "dann_expand_beam_batch")
        return out

    # -- DiskANNIndex::search -------------------------------------------------
    def search(self, params, queries, k=10):
        """nq independent Knn searches; returns (ids[nq,k], dists[nq,k], stats[nq])."""
        q = np.ascontiguousarray(queries, dtype=self.query_dtype).reshape(-1, self.query_elems)
        nq = q.shape[0]
        ids = np.empty((nq, k), np.uint32)
        dists = np.empty((nq, k), np.float32)
        stats = np.zeros(nq, STATS_DTYPE)
        check(_ffi.lib().dann_search_batch(self._h, _p(q), nq, params.l_value, params.beam_width, k, _p(ids),
                                           _p(dists), _p(stats)), "dann_search_batch")
        return ids, dists, stats

    def range_search(self, queries, starting_l, radius, beam_width=1, inner_radius=None, initial_slack=1.0,
                     range_slack=1.0, max_returned=0, out_cap=None):
        """graph::search::Range for a batch; returns (ids, dists, stats, second_round)."""
        q = np.ascontiguousarray(queries, dtype=self.query_dtype).reshape(-1, self.query_elems)
        nq = q.shape[0]
        cap = int(out_cap or max_returned or 1024)
        ids = np.empty((nq, cap), np.uint32)
        dists = np.empty((nq, cap), np.float32)
        stats = np.zeros(nq, STATS_DTYPE)
        second = np.zeros(nq, np.uint32)
        check(_ffi.lib().dann_range_search_batch(self._h, _p(q), nq, starting_l, beam_width, radius,
                                                 int(inner_radius is not None), inner_radius or 0.0, initial_slack,
                                                 range_slack, max_returned, cap, _p(ids), _p(dists), _p(stats),
                                                 _p(second)), "dann_range_search_batch")
        return ids, dists, stats, second

    # -- filtered searches (graph/ext/labeled.rs) -----------------------------------------
    def _filter(self, mode, match, nq, adaptive=None, matched_cap=0):
        """match: bool array over slots [0, capacity + nstart) (shared) or nq x nslots (per query)"""
        nslots = self.capacity + self.num_start_points
        m = np.asarray(match, dtype=bool)
        m2 = m.reshape(1, -1) if m.ndim == 1 else m
        if m2.shape[1] != nslots or m2.shape[0] not in (1, nq):
            raise ValueError("filter must have capacity + start points entries per query")
        words = (nslots + 31) // 32
        padded = np.zeros((m2.shape[0], words * 32), bool)
        padded[:, :nslots] = m2
        bits = np.ascontiguousarray(np.packbits(padded, axis=1, bitorder="little")).view(np.uint32)
        f = _ffi.Filter()
        f.mode = mode
        f.bits = bits.ctypes.data
        f.stride_words = 0 if m2.shape[0] == 1 else words
        f.adaptive_samples, f.adaptive_scale = adaptive if adaptive else (0, 1.0)
        f.matched_cap = matched_cap
        return f, bits  # keep `bits` alive for the duration of the call

    def filtered_search(self, params, queries, k, match, mode=None, adaptive=None, matched_cap=0):
        """InlineFilterSearch (default) or MultihopFilterSearch for a batch; returns (ids, dists, stats)."""
        q = np.ascontiguousarray(queries, dtype=self.query_dtype).reshape(-1, self.query_elems)
        nq = q.shape[0]
        f, keep = self._filter(mode or FILTER_INLINE, match, nq, adaptive, matched_cap)
        ids = np.empty((nq, k), np.uint32)
        dists = np.empty((nq, k), np.float32)
        stats = np.zeros(nq, STATS_DTYPE)
        check(_ffi.lib().dann_filtered_search_batch(self._h, _p(q), nq, params.l_value, params.beam_width, k,
                                                    C.byref(f), _p(ids), _p(dists), _p(stats)),
              "dann_filtered_search_batch")
        del keep
        return ids, dists, stats

    def filtered_range_search(self, queries, starting_l, radius, match, beam_width=1, inner_radius=None,
                              initial_slack=1.0, range_slack=1.0, max_returned=0, out_cap=None, matched_cap=0):
        """graph::search::FilteredRange for a batch; returns (ids, dists, stats, second_round)."""
        q = np.ascontiguousarray(queries, dtype=self.query_dtype).reshape(-1, self.query_elems)
        nq = q.shape[0]
        f, keep = self._filter(FILTER_INLINE, match, nq, None, matched_cap)
        cap = int(out_cap or max_returned or 1024)
        ids = np.empty((nq, cap), np.uint32)
        dists = np.empty((nq, cap), np.float32)
        stats = np.zeros(nq, STATS_DTYPE)
        second = np.zeros(nq, np.uint32)
        check(_ffi.lib().dann_filtered_range_search_batch(self._h, _p(q), nq, starting_l, beam_width, radius,
                                                          int(inner_radius is not None), inner_radius or 0.0,
                                                          initial_slack, range_slack, max_returned, cap, C.byref(f),
                                                          _p(ids), _p(dists), _p(stats), _p(second)),
              "dann_filtered_range_search_batch")
        del keep
        return ids, dists, stats, second

    def paged_search(self, queries, l_value, list_cap=0):
        """DiskANNIndex::paged_search for a batch: returns a PagedSearch session (next_page(k), close())."""
        return PagedSearch(self, queries, l_value, list_cap)

    def rerank(self, queries, cand_ids, k):
        """Rerank post-processor: full-precision distances for the candidates of a quantised search."""
        q = np.ascontiguousarray(queries, dtype=self.query_dtype).reshape(-1, self.query_elems)
        c = np.ascontiguousarray(cand_ids, dtype=np.uint32).reshape(q.shape[0], -1)
        ids = np.empty((q.shape[0], k), np.uint32)
        d = np.empty((q.shape[0], k), np.float32)
        check(_ffi.lib().dann_rerank_batch(self._h, _p(q), q.shape[0], _p(c), c.shape[1], k, _p(ids), _p(d)),
              "dann_rerank_batch")
        return ids, d

    def search_record(self, slots, l_value, rec_stride=None):
        s = np.ascontiguousarray(slots, dtype=np.uint32)
        rec_stride = rec_stride or 4 * (l_value + self.num_start_points) + 64
        rid = np.empty((s.size, rec_stride), np.uint32)
        rd = np.empty((s.size, rec_stride), np.float32)
        rn = np.zeros(s.size, np.uint32)
        stats = np.zeros(s.size, STATS_DTYPE)
        check(_ffi.lib().dann_search_record_batch(self._h, _p(s), s.size, l_value, _p(rid), _p(rd), rec_stride,
                                                  _p(rn), _p(stats)), "dann_search_record_batch")
        return rid, rd, rn, stats

    # -- build -------------------------------------------------------------------
    def insert(self, cfg, slot):
        """DiskANNIndex::insert of the row stored at `slot` (back-edges to the first cfg.max_backedges new neighbours)."""
        check(_ffi.lib().dann_insert(self._h, C.byref(cfg), int(slot)), "dann_insert")

    def insert_batch(self, cfg, slots):
        s = np.ascontiguousarray(slots, dtype=np.uint32)
        check(_ffi.lib().dann_insert_batch(self._h, C.byref(cfg), _p(s), s.size), "dann_insert_batch")

    def insert_batch_candidates(self, cfg, slots, lo, hi, d_pending_out):
        """phase 1 of a multi-GPU multi_insert: `d_pending_out` is a device pointer (int)."""
        s = np.ascontiguousarray(slots, dtype=np.uint32)
        check(_ffi.lib().dann_insert_batch_candidates(self._h, C.byref(cfg), _p(s), s.size, lo, hi,
                                                      C.c_void_p(d_pending_out)), "dann_insert_batch_candidates")

    def insert_batch_commit(self, cfg, slots, d_pending_all):
        """phase 2: `d_pending_all` is a device pointer to the whole batch's pending rows."""
        s = np.ascontiguousarray(slots, dtype=np.uint32)
        check(_ffi.lib().dann_insert_batch_commit(self._h, C.byref(cfg), _p(s), s.size, C.c_void_p(d_pending_all)),
              "dann_insert_batch_commit")

    def insert_batch_commit_part(self, cfg, slots, d_pending_all, rank, world, d_rows_out, rows_cap):
        """phase 2 with the prunes partitioned by target owner (id % world == rank); returns the number of rows this
        rank rewrote and exported to the device buffer `d_rows_out` (rows of max_degree + 2 u32)."""
        s = np.ascontiguousarray(slots, dtype=np.uint32)
        cnt = C.c_uint32(0)
        check(_ffi.lib().dann_insert_batch_commit_part(self._h, C.byref(cfg), _p(s), s.size, C.c_void_p(d_pending_all),
                                                       rank, world, C.c_void_p(d_rows_out), rows_cap, C.byref(cnt)),
              "dann_insert_batch_commit_part")
        return cnt.value

    def apply_neighbor_rows(self, d_rows, count):
        """adjacency rows exported by other replicas' insert_batch_commit_part (device pointer)"""
        check(_ffi.lib().dann_apply_neighbor_rows_device(self._h, C.c_void_p(d_rows), count),
              "dann_apply_neighbor_rows_device")

    def set_build_options(self, flags):
        """DANN_BUILD_* bits (dann.h): BUILD_MFMA_BACKEDGE = back-edge prunes through the Gram / MFMA path"""
        check(_ffi.lib().dann_set_build_options(self._h, int(flags)), "dann_set_build_options")

    def build_counters(self):
        out = np.zeros(11, np.uint64)  # ([10]: tied pools whose Rust-order walk reached the selection's fallback, dann.h)
        check(_ffi.lib().dann_build_counters(self._h, _p(out), 11), "dann_build_counters")
        return out

    def build(self, cfg, first, n, growth=0.02, max_batch=16384):
        return check(_ffi.lib().dann_build(self._h, C.byref(cfg), first, n, growth, max_batch), "dann_build")

    def prune_batch(self, cfg, locs, pool_ids, pool_dists, offsets, force_saturate=False):
        locs = np.ascontiguousarray(locs, dtype=np.uint32)
        pid = np.ascontiguousarray(pool_ids, dtype=np.uint32)
        pd = np.ascontiguousarray(pool_dists, dtype=np.float32)
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        out = np.zeros((locs.size, cfg.pruned_degree + 1), np.uint32)
        check(_ffi.lib().dann_prune_batch(self._h, C.byref(cfg), _p(locs), locs.size, _p(pid), _p(pd), _p(off),
                                          int(force_saturate), _p(out)), "dann_prune_batch")
        return out

    # -- the reference's on-disk formats ----------------------------------------------
    def save_graph(self, path):
        check(_ffi.lib().dann_save_graph(self._h, str(path).encode()), "dann_save_graph")

    def load_graph(self, path):
        start, nstart, npts = C.c_uint32(), C.c_uint64(), C.c_uint64()
        check(_ffi.lib().dann_load_graph(self._h, str(path).encode(), C.byref(start), C.byref(nstart), C.byref(npts)),
              "dann_load_graph")
        return start.value, nstart.value, npts.value

    def save_vectors_bin(self, path, first_slot=0, n=None):
        n = self.capacity - first_slot if n is None else n
        check(_ffi.lib().dann_save_vectors_bin(self._h, str(path).encode(), first_slot, n), "dann_save_vectors_bin")

    def load_vectors_bin(self, path, first_slot=0):
        n = C.c_uint32()
        check(_ffi.lib().dann_load_vectors_bin(self._h, str(path).encode(), first_slot, C.byref(n)),
              "dann_load_vectors_bin")
        return n.value

    # -- diagnostics ---------------------------------------------------------------
    def kernel_time(self, which=0):
        ms, n = C.c_double(), C.c_uint64()
        check(_ffi.lib().dann_kernel_time(self._h, which, C.byref(ms), C.byref(n)), "dann_kernel_time")
        return ms.value, n.value

    def kernel_time_reset(self):
        check(_ffi.lib().dann_kernel_time_reset(self._h), "dann_kernel_time_reset")

    def pq_pack_neighbors(self):
        """opt-in search layout of a PQ index: adjacency + the neighbours' code rows per node (dann_pq_pack_neighbors);
        dropped by any later mutation, never changes a result"""
        check(_ffi.lib().dann_pq_pack_neighbors(self._h), "dann_pq_pack_neighbors")

    def debug_set(self, **switches):
        """development switches of include/dann_debug.h (per index, read on every call, never change a result):
        debug_set(pair_min_queries=1, tune_off=4); None restores a default"""
        for name, value in switches.items():
            v = float("nan") if value is None else float(value)
            check(_ffi.lib().dann_debug_set(self._h, _ffi.DBG_KEYS[name], v), "dann_debug_set")

    def search_families(self):
        """{family: (launches, HIP-event ms)} of the beam-search launches since the last kernel_time_reset()"""
        n = len(_ffi.FAMILIES)
        cnt, ms = (C.c_uint64 * n)(), (C.c_double * n)()
        check(_ffi.lib().dann_debug_search_families(self._h, cnt, ms), "dann_debug_search_families")
        return {f: (int(cnt[i]), float(ms[i])) for i, f in enumerate(_ffi.FAMILIES)}

    def small_call_stats(self):
        """(launches, calls served by them) of the small host-pointer search calls so far: calls of several threads that
        arrive side by side share a launch"""
        out = (C.c_uint64 * 2)()
        check(_ffi.lib().dann_debug_small_call_stats(self._h, out), "dann_debug_small_call_stats")
        return int(out[0]), int(out[1])

    def last_family(self, fn):
        """runs fn() and returns (its result, the set of kernel families that served searches meanwhile)"""
        before = self.search_families()
        out = fn()
        after = self.search_families()
        return out, {f for f in after if after[f][0] > before[f][0]}

    def set_visited_bits(self, bits):
        check(_ffi.lib().dann_set_visited_bits(self._h, bits), "dann_set_visited_bits")

    def set_visited_format(self, entry_bits):
        """0 = automatic, 16 / 32 = width of a visited-table entry (never affects results)"""
        check(_ffi.lib().dann_set_visited_format(self._h, entry_bits), "dann_set_visited_format")

    def set_elements_device(self, first_slot, device_ptr, n, src_stride=0):
        """n rows from device memory on the index's device (device to device; src_stride 0 = packed rows)"""
        check(_ffi.lib().dann_set_elements_device(self._h, int(first_slot), int(n), C.c_void_p(int(device_ptr)),
                                                  int(src_stride) or self.layer_bytes), "dann_set_elements_device")

    def device_pointers(self):
        """(rows, adjacency) device addresses of the index's buffers (zero-copy interop; read-only)"""
        r, a = C.c_void_p(), C.c_void_p()
        check(_ffi.lib().dann_index_device_pointers(self._h, C.byref(r), C.byref(a)), "dann_index_device_pointers")
        return r.value, a.value

    def search_record_queries(self, queries, l_value, rec_stride=None):
        """VisitedSearchRecord of a Knn search (beam 1) per external query: (ids[nq, stride], dists, n[nq], stats)"""
        q = np.ascontiguousarray(queries, dtype=self.query_dtype).reshape(-1, self.query_elems)
        nq = q.shape[0]
        stride = int(rec_stride or 4 * (l_value + self.num_start_points) + 64)
        rid = np.empty((nq, stride), np.uint32)
        rd = np.empty((nq, stride), np.float32)
        rn = np.zeros(nq, np.uint32)
        stats = np.zeros(nq, STATS_DTYPE)
        check(_ffi.lib().dann_search_record_queries(self._h, _p(q), nq, int(l_value), _p(rid), _p(rd), stride, _p(rn),
                                                    _p(stats)), "dann_search_record_queries")
        return rid, rd, rn, stats

    def set_max_concurrency(self, n):
        """Queries in flight per search call (0 = all): n persistent wavefronts share the call's queries."""
        check(_ffi.lib().dann_set_max_concurrency(self._h, n), "dann_set_max_concurrency")

    def set_prune_tie_order(self, order):
        """TIE_RUST (default): equal-distance prune candidates in the order the reference's own sort leaves them in;
        TIE_POSITION: they keep their pool order (same graph on tie-free data; tied pools skip the serial walk)."""
        check(_ffi.lib().dann_set_prune_tie_order(self._h, int(order)), "dann_set_prune_tie_order")

    # -- search server: N callers on one shared index, one query per call, no kernel launch per call ----------
    def server_start(self, l_value, k=10, workers=1024, ring=0, idle_timeout_us=0):
        cfg = _ffi.ServerConfig(int(l_value), int(k), int(workers), int(ring), int(idle_timeout_us))
        check(_ffi.lib().dann_server_start(self._h, C.byref(cfg)), "dann_server_start")
        self._server_k = int(k)

    def server_stop(self):
        check(_ffi.lib().dann_server_stop(self._h), "dann_server_stop")

    def submit(self, query):
        """one query (row of query_dtype) -> ticket"""
        q = np.ascontiguousarray(query, dtype=self.query_dtype).reshape(self.query_elems)
        t = C.c_uint64(0)
        check(_ffi.lib().dann_search_submit(self._h, _p(q), C.byref(t)), "dann_search_submit")
        return t.value

    def poll(self, ticket):
        return check(_ffi.lib().dann_search_poll(self._h, C.c_uint64(ticket)), "dann_search_poll") == 1

    def wait(self, ticket):
        """(ids[k], dists[k], stats) of the query behind `ticket` (every ticket exactly once)"""
        k = self._server_k
        ids = np.empty(k, np.uint32)
        dists = np.empty(k, np.float32)
        st = np.zeros(1, STATS_DTYPE)
        check(_ffi.lib().dann_search_wait(self._h, C.c_uint64(ticket), _p(ids), _p(dists), _p(st)), "dann_search_wait")
        return ids, dists, st[0]

    def server_stats(self):
        sub, rel = C.c_uint64(0), C.c_uint64(0)
        check(_ffi.lib().dann_server_stats(self._h, C.byref(sub), C.byref(rel)), "dann_server_stats")
        return sub.value, rel.value

    def concurrent_callers(self, queries, l_value, k=10, threads=16, mode=0, depth=1):
        """`threads` native host threads issue single-query calls on this index (dann_debug_concurrent_callers):
        mode 0 = dann_search_batch(nq = 1) per call, mode 1 = submit / wait with `depth` tickets outstanding per
        thread.  Returns (ids, dists, latency_us[nq], seconds)."""
        q = np.ascontiguousarray(queries, dtype=self.query_dtype).reshape(-1, self.query_elems)
        nq = q.shape[0]
        ids = np.empty((nq, k), np.uint32)
        dists = np.empty((nq, k), np.float32)
        lat = np.zeros(nq, np.float32)
        secs = C.c_double(0.0)
        check(_ffi.lib().dann_debug_concurrent_callers(self._h, _p(q), nq, int(l_value), int(k), int(threads), int(mode),
                                                       int(depth), _p(ids), _p(dists), _p(lat), C.byref(secs)),
              "dann_debug_concurrent_callers")
        return ids, dists, lat, secs.value


def sq8_compress(x, shift, scale, device=-1):
    """ScalarQuantizer::compress_into (8 bits) on the GPU: rows of dim code bytes + f32 compensation."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    x = x.reshape(-1, x.shape[-1])
    shift = np.ascontiguousarray(shift, dtype=np.float32)
    out = np.empty((x.shape[0], x.shape[1] + 4), np.uint8)
    check(_ffi.lib().dann_sq8_compress(device, _p(x), x.shape[0], x.shape[1], _p(shift), float(scale), _p(out)),
          "dann_sq8_compress")
    return out


def pq_build_lut(metric, pivots, chunk_offsets, queries, device=-1):
    piv = np.ascontiguousarray(pivots, dtype=np.float32)
    off = np.ascontiguousarray(chunk_offsets, dtype=np.uint32)
    q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, piv.shape[1])
    lut = np.empty((q.shape[0], off.size - 1, 256), np.float32)
    check(_ffi.lib().dann_pq_build_lut(device, metric, _p(piv), _p(off), off.size - 1, piv.shape[1], _p(q), q.shape[0],
                                       _p(lut)), "dann_pq_build_lut")
    return lut


def pq_scan(lut, codes, ids, offsets, device=-1):
    lut = np.ascontiguousarray(lut, dtype=np.float32)
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    out = np.empty(ids.size, np.float32)
    check(_ffi.lib().dann_pq_scan(device, _p(lut), lut.shape[0], lut.shape[1], _p(codes), codes.shape[0], _p(ids),
                                  _p(off), _p(out)), "dann_pq_scan")
    return out


def pq_compress(pivots, chunk_offsets, rows, device=-1):
    """TransposedTable::compress_into for a batch: codes[n, nchunks] (u8)."""
    piv = np.ascontiguousarray(pivots, dtype=np.float32)
    off = np.ascontiguousarray(chunk_offsets, dtype=np.uint32)
    x = np.ascontiguousarray(rows, dtype=np.float32).reshape(-1, piv.shape[1])
    codes = np.empty((x.shape[0], off.size - 1), np.uint8)
    check(_ffi.lib().dann_pq_compress(device, _p(piv), piv.shape[0], _p(off), off.size - 1, piv.shape[1], _p(x),
                                      x.shape[0], _p(codes)), "dann_pq_compress")
    return codes


def pq_lloyds(data, chunk_offsets, centers, max_reps, device=-1):
    """Lloyd iterations of the PQ trainer on the GPU; returns (centers, assignments[nchunks, n], residuals)."""
    x = np.ascontiguousarray(data, dtype=np.float32)
    off = np.ascontiguousarray(chunk_offsets, dtype=np.uint32)
    cen = np.ascontiguousarray(centers, dtype=np.float32).copy()
    assign = np.empty((off.size - 1, x.shape[0]), np.uint32)
    res = np.empty(off.size - 1, np.float32)
    check(_ffi.lib().dann_pq_lloyds(device, _p(x), x.shape[0], x.shape[1], _p(off), off.size - 1, cen.shape[0], _p(cen),
                                    max_reps, _p(assign), _p(res)), "dann_pq_lloyds")
    return cen, assign, res


def _rng(uniform_index, uniform_f64):
    return _ffi.Rng(None, _ffi.RNG_INDEX_FN(lambda ctx, c, n: int(uniform_index(c, n))),
                    _ffi.RNG_F64_FN(lambda ctx, c, h: float(uniform_f64(c, h))))


def pq_kmeanspp(data, chunk_offsets, ncenters, uniform_index, uniform_f64, device=-1):
    """k-means++ seeding of the PQ trainer on the GPU with the caller's random draws (uniform_index(chunk, n),
    uniform_f64(chunk, high)); returns (centers[ncenters, dim], selected[nchunks])."""
    x = np.ascontiguousarray(data, dtype=np.float32)
    off = np.ascontiguousarray(chunk_offsets, dtype=np.uint32)
    cen = np.zeros((ncenters, x.shape[1]), np.float32)
    sel = np.zeros(off.size - 1, np.uint32)
    rng = _rng(uniform_index, uniform_f64)
    check(_ffi.lib().dann_pq_kmeanspp(device, _p(x), x.shape[0], x.shape[1], _p(off), off.size - 1, ncenters, C.byref(rng),
                                      _p(cen), _p(sel)), "dann_pq_kmeanspp")
    return cen, sel


def pq_train(data, chunk_offsets, ncenters, lloyds_reps, uniform_index, uniform_f64, device=-1):
    """LightPQTrainingParameters::train on the GPU: k-means++ then the Lloyd iterations; returns pivots[ncenters, dim]."""
    x = np.ascontiguousarray(data, dtype=np.float32)
    off = np.ascontiguousarray(chunk_offsets, dtype=np.uint32)
    piv = np.zeros((ncenters, x.shape[1]), np.float32)
    rng = _rng(uniform_index, uniform_f64)
    check(_ffi.lib().dann_pq_train(device, _p(x), x.shape[0], x.shape[1], _p(off), off.size - 1, ncenters, lloyds_reps,
                                   C.byref(rng), _p(piv)), "dann_pq_train")
    return piv


def pq_rolling_sum_stats(reset=False, device=-1):
    """dann_debug_pq_rolling_sum_stats: {wave_ranges, thread_ranges, walked_ranges, walked_elements} of the trainer's
    rolling f64 sums since the last reset"""
    out = (C.c_uint64 * 4)()
    check(_ffi.lib().dann_debug_pq_rolling_sum_stats(device, out, 1 if reset else 0), "dann_debug_pq_rolling_sum_stats")
    return dict(zip(("wave_ranges", "thread_ranges", "walked_ranges", "walked_elements"), (int(v) for v in out)))


def sq8_train(data, standard_deviations=2.0, device=-1):
    """ScalarQuantizationParameters::train on the GPU: (shift[dim] f32, scale, mean_norm)."""
    x = np.ascontiguousarray(data, dtype=np.float32)
    shift = np.empty(x.shape[1], np.float32)
    scale = np.zeros(1, np.float32)
    mn = np.zeros(1, np.float32)
    check(_ffi.lib().dann_sq8_train(device, _p(x), x.shape[0], x.shape[1], float(standard_deviations), _p(shift),
                                    _p(scale), _p(mn)), "dann_sq8_train")
    return shift, float(scale[0]), float(mn[0])
