"""Shared builders for parity tests: the same bytes go to the oracle and to the HIP library."""
import numpy as np

import oracle


def rand_vectors(rng, dtype, n, dim):
    if dtype == oracle.U8:
        return rng.integers(0, 256, (n, dim), dtype=np.uint8)
    if dtype == oracle.I8:
        return rng.integers(-128, 128, (n, dim), dtype=np.int8)
    return rng.uniform(-1, 1, (n, dim)).astype(oracle.NP_DTYPE[dtype])


def random_graph(rng, n, max_degree, nstart=1, min_len=None):
    """Random adjacency (unique ids per list) in the Neighbors layout [len, ids...]."""
    adj = np.zeros((n + nstart, max_degree + 1), np.uint32)
    lo = max_degree // 2 if min_len is None else min_len
    for i in range(n + nstart):
        ln = int(rng.integers(lo, max_degree + 1))
        ids = rng.choice(n, ln, replace=False).astype(np.uint32)
        adj[i, 0] = ln
        adj[i, 1:1 + ln] = ids
    return adj


def make_pair(dtype, metric, data, adj, start_rows, max_degree, row_stride=0):
    """(oracle.Index, diskann_amd.Provider) over identical rows + adjacency."""
    import diskann_amd as da
    n, dim = data.shape
    oix = oracle.Index(dtype, metric, dim, n, max_degree, start_rows, row_stride=row_stride or None)
    oix.set_rows(0, data)
    oix.adj[:] = adj
    gix = da.Provider(dtype, metric, dim, n, max_degree, start_rows, row_stride=row_stride)
    if row_stride:
        gix.upload_store(oix.rows)
    else:
        gix.set_elements(0, data)
    gix.upload_graph(adj)
    return oix, gix


def teams_on():
    """False when the suite runs with DANN_TUNE_OFF bit 4 (the whole-suite 16-bit-table mode of tests/conftest.py switches
    the teams off through the Python Provider): tests that assert the kernel family of a small launch ask this"""
    import os
    return not (int(os.environ.get("DANN_TUNE_OFF", "0") or 0, 0) & 4)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def small_calls_from_threads(gix, q, L, W, k, want_ids, want_d, nthreads=4, rounds=12):
    """dann_search_batch calls of 1 .. 8 queries from `nthreads` threads at once (they share launches: api.hip,
    small_call); every call must return the rows of want_ids / want_d (the oracle's, for all of q) for its queries"""
    import ctypes as C
    import threading
    import diskann_amd as da
    lib = da._ffi.lib()
    errs = []

    def caller(t):
        try:
            r = np.random.default_rng(500 + t)
            for _ in range(rounds):
                n = int(r.integers(1, 9))
                s0 = int(r.integers(0, len(q) - n + 1))
                qs = np.ascontiguousarray(q[s0:s0 + n])
                hi = np.zeros((n, k), np.uint32)
                hd = np.zeros((n, k), np.float32)
                da._ffi.check(lib.dann_search_batch(gix._h, qs.ctypes.data_as(C.c_void_p), n, L, W, k,
                                                    hi.ctypes.data_as(C.c_void_p), hd.ctypes.data_as(C.c_void_p), None), "batch")
                assert np.array_equal(hi, want_ids[s0:s0 + n]) and np.array_equal(bits(hd), bits(want_d[s0:s0 + n])), (t, s0, n)
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    before = gix.small_call_stats()
    ths = [threading.Thread(target=caller, args=(t,)) for t in range(nthreads)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs[:3]
    launches, calls = (a - b for a, b in zip(gix.small_call_stats(), before))
    assert calls == nthreads * rounds and 0 < launches <= calls
