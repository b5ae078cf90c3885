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
