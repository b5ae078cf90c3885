"""graph::search::Range on the GPU: the reference's golden cases and random graphs vs the oracle."""
import json
import os

import numpy as np
import pytest

import oracle
from gridutil import grid_data, grid_neighbors, grid_start_point
from helpers import bits, make_pair, rand_vectors, random_graph

pytestmark = pytest.mark.gpu
da = pytest.importorskip("diskann_amd")


def test_range_search_golden_on_gpu(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "range_search.json")))
    for c in cases:
        dims, size = c["grid_dims"], c["grid_size"]
        data = grid_data(dims, size)
        n, R = data.shape[0], 2 * dims
        adj = np.zeros((n + 1, R + 1), np.uint32)
        for i, nb in enumerate(grid_neighbors(dims, size)):
            adj[i, 0] = len(nb)
            adj[i, 1:1 + len(nb)] = nb
        adj[n, 0], adj[n, 1] = 1, n - 1
        p = da.Provider(da.F32, da.L2, dims, n, R, grid_start_point(dims, size))
        p.set_elements(0, data)
        p.upload_graph(adj)
        ids, d, st, sec = p.range_search(np.array(c["query"], np.float32), c["starting_l"], c["radius"],
                                         inner_radius=c["inner_radius"], max_returned=c["max_returned"], out_cap=256)
        k = c["result_count"]
        assert int(st["result_count"][0]) == k, c["name"]
        assert [int(i) for i in ids[0, :k]] == [r[0] for r in c["results"]], c["name"]
        assert [float(x) for x in d[0, :k]] == [r[1] for r in c["results"]], c["name"]
        assert int(st["cmps"][0]) == c["comparisons"] and int(st["hops"][0]) == c["hops"], c["name"]
        assert bool(sec[0]) == c["second_round"], c["name"]


@pytest.mark.parametrize("dtype,metric", [(oracle.F32, oracle.L2), (oracle.F16, oracle.L2), (oracle.U8, oracle.L2)])
def test_range_search_random_graph(dtype, metric):
    rng = np.random.default_rng(50 + dtype)
    n, dim, R = 3000, 16, 12
    data = rand_vectors(rng, dtype, n, dim)
    adj = random_graph(rng, n, R)
    oix, gix = make_pair(dtype, metric, data, adj, data[:1], R)
    queries = rand_vectors(rng, dtype, 24, dim)
    d0 = np.array([oracle.distance(dtype, metric, queries[0], data[i]) for i in range(200)])
    r_small, r_big = float(np.quantile(d0, 0.05)), float(np.quantile(d0, 0.4))
    for L, W, radius, inner, islack, rslack, maxret in (
            (20, 1, r_small, None, 1.0, 1.0, 0),
            (8, 2, r_big, r_small, 0.25, 1.0, 0),
            (8, 1, r_big, None, 0.5, 1.3, 40),
            (16, 3, r_big, None, 0.0, 1.0, 100)):
        cap = 1500
        gi, gd, gst, gsec = gix.range_search(queries, L, radius, W, inner, islack, rslack, maxret, out_cap=cap)
        for q in range(queries.shape[0]):
            oi, od, ost = oix.range_search(queries[q], L, radius, W, inner, islack, rslack, maxret, out_cap=cap)
            k = oi.size
            assert int(gst["result_count"][q]) == k, (L, W, q)
            assert np.array_equal(gi[q, :k], oi) and np.array_equal(bits(gd[q, :k]), bits(od)), (L, W, q)
            assert int(gst["cmps"][q]) == int(ost[0]) and int(gst["hops"][q]) == int(ost[1]), (L, W, q)
            assert int(gsec[q]) == int(ost[3])


def test_range_parameter_errors():
    p = da.Provider(da.F32, da.L2, 4, 10, 4, np.zeros((1, 4), np.float32))
    q = np.zeros((1, 4), np.float32)
    for kw in (dict(starting_l=0, radius=1.0), dict(starting_l=4, radius=1.0, initial_slack=1.5),
               dict(starting_l=4, radius=1.0, range_slack=0.5), dict(starting_l=4, radius=1.0, inner_radius=2.0),
               dict(starting_l=8, radius=1.0, max_returned=4)):
        with pytest.raises(da.DannError) as e:
            p.range_search(q, **kw)
        assert e.value.status == da._ffi.EINVAL


def test_range_searches_of_concurrent_callers_overlap():
    """The reference's callers are N workers on one shared index for every search kind
    (diskann-benchmark-core/src/search/api.rs:399-436).  Range searches take the index shared and run on leased
    contexts: 8 threads finish the same calls in well under the serial time, every result equal to the oracle's."""
    import threading
    import time
    rng = np.random.default_rng(99)
    n, dim, R = 20000, 128, 32
    data = rand_vectors(rng, oracle.F32, n, dim)
    adj = random_graph(rng, n, R)
    oix, gix = make_pair(oracle.F32, oracle.L2, data, adj, data[:1], R)
    nthreads, calls, per_call = 8, 25, 4   # few queries per call, long searches: a call is latency on the device
    queries = rand_vectors(rng, oracle.F32, nthreads * per_call, dim)
    d0 = np.array([oracle.distance(oracle.F32, oracle.L2, queries[0], data[i]) for i in range(300)])
    radius = float(np.quantile(d0, 0.02))
    Lr = 250
    want = [oix.range_search(queries[q], Lr, radius, 1, None, 1.0, 1.0, 0, out_cap=2000) for q in range(len(queries))]

    def work(t, out):
        q = queries[t * per_call:(t + 1) * per_call]
        for _ in range(calls):
            out[t] = gix.range_search(q, Lr, radius, 1, None, 1.0, 1.0, 0, out_cap=2000)
    res = [None] * nthreads
    work(0, res)  # warm: contexts, calibration
    ratios = []
    for attempt in range(3):  # a timing claim on a shared box: the best of three attempts counts
        t0 = time.perf_counter()
        for t in range(nthreads):
            work(t, res)
        serial = time.perf_counter() - t0
        serial_res = list(res)
        res = [None] * nthreads
        th = [threading.Thread(target=work, args=(t, res)) for t in range(nthreads)]
        t0 = time.perf_counter()
        [x.start() for x in th]
        [x.join() for x in th]
        conc = time.perf_counter() - t0
        ratios.append(conc / serial)
        if ratios[-1] < 0.75:
            break
    for got in (serial_res, res):
        for t in range(nthreads):
            gi, gd, gst, gsec = got[t]
            for j in range(per_call):
                oi, od, ost = want[t * per_call + j]
                k = oi.size
                assert int(gst["result_count"][j]) == k
                assert np.array_equal(gi[j, :k], oi) and np.array_equal(bits(gd[j, :k]), bits(od))
                assert int(gst["cmps"][j]) == int(ost[0]) and int(gst["hops"][j]) == int(ost[1])
    # callers that serialised on the index would take the serial time or more; 0.27 .. 0.64 of it measured from box to box
    assert min(ratios) < 0.85, ratios
