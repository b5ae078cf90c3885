"""Edge cases of the GPU search path against the oracle: empty/ragged inputs, extreme parameters,
degrees above one wavefront, isolated nodes, large L (8 queue slots per lane), wide beams."""
import numpy as np
import pytest

import oracle
from helpers import bits, make_pair, rand_vectors, random_graph

pytestmark = pytest.mark.gpu
da = pytest.importorskip("diskann_amd")


def _check(oix, gix, queries, L, W, k):
    oi, od, oc, ost = oix.search_batch(queries, L, W, k)
    gi, gd, gst = gix.search(da.Knn(L, W), queries, k)
    assert np.array_equal(oi, gi), (L, W, k)
    assert np.array_equal(bits(od), bits(gd)), (L, W, k)
    assert np.array_equal(ost[:, 0], gst["cmps"]) and np.array_equal(ost[:, 1], gst["hops"]), (L, W, k)
    assert np.array_equal(oc, gst["written"]) and np.array_equal(ost[:, 2], gst["result_count"])


@pytest.mark.parametrize("R", [1, 3, 63, 64, 65, 100])
def test_degrees_around_the_wave_width(R):
    rng = np.random.default_rng(R)
    n, dim = 1200, 20
    data = rand_vectors(rng, oracle.F32, n, dim)
    adj = random_graph(rng, n, R, min_len=0)
    oix, gix = make_pair(oracle.F32, oracle.L2, data, adj, data[:1], R)
    q = rand_vectors(rng, oracle.F32, 24, dim)
    for L, W in ((8, 1), (40, 1), (40, 3)):
        _check(oix, gix, q, L, W, 10)


def test_large_L_and_wide_beam():
    rng = np.random.default_rng(5)
    n, dim, R = 6000, 16, 24
    data = rand_vectors(rng, oracle.F32, n, dim)
    adj = random_graph(rng, n, R, nstart=2)
    oix, gix = make_pair(oracle.F32, oracle.L2, data, adj, data[:2], R)   # two start points
    q = rand_vectors(rng, oracle.F32, 12, dim)
    for L, W, k in ((510, 1, 100), (500, 16, 500), (129, 7, 1), (64, 16, 64), (600, 1, 10), (1022, 2, 300)):
        _check(oix, gix, q, L, W, k)
    with pytest.raises(da.DannError) as e:
        gix.search(da.Knn(1023), q, 10)         # L + start points > 1024: explicit, not silent
    assert e.value.status == da._ffi.EUNSUPPORTED
    with pytest.raises(da.DannError):
        gix.search(da.Knn(10, 17), q, 10)


def test_isolated_nodes_empty_graph_and_tiny_index():
    rng = np.random.default_rng(6)
    dim = 8
    # graph with no edges at all: only the start point is ever evaluated, no results (it is filtered)
    data = rand_vectors(rng, oracle.F32, 50, dim)
    adj = np.zeros((51, 5), np.uint32)
    oix, gix = make_pair(oracle.F32, oracle.L2, data, adj, data[:1], 4)
    q = rand_vectors(rng, oracle.F32, 5, dim)
    _check(oix, gix, q, 10, 1, 10)
    ids, d, st = gix.search(da.Knn(10), q, 10)
    assert (ids == 0xFFFFFFFF).all() and np.isinf(d).all() and (st["cmps"] == 1).all() and (st["hops"] == 1).all()
    # start point -> one node -> nothing
    adj[50, 0], adj[50, 1] = 1, 7
    oix, gix = make_pair(oracle.F32, oracle.L2, data, adj, data[:1], 4)
    _check(oix, gix, q, 10, 2, 3)
    # capacity 1
    one = rand_vectors(rng, oracle.F32, 1, dim)
    adj1 = np.array([[0, 0, 0], [1, 0, 0]], np.uint32)
    oix, gix = make_pair(oracle.F32, oracle.L2, one, adj1, one, 2)
    _check(oix, gix, q, 4, 1, 2)


def test_out_of_bounds_and_self_loops_in_adjacency():
    """ids beyond the slot range are inserted into the visited set but never evaluated
    (provider.rs:453-454); self loops and edges to start points are legal."""
    rng = np.random.default_rng(7)
    n, dim, R = 800, 12, 8
    data = rand_vectors(rng, oracle.F32, n, dim)
    adj = random_graph(rng, n, R)
    adj[5, 1] = 5                # self loop
    adj[6, 1] = n                # edge to the start point slot
    adj[7, 1] = n + 1000         # out of bounds
    adj[8, 2] = 0x7FFFFFF0       # far out of bounds
    oix, gix = make_pair(oracle.F32, oracle.L2, data, adj, data[:1], R)
    q = rand_vectors(rng, oracle.F32, 64, dim)
    _check(oix, gix, q, 30, 1, 10)
    _check(oix, gix, q, 30, 4, 10)


def test_empty_batches_and_zero_k():
    rng = np.random.default_rng(8)
    n, dim, R = 300, 8, 6
    data = rand_vectors(rng, oracle.F32, n, dim)
    adj = random_graph(rng, n, R)
    oix, gix = make_pair(oracle.F32, oracle.L2, data, adj, data[:1], R)
    ids, d, st = gix.search(da.Knn(10), np.zeros((0, dim), np.float32), 10)
    assert ids.shape == (0, 10)
    out = gix.expand_beam_batch(np.zeros((2, dim), np.float32), np.zeros(0, np.uint32), np.zeros(3, np.uint64))
    assert out.size == 0
    gix.insert_batch(da.build_config(4, 6, 10), np.zeros(0, np.uint32))
    with pytest.raises(ValueError):
        da.Knn(0)
    with pytest.raises(da.DannError) as e:   # l_value == 0 through the raw ABI
        import ctypes as C
        q = np.zeros((1, dim), np.float32)
        o = np.zeros((1, 1), np.uint32)
        f = np.zeros((1, 1), np.float32)
        da._ffi.check(da.lib().dann_search_batch(gix._h, q.ctypes.data, 1, 0, 1, 1, o.ctypes.data, f.ctypes.data, None),
                      "dann_search_batch")
    assert e.value.status == da._ffi.EINVAL


def test_duplicate_vectors_tie_order():
    """Many exactly equal distances: tie order in the queue is decided by insertion order
    (queue.rs:130-171) and must match the CPU path."""
    rng = np.random.default_rng(9)
    n, dim, R = 2000, 8, 16
    base = rng.integers(0, 3, (40, dim)).astype(np.float32)     # 40 distinct points, many duplicates
    data = base[rng.integers(0, 40, n)]
    adj = random_graph(rng, n, R)
    oix, gix = make_pair(oracle.F32, oracle.L2, data, adj, data[:1], R)
    q = base[rng.integers(0, 40, 32)] + 0.0
    for L, W in ((10, 1), (50, 1), (50, 4), (200, 2)):
        _check(oix, gix, q, L, W, 20)


def test_concurrent_searches_on_a_shared_index():
    """N host threads call search on one handle (the reference runs one tokio task per query block on
    a shared &DiskANNIndex, search/api.rs:409-425): results must equal the single-threaded ones."""
    import threading
    rng = np.random.default_rng(10)
    n, dim, R = 3000, 16, 12
    data = rand_vectors(rng, oracle.F32, n, dim)
    adj = random_graph(rng, n, R)
    _, gix = make_pair(oracle.F32, oracle.L2, data, adj, data[:1], R)
    qs = [rand_vectors(rng, oracle.F32, 50 + 7 * t, dim) for t in range(8)]
    want = [gix.search(da.Knn(40), q, 10) for q in qs]
    got = [None] * 8

    def work(t):
        for _ in range(5):
            got[t] = gix.search(da.Knn(40), qs[t], 10)
    th = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    for t in range(8):
        assert np.array_equal(got[t][0], want[t][0]) and np.array_equal(bits(got[t][1]), bits(want[t][1]))
