"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, bit for bit."""
import json
import os

import numpy as np
import pytest

import oracle
from gridutil import grid_data, grid_neighbors, grid_start_point
from helpers import bits, make_pair, rand_vectors, random_graph, teams_on

pytestmark = pytest.mark.gpu

da = pytest.importorskip("diskann_amd")

DTYPES = [oracle.F32, oracle.F16, oracle.U8, oracle.I8]
METRICS = [oracle.L2, oracle.INNER_PRODUCT, oracle.COSINE, oracle.COSINE_NORMALIZED]


def _prov(dtype, metric, dim, n=4):
    return da.Provider(dtype, metric, dim, n, 4, np.zeros((1, dim), oracle.NP_DTYPE[dtype]))


@pytest.mark.parametrize("dtype", DTYPES)
def test_distance_kernels_bit_exact(dtype):
    """layers::Distance (pair) and QueryDistance numerics for every metric; dims cover
    empty main loops, epilogue blocks and partial blocks (full.rs:510-703)."""
    rng = np.random.default_rng(11)
    dims = [1, 2, 7, 8, 9, 15, 16, 17, 31, 32, 33, 40, 63, 64, 65, 100, 127, 128, 129, 160, 256, 384, 771]
    for metric in METRICS:
        for dim in dims:
            p = _prov(dtype, metric, dim)
            x = rand_vectors(rng, dtype, 1, dim)[0]
            y = rand_vectors(rng, dtype, 1, dim)[0]
            got = np.float32(p.distance(x, y))
            want = np.float32(oracle.distance(dtype, metric, x, y))
            assert got.view(np.uint32) == want.view(np.uint32), (dtype, metric, dim, got, want)
            gq = np.float32(p.query_distance(x, y))
            wq = np.float32(oracle.query_distance(dtype, metric, x, y))
            assert gq.view(np.uint32) == wq.view(np.uint32), (dtype, metric, dim, gq, wq)
            p.close()


def test_distance_length_errors():
    """wrong lengths are errors, not panics (full.rs:228-241, 327-335)."""
    p = _prov(oracle.F32, oracle.L2, 16)
    with pytest.raises(da.DannError) as e:
        p.distance(np.zeros(16, np.float32), np.zeros(15, np.float32))
    assert e.value.status == da._ffi.ELENGTH
    with pytest.raises(da.DannError):
        p.query_distance(np.zeros(17, np.float32), np.zeros(16, np.float32))
    with pytest.raises(da.DannError) as e:
        p.set_neighbors(0, np.arange(5))
    assert e.value.status == da._ffi.ETOOLONG
    with pytest.raises(da.DannError) as e:
        p.set_neighbors(99, [1])
    assert e.value.status == da._ffi.EBOUNDS


def test_denormals_and_specials():
    p = _prov(oracle.F32, oracle.L2, 8)
    x = np.full(8, 1e-30, np.float32)
    y = np.zeros(8, np.float32)
    for a, b in ((x, y), (np.full(8, 1e-22, np.float32), y), (np.full(8, 3e38, np.float32), -np.full(8, 3e38, np.float32))):
        got = np.float32(p.distance(a, b))
        want = np.float32(oracle.distance(oracle.F32, oracle.L2, a, b))
        assert got.view(np.uint32) == want.view(np.uint32)


@pytest.mark.parametrize("dtype,metric,dim", [(oracle.F32, oracle.L2, 128), (oracle.F32, oracle.L2, 100),
                                               (oracle.F16, oracle.L2, 128), (oracle.F32, oracle.COSINE, 96),
                                               (oracle.U8, oracle.L2, 128), (oracle.I8, oracle.INNER_PRODUCT, 100),
                                               (oracle.F16, oracle.INNER_PRODUCT, 70)])
def test_expand_beam_and_pairs(dtype, metric, dim):
    rng = np.random.default_rng(5)
    n, R = 3000, 16
    data = rand_vectors(rng, dtype, n, dim)
    adj = random_graph(rng, n, R)
    oix, gix = make_pair(dtype, metric, data, adj, data[:1], R)
    q = rand_vectors(rng, dtype, 1, dim)[0]
    ids = rng.choice(n, 333, replace=False).astype(np.uint32)
    oi, od = oix.expand_beam(q, ids)
    gi, gd = gix.expand_beam(q, ids)
    assert np.array_equal(oi, gi) and np.array_equal(bits(od), bits(gd))
    # batched ragged form, including an empty list
    qs = rand_vectors(rng, dtype, 4, dim)
    lens = [0, 17, 600, 1]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    allids = rng.integers(0, n, int(off[-1])).astype(np.uint32)
    got = gix.expand_beam_batch(qs, allids, off)
    for i in range(4):
        _, want = oix.expand_beam(qs[i], allids[off[i]:off[i + 1]])
        assert np.array_equal(bits(want), bits(got[off[i]:off[i + 1]]))
    a = rng.integers(0, n, 500).astype(np.uint32)
    b = rng.integers(0, n, 500).astype(np.uint32)
    gp = gix.distance_pairs(a, b)
    wp = np.array([oracle.distance(dtype, metric, data[i], data[j]) for i, j in zip(a, b)], np.float32)
    assert np.array_equal(bits(gp), bits(wp))


def test_grid_search_golden_on_gpu(golden_dir):
    """The reference's 18 grid_search golden cases through dann_search_batch."""
    cases = json.load(open(os.path.join(golden_dir, "grid_search.json")))
    for case in cases:
        dims, size = case["grid_dims"], case["grid_size"]
        data = grid_data(dims, size)
        n = data.shape[0]
        R = 2 * dims
        adj = np.zeros((n + 1, R + 1), np.uint32)
        for i, nb in enumerate(grid_neighbors(dims, size)):
            adj[i, 0] = len(nb)
            adj[i, 1:1 + len(nb)] = nb
        adj[n, 0], adj[n, 1] = 1, n - 1
        p = da.Provider(da.F32, da.L2, dims, n, R, grid_start_point(dims, size))
        p.set_elements(0, data)
        p.upload_graph(adj)
        ids, dists, stats = p.search(da.Knn(case["l_value"], case["beam_width"]), np.array(case["query"], np.float32),
                                     case["k"])
        want = case["results"]
        assert [int(i) for i in ids[0][:len(want)]] == [w[0] for w in want], case
        assert [float(d) for d in dists[0][:len(want)]] == [w[1] for w in want], case
        assert int(stats["cmps"][0]) == case["comparisons"] and int(stats["hops"][0]) == case["hops"]
        # the golden's count comes from the test provider's post-processor (== entries written); the inmem2
        # Translate post-processor's count (k - 1 when the buffer fills) is `result_count`
        assert int(stats["written"][0]) == case["num_results"]
        assert int(stats["result_count"][0]) == (case["k"] - 1 if case["num_results"] == case["k"] else case["num_results"])


SEARCH_CASES = [
    (oracle.F32, oracle.L2, 128, 32, 0),
    (oracle.F32, oracle.L2, 128, 32, 544),      # diskann-inmem stride, uploaded verbatim
    (oracle.F32, oracle.L2, 100, 24, 0),
    (oracle.F32, oracle.INNER_PRODUCT, 64, 16, 0),
    (oracle.F32, oracle.COSINE, 48, 16, 0),
    (oracle.F16, oracle.L2, 128, 32, 0),
    (oracle.F16, oracle.COSINE_NORMALIZED, 72, 20, 0),
    (oracle.U8, oracle.L2, 128, 32, 0),
    (oracle.I8, oracle.COSINE, 100, 16, 0),
    (oracle.I8, oracle.L2, 128, 32, 0),          # 128-byte integer rows: query slice and norm in registers
    (oracle.I8, oracle.INNER_PRODUCT, 128, 24, 0),
    (oracle.U8, oracle.COSINE, 128, 16, 0),
]


@pytest.mark.parametrize("dtype,metric,dim,R,stride", SEARCH_CASES)
def test_search_parity_random_graph(dtype, metric, dim, R, stride):
    """ids, distances, cmps and hops equal the oracle's for every query, several L and
    beam widths, on a random graph (tie-heavy for integer rows)."""
    rng = np.random.default_rng(1234 + dim)
    n, nq = 5000, 48
    data = rand_vectors(rng, dtype, n, dim)
    adj = random_graph(rng, n, R)
    oix, gix = make_pair(dtype, metric, data, adj, data[:1], R, row_stride=stride)
    queries = rand_vectors(rng, dtype, nq, dim)
    for L, W, k in ((1, 1, 1), (10, 1, 10), (64, 1, 10), (64, 4, 10), (100, 2, 100), (200, 1, 10), (300, 3, 50)):
        oi, od, oc, ost = oix.search_batch(queries, L, W, k)
        gi, gd, gst = gix.search(da.Knn(L, W), queries, k)
        assert np.array_equal(oi, gi), (L, W)
        assert np.array_equal(bits(od), bits(gd)), (L, W)
        assert np.array_equal(ost[:, 0], gst["cmps"]) and np.array_equal(ost[:, 1], gst["hops"]), (L, W)
        assert np.array_equal(oc, gst["written"]), (L, W)
        assert np.array_equal(ost[:, 2], gst["result_count"]), (L, W)  # Translate's count (provider.rs:933-944)


def test_max_concurrency_does_not_change_results():
    """dann_set_max_concurrency: N persistent waves share the queries of a call through a counter; ids, distances
    and statistics are those of the one-wave-per-query launch (and the oracle's), in record mode too."""
    rng = np.random.default_rng(4242)
    n, dim, R, nq = 6000, 128, 32, 333
    data = rand_vectors(rng, oracle.F32, n, dim)
    adj = random_graph(rng, n, R)
    oix, gix = make_pair(oracle.F32, oracle.L2, data, adj, data[:1], R)
    # two query sets, alternating: the library reuses its staging buffers, and a launch that skipped queries (a stale
    # work counter) would hand back the previous call's rows -- identical if the previous call had the same queries
    qsets = [rand_vectors(rng, oracle.F32, nq, dim) for _ in range(2)]
    refs = [{W: oix.search_batch(q, 40, W, 10) for W in (1, 3)} for q in qsets]
    slots = rng.choice(n, 150, replace=False).astype(np.uint32)
    ref_rec = gix.search_record(slots, 30)
    turn = 0
    for cap in (0, 1, 7, 64, 332, 333, 5000, 64, 7):
        gix.set_max_concurrency(cap)
        for W in (1, 3, 1):
            turn ^= 1
            (gi, gd, gst), fam = gix.last_family(lambda: gix.search(da.Knn(40, W), qsets[turn], 10))
            oi, od, oc, ost = refs[turn][W]
            assert np.array_equal(oi, gi) and np.array_equal(bits(od), bits(gd)), (cap, W)
            assert np.array_equal(ost[:, 0], gst["cmps"]) and np.array_equal(ost[:, 1], gst["hops"]), (cap, W)
            if W == 1:  # (plain mode: persistent waves whenever the cap is below the batch; teams never)
                assert fam == ({"persistent"} if 0 < cap < nq else {"team"} if teams_on() else {"one_wave"}), (fam, cap)
        rid, rd, rn, st = gix.search_record(slots, 30)
        assert np.array_equal(rn, ref_rec[2]), cap
        for i in range(slots.size):  # entries past the record length are unspecified
            assert np.array_equal(rid[i, :rn[i]], ref_rec[0][i, :rn[i]]) and np.array_equal(bits(rd[i, :rn[i]]), bits(ref_rec[1][i, :rn[i]])), cap
    gix.set_max_concurrency(0)


def test_search_multiple_start_points_and_short_lists():
    rng = np.random.default_rng(77)
    n, dim, R = 2000, 32, 8
    data = rand_vectors(rng, oracle.F32, n, dim)
    adj = random_graph(rng, n, R, nstart=3, min_len=0)
    starts = rand_vectors(rng, oracle.F32, 3, dim)
    oix, gix = make_pair(oracle.F32, oracle.L2, data, adj, starts, R)
    queries = rand_vectors(rng, oracle.F32, 32, dim)
    for L, W in ((5, 1), (40, 2)):
        oi, od, oc, ost = oix.search_batch(queries, L, W, 10)
        gi, gd, gst = gix.search(da.Knn(L, W), queries, 10)
        assert np.array_equal(oi, gi) and np.array_equal(bits(od), bits(gd))
        assert np.array_equal(ost[:, 0], gst["cmps"]) and np.array_equal(ost[:, 1], gst["hops"])


def test_search_record_matches_oracle():
    """VisitedSearchRecord of the insert-time search (beam 1, query = stored row)."""
    rng = np.random.default_rng(3)
    n, dim, R = 3000, 64, 16
    data = rand_vectors(rng, oracle.F32, n, dim)
    adj = random_graph(rng, n, R)
    oix, gix = make_pair(oracle.F32, oracle.L2, data, adj, data[:1], R)
    slots = rng.choice(n, 40, replace=False).astype(np.uint32)
    rid, rd, rn, st = gix.search_record(slots, 50)
    for i, s in enumerate(slots):
        _, _, _, ost, orid, ord_ = oix.search(data[s], 50, 1, 10, record=True)
        assert rn[i] == orid.size
        assert np.array_equal(rid[i, :rn[i]], orid) and np.array_equal(bits(rd[i, :rn[i]]), bits(ord_))
        assert st["cmps"][i] == ost[0] and st["hops"][i] == ost[1]


def test_visited_overflow_is_retried():
    """A visited table that is too small is not an error: the overflowed queries are re-run
    with a table twice as large until they fit, and the results still equal the oracle's."""
    rng = np.random.default_rng(9)
    n, dim, R = 4000, 16, 32
    data = rand_vectors(rng, oracle.F32, n, dim)
    adj = random_graph(rng, n, R)
    oix, gix = make_pair(oracle.F32, oracle.L2, data, adj, data[:1], R)
    gix.set_visited_bits(6)
    gi, gd, gst = gix.search(da.Knn(64), data[:40], 10)
    oi, od, oc, ost = oix.search_batch(data[:40], 64, 1, 10)
    assert np.array_equal(gi, oi) and np.array_equal(bits(gd), bits(od))
    assert np.array_equal(ost[:, 0], gst["cmps"]) and not gst["status"].any()


def test_spill_tables_recycled_under_load():
    """20 000 concurrent queries with a 256-entry LDS table: nearly every query continues in a global-memory spill
    table, the 512 tables of the pool are claimed, wiped and handed on dozens of times within the launch (across XCDs).
    Results and counters must equal those of a launch with ample LDS tables, and the oracle's on a sample."""
    rng = np.random.default_rng(19)
    n, dim, R, nq = 20000, 16, 32, 20000
    data = rand_vectors(rng, oracle.F32, n, dim)
    adj = random_graph(rng, n, R)
    oix, gix = make_pair(oracle.F32, oracle.L2, data, adj, data[:1], R)
    queries = rand_vectors(rng, oracle.F32, nq, dim)
    gix.set_visited_bits(0)
    ri, rd, rst = gix.search(da.Knn(48), queries, 10)
    for rep in range(3):
        gix.set_visited_bits(256)
        gi, gd, gst = gix.search(da.Knn(48), queries, 10)
        assert not gst["status"].any()
        assert np.array_equal(gi, ri) and np.array_equal(bits(gd), bits(rd))
        assert np.array_equal(gst["cmps"], rst["cmps"]) and np.array_equal(gst["hops"], rst["hops"])
    oi, od, oc, ost = oix.search_batch(queries[:200], 48, 1, 10)
    assert np.array_equal(ri[:200], oi) and np.array_equal(ost[:, 0], rst["cmps"][:200])
    assert rst["cmps"].mean() > 256  # the forced table really was too small


@pytest.mark.parametrize("dtype,metric", [(oracle.F32, oracle.L2), (oracle.F16, oracle.L2), (oracle.U8, oracle.L2),
                                          (oracle.I8, oracle.INNER_PRODUCT), (oracle.U8, oracle.COSINE)])
def test_team_of_wavefronts_per_query_does_not_change_results(dtype, metric):
    """Latency regime: launches with few queries give every query a team of wavefronts (queue / control / visited
    filter / row gather, talking through an LDS mailbox; the control wave decides the next expansion before the merge,
    the visited wave filters the predicted one after that speculatively -- inserts that are taken back when the prediction
    fails).  ids, distances, cmps and hops equal the oracle's and the one-wave-per-query launch's (debug_set(tune_off=...): bit 4
    switches the teams off, bit 8 the speculation, bit 64 the visited wave's self-start), with several start points and for every queue size the team
    instantiations cover (L + start points <= 256) and beyond; a small explicit visited table makes a team give the query
    back (a team never spills) and the host re-run it with one wave."""
    rng = np.random.default_rng(777)
    n, dim, R, nstart = 6000, 128, 32, 3
    data = rand_vectors(rng, dtype, n, dim)
    adj = random_graph(rng, n, R, nstart=nstart)
    oix, gix = make_pair(dtype, metric, data, adj, data[:nstart], R)
    for nq, vbits in ((1, 0), (7, 0), (64, 0), (33, 8)):
        gix.set_visited_bits(vbits)   # 8: a 256-entry table -- freezes within a few hops
        queries = rand_vectors(rng, dtype, nq, dim)
        for L, k in ((1, 1), (10, 10), (26, 10), (64, 10), (125, 20), (253, 50), (300, 10)):
            oi, od, oc, ost = oix.search_batch(queries, L, 1, k)
            gix.debug_set(tune_off=0)
            (gi, gd, gst), fam = gix.last_family(lambda: gix.search(da.Knn(L, 1), queries, k))  # teams + speculative
            # expansion of the predicted node (a 256-entry table makes some teams give their query back: re-run, one wave)
            teamed = L + nstart <= 256   # beyond 256 queue entries no team instantiation exists: one wave per query
            assert ("team" in fam) == teamed and (len(fam) == 1 or vbits), (fam, nq, L)
            gix.debug_set(tune_off=8)
            (ni, nd, nst), fam = gix.last_family(lambda: gix.search(da.Knn(L, 1), queries, k))  # teams, no speculation
            assert ("team" in fam) == teamed, (fam, nq, L)
            gix.debug_set(tune_off=64)  # teams whose visited wave always waits for the control wave's words (no self-start)
            (wi, wd, wst), fam = gix.last_family(lambda: gix.search(da.Knn(L, 1), queries, k))
            assert ("team" in fam) == teamed, (fam, nq, L)
            gix.debug_set(tune_off=4)
            (si, sd, sst), fam = gix.last_family(lambda: gix.search(da.Knn(L, 1), queries, k))  # one wave per query
            assert fam == {"one_wave"}, (fam, nq, L)
            gix.debug_set(tune_off=None)
            for ids, d, st in ((gi, gd, gst), (ni, nd, nst), (wi, wd, wst), (si, sd, sst)):
                assert np.array_equal(oi, ids), (nq, L)
                assert np.array_equal(bits(od), bits(d)), (nq, L)
                assert np.array_equal(ost[:, 0], st["cmps"]) and np.array_equal(ost[:, 1], st["hops"]), (nq, L)
                assert np.array_equal(oc, st["written"])


def test_team_single_queries_at_tiny_queues_follow_the_oracle():
    """Hundreds of one-query launches at queue sizes where almost every hop takes another path through the control wave
    (no unexpanded entry left, a new candidate overtaking, a contradicted runner-up): the waves of a team only meet at one
    barrier per hop, everything else is mailbox traffic -- a lost ordering shows up here as a different number of
    comparisons long before it changes a result."""
    rng = np.random.default_rng(4242)
    n, dim, R, nstart = 6000, 128, 32, 3
    data = rand_vectors(rng, oracle.F32, n, dim)
    adj = random_graph(rng, n, R, nstart=nstart)
    oix, gix = make_pair(oracle.F32, oracle.L2, data, adj, data[:nstart], R)
    queries = rand_vectors(rng, oracle.F32, 150, dim)
    for L, k in ((1, 1), (2, 1), (5, 3), (40, 10)):
        oi, od, oc, ost = oix.search_batch(queries, L, 1, k)
        for q in range(len(queries)):
            gi, gd, st = gix.search(da.Knn(L, 1), queries[q:q + 1], k)
            assert st["status"][0] == 0
            assert np.array_equal(gi[0], oi[q]) and np.array_equal(bits(gd[0]), bits(od[q])), (L, q)
            assert st["cmps"][0] == ost[q, 0] and st["hops"][0] == ost[q, 1], (L, q, st["cmps"][0], ost[q, 0])


@pytest.mark.parametrize("R,nstart", [(63, 1), (64, 2), (17, 40), (5, 64)])
def test_team_limits_of_degree_and_start_points(R, nstart):
    """The edges of what a team takes: max_degree 63 fills the 64-dword adjacency request exactly (64 falls back to one
    wave per query), 40 and 64 start points make hop 0 a two-pass gather."""
    rng = np.random.default_rng(1000 + R + nstart)
    n, dim = 3000, 128
    data = rand_vectors(rng, oracle.F32, n, dim)
    adj = random_graph(rng, n, R, nstart=nstart)
    oix, gix = make_pair(oracle.F32, oracle.L2, data, adj, data[:nstart], R)
    queries = rand_vectors(rng, oracle.F32, 40, dim)
    for L, k in ((3, 2), (30, 10), (100, 10)):
        oi, od, oc, ost = oix.search_batch(queries, L, 1, k)
        for q0, nq in ((0, 1), (1, 1), (2, 38)):
            gi, gd, st = gix.search(da.Knn(L, 1), queries[q0:q0 + nq], k)
            assert not st["status"].any()
            assert np.array_equal(gi, oi[q0:q0 + nq]) and np.array_equal(bits(gd), bits(od[q0:q0 + nq])), (L, q0)
            assert np.array_equal(st["cmps"], ost[q0:q0 + nq, 0]) and np.array_equal(st["hops"], ost[q0:q0 + nq, 1]), (L, q0)
