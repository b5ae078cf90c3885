"""GPU parity of the filtered searches (inline filter incl. AdaptiveL, multihop, filtered range) against
the oracle and the reference's golden cases, through the C ABI."""
import json
import os

import numpy as np
import pytest

import oracle
from filtered_cases import build
from helpers import make_pair, random_graph

pytestmark = pytest.mark.gpu


def _pair(g):
    import diskann_amd as da
    dim = g.data.shape[1]
    ox = oracle.Index(oracle.F32, oracle.L2, dim, g.n, g.max_degree, g.start_vec)
    g.fill(ox)
    px = da.Provider(da.F32, da.L2, dim, g.n, g.max_degree, g.start_vec.reshape(1, -1))
    g.fill(px)
    return ox, px


@pytest.fixture(scope="module")
def cases(golden_dir):
    return json.load(open(os.path.join(golden_dir, "filtered_search.json")))


def test_inline_golden_gpu(cases):
    import diskann_amd as da
    for c in cases["inline"]:
        g = build(c["graph"])
        _, px = _pair(g)
        ids, dists, st = px.filtered_search(da.Knn(c["l"]), np.array(c["query"], np.float32), c["k"],
                                            g.match(c["filter"]),
                                            adaptive=tuple(c["adaptive"]) if c["adaptive"] else None)
        n = int(st["written"][0])
        assert [int(g.orig[i]) for i in ids[0, :n]] == c["result_ids"], c["name"]
        assert [float(d) for d in dists[0, :n]] == c["result_distances"], c["name"]
        assert (int(st["cmps"][0]), int(st["hops"][0])) == (c["comparisons"], c["hops"]), c["name"]


def test_multihop_golden_gpu(cases):
    import diskann_amd as da
    for c in cases["multihop"]:
        g = build(c["graph"], grid_size=c["grid_size"])
        _, px = _pair(g)
        ids, dists, st = px.filtered_search(da.Knn(c["l"]), np.array(c["query"], np.float32), c["k"],
                                            g.match(c["filter"]), mode=da.FILTER_MULTIHOP)
        n = int(st["written"][0])
        assert [[int(g.orig[i]), float(d)] for i, d in zip(ids[0, :n], dists[0, :n])] == c["results"], c["name"]
        assert (int(st["cmps"][0]), int(st["hops"][0])) == (c["comparisons"], c["hops"]), c["name"]


def test_filtered_range_golden_gpu(cases):
    for c in cases["filtered_range"]:
        g = build("grid", c["grid_dims"], c["grid_size"])
        ox, px = _pair(g)
        q = np.array(c["query"], np.float32)
        ids, dists, st, sec = px.filtered_range_search(q, c["starting_l"], c["radius"], g.match(c["filter"]),
                                                       inner_radius=c["inner_radius"],
                                                       max_returned=c["max_returned"], out_cap=256)
        n = int(st["written"][0])
        oi, od, ost = ox.filtered_range_search(q, c["starting_l"], c["radius"], g.match(c["filter"]),
                                               inner_radius=c["inner_radius"], max_returned=c["max_returned"])
        assert np.array_equal(ids[0, :n], oi) and np.array_equal(dists[0, :n], od), c["name"]
        assert (int(st["cmps"][0]), int(st["hops"][0]), bool(sec[0])) == (c["comparisons"], c["hops"], c["second_round"])


@pytest.mark.parametrize("frac", [0.5, 0.1, 0.01])
def test_inline_random_vs_oracle(frac):
    """random graph, random filter of the given selectivity, per-query bitmaps; fixed and adaptive L"""
    import diskann_amd as da
    rng = np.random.default_rng(11)
    n, dim, R, nq = 3000, 24, 16, 40
    data = rng.standard_normal((n, dim)).astype(np.float32)
    adj = random_graph(rng, n, R)
    ox, px = make_pair(oracle.F32, oracle.L2, data, adj, data[:1].copy(), R)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    match = rng.random((nq, n + 1)) < frac
    for adaptive in (None, (50, 8.0)):
        ids, dists, st = px.filtered_search(da.Knn(20), queries, 10, match, adaptive=adaptive)
        for qi in range(nq):
            wn, wi, wd, ws = ox.inline_filter_search(queries[qi], 20, 10, match[qi], adaptive=adaptive)
            assert np.array_equal(ids[qi], wi) and np.array_equal(dists[qi].view(np.uint32), wd.view(np.uint32)), (qi, adaptive)
            assert (int(st["cmps"][qi]), int(st["hops"][qi]), int(st["written"][qi])) == (int(ws[0]), int(ws[1]), wn)


@pytest.mark.parametrize("frac", [0.5, 0.1])
def test_multihop_random_vs_oracle(frac):
    import diskann_amd as da
    rng = np.random.default_rng(12)
    n, dim, R, nq = 3000, 24, 16, 40
    data = rng.standard_normal((n, dim)).astype(np.float32)
    adj = random_graph(rng, n, R)
    ox, px = make_pair(oracle.F32, oracle.L2, data, adj, data[:1].copy(), R)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    match = rng.random(n + 1) < frac  # one shared bitmap
    for W in (1, 2):
        ids, dists, st = px.filtered_search(da.Knn(24, W), queries, 10, match, mode=da.FILTER_MULTIHOP)
        for qi in range(nq):
            wn, wi, wd, ws = ox.multihop_search(queries[qi], 24, 10, match, beam_width=W)
            assert np.array_equal(ids[qi], wi) and np.array_equal(dists[qi].view(np.uint32), wd.view(np.uint32)), (qi, W)
            assert (int(st["cmps"][qi]), int(st["hops"][qi]), int(st["written"][qi])) == (int(ws[0]), int(ws[1]), wn)


def test_filtered_range_random_vs_oracle():
    import diskann_amd as da
    rng = np.random.default_rng(13)
    n, dim, R, nq = 3000, 8, 16, 30
    data = rng.standard_normal((n, dim)).astype(np.float32)
    adj = random_graph(rng, n, R)
    ox, px = make_pair(oracle.F32, oracle.L2, data, adj, data[:1].copy(), R)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    match = rng.random(n + 1) < 0.3
    for radius, kw in ((2.0, {}), (4.0, {"max_returned": 40}), (3.0, {"inner_radius": 1.0, "range_slack": 1.3}),
                       (3.0, {"initial_slack": 0.5, "beam_width": 2})):
        ids, dists, st, sec = px.filtered_range_search(queries, 16, radius, match, out_cap=2048, **kw)
        for qi in range(nq):
            wi, wd, ws = ox.filtered_range_search(queries[qi], 16, radius, match, **kw)
            m = int(st["written"][qi])
            assert m == len(wi), (qi, radius, kw)
            assert np.array_equal(ids[qi, :m], wi) and np.array_equal(dists[qi, :m].view(np.uint32), wd.view(np.uint32))
            assert (int(st["cmps"][qi]), int(st["hops"][qi]), int(sec[qi])) == (int(ws[0]), int(ws[1]), int(ws[3]))


def test_filter_errors():
    import diskann_amd as da
    g = build("hand_1d")
    _, px = _pair(g)
    q = np.array([2.0], np.float32)
    with pytest.raises(da.DannError):  # AdaptiveLSearchError::ScaleFactorLessThanOne
        px.filtered_search(da.Knn(5), q, 3, g.match("even"), adaptive=(5, 0.5))
    with pytest.raises(da.DannError):  # AdaptiveL is an inline-search option
        px.filtered_search(da.Knn(5), q, 3, g.match("even"), mode=da.FILTER_MULTIHOP, adaptive=(5, 2.0))
    with pytest.raises(ValueError):
        px.filtered_search(da.Knn(5), q, 3, np.ones(3, bool))


def test_paged_golden_gpu(golden_dir):
    """the reference's paged_search golden cases, page by page, through dann_paged_*"""
    cases = json.load(open(os.path.join(golden_dir, "paged_search.json")))
    for c in cases:
        g = build("grid", c["grid_dims"], c["grid_size"])
        _, px = _pair(g)
        s = px.paged_search(np.array(c["query"], np.float32), c["search_l"])
        pages = []
        while c["max_pages"] is None or len(pages) < c["max_pages"]:
            ids, dists, counts = s.next_page(c["page_size"])
            n = int(counts[0])
            if n == 0:
                break
            pages.append([[int(i), float(d)] for i, d in zip(ids[0, :n], dists[0, :n])])
        s.close()
        assert pages == c["pages"], c["name"]


@pytest.mark.parametrize("dtype", ["f32", "u8", "f16"])
def test_paged_random_vs_oracle(dtype):
    """a batch of paged sessions on a random graph: every page of every query equals the oracle's, for page sizes
    that do and do not divide L, until exhaustion; k > L and k == 0 are errors"""
    import diskann_amd as da
    from helpers import rand_vectors
    odt = {"f32": oracle.F32, "u8": oracle.U8, "f16": oracle.F16}[dtype]
    rng = np.random.default_rng(17)
    n, dim, R, nq = 1500, 20, 12, 16
    data = rand_vectors(rng, odt, n, dim)
    adj = random_graph(rng, n, R)
    ox, px = make_pair(odt, oracle.L2, data, adj, data[:1].copy(), R)
    queries = rand_vectors(rng, odt, nq, dim)
    for L, k, max_pages in ((24, 7, 40), (16, 16, 12), (40, 1, 60)):
        s = px.paged_search(queries, L)
        want = [ox.paged_search(queries[qi], L, k, max_pages=max_pages) for qi in range(nq)]
        for page in range(max_pages):
            ids, dists, counts = s.next_page(k)
            for qi in range(nq):
                if page < len(want[qi]):
                    wi, wd = want[qi][page]
                    m = int(counts[qi])
                    assert m == len(wi), (qi, page, L, k)
                    assert np.array_equal(ids[qi, :m], wi) and np.array_equal(dists[qi, :m].view(np.uint32), wd.view(np.uint32))
                else:
                    assert counts[qi] == 0
        with pytest.raises(da.DannError):
            s.next_page(L + 1)
        with pytest.raises(da.DannError):
            s.next_page(0)
        s.close()


def test_inline_many_queries_cross_chunks():
    """more queries than one scratch chunk holds (matched list + sort keys are per-launch scratch): the host entry
    walks the batch in chunks and the per-query bitmaps must follow"""
    import diskann_amd as da
    rng = np.random.default_rng(23)
    n, dim, R, nq = 3000, 16, 16, 24000
    data = rng.standard_normal((n, dim)).astype(np.float32)
    adj = random_graph(rng, n, R)
    ox, px = make_pair(oracle.F32, oracle.L2, data, adj, data[:1].copy(), R)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    match = rng.random((nq, n + 1)) < 0.2
    ids, dists, st = px.filtered_search(da.Knn(16), queries, 5, match, matched_cap=3001)
    for qi in list(range(0, nq, 997)) + [nq - 1]:
        wn, wi, wd, ws = ox.inline_filter_search(queries[qi], 16, 5, match[qi])
        assert np.array_equal(ids[qi], wi) and np.array_equal(dists[qi].view(np.uint32), wd.view(np.uint32)), qi
        assert (int(st["cmps"][qi]), int(st["hops"][qi])) == (int(ws[0]), int(ws[1]))


@pytest.mark.parametrize("mode", ["inline", "multihop"])
def test_equal_distances_follow_the_references_unstable_sort(mode):
    """`matched_results.sort_unstable_by(fast_distance)` (inline_filter_search.rs:274) and the multihop search's
    `candidates_two_hop_expansion.sort_unstable_by(..)` + truncate (multihop_filter_search.rs:207-210): byte rows with
    few distinct values give every query dozens of matched entries / rejected candidates at EQUAL distance (far beyond
    the 20 entries Rust's sort handles by insertion), and which of them come first decides the returned ids and -- in the
    multihop search -- which nodes are expanded.  Under the default order (DANN_TIE_RUST) the GPU follows the checker's
    restatement of Rust's sort exactly; under DANN_TIE_POSITION both sides keep push order."""
    import diskann_amd as da
    rng = np.random.default_rng(99)
    n, dim, R, nq = 4000, 6, 32, 64
    data = rng.integers(0, 3, (n, dim)).astype(np.uint8)  # 3^6 distinct rows: distances are small integers
    adj = random_graph(rng, n, R)
    ox, px = make_pair(oracle.U8, oracle.L2, data, adj, data[:1].copy(), R)
    queries = rng.integers(0, 3, (nq, dim)).astype(np.uint8)
    match = rng.random(n + 1) < (0.5 if mode == "inline" else 0.3)
    L, k = 40, 30
    results = {}
    try:
        for order, rule in ((da.TIE_RUST, 6), (da.TIE_POSITION, 0)):
            px.set_prune_tie_order(order)
            oracle.set_tie_rule(rule)
            if mode == "inline":
                ids, dists, st = px.filtered_search(da.Knn(L), queries, k, match, matched_cap=n)
            else:
                ids, dists, st = px.filtered_search(da.Knn(L, 2), queries, k, match, mode=da.FILTER_MULTIHOP)
            tied_lists = 0
            for qi in range(nq):
                if mode == "inline":
                    wn, wi, wd, ws = ox.inline_filter_search(queries[qi], L, k, match)
                else:
                    wn, wi, wd, ws = ox.multihop_search(queries[qi], L, k, match, beam_width=2)
                assert np.array_equal(ids[qi], wi), (order, qi)
                assert np.array_equal(dists[qi].view(np.uint32), wd.view(np.uint32)), (order, qi)
                assert (int(st["cmps"][qi]), int(st["hops"][qi]), int(st["written"][qi])) == (int(ws[0]), int(ws[1]), wn)
                tied_lists += int(len(set(wd[:wn].tolist())) < wn)
            assert tied_lists == nq  # every result list has equal distances in it
            results[order] = ids.copy()
    finally:
        oracle.set_tie_rule()
        px.set_prune_tie_order(da.TIE_RUST)
    # the two orders really are different orders on these lists
    assert not np.array_equal(results[da.TIE_RUST], results[da.TIE_POSITION])
