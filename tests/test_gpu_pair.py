"""Two queries per wavefront (search_pair_impl.h: the throughput path of 128-byte integer rows with L + start points <= 64
and max_degree <= 64 -- one to three queue entries per lane (L + start points <= 96), one or two adjacency ids per lane).  Everything it returns -- ids, distances, comparisons, hops, written, result_count -- must equal the
oracle's and beam_search_kernel's; odd batch sizes leave the upper half of the last wavefront idle; tiny explicit tables
drive queries through the frozen-table / spill path and, beyond it, through the re-run with one wavefront per query."""
import numpy as np
import pytest

import oracle
from helpers import bits, make_pair, rand_vectors, random_graph, teams_on

pytestmark = pytest.mark.gpu
da = pytest.importorskip("diskann_amd")


@pytest.fixture(autouse=True)
def _pairs_from_the_first_query(monkeypatch):
    """by default only launches beyond the latency regime are paired (20 x compute units queries): every Provider of
    this file pairs from the first query on (dann_debug_set, read by the library on every call)"""
    orig = da.Provider.__init__

    def init(self, *a, **kw):
        orig(self, *a, **kw)
        self.debug_set(pair_min_queries=1)

    monkeypatch.setattr(da.Provider, "__init__", init)


def _check(gix, oix, queries, L, k, tag, family="pair"):
    """the search equals the oracle's -- and was served by the kernel family this file is about"""
    oi, od, oc, ost = oix.search_batch(queries, L, 1, k)
    (gi, gd, gst), fam = gix.last_family(lambda: gix.search(da.Knn(L, 1), queries, k))
    assert fam == {family}, (fam, tag)
    assert not gst["status"].any(), tag
    assert np.array_equal(oi, gi), tag
    assert np.array_equal(bits(od), bits(gd)), tag
    assert np.array_equal(ost[:, 0], gst["cmps"]) and np.array_equal(ost[:, 1], gst["hops"]), tag
    assert np.array_equal(oc, gst["written"]) and np.array_equal(ost[:, 2], gst["result_count"]), tag


CASES = [
    (oracle.U8, oracle.L2, 32, 1),
    (oracle.U8, oracle.COSINE, 32, 1),
    (oracle.U8, oracle.INNER_PRODUCT, 24, 2),
    (oracle.I8, oracle.L2, 32, 1),
    (oracle.I8, oracle.INNER_PRODUCT, 17, 3),
    (oracle.I8, oracle.COSINE, 8, 1),
    # degree beyond 32: two adjacency ids per lane, the hop's candidates in two passes
    (oracle.U8, oracle.L2, 64, 1),
    (oracle.I8, oracle.INNER_PRODUCT, 33, 2),
    (oracle.U8, oracle.COSINE, 47, 32),
]


@pytest.mark.parametrize("dtype,metric,R,nstart", CASES)
def test_pair_kernel_equals_the_oracle(dtype, metric, R, nstart):
    rng = np.random.default_rng(500 + R + nstart)
    n, dim = 6000, 128
    data = rand_vectors(rng, dtype, n, dim)
    adj = random_graph(rng, n, R, nstart=nstart, min_len=0 if R in (8, 47) else None)
    oix, gix = make_pair(dtype, metric, data, adj, data[:nstart], R)
    for nq in (1, 2, 7, 64, 333):   # odd sizes: the last wavefront carries one query
        queries = rand_vectors(rng, dtype, nq, dim)
        # one queue entry per lane up to L + start points = 32, two up to 64, three up to 96 (the kernel's limit)
        for L, k in ((1, 1), (5, 5), (10, 10), (26, 10), (32 - nstart, 10), (20, 40), (33 - nstart, 10), (48, 48),
                     (64 - nstart, 10), (64, 100), (65 - nstart, 10), (80, 20), (96 - nstart, 96)):
            if L >= 1 and L + nstart <= 96:
                _check(gix, oix, queries, L, k, (nq, L, k))


def test_pair_kernel_and_one_wave_per_query_agree_beyond_the_pair_limits():
    """L + start points = 97, degree 65, other row lengths: not the pair kernel's -- same results either way"""
    rng = np.random.default_rng(9)
    n = 4000
    for dtype, dim, R, L in ((oracle.U8, 128, 32, 96), (oracle.U8, 128, 65, 20), (oracle.U8, 100, 32, 20),
                             (oracle.F32, 128, 32, 20)):
        data = rand_vectors(rng, dtype, n, dim)
        adj = random_graph(rng, n, R)
        oix, gix = make_pair(dtype, oracle.L2, data, adj, data[:1], R)
        # (50 queries: the latency regime -- a team per query where a team instantiation exists, i.e. 128-element rows)
        # (a team needs an adjacency row that fits one 64-lane request: degree <= 63)
        _check(gix, oix, rand_vectors(rng, dtype, 50, dim), L, 10, (dtype, dim, R, L),
               family="team" if dim == 128 and R <= 63 and teams_on() else "one_wave")
        gix.debug_set(tune_off=4)
        _check(gix, oix, rand_vectors(rng, dtype, 50, dim), L, 10, (dtype, dim, R, L), family="one_wave")


@pytest.mark.parametrize("R,L", [(32, 30), (64, 60), (40, 90)])
def test_pair_kernel_freezes_spills_and_gives_up_exactly_like_one_wave_per_query(R, L):
    """explicit tables of 64 .. 1024 words per query: frozen after a few hops, continued in the spill pool (20 000
    queries recycle its 512 tables many times), and a query that outgrows even that is re-run with one wave"""
    rng = np.random.default_rng(21)
    n, dim, nq = 20000, 128, 20000
    data = rand_vectors(rng, oracle.U8, n, dim)
    adj = random_graph(rng, n, R)
    oix, gix = make_pair(oracle.U8, oracle.L2, data, adj, data[:1], R)
    queries = rand_vectors(rng, oracle.U8, nq, dim)
    gix.debug_set(tune_off=20)      # one wave per query, no teams
    (ri, rd, rst), fam = gix.last_family(lambda: gix.search(da.Knn(L), queries, 10))
    assert fam == {"one_wave"}, fam
    gix.debug_set(tune_off=None)
    gix.set_visited_format(16)
    for words in (0, 64, 256, 1024):
        gix.set_visited_bits(words)
        for rep in range(2):
            (gi, gd, gst), fam = gix.last_family(lambda: gix.search(da.Knn(L), queries, 10))
            # (tiny tables: queries that outgrow table + spill pool are re-run with one wavefront per query)
            assert "pair" in fam and fam <= {"pair", "one_wave"}, (fam, words)
            assert not gst["status"].any(), words
            assert np.array_equal(gi, ri) and np.array_equal(bits(gd), bits(rd)), words
            assert np.array_equal(gst["cmps"], rst["cmps"]) and np.array_equal(gst["hops"], rst["hops"]), words
    gix.set_visited_bits(0)
    gix.set_visited_format(0)
    oi, od, oc, ost = oix.search_batch(queries[:300], L, 1, 10)
    assert np.array_equal(ri[:300], oi) and np.array_equal(ost[:, 0], rst["cmps"][:300])
    assert rst["cmps"].mean() > 300


@pytest.mark.parametrize("metric,stride", [(oracle.L2, 0), (oracle.L2, 256), (oracle.INNER_PRODUCT, 0),
                                           (oracle.COSINE_NORMALIZED, 256)])
def test_pair_kernel_sq8_rows(metric, stride):
    """SQ-8 codes (132-byte rows at a 144- or 256-byte stride, compensated epilogues) through the pair kernel"""
    from test_gpu_quant import _sq_setup
    rng = np.random.default_rng(77 + metric)
    n, dim, R = 5000, 128, 32
    data, shift, scale = _sq_setup(rng, n, dim)
    codes = da.sq8_compress(data, shift, scale)
    snorm = float(np.float32((shift.astype(np.float32) ** 2).sum(dtype=np.float32)))
    adj = random_graph(rng, n, R)
    oix = oracle.Index(oracle.SQ8, metric, dim, n, R, codes[:1], sq_scale=scale, sq_shift_norm_sq=snorm)
    oix.set_rows(0, codes)
    oix.adj[:] = adj
    gix = da.Provider(da.SQ8, metric, dim, n, R, codes[:1], sq_scale=scale, sq_shift_norm_sq=snorm, row_stride=stride)
    gix.set_elements(0, codes)
    gix.upload_graph(adj)
    for nq in (3, 40, 257):
        queries = da.sq8_compress(rng.normal(0.3, 0.5, (nq, dim)).astype(np.float32), shift, scale)
        for L, k in ((8, 5), (26, 10), (31, 10), (40, 10), (63, 63), (64, 10), (95, 10)):
            _check(gix, oix, queries, L, k, (nq, L, k))


def test_default_threshold_pairs_exactly_from_twenty_queries_per_compute_unit():
    """without the test switch: launches of 20 x CUs queries and more are paired, one query fewer is not -- and both
    sides of the threshold return the oracle's results (the last wavefront of the odd batch carries one query)"""
    import torch
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    floor = 20 * cus
    rng = np.random.default_rng(1234)
    n, dim, R = 8000, 128, 32
    data = rand_vectors(rng, oracle.I8, n, dim)
    adj = random_graph(rng, n, R)
    oix, gix = make_pair(oracle.I8, oracle.L2, data, adj, data[:1], R)
    gix.debug_set(pair_min_queries=None)
    queries = rand_vectors(rng, oracle.I8, floor + 1, dim)
    for nq, family in ((floor - 1, "one_wave"), (floor, "pair"), (floor + 1, "pair")):
        _check(gix, oix, queries[:nq], 26, 10, nq, family=family)
    _check(gix, oix, queries, 63, 10, "L = 63", family="pair")   # two queue entries per lane at the default threshold
    _check(gix, oix, queries, 64, 10, "L = 64", family="pair")   # three (SURVEY 8(a)'s C-int8 sizing: L = 64 + the start point)


def test_max_concurrency_cap_takes_pair_launches_to_persistent_waves():
    """dann_set_max_concurrency: a capped call is `cap` persistent wavefronts over the batch; pair_search_kernel launches
    nq / 2 blocks and never reads the cap, so the capped call is served by the persistent family -- same results"""
    rng = np.random.default_rng(78)
    n, dim, R = 5000, 128, 32
    data = rand_vectors(rng, oracle.U8, n, dim)
    adj = random_graph(rng, n, R)
    oix, gix = make_pair(oracle.U8, oracle.L2, data, adj, data[:1], R)
    q = rand_vectors(rng, oracle.U8, 301, dim)
    _check(gix, oix, q, 26, 10, "no cap")
    gix.set_max_concurrency(32)
    _check(gix, oix, q, 26, 10, "capped: 301 queries over 32 persistent waves", family="persistent")
    _check(gix, oix, q[:32], 26, 10, "at the cap")
    gix.set_max_concurrency(0)
    _check(gix, oix, q, 26, 10, "cap lifted")


@pytest.mark.parametrize("probes,R,L", [(1, 32, 30), (2, 64, 60), (3, 32, 90)])
def test_pair_kernel_overflow_table_of_few_probes(probes, R, L):
    """An index of 2^21 slots and more leaves a 16-bit table entry three probes per id; ids that find them all taken
    go to the query's overflow table (ov_insert) instead of freezing the table.  Small indexes reach that path through
    the probe cap (DANN_DBG_HT16_MAX_PROBES): automatic and explicit table sizes -- overflow while the table is open,
    lookups of overflowed ids once it is frozen, spill tables, re-runs -- all equal to one wave per query."""
    rng = np.random.default_rng(210 + probes)
    n, dim, nq = 20000, 128, 6000
    data = rand_vectors(rng, oracle.U8, n, dim)
    adj = random_graph(rng, n, R, nstart=2)
    oix, gix = make_pair(oracle.U8, oracle.L2, data, adj, data[:2], R)
    queries = rand_vectors(rng, oracle.U8, nq, dim)
    gix.debug_set(tune_off=20)      # one wave per query, no teams
    (ri, rd, rst), fam = gix.last_family(lambda: gix.search(da.Knn(L), queries, 10))
    assert fam == {"one_wave"}, fam
    gix.debug_set(tune_off=None, ht16_max_probes=probes)
    gix.set_visited_format(16)
    for words in (0, 128, 512, 2048):
        gix.set_visited_bits(words)
        for eighths in (6, 7):
            gix.debug_set(ht16_open_eighths=eighths)
            (gi, gd, gst), fam = gix.last_family(lambda: gix.search(da.Knn(L), queries, 10))
            assert "pair" in fam and fam <= {"pair", "one_wave"}, (fam, words)
            assert not gst["status"].any(), words
            assert np.array_equal(gi, ri) and np.array_equal(bits(gd), bits(rd)), (words, eighths)
            assert np.array_equal(gst["cmps"], rst["cmps"]) and np.array_equal(gst["hops"], rst["hops"]), (words, eighths)
    oi, od, oc, ost = oix.search_batch(queries[:200], L, 1, 10)
    assert np.array_equal(ri[:200], oi) and np.array_equal(ost[:, 0], rst["cmps"][:200])
    assert rst["cmps"].mean() > 300
