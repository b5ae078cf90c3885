"""pq_search_kernel (search_pq_impl.h): beam search over PQ code rows with the query's lookup table in registers, 16-bit
visited table, lane = neighbour, and the optional packed layout (dann_pq_pack_neighbors: adjacency + the neighbours' code
rows in one row).  Everything it returns -- ids, distances (bit for bit: table entries added in chunk order in f32),
comparisons, hops, written, result_count -- must equal the oracle's (pq_dist_lookup_single,
fixed_chunk_pq_table.rs:82-100) and beam_search_kernel's; every check also asserts which kernel family served it."""
import numpy as np
import pytest

import oracle
from helpers import bits, random_graph

pytestmark = pytest.mark.gpu
da = pytest.importorskip("diskann_amd")


def _pq_index(rng, n, dim, nchunks, R, nstart, metric, min_len=None):
    bounds = np.linspace(0, dim, nchunks + 1).round().astype(np.uint32)
    bounds[0], bounds[-1] = 0, dim
    pivots = rng.standard_normal((256, dim)).astype(np.float32)
    codes = rng.integers(0, 256, (n, nchunks), dtype=np.uint8)
    adj = random_graph(rng, n, R, nstart=nstart, min_len=min_len)
    start = rng.integers(0, 256, (nstart, nchunks), dtype=np.uint8)
    oix = oracle.Index(oracle.PQ, metric, dim, n, R, start, pq_pivots=pivots, pq_offsets=bounds)
    oix.set_rows(0, codes)
    oix.adj[:] = adj
    gix = da.Provider(da.PQ, metric, dim, n, R, start, pq_pivots=pivots, pq_offsets=bounds)
    gix.set_elements(0, codes)
    gix.upload_graph(adj)
    return oix, gix


def _check(gix, oix, queries, L, k, tag, family="pq_lut"):
    oi, od, oc, ost = oix.search_batch(queries, L, 1, k)
    (gi, gd, gst), fam = gix.last_family(lambda: gix.search(da.Knn(L, 1), queries, k))
    assert fam == {family}, (fam, tag)
    assert not gst["status"].any(), tag
    assert np.array_equal(oi, gi), tag
    assert np.array_equal(bits(od), bits(gd)), tag
    assert np.array_equal(ost[:, 0], gst["cmps"]) and np.array_equal(ost[:, 1], gst["hops"]), tag
    assert np.array_equal(oc, gst["written"]) and np.array_equal(ost[:, 2], gst["result_count"]), tag


CASES = [
    (oracle.L2, 128, 16, 32, 1),
    (oracle.INNER_PRODUCT, 128, 16, 32, 1),
    (oracle.L2, 100, 10, 64, 3),        # uneven chunks, fewer than 16 of them (zero tables beyond), the full 64 lanes
    (oracle.INNER_PRODUCT, 96, 1, 7, 2),  # one chunk, short lists
    (oracle.L2, 64, 16, 33, 64),        # 64 start points: the first merge takes the slow path
    # round 6: 17 .. 64 chunks -- two, three and four groups of 64 table registers, code rows of 32 / 48 / 64 bytes
    (oracle.L2, 128, 32, 32, 1),
    (oracle.INNER_PRODUCT, 111, 17, 40, 2),   # one chunk into the second group, uneven chunk lengths
    (oracle.L2, 192, 48, 64, 1),
    (oracle.INNER_PRODUCT, 160, 37, 20, 3),
    (oracle.L2, 256, 64, 64, 2),
    (oracle.INNER_PRODUCT, 130, 64, 33, 1),   # chunks of two and three elements
]


@pytest.mark.parametrize("metric,dim,nchunks,R,nstart", CASES)
def test_pq_lut_kernel_equals_the_oracle(metric, dim, nchunks, R, nstart):
    rng = np.random.default_rng(900 + nchunks + R)
    oix, gix = _pq_index(rng, 5000, dim, nchunks, R, nstart, metric, min_len=0 if R == 7 else None)
    wide = nchunks > 16
    for packed in (False, True):
        if packed:
            gix.pq_pack_neighbors()
        for nq in ((1, 130) if wide else (1, 33, 400)):
            q = rng.standard_normal((nq, dim)).astype(np.float32)
            for L, k in (((1, 1), (10, 10), (64 - nstart, 10), (100, 7), (129, 10), (256 - nstart, 20)) if wide else
                         ((1, 1), (10, 10), (64 - nstart, 10), (65, 65), (100, 7), (128 - nstart, 300), (129, 10),
                          (256 - nstart, 20))):
                if L < 1:
                    continue
                _check(gix, oix, q, L, k, (packed, nq, L, k))
    # beyond the kernel's queue (L + start points > 256), beams wider than one, and with the kernel switched off:
    # beam_search_kernel, same results
    q = rng.standard_normal((20, dim)).astype(np.float32)
    _check(gix, oix, q, 257, 10, "L > 256", family="one_wave")
    gix.debug_set(tune_off=32)
    _check(gix, oix, q, 48, 10, "switched off", family="one_wave")
    gix.debug_set(tune_off=None)
    oi, od, oc, ost = oix.search_batch(q, 48, 3, 10)
    (gi, gd, gst), fam = gix.last_family(lambda: gix.search(da.Knn(48, 3), q, 10))
    assert fam == {"one_wave"} and np.array_equal(oi, gi) and np.array_equal(bits(od), bits(gd))


def test_pq_lut_kernel_more_than_64_chunks_stays_on_the_lds_table():
    rng = np.random.default_rng(37)
    oix, gix = _pq_index(rng, 3000, 140, 70, 16, 1, oracle.L2)
    with pytest.raises(da.DannError) as e:
        gix.pq_pack_neighbors()
    assert e.value.status == da._ffi.EUNSUPPORTED
    _check(gix, oix, rng.standard_normal((40, 140)).astype(np.float32), 48, 10, "70 chunks", family="one_wave")


@pytest.mark.parametrize("nchunks", [32, 48, 64])
def test_wide_pq_lut_kernel_freezes_and_spills(nchunks):
    """the wide tables (2 .. 4 register groups) with explicit visited tables that freeze after a few hops: spill tables,
    the re-run of queries that outgrow them, plain and packed rows -- all equal to one wave per query"""
    rng = np.random.default_rng(nchunks)
    n, dim, R, nq = 12000, 2 * nchunks, 32, 3000
    oix, gix = _pq_index(rng, n, dim, nchunks, R, 1, oracle.L2)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    gix.debug_set(tune_off=32)
    (ri, rd, rst), fam = gix.last_family(lambda: gix.search(da.Knn(40), q, 10))
    assert fam == {"one_wave"}, fam
    gix.debug_set(tune_off=None)
    for packed in (False, True):
        if packed:
            gix.pq_pack_neighbors()
        for words in (0, 64, 256):
            gix.set_visited_bits(words)
            (gi, gd, gst), fam = gix.last_family(lambda: gix.search(da.Knn(40), q, 10))
            assert "pq_lut" in fam and fam <= {"pq_lut", "one_wave"}, (fam, words)
            assert not gst["status"].any(), words
            assert np.array_equal(gi, ri) and np.array_equal(bits(gd), bits(rd)), words
            assert np.array_equal(gst["cmps"], rst["cmps"]) and np.array_equal(gst["hops"], rst["hops"]), words
        gix.set_visited_bits(0)
    oi, od, oc, ost = oix.search_batch(q[:200], 40, 1, 10)
    assert np.array_equal(ri[:200], oi) and np.array_equal(bits(rd[:200]), bits(od))


def test_pq_lut_kernel_freezes_spills_and_gives_up_like_one_wave_per_query():
    """explicit tables of 64 .. 512 words: frozen after a few hops, continued in the spill pool (20 000 queries recycle its
    512 tables many times); a query that outgrows even that is re-run through beam_search_kernel"""
    rng = np.random.default_rng(77)
    n, dim, R, nq = 20000, 64, 32, 20000
    oix, gix = _pq_index(rng, n, dim, 16, R, 1, oracle.L2)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    gix.debug_set(tune_off=32)
    (ri, rd, rst), fam = gix.last_family(lambda: gix.search(da.Knn(40), q, 10))
    assert fam == {"one_wave"}, fam
    gix.debug_set(tune_off=None)
    for packed in (False, True):
        if packed:
            gix.pq_pack_neighbors()
        for words in (0, 64, 128, 512):
            gix.set_visited_bits(words)
            for rep in range(2):
                (gi, gd, gst), fam = gix.last_family(lambda: gix.search(da.Knn(40), q, 10))
                assert "pq_lut" in fam and fam <= {"pq_lut", "one_wave"}, (fam, words)
                assert not gst["status"].any(), words
                assert np.array_equal(gi, ri) and np.array_equal(bits(gd), bits(rd)), words
                assert np.array_equal(gst["cmps"], rst["cmps"]) and np.array_equal(gst["hops"], rst["hops"]), words
        gix.set_visited_bits(0)
    oi, od, oc, ost = oix.search_batch(q[:300], 40, 1, 10)
    assert np.array_equal(ri[:300], oi) and np.array_equal(ost[:, 0], rst["cmps"][:300])
    assert rst["cmps"].mean() > 300


def test_packed_layout_is_dropped_by_mutations_and_rebuilt_on_request():
    """the packed rows are a snapshot: a mutation of the adjacency or of a code row drops them (searches read the plain
    layout again and see the change); packing again picks the change up"""
    rng = np.random.default_rng(5)
    n, dim, R = 3000, 64, 24
    oix, gix = _pq_index(rng, n, dim, 16, R, 1, oracle.L2)
    q = rng.standard_normal((64, dim)).astype(np.float32)
    gix.pq_pack_neighbors()
    _check(gix, oix, q, 50, 10, "packed")
    # rewire the start point's list and a few hub rows; change code rows that the new lists reach
    for node in (n, 0, 1, 2):
        ids = rng.choice(n, R, replace=False).astype(np.uint32)
        gix.set_neighbors(node, ids)
        oix.adj[node, 0] = R
        oix.adj[node, 1:1 + R] = ids
    new_rows = rng.integers(0, 256, (50, 16), dtype=np.uint8)
    gix.set_elements(100, new_rows)
    oix.set_rows(100, new_rows)
    _check(gix, oix, q, 50, 10, "after the mutation (plain layout)")
    gix.pq_pack_neighbors()
    _check(gix, oix, q, 50, 10, "packed again")


def test_max_concurrency_cap_takes_pq_launches_to_persistent_waves():
    """dann_set_max_concurrency: a call with more queries than the cap runs as `cap` persistent wavefronts -- pq_search_kernel
    launches one block per query and never reads the cap, so a capped call must not be served by it (the families counter
    names the kernel that ran); results do not move"""
    rng = np.random.default_rng(77)
    oix, gix = _pq_index(rng, 4000, 64, 16, 32, 1, oracle.L2)
    q = rng.standard_normal((300, 64)).astype(np.float32)
    _check(gix, oix, q, 40, 10, "no cap")
    gix.set_max_concurrency(64)
    _check(gix, oix, q, 40, 10, "capped: 300 queries over 64 persistent waves", family="persistent")
    _check(gix, oix, q[:64], 40, 10, "at the cap: one block per query again")
    gix.set_max_concurrency(0)
    _check(gix, oix, q, 40, 10, "cap lifted")


@pytest.mark.parametrize("probes", [1, 3])
def test_pq_lut_kernel_overflow_table_of_few_probes(probes):
    """large indexes leave a 16-bit visited-table entry three probes per id; ids that find them taken go to the overflow
    table (ov_insert, search_pair_impl.h).  Reached here through the probe cap; equal to beam_search_kernel throughout"""
    rng = np.random.default_rng(79 + probes)
    n, dim, R, nq = 20000, 64, 32, 6000
    oix, gix = _pq_index(rng, n, dim, 16, R, 1, oracle.L2)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    gix.debug_set(tune_off=32)
    (ri, rd, rst), fam = gix.last_family(lambda: gix.search(da.Knn(60), q, 10))
    assert fam == {"one_wave"}, fam
    gix.debug_set(tune_off=None, ht16_max_probes=probes)
    for packed in (False, True):
        if packed:
            gix.pq_pack_neighbors()
        for words in (0, 128, 1024):
            gix.set_visited_bits(words)
            (gi, gd, gst), fam = gix.last_family(lambda: gix.search(da.Knn(60), q, 10))
            assert "pq_lut" in fam and fam <= {"pq_lut", "one_wave"}, (fam, words)
            assert not gst["status"].any(), words
            assert np.array_equal(gi, ri) and np.array_equal(bits(gd), bits(rd)), words
            assert np.array_equal(gst["cmps"], rst["cmps"]) and np.array_equal(gst["hops"], rst["hops"]), words
        gix.set_visited_bits(0)
    oi, od, oc, ost = oix.search_batch(q[:200], 60, 1, 10)
    assert np.array_equal(ri[:200], oi) and np.array_equal(ost[:, 0], rst["cmps"][:200])
