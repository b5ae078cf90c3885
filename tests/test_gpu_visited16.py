"""16-bit visited-table entries (dann_set_visited_format; search_kernel_impl.h: ht16_insert_open): an exact set at half
the LDS -- every result, distance and counter must equal the oracle's and the 32-bit table's, also when ids run out of
probes (they go to the spill table) and when the table is frozen and handed on to the spill pool under load.
tests/test_visited16_model.py holds the CPU argument; `DANN_TEST_VISITED_FORMAT=16 DANN_TUNE_OFF=4 pytest -m gpu` (read by
tests/conftest.py and the Python Provider, not by the library) runs
the whole suite on this table."""
import numpy as np
import pytest

import oracle
from helpers import bits, make_pair, rand_vectors, random_graph

pytestmark = pytest.mark.gpu
da = pytest.importorskip("diskann_amd")


@pytest.fixture(autouse=True)
def _one_wave_per_query(monkeypatch):
    """small launches would go to teams (32-bit tables only): every Provider of this file has them switched off"""
    orig = da.Provider.__init__

    def init(self, *a, **kw):
        orig(self, *a, **kw)
        self.debug_set(tune_off=4)

    monkeypatch.setattr(da.Provider, "__init__", init)


CASES = [
    (oracle.F32, oracle.L2, 128, 32, 0),
    (oracle.F32, oracle.L2, 128, 32, 544),
    (oracle.F32, oracle.INNER_PRODUCT, 64, 16, 0),
    (oracle.F16, oracle.L2, 128, 32, 0),
    (oracle.F16, oracle.COSINE_NORMALIZED, 72, 20, 0),
    (oracle.U8, oracle.L2, 128, 32, 0),
    (oracle.I8, oracle.L2, 128, 32, 0),
    (oracle.I8, oracle.COSINE, 100, 16, 0),
]


@pytest.mark.parametrize("dtype,metric,dim,R,stride", CASES)
def test_search_parity_with_16_bit_entries(dtype, metric, dim, R, stride):
    rng = np.random.default_rng(99 + dim)
    n, nq = 5000, 48
    data = rand_vectors(rng, dtype, n, dim)
    adj = random_graph(rng, n, R)
    oix, gix = make_pair(dtype, metric, data, adj, data[:1], R, row_stride=stride)
    queries = rand_vectors(rng, dtype, nq, dim)
    for fmt, vbits in ((16, 0), (16, 7), (16, 64), (32, 0)):  # automatic size; 2^7 and 64 words: frozen within a few hops
        gix.set_visited_format(fmt)
        gix.set_visited_bits(vbits)
        for L, k in ((1, 1), (10, 10), (26, 10), (64, 10), (200, 10), (300, 50)):
            oi, od, oc, ost = oix.search_batch(queries, L, 1, k)
            (gi, gd, gst), fam = gix.last_family(lambda: gix.search(da.Knn(L, 1), queries, k))
            assert fam == {"one_wave"}, (fam, fmt, vbits, L)
            assert not gst["status"].any(), (fmt, vbits, L)
            assert np.array_equal(oi, gi), (fmt, vbits, L)
            assert np.array_equal(bits(od), bits(gd)), (fmt, vbits, L)
            assert np.array_equal(ost[:, 0], gst["cmps"]) and np.array_equal(ost[:, 1], gst["hops"]), (fmt, vbits, L)
            assert np.array_equal(oc, gst["written"]), (fmt, vbits, L)


def test_ids_that_run_out_of_probes_go_to_the_spill_table():
    """2^20 ids over a table of 64 buckets (128 entries) leave 14 tag bits and 3 probes per id: long before the table is
    75 % full some ids find all their probes taken.  They freeze the table and live in the spill table from then on; the used
    ids are spread over the whole id range (the start point is the highest slot)."""
    rng = np.random.default_rng(5)
    cap, used, dim, R, nq = (1 << 20) - 1, 6000, 16, 32, 400
    ids = np.sort(rng.choice(cap, used, replace=False)).astype(np.uint32)
    data = np.zeros((cap, dim), np.float32)
    data[ids] = rand_vectors(rng, oracle.F32, used, dim)
    adj = np.zeros((cap + 1, R + 1), np.uint32)
    for i in list(ids) + [cap]:
        ln = int(rng.integers(R // 2, R + 1))
        adj[i, 0] = ln
        adj[i, 1:1 + ln] = rng.choice(ids, ln, replace=False)
    oix, gix = make_pair(oracle.F32, oracle.L2, data, adj, data[ids[:1]], R)
    queries = rand_vectors(rng, oracle.F32, nq, dim)
    oi, od, oc, ost = oix.search_batch(queries, 48, 1, 10)
    gix.set_visited_format(32)
    ri, rd, rst = gix.search(da.Knn(48), queries, 10)
    assert np.array_equal(ri, oi) and np.array_equal(rst["cmps"], ost[:, 0])
    gix.set_visited_format(16)
    for words in (64, 128, 1024):
        gix.set_visited_bits(words)
        gi, gd, gst = gix.search(da.Knn(48), queries, 10)
        assert not gst["status"].any(), words
        assert np.array_equal(gi, oi) and np.array_equal(bits(gd), bits(od)), words
        assert np.array_equal(gst["cmps"], ost[:, 0]) and np.array_equal(gst["hops"], ost[:, 1]), words
    assert ost[:, 0].mean() > 400


def test_spill_tables_recycled_under_load_with_16_bit_entries():
    rng = np.random.default_rng(19)
    n, dim, R, nq = 20000, 16, 32, 20000
    data = rand_vectors(rng, oracle.F32, n, dim)
    adj = random_graph(rng, n, R)
    oix, gix = make_pair(oracle.F32, oracle.L2, data, adj, data[:1], R)
    queries = rand_vectors(rng, oracle.F32, nq, dim)
    gix.set_visited_format(32)
    ri, rd, rst = gix.search(da.Knn(48), queries, 10)
    gix.set_visited_format(16)
    for words in (0, 128):  # automatic; 256 slots: nearly every query continues in a spill table
        gix.set_visited_bits(words)
        for rep in range(2):
            gi, gd, gst = gix.search(da.Knn(48), queries, 10)
            assert not gst["status"].any()
            assert np.array_equal(gi, ri) and np.array_equal(bits(gd), bits(rd)), words
            assert np.array_equal(gst["cmps"], rst["cmps"]) and np.array_equal(gst["hops"], rst["hops"]), words
    oi, od, oc, ost = oix.search_batch(queries[:200], 48, 1, 10)
    assert np.array_equal(ri[:200], oi) and np.array_equal(ost[:, 0], rst["cmps"][:200])


@pytest.mark.parametrize("dtype", [oracle.F32, oracle.U8])
def test_range_search_second_phase_with_16_bit_entries(dtype):
    """the second phase wipes the table and re-inserts the in-range ids (range_search.rs:297-301)"""
    rng = np.random.default_rng(50 + dtype)
    n, dim, R = 3000, 16, 12
    data = rand_vectors(rng, dtype, n, dim)
    adj = random_graph(rng, n, R)
    oix, gix = make_pair(dtype, oracle.L2, data, adj, data[:1], R)
    gix.set_visited_format(16)
    queries = rand_vectors(rng, dtype, 24, dim)
    d0 = np.array([oracle.distance(dtype, oracle.L2, queries[0], data[i]) for i in range(200)])
    r_small, r_big = float(np.quantile(d0, 0.05)), float(np.quantile(d0, 0.4))
    seconds = 0
    for L, radius, inner, islack, rslack, maxret in ((20, r_small, None, 1.0, 1.0, 0), (8, r_big, None, 0.5, 1.3, 40),
                                                     (8, r_big, r_small, 0.25, 1.0, 0)):
        cap = 1500
        gi, gd, gst, gsec = gix.range_search(queries, L, radius, 1, inner, islack, rslack, maxret, out_cap=cap)
        for q in range(queries.shape[0]):
            oi, od, ost = oix.range_search(queries[q], L, radius, 1, inner, islack, rslack, maxret, out_cap=cap)
            k = oi.size
            assert int(gst["result_count"][q]) == k, (L, q)
            assert np.array_equal(gi[q, :k], oi) and np.array_equal(bits(gd[q, :k]), bits(od)), (L, q)
            assert int(gst["cmps"][q]) == int(ost[0]) and int(gst["hops"][q]) == int(ost[1]), (L, q)
            assert int(gsec[q]) == int(ost[3])
            seconds += int(ost[3])
    assert seconds > 0


def test_insert_searches_and_build_with_16_bit_entries():
    """the insert-time search (record mode) runs on the same table: a GPU build equals the oracle's multi_insert"""
    rng = np.random.default_rng(31)
    n, dim, R = 3000, 32, 16
    data = rand_vectors(rng, oracle.F32, n, dim)
    adj = random_graph(rng, n, R)
    oix, gix = make_pair(oracle.F32, oracle.L2, data, adj, data[:1], R)
    gix.set_visited_format(16)
    slots = rng.choice(n, 300, replace=False).astype(np.uint32)
    rid, rd, rn, st = gix.search_record(slots, 50)
    for i, s in enumerate(slots[:60]):
        _, _, _, ost, orid, ord_ = oix.search(data[s], 50, 1, 10, record=True)
        assert rn[i] == orid.size
        assert np.array_equal(rid[i, :rn[i]], orid) and np.array_equal(bits(rd[i, :rn[i]]), bits(ord_))
        assert st["cmps"][i] == ost[0] and st["hops"][i] == ost[1]


def test_format_argument_is_checked():
    p = da.Provider(da.F32, da.L2, 4, 10, 4, np.zeros((1, 4), np.float32))
    for bad in (1, 8, 17, 64):
        with pytest.raises(da.DannError) as e:
            p.set_visited_format(bad)
        assert e.value.status == da._ffi.EINVAL
    for ok in (16, 32, 0):
        p.set_visited_format(ok)
