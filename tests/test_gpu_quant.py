"""Quantised variants on the GPU vs the oracle: SQ-8 (compress, distances, search, build) and
the PQ lookup-table build + scan."""
import ctypes as C

import numpy as np
import pytest

import oracle
from helpers import bits, random_graph, small_calls_from_threads

pytestmark = pytest.mark.gpu
da = pytest.importorskip("diskann_amd")


def _orc_compress(x, shift, scale):
    L = oracle.lib()
    out = np.zeros((x.shape[0], x.shape[1] + 4), np.uint8)
    for i in range(x.shape[0]):
        c = np.zeros(1, np.float32)
        L.orc_sq8_compress(x[i].ctypes.data, x.shape[1], shift.ctypes.data, C.c_float(scale),
                           out[i].ctypes.data, c.ctypes.data)
        out[i, x.shape[1]:] = c.view(np.uint8)
    return out


def _sq_setup(rng, n, dim):
    data = rng.normal(0.3, 0.5, (n, dim)).astype(np.float32)
    # ScalarQuantizer parameters as train.rs would produce them: shift = mean - 2 std, scale = 4 std
    shift = (data.mean(0) - 2.0 * data.std(0)).astype(np.float32)
    scale = float(np.float32(4.0 * data.std()))
    return data, shift, scale


def test_sq8_compress_matches_oracle():
    rng = np.random.default_rng(31)
    for dim in (7, 64, 128, 100):
        data, shift, scale = _sq_setup(rng, 300, dim)
        data[3, 0] = 1e9
        data[4, 1] = -1e9
        got = da.sq8_compress(data, shift, scale)
        want = _orc_compress(data, shift, scale)
        assert np.array_equal(got, want), dim


@pytest.mark.parametrize("metric,stride", [(oracle.L2, 0), (oracle.INNER_PRODUCT, 0), (oracle.COSINE_NORMALIZED, 0),
                                           (oracle.L2, 256), (oracle.INNER_PRODUCT, 256)])
def test_sq8_search_and_distances(metric, stride):
    """stride 256: the line-aligned SQ-8 store bench.py uses (code bytes of a row in one 128-byte line)"""
    rng = np.random.default_rng(32 + metric)
    n, dim, R = 4000, 128, 24
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
    queries = da.sq8_compress(rng.normal(0.3, 0.5, (40, dim)).astype(np.float32), shift, scale)
    oi, od_, _, _ = oix.search_batch(queries, 30, 1, 10)
    small_calls_from_threads(gix, queries, 30, 1, 10, oi, od_)
    ids = rng.choice(n, 200, replace=False).astype(np.uint32)
    _, od = oix.expand_beam(queries[0], ids)
    _, gd = gix.expand_beam(queries[0], ids)
    assert np.array_equal(bits(od), bits(gd))
    a, b = rng.integers(0, n, 100).astype(np.uint32), rng.integers(0, n, 100).astype(np.uint32)
    gp = gix.distance_pairs(a, b)
    L = oracle.lib()
    for i in range(100):
        x, y = codes[a[i]], codes[b[i]]
        want = L.orc_sq8_distance(metric if metric != oracle.COSINE_NORMALIZED else oracle.L2, x.ctypes.data,
                                  C.c_float(x[dim:].view(np.float32)[0]), y.ctypes.data,
                                  C.c_float(y[dim:].view(np.float32)[0]), dim, C.c_float(scale), C.c_float(snorm))
        if metric == oracle.COSINE_NORMALIZED:
            want = np.float32(1.0) - (np.float32(1.0) - np.float32(want) / np.float32(2.0))
        assert np.float32(want).view(np.uint32) == gp[i:i + 1].view(np.uint32)[0]
    for Lv, W in ((20, 1), (64, 2)):
        oi, od, oc, ost = oix.search_batch(queries, Lv, W, 10)
        gi, gd, gst = gix.search(da.Knn(Lv, W), queries, 10)
        assert np.array_equal(oi, gi) and np.array_equal(bits(od), bits(gd))
        assert np.array_equal(ost[:, 0], gst["cmps"]) and np.array_equal(ost[:, 1], gst["hops"])


def test_sq8_cosine_is_rejected():
    with pytest.raises(da.DannError) as e:
        da.Provider(da.SQ8, da.COSINE, 16, 10, 4, np.zeros((1, 20), np.uint8), sq_scale=1.0)
    assert e.value.status == da._ffi.EUNSUPPORTED


def test_sq8_build_matches_oracle():
    rng = np.random.default_rng(35)
    n, dim, R, maxdeg, lb = 500, 32, 8, 10, 24
    data, shift, scale = _sq_setup(rng, n, dim)
    codes = da.sq8_compress(data, shift, scale)
    snorm = float(np.float32((shift ** 2).sum(dtype=np.float32)))
    start = da.sq8_compress(data.mean(0, keepdims=True).astype(np.float32), shift, scale)
    oix = oracle.Index(oracle.SQ8, oracle.L2, dim, n, maxdeg, start, sq_scale=scale, sq_shift_norm_sq=snorm)
    oix.set_rows(0, codes)
    gix = da.Provider(da.SQ8, da.L2, dim, n, maxdeg, start, sq_scale=scale, sq_shift_norm_sq=snorm)
    gix.set_elements(0, codes)
    ocfg = oracle.build_config(R, maxdeg, lb, intra_batch_candidates=oracle.IBC_NONE)
    gcfg = da.build_config(R, maxdeg, lb, intra_batch_candidates=da.IBC_NONE)
    s = 0
    for b in (1, 2, 5, 20, 72, 400):
        slots = np.arange(s, min(s + b, n), dtype=np.uint32)
        oix.multi_insert(ocfg, slots)
        gix.insert_batch(gcfg, slots)
        s += b
    got = gix.download_graph()
    lens = oix.adj[:, 0]
    assert np.array_equal(got[:, 0], lens)
    mask = np.arange(maxdeg)[None, :] < lens[:, None]
    assert np.array_equal(got[:, 1:][mask], oix.adj[:, 1:][mask])


@pytest.mark.parametrize("metric", [oracle.L2, oracle.INNER_PRODUCT])
def test_pq_lut_and_scan(metric):
    rng = np.random.default_rng(40 + metric)
    dim = 100
    offsets = np.array([0, 7, 16, 24, 33, 48, 64, 71, 80, 92, 100], np.uint32)  # ragged chunks
    nchunks = offsets.size - 1
    pivots = rng.standard_normal((256, dim)).astype(np.float32)
    queries = rng.standard_normal((5, dim)).astype(np.float32)
    lut = da.pq_build_lut(metric, pivots, offsets, queries)
    L = oracle.lib()
    want = np.zeros((nchunks, 256), np.float32)
    for q in range(5):
        L.orc_pq_build_lut(metric, pivots.ctypes.data, None, offsets.ctypes.data, nchunks, dim,
                           queries[q].ctypes.data, want.ctypes.data)
        assert np.array_equal(bits(lut[q]), bits(want)), q
    npts = 3000
    codes = rng.integers(0, 256, (npts, nchunks), dtype=np.uint8)
    lens = [0, 5000, 17, 1, 900]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    ids = rng.integers(0, npts, int(off[-1])).astype(np.uint32)
    got = da.pq_scan(lut, codes, ids, off)
    for q in range(5):
        for j in range(int(off[q]), int(off[q + 1]), 37):
            w = L.orc_pq_lookup(lut[q].ctypes.data, codes[ids[j]].ctypes.data, nchunks)
            assert np.float32(w).view(np.uint32) == got[j:j + 1].view(np.uint32)[0]
    # known-answer test of the reference (fixed_chunk_pq_table.rs:1081-1093)
    kat = np.arange(512, dtype=np.float32).reshape(1, 2, 256)
    out = da.pq_scan(kat, np.array([[1, 3]], np.uint8), np.array([0], np.uint32), np.array([0, 1], np.uint64))
    assert out[0] == kat[0, 0, 1] + kat[0, 1, 3]


@pytest.mark.parametrize("fdtype", [oracle.F32, oracle.F16])
def test_quantised_search_with_rerank(fdtype):
    """SQ-8 graph search followed by the Rerank post-processor on the full-precision rows
    (full_precision.rs:348-397): equals oracle SQ-8 search + oracle full-precision distances,
    sorted by (distance, candidate order)."""
    rng = np.random.default_rng(60)
    n, dim, R, L, k = 3000, 64, 16, 40, 10
    data, shift, scale = _sq_setup(rng, n, dim)
    full = data.astype(oracle.NP_DTYPE[fdtype])
    codes = da.sq8_compress(data, shift, scale)
    snorm = float(np.float32((shift ** 2).sum(dtype=np.float32)))
    adj = random_graph(rng, n, R)
    sq = da.Provider(da.SQ8, da.L2, dim, n, R, codes[:1], sq_scale=scale, sq_shift_norm_sq=snorm)
    sq.set_elements(0, codes)
    sq.upload_graph(adj)
    fp = da.Provider(fdtype, da.L2, dim, n, R, full[:1])
    fp.set_elements(0, full)
    qf = rng.normal(0.3, 0.5, (30, dim)).astype(np.float32)
    qc = da.sq8_compress(qf, shift, scale)
    cand, _, _ = sq.search(da.Knn(L), qc, L)          # every non-start entry of the L-list
    qfull = qf.astype(oracle.NP_DTYPE[fdtype])
    ids, d = fp.rerank(qfull, cand, k)
    for q in range(30):
        c = [int(x) for x in cand[q] if x != 0xFFFFFFFF]
        dd = np.array([oracle.query_distance(fdtype, oracle.L2, qfull[q], full[i]) for i in c], np.float32)
        order = np.argsort(dd, kind="stable")[:k]
        assert [c[i] for i in order] == [int(x) for x in ids[q, :len(order)]]
        assert np.array_equal(bits(dd[order]), bits(d[q, :len(order)]))


@pytest.mark.parametrize("metric,nchunks", [(oracle.L2, 16), (oracle.INNER_PRODUCT, 10), (oracle.L2, 37)])
def test_pq_beam_search(metric, nchunks):
    """Beam search over PQ codes (lookup table built per query in LDS, entries added in chunk order)
    equals the oracle's search with the same table: ids, distances, cmps, hops."""
    rng = np.random.default_rng(70 + nchunks)
    n, dim, R = 3000, 100, 16
    bounds = np.linspace(0, dim, nchunks + 1).round().astype(np.uint32)
    bounds[0], bounds[-1] = 0, dim
    pivots = rng.standard_normal((256, dim)).astype(np.float32)
    data = rng.standard_normal((n, dim)).astype(np.float32)
    codes = np.empty((n, nchunks), np.uint8)
    for c in range(nchunks):
        s, e = bounds[c], bounds[c + 1]
        d = ((data[:, None, s:e] - pivots[None, :, s:e]) ** 2).sum(-1)
        codes[:, c] = d.argmin(1)
    adj = random_graph(rng, n, R)
    oix = oracle.Index(oracle.PQ, metric, dim, n, R, codes[:1], pq_pivots=pivots, pq_offsets=bounds)
    oix.set_rows(0, codes)
    oix.adj[:] = adj
    gix = da.Provider(da.PQ, metric, dim, n, R, codes[:1], pq_pivots=pivots, pq_offsets=bounds)
    gix.set_elements(0, codes)
    gix.upload_graph(adj)
    q = rng.standard_normal((32, dim)).astype(np.float32)
    for L, W in ((10, 1), (48, 1), (48, 3), (150, 2)):
        oi, od, oc, ost = oix.search_batch(q, L, W, 10)
        gi, gd, gst = gix.search(da.Knn(L, W), q, 10)
        assert np.array_equal(oi, gi), (L, W)
        assert np.array_equal(bits(od), bits(gd)), (L, W)
        assert np.array_equal(ost[:, 0], gst["cmps"]) and np.array_equal(ost[:, 1], gst["hops"])
    # small calls of several threads (one launch for a group of them; the f32 queries staged in device memory: the table
    # build reads a query 256 times)
    oi, od, _, _ = oix.search_batch(q, 48, 3, 10)
    small_calls_from_threads(gix, q, 48, 3, 10, oi, od)
    # the build path is not defined on PQ rows
    with pytest.raises(da.DannError) as e:
        gix.insert_batch(da.build_config(4, 8, 10), [0])
    assert e.value.status in (da._ffi.EUNSUPPORTED, da._ffi.EINVAL)


def test_pq_compress_reference_pattern_gpu():
    """the reference's Chunk::find_closest test pattern (pivots.rs:1295-1411) through dann_pq_compress"""
    import diskann_amd as da
    for total, dim in ((1, 1), (7, 3), (16, 8), (17, 9), (71, 15), (103, 16), (256, 7)):
        data = (np.arange(total)[:, None] + np.arange(dim)[None, :]).astype(np.float32)
        off = np.array([0, dim], np.uint32)
        codes = da.pq_compress(data, off, data + np.float32(0.125))
        assert np.array_equal(codes[:, 0], np.arange(total) % 256), (total, dim)
        assert da.pq_compress(data, off, np.zeros((1, dim), np.float32))[0, 0] == 0
        for bad in (np.inf, -np.inf, np.nan):
            with pytest.raises(da.DannError):
                da.pq_compress(data, off, np.full((3, dim), bad, np.float32))
        tied = data.copy()
        tied[0] = data[-1]
        assert da.pq_compress(tied, off, data[-1:])[0, 0] == 0


@pytest.mark.parametrize("dim,off", [(24, [0, 5, 8, 16, 24]), (128, list(range(0, 129, 8))), (100, [0, 33, 66, 100])])
def test_pq_compress_vs_oracle(dim, off):
    import diskann_amd as da
    rng = np.random.default_rng(21)
    off = np.array(off, np.uint32)
    for ncenters in (256, 37):
        piv = rng.standard_normal((ncenters, dim)).astype(np.float32)
        x = rng.standard_normal((1500, dim)).astype(np.float32)
        x[:64] = piv[rng.integers(0, ncenters, 64)]  # exact hits
        piv[5] = piv[14]                             # duplicated centres: the lane-wise tie rule decides
        rc, want = oracle.pq_compress(piv, off, x)
        assert rc == 0
        got = da.pq_compress(piv, off, x)
        assert np.array_equal(got, want)
    with pytest.raises(da.DannError):
        da.pq_compress(np.zeros((300, dim), np.float32), off, x)  # CannotCompressToByte


@pytest.mark.parametrize("n,dim,off,k", [(1003, 7, [0, 3, 7], 9), (4096, 32, list(range(0, 33, 8)), 256), (777, 5, [0, 5], 2),
                                          (20000, 16, [0, 4, 8, 12, 16], 64)])
def test_pq_lloyds_vs_oracle(n, dim, off, k):
    """GPU Lloyd iterations == oracle: centres, last assignments and residuals, bit for bit"""
    import diskann_amd as da
    rng = np.random.default_rng(31)
    true = (rng.standard_normal((max(k // 2, 1), dim)) * 4).astype(np.float32)
    x = (true[rng.integers(0, true.shape[0], n)] + rng.standard_normal((n, dim))).astype(np.float32)
    init = x[rng.choice(n, k, replace=False)].copy()
    if k > 4:
        init[3] = 1e5  # an empty cluster
    for reps in (1, 4):
        wc, wa, wr = oracle.pq_lloyds(x, off, init, reps)
        gc, ga, gr = da.pq_lloyds(x, off, init, reps)
        assert np.array_equal(ga, wa), reps
        assert np.array_equal(gc.view(np.uint32), wc.view(np.uint32)), reps
        assert np.array_equal(gr.view(np.uint32), wr.view(np.uint32)), reps
    # trained pivots feed the compressor: every row lands on its assigned centre or a closer one after the update
    codes = da.pq_compress(gc, off, x)
    assert codes.shape == (n, len(off) - 1)


@pytest.mark.parametrize("n,dim", [(1000, 12), (4097, 128), (333, 7)])
def test_sq8_train_vs_oracle(n, dim):
    """ScalarQuantizationParameters::train: shift, scale and mean norm equal the oracle's bit for bit"""
    import diskann_amd as da
    rng = np.random.default_rng(41)
    x = (rng.standard_normal((n, dim)) * rng.uniform(0.1, 5.0, dim) + rng.uniform(-3, 3, dim)).astype(np.float32)
    for sd in (1.0, 1.5, 2.0):
        ws, wc, wm = oracle.sq8_train(x, sd)
        gs, gc, gm = da.sq8_train(x, sd)
        assert np.array_equal(gs.view(np.uint32), ws.view(np.uint32))
        assert np.float32(gc) == wc and np.float32(gm) == wm
    with pytest.raises(da.DannError):
        da.sq8_train(x, 0.0)


class _Draws:
    def __init__(self, seed, nchunks):
        self.g = [np.random.default_rng(seed + c) for c in range(nchunks)]

    def index(self, c, n):
        return int(self.g[c].integers(0, n))

    def f64(self, c, high):
        return float(self.g[c].random() * high)


@pytest.mark.parametrize("n,dim,off,k", [(3000, 24, [0, 8, 16, 24], 64), (1000, 17, [0, 5, 17], 16), (777, 9, [0, 9], 256)])
def test_pq_kmeanspp_and_train_vs_oracle(n, dim, off, k):
    """k-means++ seeding and the whole LightPQTrainingParameters::train on the GPU, bit-identical to the oracle given the
    same random draws (n not a multiple of 16: the partial block of update_distances; k = 256: the PQ default)."""
    rng = np.random.default_rng(n + dim)
    x = (rng.standard_normal((n, dim)) * rng.uniform(0.5, 3.0, (1, dim))).astype(np.float32)
    d1, d2 = _Draws(77, len(off) - 1), _Draws(77, len(off) - 1)
    rc, ocen, osel = oracle.pq_kmeanspp(x, off, k, d1.index, d1.f64)
    gcen, gsel = da.pq_kmeanspp(x, off, k, d2.index, d2.f64)
    assert rc == 0 and np.array_equal(osel, gsel) and np.array_equal(bits(ocen), bits(gcen))
    d3 = _Draws(78, len(off) - 1)
    piv = da.pq_train(x, off, k, 6, d3.index, d3.f64)
    d4 = _Draws(78, len(off) - 1)
    _, seed_c, _ = oracle.pq_kmeanspp(x, off, k, d4.index, d4.f64)
    want, _, _ = oracle.pq_lloyds(x, off, seed_c, 6)
    assert np.array_equal(bits(piv), bits(want))


def test_pq_lloyds_chunk_beyond_the_mfma_slab_takes_the_row_kernel():
    """a chunk whose transposed centres do not fit 64 KiB of LDS keeps the row kernel; same bits as the oracle either way"""
    rng = np.random.default_rng(41)
    n, dim, k = 900, 200, 64
    x = rng.standard_normal((n, dim)).astype(np.float32)
    init = x[rng.choice(n, k, replace=False)].copy()
    wc, wa, wr = oracle.pq_lloyds(x, [0, dim], init, 2)
    gc, ga, gr = da.pq_lloyds(x, [0, dim], init, 2)
    assert np.array_equal(ga, wa) and np.array_equal(bits(gc), bits(wc)) and np.array_equal(bits(gr), bits(wr))


@pytest.mark.parametrize("kind", ["scales", "duplicates", "tiny", "large"])
def test_pq_kmeanspp_rolling_sums_on_hard_inputs(kind):
    """The D^2 draw and its totals are rolling f64 sums in row order; the kernels evaluate them in parallel only over
    ranges where no addition can round (seq_exact) and walk the rest element by element.  Inputs that force the walk --
    rows on scales 12 orders of magnitude apart, thousands of copies of the centres picked (distances of rounding-noise
    size, some negative), distances below the running sum's last place -- and a set large enough for full ranges:
    seeds and selected counts bit-identical to the oracle's sequential loops."""
    rng = np.random.default_rng(len(kind))
    dim, off, k = 12, [0, 4, 12], 24
    if kind == "scales":
        n = 5000
        x = rng.standard_normal((n, dim)).astype(np.float32) * np.float32(10.0) ** rng.integers(-6, 6, (n, 1)).astype(np.float32)
    elif kind == "duplicates":
        n = 6000
        base = (rng.standard_normal((40, dim)) * 100).astype(np.float32) + np.float32(1000.0)
        x = base[rng.integers(0, 40, n)].copy()
        x[::7] += (rng.standard_normal((len(x[::7]), dim)) * 1e-3).astype(np.float32)
    elif kind == "tiny":
        n = 4000
        x = rng.standard_normal((n, dim)).astype(np.float32)
        x[rng.integers(0, n, 300)] *= np.float32(1e-12)
        x[rng.integers(0, n, 300)] *= np.float32(1e-30)
    else:
        n, k = 70001, 40
        x = (rng.standard_normal((n, dim)) * rng.uniform(0.2, 2.0, (1, dim))).astype(np.float32)
    d1, d2 = _Draws(11, len(off) - 1), _Draws(11, len(off) - 1)
    rc, ocen, osel = oracle.pq_kmeanspp(x, off, k, d1.index, d1.f64)
    da.pq_rolling_sum_stats(reset=True)
    gcen, gsel = da.pq_kmeanspp(x, off, k, d2.index, d2.f64)
    st = da.pq_rolling_sum_stats()
    assert rc == 0 and np.array_equal(osel, gsel), (osel, gsel)
    assert np.array_equal(bits(ocen), bits(gcen))
    # which paths the rolling sums took: the hard inputs reach the element walk, the large plain one sums in parallel
    assert st["wave_ranges"] + st["thread_ranges"] > 0, st
    if kind in ("scales", "tiny"):
        assert st["walked_ranges"] > 0, st
    if kind == "large":
        assert st["walked_elements"] < 0.01 * 2 * (k - 1) * 2 * n, st


def test_pq_kmeanspp_degenerate_inputs():
    rng = np.random.default_rng(3)
    few = np.repeat(rng.integers(-3, 4, (4, 6)).astype(np.float32), 30, axis=0)   # 4 distinct rows (exact arithmetic), 10 centres wanted
    d1, d2 = _Draws(1, 1), _Draws(1, 1)
    rc, ocen, osel = oracle.pq_kmeanspp(few, [0, 6], 10, d1.index, d1.f64)
    gcen, gsel = da.pq_kmeanspp(few, [0, 6], 10, d2.index, d2.f64)
    assert rc == 0 and osel[0] == 4 and np.array_equal(osel, gsel) and np.array_equal(bits(ocen), bits(gcen))
    bad = rng.standard_normal((40, 6)).astype(np.float32)
    bad[3, 2] = np.inf
    with pytest.raises(da.DannError):
        da.pq_kmeanspp(bad, [0, 6], 5, _Draws(2, 1).index, _Draws(2, 1).f64)
