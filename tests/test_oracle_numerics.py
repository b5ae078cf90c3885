"""Distance numerics of the oracle: the contracts the reference's own kernel tests state
(diskann-vector/src/distance/implementations.rs:514-594,719-759;
 diskann-inmem/src/layers/full.rs:573-575)."""
import numpy as np
import pytest

import oracle

METRICS = [oracle.L2, oracle.INNER_PRODUCT, oracle.COSINE, oracle.COSINE_NORMALIZED]


def _rand(rng, dtype, dim):
    if dtype == oracle.U8:
        return rng.integers(0, 256, dim, dtype=np.uint8)
    if dtype == oracle.I8:
        return rng.integers(-128, 128, dim, dtype=np.int8)
    return rng.uniform(-1, 1, dim).astype(oracle.NP_DTYPE[dtype])


@pytest.mark.parametrize("dtype", [oracle.F32, oracle.F16])
def test_float_kernels_vs_scalar_reference(dtype):
    rng = np.random.default_rng(0x12345)
    for dim in list(range(0, 70)) + [100, 127, 128, 129, 160, 255, 256, 384, 768]:
        for metric in METRICS:
            x, y = _rand(rng, dtype, dim), _rand(rng, dtype, dim)
            got = oracle.distance(dtype, metric, x, y)
            want = oracle.distance_scalar_ref(dtype, metric, x, y)
            assert abs(got - want) <= 1e-4 + 1e-4 * abs(want), (dtype, metric, dim)
            q = oracle.query_distance(dtype, metric, x, y)
            assert abs(q - got) <= 1e-3 + 1e-4 * abs(got)  # full.rs:573-575


@pytest.mark.parametrize("dtype", [oracle.U8, oracle.I8])
def test_integer_kernels_exact(dtype):
    rng = np.random.default_rng(1)
    for dim in list(range(0, 70)) + [100, 128, 160]:
        x, y = _rand(rng, dtype, dim), _rand(rng, dtype, dim)
        xi, yi = x.astype(np.int64), y.astype(np.int64)
        assert oracle.distance(dtype, oracle.L2, x, y) == float(np.float32(((xi - yi) ** 2).sum()))
        assert oracle.distance(dtype, oracle.INNER_PRODUCT, x, y) == float(np.float32(-(xi * yi).sum()))
        # CosineNormalized == Cosine for integers (distance_provider.rs:274-297)
        assert oracle.distance(dtype, oracle.COSINE_NORMALIZED, x, y) == oracle.distance(dtype, oracle.COSINE, x, y)
        assert oracle.query_distance(dtype, oracle.L2, x, y) == oracle.distance(dtype, oracle.L2, x, y)


def test_avx2_twin_is_bit_identical():
    rng = np.random.default_rng(2)
    for dtype in (oracle.F32, oracle.F16):
        for dim in list(range(0, 80)) + [100, 128, 384, 768, 771]:
            for metric in (oracle.L2, oracle.INNER_PRODUCT, oracle.COSINE_NORMALIZED):
                q, r = _rand(rng, dtype, dim), _rand(rng, dtype, dim)
                a = np.float32(oracle.query_distance(dtype, metric, q, r))
                b = np.float32(oracle.query_distance(dtype, metric, q, r, fast=True))
                assert a.view(np.uint32) == b.view(np.uint32), (dtype, dim, metric)


def test_l2_association_order_f32():
    """Appendix A rule 14 spelled out in numpy float32 for dim=128."""
    rng = np.random.default_rng(3)
    x, y = rng.standard_normal(128).astype(np.float32), rng.standard_normal(128).astype(np.float32)
    acc = np.zeros(32, np.float32)
    for t in range(4):
        c = (x[32 * t: 32 * t + 32] - y[32 * t: 32 * t + 32]).astype(np.float32)
        # fused multiply-add: exact product in float64, single rounding
        acc = (c.astype(np.float64) * c.astype(np.float64) + acc.astype(np.float64)).astype(np.float32)
    s = acc.reshape(4, 8)
    v = (s[0] + s[1]) + (s[2] + s[3])
    want = ((v[0] + v[4]) + (v[2] + v[6])) + ((v[1] + v[5]) + (v[3] + v[7]))
    got = np.float32(oracle.distance(oracle.F32, oracle.L2, x, y))
    assert got.view(np.uint32) == np.float32(want).view(np.uint32)


def test_cosine_edge_cases():
    z = np.zeros(16, np.float32)
    o = np.ones(16, np.float32)
    assert oracle.distance(oracle.F32, oracle.COSINE, z, o) == 1.0  # zero norm -> similarity 0
    assert oracle.distance(oracle.F32, oracle.COSINE, o, o) == 0.0
    assert oracle.distance(oracle.F32, oracle.COSINE, o, -o) == 2.0


def test_medoid_rule():
    rng = np.random.default_rng(4)
    data = rng.standard_normal((500, 32)).astype(np.float32)
    r, mean = oracle.medoid_f32(data)
    m = (data.astype(np.float64).sum(0) / 500).astype(np.float32)
    assert np.array_equal(mean, m)
    d = np.array([oracle.distance(oracle.F32, oracle.L2, m, row) for row in data], np.float32)
    assert r == int(np.argmin(d))


def test_pq_lookup_kat():
    """fixed_chunk_pq_table.rs:1081-1093 pq_dist_lookup_test: out = LUT[0*256+1] + LUT[1*256+3]."""
    lut = np.arange(512, dtype=np.float32)
    code = np.array([1, 3], np.uint8)
    got = oracle.lib().orc_pq_lookup(lut.ctypes.data, code.ctypes.data, 2)
    assert got == float(lut[1] + lut[256 + 3])


def test_sq8_formulas():
    rng = np.random.default_rng(5)
    dim = 64
    shift = rng.uniform(-1, 0, dim).astype(np.float32)
    scale = np.float32(2.5)
    xs = rng.uniform(-1, 1, (2, dim)).astype(np.float32)
    codes = np.zeros((2, dim), np.uint8)
    comp = np.zeros(2, np.float32)
    L = oracle.lib()
    for i in range(2):
        c = np.zeros(1, np.float32)
        L.orc_sq8_compress(xs[i].ctypes.data, dim, shift.ctypes.data, scale, codes[i].ctypes.data, c.ctypes.data)
        comp[i] = c[0]
        want = np.round(np.clip((xs[i] - shift) * np.float32(255.0 / scale), 0, 255))
        assert np.abs(codes[i].astype(np.float32) - want).max() <= 1  # float32 vs float64 rounding at .5
    recon = codes.astype(np.float32) * (scale / 255) + shift
    l2 = L.orc_sq8_distance(oracle.L2, codes[0].ctypes.data, comp[0], codes[1].ctypes.data, comp[1], dim, scale,
                            float((shift * shift).sum()))
    assert abs(l2 - float(((recon[0] - recon[1]) ** 2).sum())) <= 1e-3 * max(1.0, abs(l2))
    ip = L.orc_sq8_distance(oracle.INNER_PRODUCT, codes[0].ctypes.data, comp[0], codes[1].ctypes.data, comp[1], dim,
                            scale, float((shift * shift).sum()))
    assert abs(-ip - float((recon[0] * recon[1]).sum())) <= 1e-3 * max(1.0, abs(ip))


def test_pq_compress_reference_pattern():
    """Chunk::find_closest known answers, restated from the reference's own test
    (diskann-quantization/src/product/tables/transposed/pivots.rs:1165-1170, 1295-1411): centres
    data[i][j] = i + j, query (i + j) + 0.125 -> i; zero query -> 0; non-finite queries are errors; a copy of the
    last centre in slot 0 wins the tie."""
    for total in list(range(1, 18)) + [64, 71, 96, 103, 255, 256]:
        for dim in (1, 2, 3, 7, 8, 9, 15, 16):
            data = (np.arange(total)[:, None] + np.arange(dim)[None, :]).astype(np.float32)
            off = np.array([0, dim], np.uint32)
            queries = data + np.float32(0.125)
            rc, codes = oracle.pq_compress(data, off, queries)
            assert rc == 0 and np.array_equal(codes[:, 0], np.arange(total) % 256), (total, dim)
            rc, codes = oracle.pq_compress(data, off, np.zeros((1, dim), np.float32))
            assert rc == 0 and codes[0, 0] == 0
            for bad in (np.inf, -np.inf, np.nan):
                rc, _ = oracle.pq_compress(data, off, np.full((1, dim), bad, np.float32))
                assert rc < -1, (total, dim, bad)
            tied = data.copy()
            tied[0] = data[-1]
            rc, codes = oracle.pq_compress(tied, off, data[-1:])
            assert rc == 0 and codes[0, 0] == 0, (total, dim)


def test_pq_compress_matches_bruteforce_and_chunking():
    rng = np.random.default_rng(5)
    dim, off = 24, np.array([0, 5, 8, 16, 24], np.uint32)
    piv = rng.standard_normal((256, dim)).astype(np.float32)
    x = rng.standard_normal((300, dim)).astype(np.float32)
    rc, codes = oracle.pq_compress(piv, off, x)
    assert rc == 0
    for c in range(4):
        s, e = int(off[c]), int(off[c + 1])
        d = ((x[:, None, s:e].astype(np.float64) - piv[None, :, s:e].astype(np.float64)) ** 2).sum(-1)
        best = d.min(1)
        got = d[np.arange(x.shape[0]), codes[:, c]]
        assert np.all(got <= best * (1 + 1e-5) + 1e-6)
    # lane-wise tie rule: equal scores in lanes 1 (index 9) and 2 (index 2) -> the lower lane wins, not the lower index
    piv2 = np.full((16, 2), 10.0, np.float32)
    piv2[9] = piv2[2] = [1.0, 1.0]
    rc, codes = oracle.pq_compress(piv2, np.array([0, 2], np.uint32), np.array([[1.0, 1.0]], np.float32))
    assert rc == 0 and codes[0, 0] == 9


def test_pq_lloyds_against_naive():
    """lloyds.rs is checked in the reference against a naive k-means within rounding; same here, plus the corner
    rules read off the source: empty clusters become the zero vector, assignments are those of the last step."""
    rng = np.random.default_rng(3)
    k, dim, n = 9, 7, 1003
    true = (rng.standard_normal((k, dim)) * 8).astype(np.float32)
    x = (true[rng.integers(0, k, n)] + 0.2 * rng.standard_normal((n, dim))).astype(np.float32)
    init = x[rng.choice(n, k, replace=False)].copy()
    off = [0, 3, 7]
    cen, asg, res = oracle.pq_lloyds(x, off, init, 6)
    for c in range(2):
        s, e = off[c], off[c + 1]
        cc, xs = init[:, s:e].astype(np.float64).copy(), x[:, s:e].astype(np.float64)
        for _ in range(6):
            d = ((xs[:, None, :] - cc[None]) ** 2).sum(-1)
            a = d.argmin(1)
            for j in range(k):
                cc[j] = xs[a == j].mean(0) if (a == j).any() else 0.0
        assert np.abs(cc - cen[:, s:e]).max() < 1e-4
        assert (a == asg[c]).mean() > 0.999
        assert abs(res[c] - d.min(1).sum()) <= 1e-3 * d.min(1).sum() + 1e-3
    # an unreachable initial centre empties out and becomes zero
    init2 = init.copy()
    init2[0] = 1e6
    cen2, asg2, _ = oracle.pq_lloyds(x, [0, dim], init2, 2)
    assert not (asg2 == 0).any() and np.all(cen2[0] == 0)


def test_sq8_train_matches_reference_contract():
    """scalar/train.rs:66-128 (test_train): scale = 2 * std * max column std within 1e-7 relative, shift =
    (mean - std * max column std) as f32 exactly, mean_norm = mean row norm as f32"""
    rng = np.random.default_rng(2)
    for nrows, ncols in ((10, 16), (7, 8), (500, 33)):
        x = (rng.standard_normal((nrows, ncols)) * rng.uniform(0.5, 3, ncols)).astype(np.float32)
        xd = x.astype(np.float64)
        means = np.zeros(ncols)
        for r in range(nrows):
            means += xd[r]
        means /= nrows
        var = np.zeros(ncols)
        for r in range(nrows):
            var += (xd[r] - means) ** 2
        var /= nrows
        smax = np.sqrt(var.max())
        for sd in (1.0, 1.5, 2.0):
            shift, scale, mn = oracle.sq8_train(x, sd)
            assert abs(float(scale) - sd * 2.0 * smax) / (sd * 2.0 * smax) < 1e-7
            assert np.array_equal(shift, (means - sd * smax).astype(np.float32))
        norms = 0.0
        for r in range(nrows):
            norms += np.sqrt((xd[r] * xd[r]).cumsum()[-1])
        assert mn == np.float32(norms / nrows)


def test_unreadable_slots_equal_a_graph_without_them():
    """Tag rule of the oracle (expand_beam_inner skips a slot whose tag is below PUBLISHED after the visited insert,
    uncounted: provider.rs:448-473, 681-686) against an independent formulation: the same search on a graph whose
    adjacency lists never mention the unreadable slots returns the same ids, distances, cmps and hops."""
    from helpers import rand_vectors, random_graph
    rng = np.random.default_rng(4242)
    n, dim, R = 3000, 24, 16
    stride = oracle.inmem2_stride(oracle.F32, dim)
    data = rand_vectors(rng, oracle.F32, n, dim)
    adj = random_graph(rng, n, R)
    holes = rng.choice(n, 700, replace=False)
    tagged = oracle.Index(oracle.F32, oracle.L2, dim, n, R, data[:1], row_stride=stride, tags=True)
    tagged.set_rows(0, data)
    tagged.adj[:] = adj
    for h in holes:
        tagged.set_tags(int(h), [int(rng.integers(0, 254))])
    clean = oracle.Index(oracle.F32, oracle.L2, dim, n, R, data[:1])
    clean.set_rows(0, data)
    hole_mask = np.zeros(n + 1, bool)
    hole_mask[holes] = True
    for i in range(n + 1):
        ids = adj[i, 1:1 + adj[i, 0]]
        clean.set_neighbors(i, ids[~hole_mask[ids]])
    q = rand_vectors(rng, oracle.F32, 30, dim)
    for L, W in ((10, 1), (50, 1), (50, 3)):
        a = tagged.search_batch(q, L, W, 10)
        b = clean.search_batch(q, L, W, 10)
        for x, y in zip(a, b):
            assert np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x,
                                  y.view(np.uint32) if y.dtype == np.float32 else y), (L, W)
    # all slots published: tags change nothing
    tagged.set_tags(0, np.full(n, 254, np.uint8))
    plain = oracle.Index(oracle.F32, oracle.L2, dim, n, R, data[:1])
    plain.set_rows(0, data)
    plain.adj[:] = adj
    a, b = tagged.search_batch(q, 40, 2, 10), plain.search_batch(q, 40, 2, 10)
    assert all(np.array_equal(x, y) for x, y in zip(a[::2], b[::2]))


class _Draws:
    """deterministic stand-in for the per-chunk StdRng of the PQ trainer (the same draws go to the oracle and the GPU)"""

    def __init__(self, seed, nchunks):
        self.g = [np.random.default_rng(seed + c) for c in range(nchunks)]
        self.log = []

    def index(self, c, n):
        v = int(self.g[c].integers(0, n))
        self.log.append(("i", c, n, v))
        return v

    def f64(self, c, high):
        v = float(self.g[c].random() * high)
        self.log.append(("f", c, high, v))
        return v


def test_kmeanspp_post_conditions_and_d2_sampling():
    """kmeans_plusplus_into_inner (plusplus.rs:366-497) as the reference's own tests state it: every selected centre is a
    dataset row, no row twice; fewer distinct rows than centres -> the remaining centres stay zero (recoverable
    InsufficientDiversity / DatasetTooSmall); and the D^2 rule itself, checked against a straight numpy re-derivation of
    the running sums (same draws)."""
    rng = np.random.default_rng(11)
    x = rng.standard_normal((700, 20)).astype(np.float32)
    off = [0, 7, 12, 20]
    d = _Draws(5, 3)
    rc, cen, sel = oracle.pq_kmeanspp(x, off, 32, d.index, d.f64)
    assert rc == 0 and sel.tolist() == [32, 32, 32]
    for c in range(3):
        a, b = off[c], off[c + 1]
        rows = {v.tobytes() for v in x[:, a:b]}
        got = [v.tobytes() for v in cen[:, a:b]]
        assert all(g in rows for g in got) and len(set(got)) == 32
    # D^2 sampling re-derived: exact squared distances in f64 differ from the kernel's f32 values only in the last
    # bits, so the pick is the same except when the threshold falls within rounding of a running sum (not the case here)
    c, (a, b) = 0, (off[0], off[1])
    xs = x[:, a:b].astype(np.float64)
    draws = [e for e in d.log if e[1] == c]
    first = draws[0][3]
    assert np.array_equal(cen[0, a:b], x[first, a:b])
    mins = ((xs - xs[first]) ** 2).sum(1)
    picked = {first}
    for k in range(1, 6):
        thr = draws[k][3]
        run = np.cumsum(mins)
        cand = [i for i in range(x.shape[0]) if run[i] >= thr and mins[i] > 0 and i not in picked]
        assert np.array_equal(cen[k, a:b], x[cand[0], a:b]), k
        picked.add(cand[0])
        mins = np.minimum(mins, ((xs - xs[cand[0]]) ** 2).sum(1))
    # too few distinct rows
    few = np.repeat(rng.integers(-3, 4, (5, 8)).astype(np.float32), 20, axis=0)  # integer entries: duplicate rows are at distance exactly 0
    d2 = _Draws(9, 1)
    rc, cen, sel = oracle.pq_kmeanspp(few, [0, 8], 12, d2.index, d2.f64)
    assert rc == 0 and sel[0] == 5 and not cen[5:].any() and len({v.tobytes() for v in cen[:5]}) == 5
    # fewer rows than centres
    rc, cen, sel = oracle.pq_kmeanspp(x[:6], off, 9, _Draws(1, 3).index, _Draws(1, 3).f64)
    assert rc == 0 and (sel == 6).all() and not cen[6:].any()
    # a non-finite total is the unrecoverable SawInfinity
    bad = x[:50].copy()
    bad[7, 3] = np.inf
    rc, _, _ = oracle.pq_kmeanspp(bad, off, 4, _Draws(2, 3).index, _Draws(2, 3).f64)
    assert rc == -2
