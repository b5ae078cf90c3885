"""Slot readability (inline concurrency tags of the diskann-inmem Store, store.rs:133-158, tag.rs:86-133): a slot whose
tag is below PUBLISHED is skipped by expand_beam *after* the visited insert and is not counted
(provider.rs:448-473, 681-686).  A Store snapshot with holes (never-published or retired slots) uploaded verbatim
must search exactly like the CPU path."""
import numpy as np
import pytest

import oracle
from helpers import bits, rand_vectors, random_graph

pytestmark = pytest.mark.gpu
da = pytest.importorskip("diskann_amd")


def _pair_with_holes(dtype, metric, dim, R, n, rng, hole_frac, nstart=1):
    stride = oracle.inmem2_stride(dtype, dim)
    data = rand_vectors(rng, dtype, n, dim)
    adj = random_graph(rng, n, R, nstart=nstart)
    oix = oracle.Index(dtype, metric, dim, n, R, data[:nstart], row_stride=stride, tags=True)
    oix.set_rows(0, data)
    oix.adj[:] = adj
    holes = rng.choice(n, int(n * hole_frac), replace=False)
    tags = rng.choice(np.array([0, 1, 2, 253], np.uint8), holes.size)  # AVAILABLE / OWNED / RETIRING / below PUBLISHED
    for h, t in zip(holes, tags):
        oix.set_tags(int(h), [t])
    gix = da.Provider(dtype, metric, dim, n, R, data[:nstart], row_stride=stride, inline_tags=True)
    gix.upload_store(oix.rows)   # verbatim Store buffer: payloads and tag bytes
    gix.upload_graph(adj)
    return oix, gix, holes


@pytest.mark.parametrize("dtype,metric,dim,R", [
    (oracle.F32, oracle.L2, 128, 32),            # the fixed-length f32 kernel, 544-byte stride
    (oracle.F32, oracle.INNER_PRODUCT, 100, 24),
    (oracle.F16, oracle.L2, 128, 32),
    (oracle.U8, oracle.L2, 128, 32),
    (oracle.I8, oracle.COSINE, 100, 16),
])
def test_search_skips_unreadable_slots(dtype, metric, dim, R):
    rng = np.random.default_rng(900 + dim + R)
    oix, gix, holes = _pair_with_holes(dtype, metric, dim, R, 4000, rng, 0.2)
    assert np.array_equal(gix.get_tags(0, 4000), oix.rows[:4000, oix.tag_offset])
    q = rand_vectors(rng, dtype, 40, dim)
    for L, W, k in ((10, 1, 10), (64, 1, 10), (64, 4, 20), (200, 2, 50)):
        oi, od, oc, ost = oix.search_batch(q, L, W, k)
        gi, gd, gst = gix.search(da.Knn(L, W), q, k)
        assert np.array_equal(oi, gi), (L, W)
        assert np.array_equal(bits(od), bits(gd)), (L, W)
        assert np.array_equal(ost[:, 0], gst["cmps"]) and np.array_equal(ost[:, 1], gst["hops"]), (L, W)
        assert np.array_equal(oc, gst["written"]) and np.array_equal(ost[:, 2], gst["result_count"])
        assert not np.isin(gi[gi != 0xFFFFFFFF], holes).any()
    # the same store with every slot published again searches like a store without tags
    oix.set_tags(0, np.full(4000, 254, np.uint8))
    gix.set_tags(0, np.full(4000, 254, np.uint8))
    oi, od, oc, ost = oix.search_batch(q, 64, 1, 10)
    gi, gd, gst = gix.search(da.Knn(64, 1), q, 10)
    assert np.array_equal(oi, gi) and np.array_equal(ost[:, 0], gst["cmps"])


def test_expand_beam_seam_and_set_elements_publish():
    rng = np.random.default_rng(31)
    n, dim, R = 600, 48, 12
    stride = oracle.inmem2_stride(oracle.F32, dim)
    data = rand_vectors(rng, oracle.F32, n, dim)
    oix = oracle.Index(oracle.F32, oracle.L2, dim, n, R, data[:1], row_stride=stride, tags=True)
    gix = da.Provider(oracle.F32, oracle.L2, dim, n, R, data[:1], row_stride=stride, inline_tags=True)
    # only the first 400 slots are ever set: the rest stay AVAILABLE (never published)
    oix.set_rows(0, data[:400])
    gix.set_elements(0, data[:400])
    assert (gix.get_tags(0, 400) == 254).all() and (gix.get_tags(400, 200) == 0).all() and gix.get_tags(n, 1)[0] == 255
    ids = rng.choice(n, 64, replace=False).astype(np.uint32)
    q = rand_vectors(rng, oracle.F32, 1, dim)[0]
    oi, od = oix.expand_beam(q, ids)
    gi, gd = gix.expand_beam(q, ids)
    assert np.array_equal(oi, gi) and np.array_equal(bits(od), bits(gd))
    assert oi.size == int((ids < 400).sum()) and (oi < 400).all()
    # positional batch form: unreadable slots report NaN
    out = gix.expand_beam_batch(q[None, :], ids, np.array([0, ids.size], np.uint64))
    assert np.array_equal(np.isnan(out), ids >= 400)
    # searches over a graph that points into the unpublished region
    adj = random_graph(rng, n, R)
    oix.adj[:] = adj
    gix.upload_graph(adj)
    qs = rand_vectors(rng, oracle.F32, 16, dim)
    oi, od, oc, ost = oix.search_batch(qs, 32, 1, 10)
    gi, gd, gst = gix.search(da.Knn(32, 1), qs, 10)
    assert np.array_equal(oi, gi) and np.array_equal(bits(od), bits(gd)) and (gi[gi != 0xFFFFFFFF] < 400).all()
    assert np.array_equal(ost[:, 0], gst["cmps"]) and np.array_equal(ost[:, 1], gst["hops"])


def test_range_filtered_and_paged_search_respect_tags():
    rng = np.random.default_rng(77)
    dim, R, n = 32, 16, 3000
    oix, gix, holes = _pair_with_holes(oracle.F32, oracle.L2, dim, R, n, rng, 0.25)
    qs = rand_vectors(rng, oracle.F32, 8, dim)
    # range search
    radius = 3.2
    gi, gd, gst, gsec = gix.range_search(qs, 40, radius, out_cap=2048)
    for qi in range(qs.shape[0]):
        oi, od, ost = oix.range_search(qs[qi], 40, radius)
        m = int(gst["written"][qi])
        assert m == oi.size and np.array_equal(gi[qi, :m], oi) and np.array_equal(bits(gd[qi, :m]), bits(od))
        assert (int(gst["cmps"][qi]), int(gst["hops"][qi])) == (int(ost[0]), int(ost[1]))
    # inline-filter search
    match = rng.random(n + 1) < 0.5
    gi, gd, gst = gix.filtered_search(da.Knn(40, 1), qs, 10, match)
    for qi in range(qs.shape[0]):
        wn, oi, od, ost = oix.inline_filter_search(qs[qi], 40, 10, match)
        assert int(gst["written"][qi]) == wn and np.array_equal(gi[qi, :wn], oi[:wn])
        assert (int(gst["cmps"][qi]), int(gst["hops"][qi])) == (int(ost[0]), int(ost[1]))
    # paged search
    sess = gix.paged_search(qs[:3], 24)
    try:
        for qi in range(3):
            pass
        pages = [oix.paged_search(qs[qi], 24, 10, max_pages=4) for qi in range(3)]
        for p in range(4):
            ids, d, cnt = sess.next_page(10)
            for qi in range(3):
                if p < len(pages[qi]):
                    oi, od = pages[qi][p]
                    assert int(cnt[qi]) == oi.size and np.array_equal(ids[qi, :oi.size], oi)
                    assert np.array_equal(bits(d[qi, :oi.size]), bits(od))
                else:
                    assert int(cnt[qi]) == 0
    finally:
        sess.close()


def test_inline_tags_need_room_for_the_tag_byte():
    with pytest.raises(da.DannError):
        da.Provider(oracle.F32, oracle.L2, 128, 10, 4, np.zeros((1, 128), np.float32), inline_tags=True)  # packed stride
