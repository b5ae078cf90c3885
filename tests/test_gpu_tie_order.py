"""dann_set_prune_tie_order: by default (DANN_TIE_RUST) the GPU prune orders equal-distance candidates the way the reference's
own sort does (csrc/rust_order.h).  The headline: the reference's fifteen grid_insert goldens -- twelve of them integer
lattices on which nearly every pool has ties -- built on the GPU and searched on the GPU reproduce the golden files:
ids, distances, comparisons and hops of both post-build searches, and an adjacency identical to the oracle's, whose
set_neighbors / append_neighbors / get_neighbors counters are the goldens' (tests/test_oracle_build.py).  Under
DANN_TIE_POSITION (the faster sort) the same builds equal the oracle under the position rule."""
import json
import os
import re

import numpy as np
import pytest

import oracle
from gridutil import grid_data, grid_start_point
from helpers import make_pair, rand_vectors, random_graph

pytestmark = pytest.mark.gpu
da = pytest.importorskip("diskann_amd")


def _case(f):
    p = f["payload"]
    dims, size = p["grid_dims"], p["grid_size"]
    m = re.search(r"insert_\d+_\d+_(single|batch_(\d+))/ibc_(\w+)\.json", f["source"])
    batch = None if m.group(1) == "single" else int(m.group(2))
    ibc = {"none": oracle.IBC_NONE, "all": oracle.IBC_ALL, "max_4": 4}[m.group(3)]
    data = grid_data(dims, size)
    deg = 2 * dims
    target = min(max(deg - 2, 2), deg)  # grid_insert.rs:83-86
    return p, dims, size, batch, ibc, data, deg, target


def _build_both(f, gpu_order, oracle_rule):
    p, dims, size, batch, ibc, data, deg, target = _case(f)
    n = data.shape[0]
    start = grid_start_point(dims, size)
    oix = oracle.Index(oracle.F32, oracle.L2, dims, n, deg, start)
    oix.set_rows(0, data)
    gix = da.Provider(da.F32, da.L2, dims, n, deg, start)
    gix.set_elements(0, data)
    gix.set_prune_tie_order(gpu_order)
    ocfg = oracle.build_config(target, deg, 100, intra_batch_candidates=ibc)
    gcfg = da.build_config(target, deg, 100, intra_batch_candidates=ibc)
    cnt = np.zeros(5, np.uint64)
    oracle.set_tie_rule(oracle_rule, 0)
    if batch is None:
        for i in range(n):
            oix.insert(ocfg, i, cnt)
            gix.insert_batch(gcfg, [i])     # DiskANNIndex::insert == multi_insert of one point
    else:
        for s in range(0, n, batch):
            slots = np.arange(s, min(s + batch, n), dtype=np.uint32)
            oix.multi_insert(ocfg, slots, cnt)
            gix.insert_batch(gcfg, slots)
    return p, oix, gix, cnt


def _same_graph(gix, oix):
    g, o = gix.download_graph(), oix.adj
    if not np.array_equal(g[:, 0], o[:, 0]):
        return False
    mask = np.arange(o.shape[1] - 1)[None, :] < o[:, :1]
    return bool(np.array_equal(g[:, 1:][mask], o[:, 1:][mask]))


def test_grid_insert_goldens_on_the_gpu_in_the_references_tie_order(golden_dir):
    files = json.load(open(os.path.join(golden_dir, "grid_insert.json")))
    assert len(files) == 15
    try:
        for f in files:
            p, oix, gix, cnt = _build_both(f, da.TIE_RUST, 6)
            m = p["insert_metrics"]
            # the oracle under Rust's order holds the golden's counters ...
            assert [int(cnt[2]), int(cnt[3]), int(cnt[4])] == [m["set_neighbors"], m["append_neighbors"], m["get_neighbors"]], f["test"]
            # ... and the GPU built the same graph, list by list in the same order
            assert _same_graph(gix, oix), f["test"]
            # no pool's walk ran into the selection's last resort (restated as a sort: dann_build_counters()[10])
            assert int(gix.build_counters()[10]) == 0, f["test"]
            for sc in p["searches"]:
                ids, dists, st = gix.search(da.Knn(10, sc["beam_width"]), np.array(sc["query"], np.float32), 10)
                k = sc["num_results"]
                assert int(st["written"][0]) == k
                assert [int(i) for i in ids[0][:k]] == [w[0] for w in sc["results"]], f["test"]
                assert [float(d) for d in dists[0][:k]] == [w[1] for w in sc["results"]], f["test"]
                assert int(st["cmps"][0]) == sc["comparisons"] and int(st["hops"][0]) == sc["hops"], f["test"]
    finally:
        oracle.set_tie_rule()


def test_grid_insert_goldens_on_the_gpu_in_pool_order(golden_dir):
    """DANN_TIE_POSITION (the faster sort) on the same tie-heavy builds: the GPU equals the oracle under the position rule"""
    files = json.load(open(os.path.join(golden_dir, "grid_insert.json")))
    try:
        for f in files:
            _, oix, gix, _ = _build_both(f, da.TIE_POSITION, oracle.POSITION_TIE_RULE)
            assert _same_graph(gix, oix), f["test"]
    finally:
        oracle.set_tie_rule()


def test_the_default_order_is_the_references():
    """nothing set on either side: oracle and product both follow Rust's order (a lattice batch with ties in every pool)"""
    data = grid_data(3, 5)
    n, deg = data.shape[0], 6
    start = grid_start_point(3, 5)
    oix = oracle.Index(oracle.F32, oracle.L2, 3, n, deg, start)
    oix.set_rows(0, data)
    gix = da.Provider(da.F32, da.L2, 3, n, deg, start)
    gix.set_elements(0, data)
    cnt = np.zeros(5, np.uint64)
    for s in range(0, n, 25):
        slots = np.arange(s, min(s + 25, n), dtype=np.uint32)
        oix.multi_insert(oracle.build_config(4, deg, 100, intra_batch_candidates=oracle.IBC_NONE), slots, cnt)
        gix.insert_batch(da.build_config(4, deg, 100, intra_batch_candidates=da.IBC_NONE), slots)
    assert [int(cnt[2]), int(cnt[3])] == [133, 131]   # insert_3_5_batch_25/ibc_none of the reference's goldens
    assert _same_graph(gix, oix)


@pytest.mark.parametrize("dtype,metric,dim", [(oracle.F32, oracle.L2, 24), (oracle.U8, oracle.L2, 16),
                                              (oracle.F32, oracle.INNER_PRODUCT, 12), (oracle.F32, oracle.L2, 256)])
def test_prune_batch_in_the_references_tie_order(dtype, metric, dim):
    """explicit pools with few distinct distances (small integer coordinates, repeated rows), lengths on both sides of
    every threshold of the sort (20 / 21 insertion, 32 / 33 small sort, 64 pseudo-median, max_occlusion cut), row kernel
    (dim 24 / 16 / 12) and the matrix-core path (1 KiB rows): GPU == oracle under tie rule 6"""
    rng = np.random.default_rng(5)
    n, R, maxdeg = 1600, 16, 20
    if dtype == oracle.U8:
        data = rng.integers(0, 3, (n, dim)).astype(np.uint8)
    else:
        data = rng.integers(0, 3, (n, dim)).astype(np.float32)
    data[1::4] = data[0:-1:4][: data[1::4].shape[0]]
    adj = random_graph(rng, n, maxdeg)
    oix, gix = make_pair(dtype, metric, data, adj, data[:1], maxdeg)
    gix.set_prune_tie_order(da.TIE_RUST)
    try:
        oracle.set_tie_rule(6, 0)
        for occl in (750, 100):
            ocfg = oracle.build_config(R, maxdeg, 50, max_occlusion_size=occl)
            gcfg = da.build_config(R, maxdeg, 50, max_occlusion_size=occl)
            sizes = [1, 2, 9, 13, 20, 21, 22, 32, 33, 40, 63, 64, 65, 100, 129, 257, 400, 700, 900]
            locs = rng.choice(n, len(sizes), replace=False).astype(np.uint32)
            pools, dists, off = [], [], [0]
            for loc, m in zip(locs, sizes):
                ids = rng.choice(n, m, replace=False).astype(np.uint32)
                d = np.array([oracle.distance(dtype, metric, data[loc], data[j]) for j in ids], np.float32)
                pools.append(ids)
                dists.append(d)
                off.append(off[-1] + m)
            pid, pdd = np.concatenate(pools), np.concatenate(dists)
            for sat in (False, True):
                got = gix.prune_batch(gcfg, locs, pid, pdd, np.array(off, np.uint64), force_saturate=sat)
                for i, loc in enumerate(locs):
                    want, _ = oix.prune_pool(ocfg, int(loc), pools[i], dists[i], force_saturate=sat)
                    assert got[i, 0] == want.size, (i, sat, occl)
                    assert np.array_equal(got[i, 1:1 + want.size], want), (i, sat, occl)
    finally:
        oracle.set_tie_rule()


@pytest.mark.parametrize("dtype,dim,ibc", [(oracle.U8, 16, 4), (oracle.F32, 256, oracle.IBC_NONE), (oracle.F32, 32, oracle.IBC_ALL)])
def test_batched_build_on_integer_data_in_the_references_tie_order(dtype, dim, ibc):
    """multi_insert batches over rows with small integer coordinates (u8 rows: what SIFT-like byte data looks like to
    the prune; 1 KiB f32 rows: pool and back-edge prunes through the Gram kernels): bootstrap, pool prunes with
    intra-batch candidates and back-edge prunes all see tied pools.  GPU (DANN_TIE_RUST) == oracle (tie rule 6)."""
    rng = np.random.default_rng(9)
    n, R, maxdeg, lb = 700, 8, 10, 40
    data = rng.integers(0, 4, (n, dim)).astype(np.uint8 if dtype == oracle.U8 else np.float32)
    adj = np.zeros((n + 1, maxdeg + 1), np.uint32)
    oix, gix = make_pair(dtype, oracle.L2, data, adj, data[:1], maxdeg)
    gix.set_prune_tie_order(da.TIE_RUST)
    ocfg = oracle.build_config(R, maxdeg, lb, intra_batch_candidates=ibc)
    gcfg = da.build_config(R, maxdeg, lb, intra_batch_candidates=ibc)
    try:
        oracle.set_tie_rule(6, 0)
        s0 = 0
        for b in (1, 3, 30, 66, 200, 400):
            slots = np.arange(s0, s0 + b, dtype=np.uint32)
            oix.multi_insert(ocfg, slots)
            gix.insert_batch(gcfg, slots)
            assert _same_graph(gix, oix), (b, s0)
            s0 += b
        assert s0 == n
        assert int(gix.build_counters()[10]) == 0
    finally:
        oracle.set_tie_rule()


def test_the_two_orders_differ_only_where_distances_tie():
    """continuous data (and a start point that is not a copy of a row): no pool holds two equal distances, and the graph
    does not depend on the order"""
    rng = np.random.default_rng(31)
    n, dim, R, maxdeg, lb = 600, 20, 8, 10, 40
    data = rand_vectors(rng, oracle.F32, n + 1, dim)
    data, start = data[:n], data[n:]
    graphs = []
    for order in (da.TIE_POSITION, da.TIE_RUST):
        gix = da.Provider(da.F32, da.L2, dim, n, maxdeg, start)
        gix.set_elements(0, data)
        gix.set_prune_tie_order(order)
        gix.build(da.build_config(R, maxdeg, lb), 0, n, 2.0, 256)
        g = gix.download_graph()
        g[:, 1:][np.arange(maxdeg)[None, :] >= g[:, :1]] = 0   # slots beyond a list's length are not part of it
        graphs.append(g)
    assert np.array_equal(graphs[0], graphs[1])
    with pytest.raises(da.DannError):
        gix.set_prune_tie_order(2)
