"""Oracle build path (insert / multi_insert / prune) against the reference's grid_insert
golden files.  1-D lattices pin the build bit-for-bit (ids, distances, comparisons, hops,
set/append counters); 3-D/4-D lattices are tie-heavy and the reference's own result
depends on Rust's unstable sort (internal/sorted_neighbors.rs:36-40), so they are soft
pins (counters within 5 %, nearest neighbour found)."""
import json
import os
import re

import numpy as np
import pytest

import oracle
from gridutil import grid_data, grid_start_point


def _build(src, payload):
    dims, size = payload["grid_dims"], payload["grid_size"]
    m = re.search(r"insert_\d+_\d+_(single|batch_(\d+))/ibc_(\w+)\.json", src)
    batch = None if m.group(1) == "single" else int(m.group(2))
    ibc = {"none": oracle.IBC_NONE, "all": oracle.IBC_ALL, "max_4": 4}[m.group(3)]
    data = grid_data(dims, size)
    n = data.shape[0]
    deg = 2 * dims
    target = min(max(deg - 2, 2), deg)  # grid_insert.rs:83-86
    ix = oracle.Index(oracle.F32, oracle.L2, dims, n, deg, grid_start_point(dims, size))
    cfg = oracle.build_config(target, deg, 100, intra_batch_candidates=ibc)
    ix.set_rows(0, data)
    cnt = np.zeros(4, np.uint64)
    if batch is None:
        for i in range(n):
            ix.insert(cfg, i, cnt)
    else:
        for s in range(0, n, batch):
            ix.multi_insert(cfg, np.arange(s, min(s + batch, n)), cnt)
    return ix, cnt


def _files(golden_dir):
    return json.load(open(os.path.join(golden_dir, "grid_insert.json")))


def test_grid_insert_1d_exact(golden_dir):
    seen = 0
    for f in _files(golden_dir):
        p = f["payload"]
        if p["grid_dims"] != 1:
            continue
        seen += 1
        ix, cnt = _build(f["source"], p)
        assert int(cnt[2]) == p["insert_metrics"]["set_neighbors"]
        assert int(cnt[3]) == p["insert_metrics"]["append_neighbors"]
        for sc in p["searches"]:
            k, ids, dists, st = ix.search(np.array(sc["query"], np.float32), 10, sc["beam_width"], 10)
            assert [int(i) for i in ids[:k]] == [w[0] for w in sc["results"]]
            assert [float(d) for d in dists[:k]] == [w[1] for w in sc["results"]]
            assert int(st[0]) == sc["comparisons"] and int(st[1]) == sc["hops"]
    assert seen == 3


def test_grid_insert_lattice_soft(golden_dir):
    for f in _files(golden_dir):
        p = f["payload"]
        if p["grid_dims"] == 1:
            continue
        ix, cnt = _build(f["source"], p)
        want_set = p["insert_metrics"]["set_neighbors"]
        assert abs(int(cnt[2]) - want_set) <= max(3, 0.05 * want_set)
        for sc in p["searches"]:
            k, ids, dists, st = ix.search(np.array(sc["query"], np.float32), 10, sc["beam_width"], 10)
            assert k == sc["num_results"]
            # the exact nearest neighbour and its distance are tie-free
            assert int(ids[0]) == sc["results"][0][0] and float(dists[0]) == sc["results"][0][1]
            # same multiset of distances at the head of the list
            assert sorted(float(d) for d in dists[:5]) == sorted(w[1] for w in sc["results"][:5])


def test_prune_matches_bruteforce_rule():
    """RobustPrune result == the eager definition (Appendix A rule 10): candidate i is kept
    iff max_j d(q,i)/d(i,j) <= alpha over already kept j earlier in the pool, with the
    1.0 -> 1.2 alpha sweep.  Lazy evaluation must not change the result."""
    rng = np.random.default_rng(7)
    n, dim, R = 400, 24, 12
    data = rng.standard_normal((n, dim)).astype(np.float32)
    ix = oracle.Index(oracle.F32, oracle.L2, dim, n, 16, np.zeros(dim, np.float32))
    ix.set_rows(0, data)
    cfg = oracle.build_config(R, 16, 50)
    for loc in (0, 17, 123):
        pool = rng.choice(n, 120, replace=False).astype(np.uint32)
        pd = np.array([oracle.distance(oracle.F32, oracle.L2, data[loc], data[i]) for i in pool], np.float32)
        got, evals = ix.prune_pool(cfg, loc, pool, pd)
        order = np.argsort(pd, kind="stable")
        spool, sd = pool[order], pd[order]
        kept, state = [], {}
        for alpha in (1.0, 1.2):
            for i in range(len(spool)):
                if len(kept) >= R or spool[i] == loc or i in kept:
                    continue
                ok = True
                for j in kept:
                    if j < i:
                        dij = oracle.distance(oracle.F32, oracle.L2, data[spool[i]], data[spool[j]])
                        f = np.float32(3.4028235e38) if dij == 0 else np.float32(sd[i]) / np.float32(dij)
                        if f > np.float32(alpha):
                            ok = False
                            break
                if ok:
                    kept.append(i)
        assert [int(spool[i]) for i in kept] == [int(g) for g in got]
        assert evals > 0
