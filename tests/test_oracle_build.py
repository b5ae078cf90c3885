"""Oracle build path (insert / multi_insert / prune) against the reference's grid_insert
golden files.  1-D lattices pin the build bit-for-bit (ids, distances, comparisons, hops,
set/append counters) under every tie rule; 3-D/4-D lattices are tie-heavy and the reference's
own result depends on the order Rust's unstable sort leaves equal distances in
(internal/sorted_neighbors.rs:36-40).  With that sort restated (oracle/rust_unstable_sort.h,
tie rule 6) the oracle reproduces ALL fifteen goldens exactly: set_neighbors, append_neighbors
and get_neighbors of the build, and ids, distances, comparisons and hops of every post-build
search (test_grid_insert_all_goldens_exact_with_rust_sort); rule 6 is the oracle's default and
the product's (DANN_TIE_RUST).  Under the position rule (the product's DANN_TIE_POSITION) the
oracle's counters are recorded next to the reference's, and every counter of every golden lies
inside the range alternative tie orders span."""
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
    cnt = np.zeros(5, np.uint64)
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


def test_grid_insert_all_goldens_exact_with_rust_sort(golden_dir):
    """Tie rule 6 = Rust's select_nth_unstable_by + sort_unstable_by restated (ipnsort; oracle/rust_unstable_sort.h) and
    the bootstrap's candidate list in AdjacencyList::from_iter_untrusted's ascending order (adjacencylist.rs:181-190).
    Every one of the fifteen grid_insert goldens -- the twelve tie-heavy 3-D / 4-D lattices included -- is reproduced
    exactly: the build's set_neighbors / append_neighbors / get_neighbors counters and, for both post-build searches,
    the result ids, the distances, comparisons and hops.  The Rust standard library is not in the image; these
    reference-held vectors are what pins the restatement (1096 appends and 22 321 adjacency reads of the 4-D single-insert
    build hang on the order of every tied pool)."""
    try:
        oracle.set_tie_rule(6, 0)
        before = oracle.rust_sort_fallbacks()
        paths0 = oracle.rust_sort_paths()
        seen = 0
        for f in _files(golden_dir):
            p = f["payload"]
            ix, cnt = _build(f["source"], p)
            m = p["insert_metrics"]
            assert [int(cnt[2]), int(cnt[3]), int(cnt[4])] == [m["set_neighbors"], m["append_neighbors"], m["get_neighbors"]], f["test"]
            for sc in p["searches"]:
                k, ids, dists, st = ix.search(np.array(sc["query"], np.float32), 10, sc["beam_width"], 10)
                assert k == sc["num_results"]
                assert [int(i) for i in ids[:k]] == [w[0] for w in sc["results"]], f["test"]
                assert [float(d) for d in dists[:k]] == [w[1] for w in sc["results"]], f["test"]
                assert int(st[0]) == sc["comparisons"] and int(st[1]) == sc["hops"], f["test"]
            seen += 1
        assert seen == 15
        assert oracle.rust_sort_fallbacks() == before  # the selection's median-of-medians fallback is never reached
        # which parts of the restated sort these vectors pin, and which they do not (oracle/rust_unstable_sort.h)
        ran = {k: v - paths0[k] for k, v in oracle.rust_sort_paths().items()}
        for part in ("insertion_20", "run_kept", "quicksort", "small_network", "sort9", "sort13", "merge", "partition_lt",
                     "partition_le", "median3", "median3_rec", "select_max"):
            assert ran[part] > 0, part
        assert ran["partition_le"] > 1000 and ran["median3_rec"] > 5000 and ran["small_network"] > 10000
        for part in ("run_reversed", "heapsort", "select_min", "select_loop", "select_fallback"):
            assert ran[part] == 0, part   # not reached by any reference-held vector: restated from the published algorithm
    finally:
        oracle.set_tie_rule()


# (set_neighbors, append_neighbors) of the oracle under the position rule (equal distances keep their pool order: the
# product's DANN_TIE_POSITION) next to the reference's, per lattice golden -- measured, not a tolerance: append_neighbors
# is off by up to 12 % on the 3-D lattices.  The next test shows that this is the freedom the reference's unstable sort
# has, not a different rule; under the default rule (Rust's own order) the goldens are reproduced exactly, see above.
ORACLE_VS_REFERENCE = {
    "insert_3_5_batch_125/ibc_all": ([125, 83], [125, 81]),
    "insert_3_5_batch_125/ibc_max_4": ([125, 83], [125, 74]),
    "insert_3_5_batch_125/ibc_none": ([125, 83], [125, 76]),
    "insert_3_5_batch_25/ibc_all": ([127, 142], [127, 132]),
    "insert_3_5_batch_25/ibc_none": ([133, 139], [133, 131]),
    "insert_3_5_single/ibc_none": ([205, 389], [206, 388]),
    "insert_4_4_batch_25/ibc_all": ([292, 348], [293, 360]),
    "insert_4_4_batch_25/ibc_none": ([376, 357], [374, 362]),
    "insert_4_4_batch_256/ibc_all": ([257, 96], [257, 96]),
    "insert_4_4_batch_256/ibc_max_4": ([272, 96], [272, 96]),
    "insert_4_4_batch_256/ibc_none": ([272, 96], [272, 96]),
    "insert_4_4_single/ibc_none": ([497, 1102], [503, 1096]),
}


def _lattice_files(golden_dir):
    return [f for f in _files(golden_dir) if f["payload"]["grid_dims"] != 1]


def _tuple(f):
    """(set_neighbors, append_neighbors, then comparisons and hops of every post-build search) + exact searches"""
    p = f["payload"]
    ix, cnt = _build(f["source"], p)
    tup = [int(cnt[2]), int(cnt[3])]
    exact = 0
    for sc in p["searches"]:
        k, ids, dists, st = ix.search(np.array(sc["query"], np.float32), 10, sc["beam_width"], 10)
        tup += [int(st[0]), int(st[1])]
        assert k == sc["num_results"]
        # the exact nearest neighbour and its distance are tie-free, and so is the multiset of distances at the head
        assert int(ids[0]) == sc["results"][0][0] and float(dists[0]) == sc["results"][0][1]
        assert sorted(float(d) for d in dists[:5]) == sorted(w[1] for w in sc["results"][:5])
        exact += int([int(i) for i in ids[:k]] == [w[0] for w in sc["results"]])
    return tup, exact


def _reference_tuple(p):
    ref = [p["insert_metrics"]["set_neighbors"], p["insert_metrics"]["append_neighbors"]]
    for sc in p["searches"]:
        ref += [sc["comparisons"], sc["hops"]]
    return ref


def test_grid_insert_lattice_counters_as_measured(golden_dir):
    """under the position rule (the product's DANN_TIE_POSITION) the oracle's counters on the 12 tie-heavy lattice goldens
    are exactly the recorded ones (a regression pin of that rule) and the reference's are the recorded ones too (a pin of
    the table above)"""
    try:
        oracle.set_tie_rule(oracle.POSITION_TIE_RULE)
        seen = 0
        for f in _lattice_files(golden_dir):
            name = f["test"].split("grid_insert/")[1]
            mine, ref = ORACLE_VS_REFERENCE[name]
            tup, _ = _tuple(f)
            assert tup[:2] == mine, name
            assert _reference_tuple(f["payload"])[:2] == ref, name
            seen += 1
        assert seen == 12
    finally:
        oracle.set_tie_rule()


def test_grid_insert_lattice_tie_envelope(golden_dir):
    """SortedNeighbors::new (internal/sorted_neighbors.rs:26-44) sorts with select_nth_unstable_by + sort_unstable_by:
    the order of candidates at equal distance is unspecified, and on integer lattices nearly every prune has such ties.
    For every lattice golden the oracle is re-run under other tie orders (pool position descending, id ascending /
    descending, a hypothesis about Rust's small-slice path, 200 seeded shuffles): every counter the golden holds --
    set_neighbors, append_neighbors, comparisons and hops of each post-build search -- lies inside the range those
    orders span.  In particular the reference's 81 / 74 / 76 appends for the three intra-batch-candidate modes of the
    all-at-once 5x5x5 batch (the oracle: 83 / 83 / 83, because position order is id order there) are three draws from a
    60..83 range.  profiles/r04_tie_envelope.json holds the run (scratch/tie_envelope.py)."""
    try:
        for f in _lattice_files(golden_dir):
            ref = _reference_tuple(f["payload"])
            lo = hi = None
            runs = [(r, 0) for r in (0, 1, 2, 3, 5)] + [(4, s + 1) for s in range(200)]
            for rule, seed in runs:
                oracle.set_tie_rule(rule, seed)
                t, _ = _tuple(f)
                lo = t if lo is None else [min(a, b) for a, b in zip(lo, t)]
                hi = t if hi is None else [max(a, b) for a, b in zip(hi, t)]
            assert all(l <= r <= h for l, r, h in zip(lo, ref, hi)), (f["test"], ref, lo, hi)
    finally:
        oracle.set_tie_rule()


def test_small_pool_hypothesis_reproduces_the_single_insert_3d_counters(golden_dir):
    """Tie rule 5 (whole pool kept: first maximum swapped to the end, prefixes of <= 20 entries sorted by insertion --
    public descriptions of Rust >= 1.81's select_nth_unstable / sort_unstable, not checked against their source)
    reproduces both build counters of insert_3_5_single exactly where pool position is off by one in each.  Evidence
    about where the residual differences come from (pools beyond 20 entries), nothing the product builds on."""
    try:
        oracle.set_tie_rule(5, 0)
        f = [f for f in _lattice_files(golden_dir) if "insert_3_5_single" in f["test"]][0]
        tup, _ = _tuple(f)
        assert tup[:2] == _reference_tuple(f["payload"])[:2] == [206, 388]
    finally:
        oracle.set_tie_rule()


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
