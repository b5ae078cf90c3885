"""The beam-search kernel merges a batch of candidates into the L-queue by rank (csrc/search_kernel_impl.h, `merge`)
instead of calling NeighborPriorityQueue::insert once per candidate (diskann/src/neighbor/queue.rs:130-171).  This
file restates both in plain Python and checks, on tie-heavy random inputs, that they produce the same queue -- the
exactness argument of DESIGN.md section 4 as a test (host logic, no GPU)."""
import math

import numpy as np


def sequential_insert(queue, cap, cands):
    """queue.rs:130-171 -- NaN dropped; a full queue drops a candidate worse than its last entry (`last < d`); the
    position is the first entry with distance >= d (a new element goes BEFORE equal old ones); then truncate."""
    q = list(queue)
    for d, i in cands:
        if math.isnan(d):
            continue
        if len(q) == cap and q[-1][0] < d:
            continue
        pos = 0
        while pos < len(q) and q[pos][0] < d:
            pos += 1
        q.insert(pos, (d, i))
        del q[cap:]
    return q


def rank_merge(queue, cap, cands):
    """the kernel's rule: (1) survivors = not NaN and, if the queue is full, not worse than its OLD last entry;
    (2) survivor j lands at #{old e: d_e < d_j} + #{survivors i: d_i < d_j or (d_i == d_j and i emitted after j)};
    (3) old entry e moves up by #{survivors j: d_j <= d_e}; entries at positions >= cap fall off."""
    full = len(queue) == cap and cap > 0
    surv = [(d, i) for d, i in cands if not math.isnan(d) and not (full and queue[-1][0] < d)]
    out = {}
    for e, (de, ie) in enumerate(queue):
        out[e + sum(1 for dj, _ in surv if dj <= de)] = (de, ie)
    for j, (dj, ij) in enumerate(surv):
        before = sum(1 for t, (dt, _) in enumerate(surv) if dt < dj or (dt == dj and t > j))
        out[sum(1 for de, _ in queue if de < dj) + before] = (dj, ij)
    assert sorted(out) == list(range(len(out)))  # a permutation: no two entries share a position
    return [out[p] for p in range(min(len(out), cap))]


def test_rank_merge_equals_sequential_inserts():
    rng = np.random.default_rng(20260924)
    for trial in range(3000):
        cap = int(rng.integers(1, 40))
        nq = int(rng.integers(0, cap + 1))
        levels = int(rng.integers(1, 12))  # few distinct distances: ties everywhere
        qd = np.sort(rng.integers(0, levels, nq)).astype(np.float32)
        queue = [(float(qd[e]), 1000 + e) for e in range(nq)]
        nc = int(rng.integers(0, 70))
        cd = rng.integers(0, levels + 2, nc).astype(np.float32)
        cd[rng.random(nc) < 0.05] = np.nan
        cands = [(float(cd[j]), j) for j in range(nc)]
        assert rank_merge(queue, cap, cands) == sequential_insert(queue, cap, cands), (trial, cap, queue, cands)


def test_rank_merge_keeps_insertion_time_order_among_equals():
    # three equal candidates into an empty queue: the last emitted ends up first (each insert goes before its equals)
    q = sequential_insert([], 8, [(1.0, 0), (1.0, 1), (1.0, 2)])
    assert [i for _, i in q] == [2, 1, 0]
    assert rank_merge([], 8, [(1.0, 0), (1.0, 1), (1.0, 2)]) == q
    # a full queue: a candidate equal to the last entry is inserted (not `last < d`) and pushes it out
    full = [(0.0, 10), (1.0, 11), (2.0, 12)]
    assert sequential_insert(full, 3, [(2.0, 0)]) == [(0.0, 10), (1.0, 11), (2.0, 0)] == rank_merge(full, 3, [(2.0, 0)])
