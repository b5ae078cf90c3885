"""The restated Rust unstable sort / selection (oracle/rust_unstable_sort.h; tie rule 6 of the oracle) on its own.

The standard library source is not in the image: what pins the restatement against the reference are the fifteen
grid_insert goldens (tests/test_oracle_build.py).  This file checks what can be checked without them: the two
comparator lists are sorting networks (0-1 principle), every entry point returns a sorted permutation whose head is the
head of a stable sort wherever distances are distinct (so on tie-free pools rule 6 and the oracle's own rule agree), and
the small documented cases (insertion sort up to 20 entries, first maximum to the back, runs kept / reversed)."""
import itertools

import numpy as np
import pytest

import oracle


def _pairs(d):
    d = np.asarray(d, np.float32)
    return np.arange(d.size, dtype=np.uint32), d


@pytest.mark.parametrize("n", [9, 13])
def test_networks_sort_every_zero_one_input(n):
    """0-1 principle: a comparator network that sorts all 2^n inputs of zeros and ones sorts everything"""
    for bits in itertools.product((0.0, 1.0), repeat=n):
        ids, d = _pairs(bits)
        _, out = oracle.rust_sort(oracle.RUST_SMALL_SORT, ids, d)
        assert np.all(out[:-1] <= out[1:]), bits


def test_every_entry_point_returns_a_sorted_permutation():
    rng = np.random.default_rng(11)
    for trial in range(600):
        n = int(rng.integers(1, 33 if trial % 3 == 0 else 700))
        levels = int(rng.choice([2, 3, 8, 50, 10 ** 6]))
        ids, d = _pairs(rng.integers(0, levels, n))
        if trial % 5 == 0:
            d = np.sort(d)[::-1].copy() if trial % 10 == 0 else np.sort(d)
        if n <= 32:
            si, sd = oracle.rust_sort(oracle.RUST_SMALL_SORT, ids, d)
            assert np.all(sd[:-1] <= sd[1:]) and sorted(si.tolist()) == ids.tolist() and np.array_equal(d[si], sd)
        si, sd = oracle.rust_sort(oracle.RUST_SORT_UNSTABLE, ids, d)
        assert np.all(sd[:-1] <= sd[1:]) and sorted(si.tolist()) == ids.tolist() and np.array_equal(d[si], sd)
        idx = int(rng.integers(0, n))
        si, sd = oracle.rust_sort(oracle.RUST_SELECT_NTH, ids, d, idx)
        assert sorted(si.tolist()) == ids.tolist() and np.array_equal(d[si], sd)
        assert sd[idx] == np.sort(d)[idx] and np.all(sd[:idx] <= sd[idx]) and np.all(sd[idx:] >= sd[idx])
        mx = int(rng.integers(0, n + 5))
        si, sd = oracle.rust_sort(oracle.RUST_SORTED_NEIGHBORS, ids, d, mx)
        assert si.size == min(mx, n) and np.array_equal(sd, np.sort(d)[:si.size]) and np.array_equal(d[si], sd)
        assert len(set(si.tolist())) == si.size


def test_tie_free_pools_do_not_depend_on_the_rule():
    """with distinct distances SortedNeighbors::new has one answer: Rust's order == the stable sort == the product's"""
    rng = np.random.default_rng(5)
    for n in (1, 2, 20, 21, 33, 64, 65, 257, 1000):
        d = rng.permutation(n).astype(np.float32)
        ids = rng.permutation(n).astype(np.uint32)
        for mx in (1, n // 2 + 1, n, n + 3):
            si, sd = oracle.rust_sort(oracle.RUST_SORTED_NEIGHBORS, ids, d, mx)
            order = np.argsort(d, kind="stable")[:mx]
            assert np.array_equal(si, ids[order]) and np.array_equal(sd, d[order])


def test_documented_small_cases():
    # up to 20 entries sort_unstable is an insertion sort: equal distances keep their order
    ids, d = _pairs([2, 1, 2, 1, 2, 1, 0, 0, 2, 1, 0, 1, 2, 0, 1, 2, 0, 1, 2, 0])
    si, _ = oracle.rust_sort(oracle.RUST_SORT_UNSTABLE, ids, d)
    assert si.tolist() == np.argsort(d, kind="stable").tolist()
    # select_nth_unstable(len - 1) swaps the FIRST maximum with the last element and touches nothing else
    ids, d = _pairs([3, 9, 1, 9, 4])
    si, _ = oracle.rust_sort(oracle.RUST_SELECT_NTH, ids, d, 4)
    assert si.tolist() == [0, 4, 2, 3, 1]
    # ... (0) swaps the FIRST minimum to the front
    ids, d = _pairs([3, 1, 7, 1, 4])
    si, _ = oracle.rust_sort(oracle.RUST_SELECT_NTH, ids, d, 0)
    assert si.tolist() == [1, 0, 2, 3, 4]
    # a non-descending slice of more than 20 entries is kept as it is, a strictly descending one is reversed
    ids, d = _pairs([0] * 10 + [1] * 10 + [2] * 10)
    si, _ = oracle.rust_sort(oracle.RUST_SORT_UNSTABLE, ids, d)
    assert si.tolist() == list(range(30))
    ids, d = _pairs(np.arange(30, 0, -1))
    si, _ = oracle.rust_sort(oracle.RUST_SORT_UNSTABLE, ids, d)
    assert si.tolist() == list(range(29, -1, -1))
    # a whole pool kept by SortedNeighbors::new: the first maximum (entry 1) changes places with the last entry (4),
    # which is why the two entries at distance 1 come out as 4, 3; the rest is an insertion sort
    ids, d = _pairs([5, 7, 7, 1, 1])
    si, sd = oracle.rust_sort(oracle.RUST_SORTED_NEIGHBORS, ids, d, 10)
    assert si.tolist() == [4, 3, 0, 2, 1] and sd.tolist() == [1, 1, 5, 7, 7]


def test_rule_six_is_not_stable_on_tied_pools():
    """the reason the rule exists: beyond 20 entries Rust's order of equal distances is not pool order"""
    rng = np.random.default_rng(3)
    differs = 0
    for _ in range(50):
        ids, d = _pairs(rng.integers(0, 4, 60))
        si, _ = oracle.rust_sort(oracle.RUST_SORTED_NEIGHBORS, ids, d, 60)
        differs += int(si.tolist() != np.argsort(d, kind="stable").tolist())
    assert differs > 40


def test_reference_sorted_neighbors_unit_test():
    """sorted_neighbors.rs:77-109 (`test_sorted_neighbors`): ten neighbours (id i, distance i / 10) shuffled, max = 0 .. 11:
    SortedNeighbors::new leaves the first min(10, max) of the sorted list and truncates the vector (any shuffle will do:
    the distances are distinct); :68-75 (`test_empty_neighbors`): an empty vector stays empty for every max."""
    ids = np.arange(1, 11, dtype=np.uint32)
    d = (ids / 10.0).astype(np.float32)
    rng = np.random.default_rng(0xD6152FB9)
    for mx in range(0, 12):
        for _ in range(10):
            p = rng.permutation(10)
            si, sd = oracle.rust_sort(oracle.RUST_SORTED_NEIGHBORS, ids[p], d[p], mx)
            n = min(10, mx)
            assert si.tolist() == ids[:n].tolist() and sd.tolist() == d[:n].tolist()
    for mx in range(10):
        si, _ = oracle.rust_sort(oracle.RUST_SORTED_NEIGHBORS, np.zeros(0, np.uint32), np.zeros(0, np.float32), mx)
        assert si.size == 0
