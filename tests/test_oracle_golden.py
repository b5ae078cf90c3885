"""Pin the CPU oracle against the reference's own golden vectors."""
import json
import os

import numpy as np
import pytest

import oracle
from gridutil import grid_data, grid_neighbors, grid_start_point


def _grid_index(dims, size, row_stride=None):
    data = grid_data(dims, size)
    n = data.shape[0]
    # Provider::grid: max_degree = 2*dims, start id linked to the last point
    ix = oracle.Index(oracle.F32, oracle.L2, dims, n, 2 * dims, grid_start_point(dims, size), row_stride=row_stride)
    ix.set_rows(0, data)
    for i, nb in enumerate(grid_neighbors(dims, size)):
        ix.set_neighbors(i, nb)
    ix.set_neighbors(n, [n - 1])
    return ix


def _cases(golden_dir):
    return json.load(open(os.path.join(golden_dir, "grid_search.json")))


def test_grid_search_golden(golden_dir):
    cases = _cases(golden_dir)
    assert len(cases) == 18
    for case in cases:
        ix = _grid_index(case["grid_dims"], case["grid_size"])
        n, ids, dists, stats = ix.search(np.array(case["query"], np.float32), case["l_value"],
                                         case["beam_width"], case["k"])
        want = case["results"]
        assert n == len(want) == case["num_results"]
        assert [int(i) for i in ids[:n]] == [w[0] for w in want], case
        assert [float(d) for d in dists[:n]] == [w[1] for w in want], case
        assert int(stats[0]) == case["comparisons"], case
        assert int(stats[1]) == case["hops"], case


def test_grid_search_inmem2_stride(golden_dir):
    """Same answers when rows use the diskann-inmem stride (tag byte + pad to 32 B)."""
    for case in _cases(golden_dir)[:6]:
        stride = oracle.inmem2_stride(oracle.F32, case["grid_dims"])
        ix = _grid_index(case["grid_dims"], case["grid_size"], row_stride=stride)
        n, ids, dists, stats = ix.search(np.array(case["query"], np.float32), case["l_value"],
                                         case["beam_width"], case["k"])
        assert [int(i) for i in ids[:n]] == [w[0] for w in case["results"]]
        assert int(stats[0]) == case["comparisons"] and int(stats[1]) == case["hops"]


def test_f16_table(golden_dir):
    """diskann-wide/test_data/float16_conversion.txt: exhaustive f16 -> f32."""
    bits = np.load(os.path.join(golden_dir, "f16_to_f32.npz"))["f32_bits"]
    L = oracle.lib()
    got = np.array([L.orc_f16_to_f32(h) for h in range(65536)], dtype=np.float32).view(np.uint32)
    want = bits
    nan = np.isnan(want.view(np.float32))
    assert np.array_equal(got[~nan], want[~nan])
    assert np.isnan(got.view(np.float32)[nan]).all()
    # and numpy agrees with both (IEEE widening)
    assert np.array_equal(np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float32).view(np.uint32)[~nan],
                          want[~nan])
    # f32 -> f16 round-to-nearest-even round-trips every non-NaN pattern
    back = np.array([L.orc_f32_to_f16(float(np.uint32(b).view(np.float32))) for b in want[~nan]], dtype=np.uint16)
    assert np.array_equal(back, np.arange(65536, dtype=np.uint16)[~nan])


def test_provider_smoke_search():
    """diskann-inmem/src/provider.rs:1080-1256 `smoke`: 5x5 grid, degree 6, l_build 10,
    pruned_degree 4, inserted one by one; searching [0,0] with L=10 returns
    (0,0.0), then the two distance-1 points, then the distance-2 point."""
    dims, size = 2, 5
    data = grid_data(dims, size)
    ix = oracle.Index(oracle.F32, oracle.L2, dims, size ** dims, 6, grid_start_point(dims, size))
    cfg = oracle.build_config(pruned_degree=4, max_degree=6, l_build=10)
    for i in range(data.shape[0]):
        ix.set_row(i, data[i])
        ix.insert(cfg, i)
    n, ids, dists, stats = ix.search(np.zeros(2, np.float32), 10, 1, 10)
    ext = [10 * int(i) + 1 for i in ids[:n]]
    assert (ext[0], dists[0]) == (1, 0.0)
    assert sorted(ext[1:3]) == [11, 51] and dists[1] == 1.0 and dists[2] == 1.0
    assert (ext[3], dists[3]) == (61, 2.0)


def test_range_search_golden(golden_dir):
    """diskann/test/generated/graph/test/cases/range_search/*.json through the oracle's Range search."""
    cases = json.load(open(os.path.join(golden_dir, "range_search.json")))
    assert len(cases) == 5
    for c in cases:
        ix = _grid_index(c["grid_dims"], c["grid_size"])
        ids, dists, stats = ix.range_search(np.array(c["query"], np.float32), c["starting_l"], c["radius"],
                                            inner_radius=c["inner_radius"], max_returned=c["max_returned"])
        assert [int(i) for i in ids] == [r[0] for r in c["results"]], c["name"]
        assert [float(d) for d in dists] == [r[1] for r in c["results"]], c["name"]
        assert int(stats[0]) == c["comparisons"] and int(stats[1]) == c["hops"], (c["name"], stats)
        assert int(stats[2]) == c["result_count"] and bool(stats[3]) == c["second_round"], c["name"]
