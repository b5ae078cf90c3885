"""Synthetic lattice used by the reference's algorithm tests, restated in numpy.

Follows diskann/src/graph/test/synthetic.rs:21-346: points in row-major order with the
last coordinate varying fastest, neighbours along each axis in (-, +) order, one start
point at (size, ..., size) linked to the last grid point.
"""
import itertools

import numpy as np


def grid_data(dims, size):
    pts = np.array(list(itertools.product(range(size), repeat=dims)), dtype=np.float32)
    return pts.reshape(-1, dims)


def grid_neighbors(dims, size):
    strides = [size ** (dims - 1 - a) for a in range(dims)]
    lists = []
    for coord in itertools.product(range(size), repeat=dims):
        idx = sum(c * s for c, s in zip(coord, strides))
        nb = []
        for a in range(dims):
            if coord[a] > 0:
                nb.append(idx - strides[a])
            if coord[a] < size - 1:
                nb.append(idx + strides[a])
        lists.append(nb)
    return lists


def grid_start_point(dims, size):
    return np.full(dims, float(size), dtype=np.float32)
