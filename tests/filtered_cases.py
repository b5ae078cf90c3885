"""Graphs and filters of the reference's filtered-search tests (diskann/src/graph/test/cases/
{inline,multihop,filtered_range_search}.rs), restated as arrays in the diskann-inmem slot layout:
points in slots [0, n), the start point in slot n.  `orig` maps slot -> id used by the reference test."""
import numpy as np

from gridutil import grid_data, grid_neighbors, grid_start_point

U32MAX = 0xFFFFFFFF


class Graph:
    def __init__(self, data, lists, start_vec, start_list, max_degree, orig):
        self.data = np.asarray(data, np.float32)
        self.lists = lists              # adjacency of slots 0..n-1 (slot ids)
        self.start_vec = np.asarray(start_vec, np.float32)
        self.start_list = start_list    # adjacency of the start slot
        self.max_degree = max_degree
        self.orig = np.asarray(orig, np.uint64)  # slot -> reference id (start slot last)
        self.n = self.data.shape[0]

    def match(self, rule):
        """boolean array over slots [0, n] for a filter given as a list of reference ids or a rule name"""
        if rule == "all":
            return np.ones(self.n + 1, bool)
        if rule == "even":
            return self.orig % 2 == 0
        if rule == "div4":
            return self.orig % 4 == 0
        return np.isin(self.orig, np.asarray(rule, np.uint64))

    def fill(self, ix):
        """ix: oracle.Index or diskann_amd.Provider created with capacity n, max_degree, start_vec"""
        ix.set_rows(0, self.data) if hasattr(ix, "set_rows") else ix.set_elements(0, self.data)
        for i, nb in enumerate(self.lists):
            ix.set_neighbors(i, nb)
        ix.set_neighbors(self.n, self.start_list)


def grid(dims, size):
    data = grid_data(dims, size)
    n = data.shape[0]
    return Graph(data, grid_neighbors(dims, size), grid_start_point(dims, size), [n - 1], 2 * dims,
                 list(range(n)) + [U32MAX])


def three_level():
    """inline.rs:56-110: start id 0 (coord 0), ids 1..14 below it; slot = id - 1, start -> slot 14"""
    adj = {0: [1, 2], 1: [0, 3, 4], 2: [0, 5, 6], 3: [1, 7, 8], 4: [1, 9, 10], 5: [2, 11, 12], 6: [2, 13, 14]}
    for leaf, parent in zip(range(7, 15), [3, 3, 4, 4, 5, 5, 6, 6]):
        adj[leaf] = [parent]
    coord = {**{i: 0.0 for i in (0, 1, 2)}, **{i: 1.0 for i in (3, 4, 5, 6)}, **{i: 2.0 for i in range(7, 15)}}
    slot = lambda i: 14 if i == 0 else i - 1
    data = [[coord[i]] for i in range(1, 15)]
    lists = [[slot(j) for j in adj[i]] for i in range(1, 15)]
    return Graph(data, lists, [coord[0]], [slot(j) for j in adj[0]], 3, list(range(1, 15)) + [0])


def hand_1d():
    """inline.rs:598-632 / multihop.rs:266-296: start id 10 at 5.0 -> slot 5"""
    s = 5
    lists = [[1, s], [0, 2, s], [1, 3], [0, 4, s], [3, 2]]
    return Graph([[0.0], [1.0], [2.0], [3.0], [4.0]], lists, [5.0], [0, 1, 3], 4, [0, 1, 2, 3, 4, 10])


def build(name, grid_dims=None, grid_size=None):
    if name == "grid1d_100":
        return grid(1, 100)
    if name == "three_level":
        return three_level()
    if name == "hand_1d":
        return hand_1d()
    if name == "grid3d":
        return grid(3, grid_size)
    if name == "grid":
        return grid(grid_dims, grid_size)
    raise KeyError(name)
