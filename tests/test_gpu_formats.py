"""The reference's on-disk formats: graph file (diskann-providers/src/storage/bin.rs:234-380) and
`.bin` vectors (diskann-utils/src/io.rs:24), written by the library and re-read by an independent
Python parser, and the other way round."""
import struct

import numpy as np
import pytest

import oracle
from helpers import make_pair, rand_vectors, random_graph

pytestmark = pytest.mark.gpu
da = pytest.importorskip("diskann_amd")


def _parse_graph(path):
    raw = open(path, "rb").read()
    file_size, max_degree, start, nstart = struct.unpack_from("<QIIQ", raw, 0)
    assert file_size == len(raw)
    pos, lists = 24, []
    while pos < file_size:
        (ln,) = struct.unpack_from("<I", raw, pos)
        lists.append(np.frombuffer(raw, np.uint32, ln, pos + 4))
        pos += 4 * (1 + ln)
    return max_degree, start, nstart, lists


def test_graph_and_bin_round_trip(tmp_path):
    rng = np.random.default_rng(3)
    n, dim, R = 700, 20, 9
    data = rand_vectors(rng, oracle.F32, n, dim)
    adj = random_graph(rng, n, R, nstart=2, min_len=0)
    _, gix = make_pair(oracle.F32, oracle.L2, data, adj, data[:2], R)
    gpath, vpath = tmp_path / "index.graph", tmp_path / "vectors.bin"
    gix.save_graph(gpath)
    gix.save_vectors_bin(vpath)
    max_degree, start, nstart, lists = _parse_graph(gpath)
    assert (max_degree, start, nstart, len(lists)) == (R, n, 2, n + 2)
    for i, l in enumerate(lists):
        assert np.array_equal(l, adj[i, 1:1 + adj[i, 0]])
    raw = open(vpath, "rb").read()
    npts, d = struct.unpack_from("<II", raw, 0)
    assert (npts, d) == (n, dim)
    assert np.array_equal(np.frombuffer(raw, np.float32, n * dim, 8).reshape(n, dim), data)
    # load into a fresh index: identical search results
    g2 = da.Provider(da.F32, da.L2, dim, n, R, data[:2])
    assert g2.load_vectors_bin(vpath) == n
    assert g2.load_graph(gpath) == (n, 2, n + 2)
    q = rand_vectors(rng, oracle.F32, 16, dim)
    a = gix.search(da.Knn(30), q, 10)
    b = g2.search(da.Knn(30), q, 10)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    # dimension mismatch is an error
    g3 = da.Provider(da.F32, da.L2, dim + 1, n, R, np.zeros((1, dim + 1), np.float32))
    with pytest.raises(da.DannError):
        g3.load_vectors_bin(vpath)
