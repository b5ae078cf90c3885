"""Repro loop for the team kernels: many single-query searches at small L against the oracle; prints every mismatch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
import diskann_amd as da
from helpers import bits, make_pair, rand_vectors, random_graph
rng = np.random.default_rng(777)
n, dim, R, nstart = 6000, 128, 32, 3
dtype, metric = oracle.F32, oracle.L2
data = rand_vectors(rng, dtype, n, dim)
adj = random_graph(rng, n, R, nstart=nstart)
oix, gix = make_pair(dtype, metric, data, adj, data[:nstart], R)
queries = rand_vectors(rng, dtype, 300, dim)
bad = 0
for tune, vb in (("4", 0), ("0", 12), ("8", 12), ("0", 0)):
    os.environ["DANN_TUNE_OFF"] = tune
    gix.set_visited_bits(vb)
    for L, k in ((1, 1), (2, 1), (5, 3), (26, 10), (64, 10), (130, 10)):
        oi, od, oc, ost = oix.search_batch(queries, L, 1, k)
        for rep in range(2):
            for q in range(len(queries)):
                gi, gd, st = gix.search(da.Knn(L, 1), queries[q:q + 1], k)
                ok = (np.array_equal(gi[0], oi[q]) and st["cmps"][0] == ost[q, 0] and st["hops"][0] == ost[q, 1]
                      and st["status"][0] == 0)
                if not ok:
                    bad += 1
                    if bad < 12:
                        print(f"tune {tune} L {L} q {q} rep {rep}: ids_eq {np.array_equal(gi[0], oi[q])} cmps {st['cmps'][0]} vs {ost[q, 0]} "
                              f"hops {st['hops'][0]} vs {ost[q, 1]} status {st['status'][0]}", flush=True)
    print(f"tune {tune} vbits {vb}: mismatches so far {bad}", flush=True)
print("BAD", bad)
