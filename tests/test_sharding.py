"""Multi-GPU path on CPU: world_size-2 gloo run of the query-sharding logic.  The search callable
is the oracle here (no GPU in this container); on the GPU box the same function wraps
Provider.search (tests/test_gpu_sharding.py, two ranks on one GPU)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_matches_reference_rule():
    from diskann_amd.sharding import partition
    for n in (0, 1, 7, 10, 100, 10001):
        for t in (1, 2, 3, 8):
            ranges = [partition(n, t, i) for i in range(t)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(t - 1))
            lens = [b - a for a, b in ranges]
            assert max(lens) - min(lens) <= 1 and lens == sorted(lens, reverse=True)
    import pytest
    with pytest.raises(ValueError):
        partition(10, 2, 2)


WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch.distributed as dist
import oracle
from helpers import rand_vectors, random_graph
from diskann_amd.sharding import search_sharded
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(42)
n, dim, R = 1500, 24, 12
data = rand_vectors(rng, oracle.F32, n, dim)
adj = random_graph(rng, n, R)
ix = oracle.Index(oracle.F32, oracle.L2, dim, n, R, data[:1]); ix.set_rows(0, data); ix.adj[:] = adj
q = rand_vectors(rng, oracle.F32, 101, dim)
fn = lambda qs: ix.search_batch(qs, 20, 1, 5)[:2] if len(qs) else (np.zeros((0, 5), np.uint32), np.zeros((0, 5), np.float32))
ids, d = search_sharded(fn, q, 5, rank, world)
ref_ids, ref_d = ix.search_batch(q, 20, 1, 5)[:2]
assert np.array_equal(ids, ref_ids) and np.array_equal(d.view(np.uint32), ref_d.view(np.uint32))
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_sharded_search_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", str(script), ROOT]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


def test_bench_launcher_rules(monkeypatch):
    """bench.py --gpus N: under a launcher it must agree with WORLD_SIZE; without one and N <= 1 it runs in-process."""
    import argparse
    import sys as _sys
    _sys.path.insert(0, ROOT)
    import bench
    import pytest
    monkeypatch.setenv("WORLD_SIZE", "2")
    bench.maybe_spawn(argparse.Namespace(gpus=2))
    bench.maybe_spawn(argparse.Namespace(gpus=None))
    with pytest.raises(SystemExit):
        bench.maybe_spawn(argparse.Namespace(gpus=8))
    monkeypatch.delenv("WORLD_SIZE")
    bench.maybe_spawn(argparse.Namespace(gpus=1))
    bench.maybe_spawn(argparse.Namespace(gpus=None))


WORKER_RERANK = r"""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch.distributed as dist
import oracle
from diskann_amd.sharding import rerank_sharded, partition
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(9)
n, dim, L, k = 900, 20, 24, 7
rows = rng.standard_normal((n, dim)).astype(np.float32)
bounds = [partition(n, world, r)[0] for r in range(world)] + [n]


class OracleShard:  # stands in for the HIP provider holding this rank's f32 rows (no GPU in this container)
    def __init__(self, lo, hi):
        self.dim = dim
        self.ix = oracle.Index(oracle.F32, oracle.L2, dim, hi - lo, 1, rows[lo:lo + 1])
        self.ix.set_rows(0, rows[lo:hi])

    def expand_beam_batch(self, queries, ids, offsets):
        out = np.empty(ids.size, np.float32)
        for qi in range(len(offsets) - 1):
            a, b = int(offsets[qi]), int(offsets[qi + 1])
            if b > a:
                out[a:b] = self.ix.expand_beam(queries[qi], ids[a:b])[1]
        return out


shard = OracleShard(bounds[rank], bounds[rank + 1])
queries = rng.standard_normal((31, dim)).astype(np.float32)
cand = rng.integers(0, n, (31, L)).astype(np.uint32)
cand[:, -3:] = 0xFFFFFFFF                     # padding
cand[5, :4] = cand[5, 4]                      # duplicates keep their list order
qlo, qhi = partition(31, world, rank)
ids, d = rerank_sharded(shard, bounds, queries[qlo:qhi], cand[qlo:qhi], k, rank, world)
for qi in range(qlo, qhi):
    c = cand[qi][cand[qi] != 0xFFFFFFFF]
    dd = np.array([oracle.query_distance(oracle.F32, oracle.L2, queries[qi], rows[j]) for j in c], np.float32)
    order = np.lexsort((np.arange(c.size), dd))[:k]
    assert np.array_equal(ids[qi - qlo], c[order]) and np.array_equal(d[qi - qlo], dd[order]), qi
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_sharded_rerank_gloo_world2(tmp_path):
    """owner-computes Rerank over partitioned f32 rows (config 5's layout), two ranks over gloo, oracle-backed shards"""
    script = tmp_path / "worker_rerank.py"
    script.write_text(WORKER_RERANK)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29537", str(script), ROOT]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("ok") == 2


BUILD_WORKER = r"""
import ctypes as C, os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from diskann_amd.sharding import build_sharded
from diskann_amd._ffi import BuildConfig


def view(ptr, rows, cols):  # the int32 matrix behind a tensor's data_ptr
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int32)), shape=(rows, cols))


class HostGraph:
    # Host-memory stand-in for Provider (device -1) with the same two-phase insert interface: the "candidates" of a
    # slot are a fixed pseudo-random list of earlier slots, a list that overflows is "pruned" to its smallest ids.
    # What the test checks is the exchange protocol of build_sharded, not RobustPrune.
    def __init__(self, n, R, deg):
        self.device, self.max_degree, self.deg = -1, R, deg
        self.adj = np.zeros((n, R + 1), np.int32)

    def cands(self, s):
        if s == 0:
            return []
        rng = np.random.default_rng(1000 + int(s))
        return sorted(set(int(x) for x in rng.integers(0, s, self.deg)))

    def insert_batch_candidates(self, cfg, slots, lo, hi, d_out):
        out = view(d_out, hi - lo, self.deg + 1) if hi > lo else None
        for i in range(lo, hi):
            c = self.cands(slots[i])
            out[i - lo, 0] = len(c)
            out[i - lo, 1:1 + len(c)] = c

    def _commit(self, slots, d_pending, rank, world):
        pend = view(d_pending, len(slots), self.deg + 1).copy()
        for i, s in enumerate(slots):
            self.adj[s, 0] = pend[i, 0]
            self.adj[s, 1:1 + pend[i, 0]] = pend[i, 1:1 + pend[i, 0]]
        back = {}
        for i, s in enumerate(slots):
            for t in pend[i, 1:1 + pend[i, 0]]:
                back.setdefault(int(t), []).append(int(s))
        rewritten = []
        for t in sorted(back):
            cur = list(self.adj[t, 1:1 + self.adj[t, 0]])
            new = cur + [s for s in back[t] if s not in cur]
            if len(new) <= self.max_degree:  # fits: every replica appends
                self.adj[t, 0] = len(new); self.adj[t, 1:1 + len(new)] = new
            elif world == 1 or t % world == rank:  # "prune": only the owner
                new = sorted(new)[: self.max_degree - 2]
                self.adj[t, :] = 0
                self.adj[t, 0] = len(new); self.adj[t, 1:1 + len(new)] = new
                rewritten.append(t)
        return rewritten

    def insert_batch_commit(self, cfg, slots, d_pending):
        self._commit(slots, d_pending, 0, 1)

    def insert_batch_commit_part(self, cfg, slots, d_pending, rank, world, d_rows, cap):
        rw = self._commit(slots, d_pending, rank, world)
        assert len(rw) <= cap
        if rw:
            out = view(d_rows, len(rw), self.max_degree + 2)
            for i, t in enumerate(rw):
                out[i, 0] = t
                out[i, 1:] = self.adj[t]
        return len(rw)

    def apply_neighbor_rows(self, d_rows, count):
        rows = view(d_rows, count, self.max_degree + 2)
        for r in rows:
            self.adj[r[0]] = r[1:]


dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n, R, deg = 700, 12, 6
cfg = BuildConfig(); cfg.pruned_degree = deg; cfg.max_degree = R
ref = HostGraph(n, R, deg)
nb_ref = build_sharded(ref, cfg, 0, n, 0.2, 64)  # one rank: the plain commit
assert ref.adj[:, 0].max() >= R - 2, "the test must exercise the prune branch"
for owner in (True, False):
    g = HostGraph(n, R, deg)
    st = {}
    nb = build_sharded(g, cfg, 0, n, 0.2, 64, rank, world, stats=st, owner_prunes=owner)
    assert nb == nb_ref and np.array_equal(g.adj, ref.adj), (rank, owner)
    assert (st.get("rows_rewritten", 0) > 0) == owner and st["rounds"] == nb
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_sharded_build_protocol_gloo_world2(tmp_path):
    """build_sharded's two exchanges (pending rows; rows rewritten by their owners) at world = 2 over gloo, driven by a
    host-memory stand-in for the provider: every replica ends identical to the single-rank build, with and without
    the owner-partitioned prunes.  (The HIP provider takes the same path in tests/test_gpu_sharding.py.)"""
    script = tmp_path / "build_worker.py"
    script.write_text(BUILD_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29537", str(script), ROOT]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2
