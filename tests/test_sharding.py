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
