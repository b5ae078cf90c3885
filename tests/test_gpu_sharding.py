"""The N > 1 code paths with the HIP provider, on a 1-GPU box: two ranks (torch.distributed.run, gloo) share cuda:0.
  * search: Provider.search through sharding.search_sharded, gathered output == the oracle's single-stream output;
  * build: sharding.build_sharded at world = 2 (candidates partitioned, pending rows all-gathered, identical commits),
    every replica's adjacency == the oracle's multi_insert over the same batch schedule;
  * bench.py --gpus 2 spawns its own ranks and reports n_gpus 2 (DANN_BENCH_ONE_DEVICE test hook)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
import oracle, diskann_amd as da
from helpers import rand_vectors, random_graph, make_pair
from diskann_amd.sharding import search_sharded, build_sharded, batch_schedule
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
rng = np.random.default_rng(42)
# ---- sharded search ----------------------------------------------------------------------------
n, dim, R = 3000, 40, 16
data = rand_vectors(rng, oracle.F32, n, dim)
adj = random_graph(rng, n, R)
oix, gix = make_pair(oracle.F32, oracle.L2, data, adj, data[:1], R)
q = rand_vectors(rng, oracle.F32, 203, dim)
k = 7
fn = lambda qs: gix.search(da.Knn(48, 2), qs, k)[:2] if len(qs) else (np.zeros((0, k), np.uint32), np.zeros((0, k), np.float32))
ids, d = search_sharded(fn, q, k, rank, world)
ref_ids, ref_d = oix.search_batch(q, 48, 2, k)[:2]
assert np.array_equal(ids, ref_ids) and np.array_equal(d.view(np.uint32), ref_d.view(np.uint32)), "sharded search != oracle"
# ---- sharded build -----------------------------------------------------------------------------
n, dim, Rp, maxdeg, lb = 2500, 24, 8, 10, 24
data = rand_vectors(rng, oracle.F32, n, dim)
start = data.mean(0, keepdims=True).astype(np.float32)
for ibc in (da.IBC_NONE, 4):
    gcfg = da.build_config(Rp, maxdeg, lb, intra_batch_candidates=ibc)
    ocfg = oracle.build_config(Rp, maxdeg, lb, intra_batch_candidates=ibc)
    p = da.Provider(da.F32, da.L2, dim, n, maxdeg, start)
    p.set_elements(0, data)
    growth, max_batch = 0.1, 300
    nb = build_sharded(p, gcfg, 0, n, growth, max_batch, rank, world)
    got = p.download_graph()
    o = oracle.Index(oracle.F32, oracle.L2, dim, n, maxdeg, start)
    o.set_rows(0, data)
    nb_o = 0
    for s0, b in batch_schedule(0, n, growth, max_batch):
        o.multi_insert(ocfg, np.arange(s0, s0 + b, dtype=np.uint32)); nb_o += 1
    assert nb == nb_o and np.array_equal(got, o.adj), f"rank {rank}: sharded build != oracle multi_insert (ibc {ibc})"
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def _run(cmd, env, timeout=600):
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r


def test_two_ranks_search_and_build_against_the_oracle(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29541", str(script), ROOT]
    r = _run(cmd, env)
    assert r.stdout.count("ok") == 2


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without a launcher: bench.py re-executes itself under torch.distributed.run; the test
    hook puts both ranks on cuda:0 over gloo.  The line must say n_gpus 2 and carry the strong-scaling identity check."""
    env = dict(os.environ, DANN_BENCH_ONE_DEVICE="1", OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--n", "60000",
           "--nq", "4000", "--nq-shared", "1001", "--no-extras", "--no-cpu-baseline", "--max-batch", "4096"]
    r = _run(cmd, env, timeout=900)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    s = out["other_configs"]["strong_scaling_shared_set"]
    assert s["ranks"] == 2 and s["queries"] == 1001 and s["identical_to_single_rank"] is True
    # a launcher whose WORLD_SIZE disagrees with --gpus is an error, not a silent 1-rank run
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run(cmd, env=env2, capture_output=True, text=True, timeout=300)
    assert r2.returncode != 0 and "disagrees" in (r2.stdout + r2.stderr)
