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
for ibc, owner_prunes in ((da.IBC_NONE, True), (4, True), (da.IBC_NONE, False)):
    gcfg = da.build_config(Rp, maxdeg, lb, intra_batch_candidates=ibc)
    ocfg = oracle.build_config(Rp, maxdeg, lb, intra_batch_candidates=ibc)
    p = da.Provider(da.F32, da.L2, dim, n, maxdeg, start)
    p.set_elements(0, data)
    growth, max_batch = 0.1, 300
    st = {}
    # owner_prunes: prunes of back-edge targets partitioned by id % world, rewritten rows all-gathered and applied
    nb = build_sharded(p, gcfg, 0, n, growth, max_batch, rank, world, stats=st, owner_prunes=owner_prunes)
    assert (st.get("rows_rewritten", 0) > 0) == owner_prunes, st
    got = p.download_graph()
    o = oracle.Index(oracle.F32, oracle.L2, dim, n, maxdeg, start)
    o.set_rows(0, data)
    nb_o = 0
    for s0, b in batch_schedule(0, n, growth, max_batch):
        o.multi_insert(ocfg, np.arange(s0, s0 + b, dtype=np.uint32)); nb_o += 1
    assert nb == nb_o and np.array_equal(got, o.adj), f"rank {rank}: sharded build != oracle multi_insert (ibc {ibc}, owner_prunes {owner_prunes})"
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


WORKER_CONFIG5 = r"""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
import oracle, diskann_amd as da
from diskann_amd.sharding import build_sharded, rerank_sharded, partition, batch_schedule
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
rng = np.random.default_rng(5)
# config 5 in miniature: every rank holds an f16 replica of all rows + the whole graph, the f32 rows are partitioned
n, dim, Rp, maxdeg, lb = 4000, 64, 12, 16, 32
centers = rng.random((16, dim)).astype(np.float32)
f32 = (centers[rng.integers(0, 16, n)] + 0.2 * rng.standard_normal((n, dim))).astype(np.float32)
f16 = f32.astype(np.float16)
start16 = f16.astype(np.float32).mean(0, keepdims=True).astype(np.float16)
gcfg = da.build_config(Rp, maxdeg, lb, intra_batch_candidates=da.IBC_NONE)
ocfg = oracle.build_config(Rp, maxdeg, lb, intra_batch_candidates=oracle.IBC_NONE)
rep = da.Provider(da.F16, da.L2, dim, n, maxdeg, start16)
rep.set_elements(0, f16)
st = {}
nb = build_sharded(rep, gcfg, 0, n, 0.1, 512, rank, world, stats=st)
o = oracle.Index(oracle.F16, oracle.L2, dim, n, maxdeg, start16)
o.set_rows(0, f16)
for s0, b in batch_schedule(0, n, 0.1, 512):
    o.multi_insert(ocfg, np.arange(s0, s0 + b, dtype=np.uint32))
assert np.array_equal(rep.download_graph(), o.adj), "f16 replica build != oracle"
assert st["rounds"] == nb and st["bytes_gathered"] > 0
bounds = [partition(n, world, r)[0] for r in range(world)] + [n]
lo, hi = bounds[rank], bounds[rank + 1]
shard = da.Provider(da.F32, da.L2, dim, hi - lo, 1, f32[lo:lo + 1])     # the f32 rows this rank owns
shard.set_elements(0, f32[lo:hi])
queries = (centers[rng.integers(0, 16, 64)] + 0.2 * rng.standard_normal((64, dim))).astype(np.float32)
qlo, qhi = partition(64, world, rank)
myq = queries[qlo:qhi]
L, k = 32, 10
cand, _, _ = rep.search(da.Knn(L, 1), myq.astype(np.float16), L)        # graph walk on the replica
rs = {}
ids, d = rerank_sharded(shard, bounds, myq, cand, k, rank, world, stats=rs)
full = da.Provider(da.F32, da.L2, dim, n, 1, f32[:1]); full.set_elements(0, f32)  # checker: un-partitioned f32 rows
want_i, want_d = full.rerank(myq, cand, k)
assert np.array_equal(ids, want_i) and np.array_equal(d.view(np.uint32), want_d.view(np.uint32)), "sharded rerank != Rerank"
for qi in range(myq.shape[0]):
    for j in range(k):
        assert d[qi, j] == np.float32(oracle.query_distance(oracle.F32, oracle.L2, myq[qi], f32[ids[qi, j]]))
assert rs["bytes_gathered"] > 0
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok", st, rs)
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
    # with several ranks the benchmark index comes from the library's sharded build (here over the callback
    # communicator: both ranks share one device): exchange statistics in the line, replicas byte-identical
    ex = out["config"]["build_exchange"]
    assert "error" not in ex and ex["digest_identical_across_ranks"] is True and ex["rounds"] > 0 and ex["bytes_gathered"] > 0
    s = out["other_configs"]["strong_scaling_shared_set"]
    assert s["ranks"] == 2 and s["queries"] == 1001 and s["identical_to_single_rank"] is True
    # a launcher whose WORLD_SIZE disagrees with --gpus is an error, not a silent 1-rank run
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run(cmd, env=env2, capture_output=True, text=True, timeout=300)
    assert r2.returncode != 0 and "disagrees" in (r2.stdout + r2.stderr)


def test_config5_layout_f16_replica_with_f32_owner_rerank(tmp_path):
    """SURVEY.md 7, option (a) for 100 M x 768 at reduced scale: f16 replica + whole graph on every rank (built with
    build_sharded, identical to the oracle's f16 build), f32 rows partitioned, Rerank computed by the owners."""
    script = tmp_path / "worker5.py"
    script.write_text(WORKER_CONFIG5)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29543", str(script), ROOT]
    r = _run(cmd, env)
    assert r.stdout.count("ok") == 2


def _oracle_build(dtype, metric, data, start, Rp, maxdeg, lb, growth, max_batch):
    import numpy as np
    import oracle
    from diskann_amd.sharding import batch_schedule
    n, dim = data.shape
    o = oracle.Index(dtype, metric, dim, n, maxdeg, start)
    o.set_rows(0, data)
    ocfg = oracle.build_config(Rp, maxdeg, lb, intra_batch_candidates=oracle.IBC_NONE)
    nb = 0
    for s0, b in batch_schedule(0, n, growth, max_batch):
        o.multi_insert(ocfg, np.arange(s0, s0 + b, dtype=np.uint32))
        nb += 1
    return o, nb


def test_one_process_several_replicas_through_the_c_abi():
    """dann_multi (one process driving several devices; here two replicas on device 0): dann_multi_build runs
    dann_build_sharded on one host thread per replica over the in-process communicator -- every replica's graph equals
    the oracle's multi_insert -- and dann_multi_search_batch partitions the query block over the replicas."""
    import numpy as np
    import oracle
    import diskann_amd as da
    from helpers import bits, rand_vectors
    rng = np.random.default_rng(99)
    n, dim, Rp, maxdeg, lb = 3000, 32, 10, 12, 32
    data = rand_vectors(rng, oracle.F32, n, dim)
    start = data.mean(0, keepdims=True).astype(np.float32)
    growth, max_batch = 0.1, 400
    o, nb_o = _oracle_build(oracle.F32, oracle.L2, data, start, Rp, maxdeg, lb, growth, max_batch)
    # one GPU: the replicas share device 0; a box with more devices spreads them (and adds a run over all of them, at
    # most 8): the first multi-GPU box that runs `-m gpu` exercises hipMemcpyPeerAsync between real devices
    import torch
    visible = min(torch.cuda.device_count(), 8)
    worlds = [2, 3] + ([visible] if visible > 3 else [])
    for ndev in worlds:
        m = da.MultiProvider(da.F32, da.L2, dim, n, maxdeg, start, [r % visible for r in range(ndev)])
        m.set_elements(0, data)
        nb, st = m.build(da.build_config(Rp, maxdeg, lb, intra_batch_candidates=da.IBC_NONE), 0, n, growth, max_batch)
        assert nb == nb_o
        assert st["rounds"] == nb and st["bytes_gathered"] > 0 and st["rows_rewritten"] > 0
        for r in range(ndev):
            assert np.array_equal(m.download_graph(r), o.adj), (ndev, r)
        q = rand_vectors(rng, oracle.F32, 301, dim)
        ids, d, stats = m.search(da.Knn(40, 1), q, 10)
        oi, od, _, ost = o.search_batch(q, 40, 1, 10)
        assert np.array_equal(ids, oi) and np.array_equal(bits(d), bits(od)) and np.array_equal(stats["cmps"], ost[:, 0])
        m.close()


def test_rccl_communicator_preflight_on_one_gpu():
    """RCCL itself (librccl bound at run time): unique id, a world-1 communicator on device 0, ncclAllGather of a device
    buffer, and dann_build_sharded over it == dann_build.  What one GPU allows of the path the 8-GPU run takes."""
    import ctypes as C
    import numpy as np
    import oracle
    import diskann_amd as da
    from diskann_amd import _ffi
    from diskann_amd.sharding import Comm, build_sharded_native
    from helpers import rand_vectors
    lib = _ffi.lib()
    uid = (C.c_char * 128)()
    _ffi.check(lib.dann_comm_rccl_unique_id(uid), "dann_comm_rccl_unique_id")
    h = C.c_void_p()
    _ffi.check(lib.dann_comm_create_rccl(uid, 0, 1, 0, C.byref(h)), "dann_comm_create_rccl")
    comm = Comm(h)
    assert lib.dann_comm_world(h) == 1 and lib.dann_comm_rank(h) == 0
    import torch
    src = torch.arange(1000, dtype=torch.int32, device="cuda:0")
    dst = torch.zeros(1000, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    _ffi.check(lib.dann_comm_all_gather_device(h, 0, C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), 4000),
               "dann_comm_all_gather_device")
    assert torch.equal(src, dst)
    rng = np.random.default_rng(8)
    n, dim, Rp, maxdeg, lb = 2000, 24, 8, 10, 24
    data = rand_vectors(rng, oracle.F32, n, dim)
    start = data.mean(0, keepdims=True).astype(np.float32)
    cfg = da.build_config(Rp, maxdeg, lb, intra_batch_candidates=da.IBC_NONE)
    a = da.Provider(da.F32, da.L2, dim, n, maxdeg, start)
    a.set_elements(0, data)
    nb = build_sharded_native(a, cfg, 0, n, 0.1, 256, comm)
    b = da.Provider(da.F32, da.L2, dim, n, maxdeg, start)
    b.set_elements(0, data)
    assert b.build(cfg, 0, n, 0.1, 256) == nb
    assert np.array_equal(a.download_graph(), b.download_graph())
    comm.close()


def test_rccl_world_n_over_the_visible_gpus():
    """RCCL at world > 1, as soon as the box has more than one GPU (skipped on one): one host thread per device creates
    its rank of a dann_comm_create_rccl communicator, all-gathers a device buffer (ncclAllGather over xGMI), builds its
    replica with dann_build_sharded over it and searches the shared block with dann_search_sharded -- every replica's
    graph == a single-GPU dann_build, every rank's gathered results == one rank searching alone."""
    import ctypes as C
    import threading
    import numpy as np
    import torch
    import oracle
    import diskann_amd as da
    from diskann_amd import _ffi
    from diskann_amd.sharding import Comm, build_sharded_native, search_sharded_native
    from helpers import bits, rand_vectors
    world = min(torch.cuda.device_count(), 8)
    if world < 2:
        pytest.skip("one visible GPU: RCCL at world > 1 needs at least two")
    lib = _ffi.lib()
    uid = (C.c_char * 128)()
    _ffi.check(lib.dann_comm_rccl_unique_id(uid), "dann_comm_rccl_unique_id")
    rng = np.random.default_rng(18)
    n, dim, Rp, maxdeg, lb = 4000, 32, 10, 12, 32
    data = rand_vectors(rng, oracle.F32, n, dim)
    start = data.mean(0, keepdims=True).astype(np.float32)
    q = rand_vectors(rng, oracle.F32, 257, dim)
    cfg = da.build_config(Rp, maxdeg, lb, intra_batch_candidates=da.IBC_NONE)
    ref = da.Provider(da.F32, da.L2, dim, n, maxdeg, start, device=0)
    ref.set_elements(0, data)
    nb_ref = ref.build(cfg, 0, n, 0.1, 512)
    g_ref = ref.download_graph()
    i_ref, d_ref, _ = ref.search(da.Knn(40, 1), q, 10)
    out, err = [None] * world, [None] * world

    def rank_main(r):
        try:
            h = C.c_void_p()
            _ffi.check(lib.dann_comm_create_rccl(uid, r, world, r, C.byref(h)), "dann_comm_create_rccl")
            comm = Comm(h)
            src = torch.full((1000,), r, dtype=torch.int32, device=f"cuda:{r}")
            dst = torch.zeros(1000 * world, dtype=torch.int32, device=f"cuda:{r}")
            torch.cuda.synchronize(r)
            _ffi.check(lib.dann_comm_all_gather_device(h, r, C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), 4000),
                       "dann_comm_all_gather_device")
            gathered = dst.cpu().numpy().reshape(world, 1000)
            p = da.Provider(da.F32, da.L2, dim, n, maxdeg, start, device=r)
            p.set_elements(0, data)
            st = {}
            nb = build_sharded_native(p, cfg, 0, n, 0.1, 512, comm, stats=st)
            ids, dd = search_sharded_native(p, comm, q, 40, 1, 10)
            out[r] = (gathered, nb, st, p.download_graph(), ids, dd)
            comm.close()
        except Exception as e:  # noqa: BLE001
            err[r] = repr(e)

    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a rank did not return within 300 s"
    assert err == [None] * world, err
    for r in range(world):
        gathered, nb, st, g, ids, dd = out[r]
        assert np.array_equal(gathered, np.repeat(np.arange(world, dtype=np.int32)[:, None], 1000, 1)), r
        assert nb == nb_ref and st["rounds"] == nb and st["bytes_gathered"] > 0, (r, nb, nb_ref, st)
        assert np.array_equal(g, g_ref), f"replica {r} of the RCCL build differs from dann_build"
        assert np.array_equal(ids, i_ref) and np.array_equal(bits(dd), bits(d_ref)), r
