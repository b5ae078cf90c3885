#!/usr/bin/env python3
"""bench.py -- QPS @ recall@10 >= 0.95 on a SIFT-1M-shaped index resident in HBM.

One "step" = one pass of the hot path (batched Vamana beam search, dann_search_batch_device)
over one batch of `--nq` synthetic queries that are already resident in HBM.  The index
(1 M x 128-d f32, L2, R = 32) is built on the GPU by the library's own multi_insert path
(untimed setup), the search list size L is the first value of the sweep whose recall@10
against exact brute-force ground truth is >= 0.95, and the timed region is exactly K steps
bracketed by a barrier + torch.cuda.synchronize().

Multi-GPU (launched by torch.distributed.run, one rank per GPU): the index is replicated in
every GPU's HBM, each rank searches its own query stream (no data-path collective), `value`
is the whole-job QPS = N * nq * K / max-over-ranks time ("weak" scaling).

Rank 0 prints ONE JSON line with the `roofline` (HIP-event time of the beam-search kernel vs
its algorithmic bytes) and `cpu_baseline` (the CPU oracle on this box's host cores) objects.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
# working sets far beyond the 256 MiB Infinity Cache, "n:dim:dist:R:pruned:l_build": the headline's row shape at 10 M
# points (6.4 GB; the generator's per-blob density kept: 2 560 blobs) and config 5's row shape (1 M x 768, 3.3 GB)
LARGE_DEFAULT = ("10000000:128:sift_like:1:2560:32:28:100,1000000:768:sift_like:64:56:128,"
                 "1000000:768:sift_like:64:56:128:f16")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="GPUs of this node (one rank each).  Without a launcher bench.py spawns the ranks itself; "
                         "under torch.distributed.run it must equal WORLD_SIZE.  Default: WORLD_SIZE or 1")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank searches its own --nq queries per step.  strong: one shared set of "
                         "--nq-shared queries is block-partitioned over the ranks (diskann-benchmark-core "
                         "search/api.rs:399-436) and the gathered output must equal the 1-rank output byte for byte")
    ap.add_argument("--nq-shared", type=int, default=10000, help="size of the shared query set (the protocol's nq)")
    ap.add_argument("--only-large", action="store_true",
                    help="run only the roofline_large workload and print its object (used under rocprofv3)")
    ap.add_argument("--no-pq-pack", action="store_true", help="PQ leg: search the plain rows (no dann_pq_pack_neighbors)")
    ap.add_argument("--large-int-n", type=int, default=0, help="large_u8 / large_sq8 legs: number of rows (0 = 10 M)")
    ap.add_argument("--only", default="", choices=["", "large", "large768", "large768f16", "large_u8", "large_sq8", "gather", "sq8", "u8", "pq",
                                                   "pq768", "build768", "cpu-distance"],
                    help="run ONE secondary workload and print its object (profiles/run_profiles_r02.sh): the large "
                         "index (first / second --large spec), the gather-distance kernel on a 5 GB store, the SQ-8 or "
                         "u8 search kernel")
    ap.add_argument("--large", default="auto",
                    help="roofline_large workloads, comma-separated 'n:dim:dist:R:pruned:l_build', or 'none'")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--nq", type=int, default=100_000, help="queries per step per GPU (one launch)")
    ap.add_argument("--dist", default="sift_like", help="sift_like | sift_like:<centre scale> | uniform (benchdata.py)")
    ap.add_argument("--max-degree", type=int, default=32)
    ap.add_argument("--pruned-degree", type=int, default=28)
    ap.add_argument("--l-build", type=int, default=100)
    ap.add_argument("--growth", type=float, default=0.05)
    ap.add_argument("--max-batch", type=int, default=16384)
    ap.add_argument("--beam-width", type=int, default=1)
    ap.add_argument("--L", type=int, default=0, help="fixed L (0 = first L of the sweep with recall >= target)")
    ap.add_argument("--target-recall", type=float, default=0.95)
    ap.add_argument("--cpu-queries", type=int, default=20000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sq8", action="store_true")
    ap.add_argument("--no-pq", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the headline workload (used under rocprofv3 so that every launch of the "
                         "beam-search kernel is the timed one)")
    ap.add_argument("--pq-chunks", type=int, default=16)
    ap.add_argument("--replicated-build", action="store_true",
                    help="with --gpus N > 1: every rank builds its own replica with dann_build (no exchange); default "
                         "is the library's sharded build over RCCL (dann_build_sharded)")
    ap.add_argument("--sharded-build", action="store_true",
                    help="N>1: build with diskann_amd.sharding.build_sharded (batch partitioned across ranks, RCCL "
                         "all-gather of the pending adjacency rows) instead of one independent build per rank")
    ap.add_argument("--graph-cache", default="",
                    help="file: load the built graph from it when present, else build and save it (profiling passes "
                         "under rocprofv3 --pmc skip the thousands of build dispatches this way)")
    ap.add_argument("--visited-bits", type=int, default=0)
    ap.add_argument("--prune-tie-order", default="default", choices=["default", "position", "rust"],
                    help="dann_set_prune_tie_order on every index the run builds (A/B of the build legs; searches and "
                         "recall do not depend on it on this continuous data)")
    ap.add_argument("--visited-format", type=int, default=0, choices=[0, 16, 32],
                    help="experiment knob: width of a visited-table entry on every index of the run (0 = automatic)")
    ap.add_argument("--sq8-stride", type=int, default=256,
                    help="row stride of the SQ-8 store: 256 keeps the 128 code bytes of a row in one 128-byte line (the "
                         "L2 kernel never reads the compensation); 0 = payload rounded to 16 B (144: rows straddle lines)")
    ap.add_argument("--build-spec", default="10000000:768:64:56:128:f32",
                    help="--only build768: 'n:dim:R:pruned:l_build:f32|f16' -- the index-build workload at the size one "
                         "MI355X holds (config 5's row shape): data generated on the device chunk by chunk, built by "
                         "dann_build, recall on a 1 000-query exact (f64) ground truth, a 256-query replay through the CPU "
                         "oracle on the bytes the searches touch")
    ap.add_argument("--build-blobs", type=int, default=0, help="--only build768: number of blobs (0 = n / 3906, at least 256)")
    ap.add_argument("--build-centres", default="auto", choices=["auto", "iid", "hier"],
                    help="--only build768: blob centres i.i.d. U(0,1)^dim, or hierarchical (256 topic centres, the blobs "
                         "of a topic on a shared 8-d subspace around it); auto = hier beyond 4096 blobs")
    ap.add_argument("--sweep", action="store_true", help="print the whole recall/QPS sweep to stderr")
    ap.add_argument("--query-sets", type=int, default=4,
                    help="distinct query sets of --nq queries each; timed step i searches set i %% query_sets")
    if len(sys.argv) == 1 and os.environ.get("DANN_BENCH_ARGV"):  # a rank spawned by maybe_spawn()
        return ap.parse_args(json.loads(os.environ["DANN_BENCH_ARGV"]))
    return ap.parse_args()


def log(*a):
    print(*a, file=sys.stderr, flush=True)


from benchdata import ground_truth, make_data, recall_at_k  # noqa: E402  (synthetic data + exact ground truth)


def _strict(o):
    """NaN / inf -> null so that the printed line is strict JSON (e.g. recall when --L skips the ground truth)."""
    if isinstance(o, float):
        return o if o == o and abs(o) != float("inf") else None
    if isinstance(o, dict):
        return {k: _strict(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_strict(v) for v in o]
    return o


def maybe_spawn(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: re-exec N ranks, one per GPU, under
    torch.distributed.run on 127.0.0.1.  Under a launcher (WORLD_SIZE set) --gpus must agree with it."""
    ws = os.environ.get("WORLD_SIZE")
    if ws is None:
        if args.gpus is None or args.gpus <= 1:
            return
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        # the script's own flags travel in the environment: torch.distributed.run's argparse rejects script flags that
        # abbreviate one of its options (--n is "ambiguous" with --nnodes / --nproc-per-node) even after the script path
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", DANN_BENCH_ARGV=json.dumps(sys.argv[1:]),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(cmd, env=env))
    if args.gpus is not None and int(ws) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} disagrees with the launcher's WORLD_SIZE={ws}")


def main():
    args = parse()
    if args.nq <= 2048:
        # launches that small are not bracketed by HIP events by default (dann_debug.h: DANN_DBG_TIME_SMALL_LAUNCHES); the
        # roofline legs divide by the kernel time, so an odd run with a tiny --nq asks for the events on every index
        os.environ.setdefault("DANN_TIME_SMALL_LAUNCHES", "1")
    if args.only == "cpu-distance":  # host only
        print(json.dumps(_strict({"cpu_distance_kernels": cpu_distance_microbench()})), flush=True)
        return
    maybe_spawn(args)
    # (multi-process GPU work on this pool needs dmabuf IPC; exported on the boxes already -- kept in any launch)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import diskann_amd as da
    from diskann_amd import _ffi
    if args.visited_format:  # A/B knob (results never depend on it)
        _init = da.Provider.__init__

        def _init_fmt(self, *a, **kw):
            _init(self, *a, **kw)
            self.set_visited_format(args.visited_format)
        da.Provider.__init__ = _init_fmt

    if args.prune_tie_order != "default":
        _init_t = da.Provider.__init__

        def _init_tie(self, *a, **kw):
            _init_t(self, *a, **kw)
            self.set_prune_tie_order(da.TIE_RUST if args.prune_tie_order == "rust" else da.TIE_POSITION)
        da.Provider.__init__ = _init_tie

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if os.environ.get("DANN_BENCH_ONE_DEVICE") != "1" and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: {world} ranks need {world} visible GPUs, found {torch.cuda.device_count()}")
    # test hook: DANN_BENCH_ONE_DEVICE=1 runs every rank on cuda:0 over gloo (exercises the multi-rank code path on a
    # 1-GPU box; never used for reported numbers)
    one_dev = os.environ.get("DANN_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.only == "build768":
        print(json.dumps(_strict({"build_large": build_large_variant(args, torch, da, _ffi.lib(), _ffi, dev, local)})), flush=True)
        return
    if args.only == "pq768":
        # config 5's row shape: 1 M x 768, R = 64 / 56, l_build 128; codes of --pq-chunks bytes (default 48 here), searched
        # with the register-resident table, reranked on the f32 rows -- beside the f16 rows searched at full precision
        args.dim, args.max_degree, args.pruned_degree, args.l_build = 768, 64, 56, 128
        if args.pq_chunks == 16:
            args.pq_chunks = 48
        args.pq768 = True
        args.only = "pq"
        print(json.dumps(_strict({"pq768": only_variant(args, torch, da, _ffi.lib(), _ffi, dev, local)})), flush=True)
        return
    if args.only in ("gather", "sq8", "u8", "pq"):
        print(json.dumps(_strict({args.only: only_variant(args, torch, da, _ffi.lib(), _ffi, dev, local)})), flush=True)
        return
    if args.only in ("large_u8", "large_sq8"):
        rd = C.c_double(0.0)
        _ffi.lib().dann_debug_stream_read_gbps(local, 4 << 30, 10, C.byref(rd))
        kind = args.only.split("_")[1]
        print(json.dumps(_strict({"roofline_large_" + kind: large_int_variant(args, kind, torch, da, _ffi.lib(), _ffi, dev, local,
                                                                             10, args.beam_width, rd.value or None)})), flush=True)
        return
    if args.only in ("large", "large768", "large768f16"):
        args.only_large = True
        if args.only == "large768":
            specs = (LARGE_DEFAULT if args.large == "auto" else args.large).split(",")
            args.large = specs[1] if len(specs) > 1 else specs[0]
        if args.only == "large768f16":  # config 5's replica: the same rows stored as f16 (Full<f16>)
            args.large = "1000000:768:sift_like:64:56:128:f16"
    if args.only_large:
        rd = C.c_double(0.0)
        _ffi.lib().dann_debug_stream_read_gbps(local, 4 << 30, 10, C.byref(rd))
        spec = (LARGE_DEFAULT if args.large == "auto" else args.large).split(",")[0]
        print(json.dumps(_strict({"roofline_large": large_variant(args, spec, torch, da, _ffi.lib(), _ffi, dev, local, 10,
                                                                  args.beam_width, rd.value or None)})), flush=True)
        return

    # ---- setup (untimed): data, index build on the GPU, ground truth ---------------------
    t0 = time.time()
    base, queries = make_data(torch, dev, args.n, args.dim, args.nq, args.dist, 0xD15CA11, 0xD15CA12 + rank)
    # the timed steps rotate through several distinct query sets (set 0 is the one the recall sweep uses): no two
    # consecutive steps replay the same address stream
    nsets = max(1, args.query_sets)
    qsets = [queries] + [make_data(torch, dev, 0, args.dim, args.nq, args.dist, 0xD15CA11, 0xD15CB00 + 97 * i + rank)[1]
                         for i in range(1, nsets)]
    # medoid start point (diskann-utils/src/sampling/medoid.rs:15-48): f64 mean, nearest row
    mean = base.double().mean(0).float()
    medoid = int(torch.argmin(((base - mean[None, :]) ** 2).sum(1)).item())
    start = base[medoid:medoid + 1].cpu().numpy()
    prov = da.Provider(da.F32, da.L2, args.dim, args.n, args.max_degree, start, device=local)
    if args.visited_bits:
        prov.set_visited_bits(args.visited_bits)
    lib = _ffi.lib()
    base_h = base.cpu().numpy()
    prov.set_elements(0, base_h)
    build_stats = None
    t1 = time.time()
    cfg = da.build_config(args.pruned_degree, args.max_degree, args.l_build, intra_batch_candidates=da.IBC_NONE)
    if args.graph_cache and os.path.exists(args.graph_cache):
        prov.load_graph(args.graph_cache)
        nb = 0
    elif world > 1 and not args.replicated_build:
        # The reference's multi_insert has one exchange step (index.rs:911-1024); with several ranks the benchmark index is
        # built by the library's own sharded build: dann_comm_create_rccl (the unique id travels over torch.distributed),
        # candidates per rank, ncclAllGather of the pending rows, owner-partitioned prunes, second all-gather of the
        # rewritten rows -- every replica byte-identical to a single-GPU dann_build.  Should the communicator fail on
        # this node the run falls back to replicated builds and says so (build_exchange.error).
        from diskann_amd.sharding import build_sharded
        import hashlib
        build_stats = {}
        try:
            nb = build_sharded(prov, cfg, 0, args.n, args.growth, args.max_batch, rank, world, stats=build_stats)
            ok_local = 1
        except Exception as e:  # noqa: BLE001
            build_stats = {"error": str(e)[:300]}
            ok_local = 0
        flag = torch.tensor([ok_local], dtype=torch.int32, device=torch.device("cpu") if one_dev else dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            log(f"[rank {rank}] sharded build failed on some rank ({build_stats.get('error', 'another rank')}): "
                "replicated builds instead")
            err = build_stats.get("error", "failed on another rank")
            prov.close()
            prov = da.Provider(da.F32, da.L2, args.dim, args.n, args.max_degree, start, device=local)
            prov.set_elements(0, base_h)
            nb = prov.build(cfg, 0, args.n, args.growth, args.max_batch)
            build_stats = {"error": err, "fallback": "replicated dann_build per rank"}
        else:
            # the replicas must be byte-identical: compare a digest of every rank's adjacency
            dig = np.frombuffer(hashlib.sha256(prov.download_graph().tobytes()).digest()[:8], dtype=np.int64).copy()
            mine_d = torch.from_numpy(dig).to(torch.device("cpu") if one_dev else dev)
            all_d = torch.empty(world, dtype=torch.int64, device=mine_d.device)
            dist.all_gather_into_tensor(all_d, mine_d)
            if len(set(all_d.cpu().tolist())) != 1:
                raise SystemExit("sharded build: the replicas' graphs differ")
            build_stats["digest_identical_across_ranks"] = True
            build_stats["replicas_identical"] = True
            build_stats["rccl_ranks"] = 0 if one_dev else world
            build_stats["communicator"] = "gloo callback (one-device test hook)" if one_dev else "RCCL (dann_comm_create_rccl)"
    else:
        nb = prov.build(cfg, 0, args.n, args.growth, args.max_batch)
        if args.graph_cache and rank == 0:
            prov.save_graph(args.graph_cache)
    torch.cuda.synchronize()
    t_build = time.time() - t1
    gt = ground_truth(torch, base, queries, 10)
    if rank == 0:
        log(f"[setup] data {t1 - t0:.1f}s, GPU build {t_build:.1f}s ({nb} batches, {args.n / t_build:,.0f} pts/s), "
            f"medoid {medoid}")

    k = 10
    d_ids = torch.empty((args.nq, k), dtype=torch.int32, device=dev)
    d_dists = torch.empty((args.nq, k), dtype=torch.float32, device=dev)
    d_stats = torch.empty((args.nq, 5), dtype=torch.int32, device=dev)

    def run_search(L, W, qset=0):
        _ffi.check(lib.dann_search_batch_device(prov._h, C.c_void_p(qsets[qset].data_ptr()), args.nq, L, W, k,
                                                C.c_void_p(d_ids.data_ptr()), C.c_void_p(d_dists.data_ptr()),
                                                C.c_void_p(d_stats.data_ptr())), "dann_search_batch_device")

    def evaluate(L, W):
        run_search(L, W)
        st = d_stats.cpu().numpy().view(np.uint32)
        if st[:, 3].any():
            raise RuntimeError(f"per-query scratch overflow at L={L}")
        ids = d_ids.cpu().numpy().view(np.uint32)
        evaluate.last_ids = ids
        return recall_at_k(ids, gt, k), st

    # ---- choose L: first L of the sweep with recall@10 >= target (reference protocol) --------
    sweep = [10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30, 32, 36, 40, 48, 56, 64, 80, 96, 128, 160, 192, 256, 320, 400, 500]
    W = args.beam_width
    chosen, rec, st = None, 0.0, None
    if args.L:
        chosen = args.L
        rec, st = evaluate(chosen, W)
    else:
        for L in sweep:
            rec, st = evaluate(L, W)
            if rank == 0 and args.sweep:
                prov.kernel_time_reset()
                run_search(L, W)
                ms, _ = prov.kernel_time(0)
                log(f"[sweep] L={L} recall@10={rec:.4f} cmps={st[:, 0].mean():.0f} hops={st[:, 1].mean():.0f} "
                    f"kernel={ms:.3f} ms QPS={args.nq / ms * 1e3:,.0f}")
            if rec >= args.target_recall and chosen is None:
                chosen = L
                if not args.sweep:
                    break
        if chosen is None:
            chosen = sweep[-1]
        rec, st = evaluate(chosen, W)
    if world > 1:  # all ranks use rank 0's L so the work per GPU is the same
        t = torch.tensor([chosen], device=torch.device("cpu") if one_dev else dev)
        dist.broadcast(t, 0)
        if int(t.item()) != chosen:
            chosen = int(t.item())
            rec, st = evaluate(chosen, W)
    # algorithmic bytes of a launch: the mean over the query sets the timed steps cycle through
    set_cmps, set_hops = [int(st[:, 0].sum())], [int(st[:, 1].sum())]
    for i in range(1, nsets):
        run_search(chosen, W, i)
        sti = d_stats.cpu().numpy().view(np.uint32)
        if sti[:, 3].any():
            raise RuntimeError(f"per-query scratch overflow at L={chosen} (query set {i})")
        set_cmps.append(int(sti[:, 0].sum()))
        set_hops.append(int(sti[:, 1].sum()))
    used = [i % nsets for i in range(args.steps)] or [0]
    cmps_sum = sum(set_cmps[i] for i in used) / len(used)
    hops_sum = sum(set_hops[i] for i in used) / len(used)

    # ---- timed region ------------------------------------------------------------------------------
    for w_ in range(args.warmup):
        run_search(chosen, W, w_ % nsets)
    prov.kernel_time_reset()
    barrier()
    tstart = time.perf_counter()
    for i in range(args.steps):
        run_search(chosen, W, i % nsets)
    barrier()
    elapsed = time.perf_counter() - tstart
    kernel_ms, launches = prov.kernel_time(0)
    per_rank_qps = [args.nq * args.steps / elapsed]
    if world > 1:
        cdev = torch.device("cpu") if one_dev else dev
        mine_t = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        all_t = torch.empty(world, device=cdev, dtype=torch.float64)
        dist.all_gather_into_tensor(all_t, mine_t)  # every rank's own time over the same K steps (self-checking record)
        per_rank_qps = [args.nq * args.steps / float(x) for x in all_t.cpu().tolist()]
        t = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- strong scaling: ONE shared query set, block-partitioned over the ranks (the reference's protocol:
    # diskann-benchmark-core/src/search/api.rs:399-436); the gathered g-rank output must equal the 1-rank output
    from diskann_amd.sharding import partition, search_sharded
    # (skipped in single-rank --no-extras runs: under rocprofv3 every beam-search launch is then the timed workload)
    nqs = args.nq_shared if (world > 1 or not args.no_extras or args.scaling == "strong") else 0
    _, shared = make_data(torch, dev, 0, args.dim, nqs, args.dist, 0xD15CA11, 0xD15CA20)  # same set on every rank
    lo, hi = partition(nqs, world, rank)
    s_ids = torch.empty((max(hi - lo, 1), k), dtype=torch.int32, device=dev)
    s_d = torch.empty((max(hi - lo, 1), k), dtype=torch.float32, device=dev)
    s_st = torch.empty((max(hi - lo, 1), 5), dtype=torch.int32, device=dev)

    def run_shared():
        if hi > lo:
            _ffi.check(lib.dann_search_batch_device(prov._h, C.c_void_p(shared[lo:hi].data_ptr()), hi - lo, chosen, W, k,
                                                    C.c_void_p(s_ids.data_ptr()), C.c_void_p(s_d.data_ptr()),
                                                    C.c_void_p(s_st.data_ptr())), "dann_search_batch_device")
    for _ in range(max(args.warmup, 1)):
        run_shared()
    barrier()
    ts = time.perf_counter()
    for _ in range(args.steps):
        run_shared()
    barrier()
    strong_elapsed = time.perf_counter() - ts
    if world > 1:
        t = torch.tensor([strong_elapsed], device=torch.device("cpu") if one_dev else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        strong_elapsed = float(t.item())
    shared_h = shared.cpu().numpy()
    empty = (np.zeros((0, k), np.uint32), np.zeros((0, k), np.float32))
    g_ids, g_d = empty if nqs == 0 else search_sharded(
        lambda qs: prov.search(da.Knn(chosen, W), qs, k)[:2] if len(qs) else empty, shared_h, k, rank, world)
    strong = {"queries": nqs, "qps": nqs * args.steps / max(strong_elapsed, 1e-9), "ms_per_pass": strong_elapsed / args.steps * 1e3,
              "ranks": world}
    if rank == 0 and nqs:
        one_ids, one_d, _ = prov.search(da.Knn(chosen, W), shared_h, k)  # the same set through ONE rank
        strong["identical_to_single_rank"] = bool(np.array_equal(one_ids, g_ids) and
                                                  np.array_equal(one_d.view(np.uint32), g_d.view(np.uint32)))
        if not strong["identical_to_single_rank"]:
            raise SystemExit("bench.py: sharded output differs from the single-rank output")

    if rank == 0:
        qps = world * args.nq * args.steps / elapsed
        row_bytes = args.dim * 4
        adj_bytes = (args.max_degree + 1) * 4
        alg_bytes = cmps_sum * row_bytes + hops_sum * adj_bytes  # per launch
        avg_kernel_ms = kernel_ms / max(launches, 1)
        achieved = alg_bytes / (avg_kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "QPS @ recall@10>=0.95, SIFT-1M-shaped d=128 f32 L2",
            "value": qps,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",  # per-GPU work fixed (nq queries per rank per step); --scaling strong: shared set
            "vs_baseline": None,
            "dtype": "f32",
            "data": f"synthetic ({args.dist}), no SIFT files on the box",
            "config": {
                "workload": f"batched beam search over a {args.n}x{args.dim} f32 index resident in HBM, "
                            f"{args.nq} queries/step/GPU, k=10, L={chosen}, beam_width={W}",
                "index": f"Vamana R={args.max_degree} (pruned {args.pruned_degree}), l_build={args.l_build}, "
                         f"alpha=1.2, built on GPU by " + (f"dann_build_sharded over {world} ranks" if build_stats
                                                           is not None and "error" not in build_stats else "dann_build") +
                         f" (growth {args.growth}, max_batch {args.max_batch})",
                "recall_at_10": round(rec, 4),
                "L": chosen,
                "beam_width": W,
                "mean_cmps": cmps_sum / args.nq,
                "mean_hops": hops_sum / args.nq,
                "query_sets_rotated": nsets,
                "build_seconds": round(t_build, 2),
                **({"build_exchange": build_stats} if build_stats is not None else {}),
                "parallelism": f"replicated index x{world}, query streams sharded, no collective",
                # multi-GPU records check themselves: the collective backend the ranks ran on, every rank's own rate over
                # the timed steps (value = world * nq * steps / the slowest rank's time) and, above, the build's exchange
                "rccl_world": 0 if (world == 1 or one_dev) else world,
                "per_rank_qps": per_rank_qps,
            },
            "roofline": {
                "kernel": "beam_search_kernel",
                # a 644 MB working set against a 256 MiB Infinity Cache: the rate below is what the fabric delivers
                # (HBM + cache hits); the HBM-side fraction of the same kernel is `hbm_side_frac` (the 6.4 GB index)
                "bound": "fabric (HBM + Infinity Cache)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "algorithmic_bytes_per_launch": alg_bytes,
                "avg_kernel_ms": avg_kernel_ms,
            },
        }
        # the other BASELINE.json configurations, measured on the same index (not the headline):
        # configs[1] single-query beam search at L=64, configs[2] 1024 concurrent queries
        def timed_small(nq_small, L_small, reps):
            d_ids.zero_()  # (a launch that skipped queries must not pass the oracle check on a previous call's rows)
            _ffi.check(lib.dann_search_batch_device(prov._h, C.c_void_p(queries.data_ptr()), nq_small, L_small, W, k,
                                                    C.c_void_p(d_ids.data_ptr()), C.c_void_p(d_dists.data_ptr()),
                                                    C.c_void_p(d_stats.data_ptr())), "dann_search_batch_device")
            torch.cuda.synchronize()
            t_0 = time.perf_counter()
            for r in range(reps):
                qptr = queries.data_ptr() + (r % max(1, min(64, args.nq // nq_small))) * nq_small * args.dim * 4
                lib.dann_search_batch_device(prov._h, C.c_void_p(qptr), nq_small, L_small, W, k,
                                             C.c_void_p(d_ids.data_ptr()), C.c_void_p(d_dists.data_ptr()),
                                             C.c_void_p(d_stats.data_ptr()))
            torch.cuda.synchronize()
            return (time.perf_counter() - t_0) / reps
        out["other_configs"] = {"strong_scaling_shared_set": strong}
        if args.scaling == "strong":  # headline = the shared set; the weak number moves to other_configs
            out["other_configs"]["weak_scaling"] = {"qps": qps, "ms_per_step": out["ms_per_step"],
                                                    "queries_per_step_per_gpu": args.nq}
            out["value"] = strong["qps"]
            out["ms_per_step"] = strong["ms_per_pass"]
            out["scaling"] = "strong"
            out["config"]["workload"] = (f"batched beam search over a {args.n}x{args.dim} f32 index resident in HBM, ONE "
                                         f"shared set of {nqs} queries per step block-partitioned over {world} GPU(s), "
                                         f"k=10, L={chosen}, beam_width={W}")
        oix_head, qh_all, ref_small = None, None, None
        if not args.no_cpu_baseline and world == 1:
            import oracle
            oix_head = oracle.Index(oracle.F32, oracle.L2, args.dim, args.n, args.max_degree, start)
            oix_head.rows[:args.n, :] = base_h.view(np.uint8).reshape(args.n, -1)
            oix_head.adj[:] = prov.download_graph()
            qh_all = queries.cpu().numpy()
            # the oracle's answer for the first 2048 queries at the chosen L and at L = 64: every secondary leg is
            # checked against it (ids and distance bits)
            ref_small = {L_: oix_head.search_batch(qh_all[:2048], L_, W, k, threads=host_cores()[0], fast=True)[:2]
                         for L_ in {chosen, 64}}

        def same_as_oracle(ids, dists, L_, nfirst):
            if ref_small is None:
                return None
            m = min(nfirst, 2048)
            return bool(np.array_equal(ids[:m], ref_small[L_][0][:m]) and
                        np.array_equal(dists[:m].view(np.uint32), ref_small[L_][1][:m].view(np.uint32)))

        def out_small(nq_small):
            return (d_ids[:nq_small].cpu().numpy().view(np.uint32), d_dists[:nq_small].cpu().numpy())
        if not args.no_extras:
            lat = timed_small(1, 64, 200)
            t1024 = timed_small(1024, chosen, 50)
            t10k = timed_small(10000, chosen, 20)
            # configs[2] as a server sees it: 1024 searches kept in flight (dann_set_max_concurrency: 1024 persistent
            # wavefronts share the queries of a call; a finished search is replaced at once, not at the batch's end)
            prov.set_max_concurrency(1024)
            nsus = min(20480, args.nq)
            tsus = timed_small(nsus, chosen, 20)
            tsus64 = timed_small(nsus, 64, 10)
            timed_small(nsus, chosen, 1)
            i_, d_ = out_small(2048)
            sus_same = same_as_oracle(i_, d_, chosen, 2048)
            prov.set_max_concurrency(0)
            out["other_configs"].update({
                "sustained_1024_in_flight_identical_to_oracle": sus_same,
                "sustained_1024_in_flight_qps_at_L": nsus / tsus,
                "sustained_1024_in_flight_mean_latency_us_at_L": 1024 * tsus / nsus * 1e6,
                "sustained_1024_in_flight_qps_L64": nsus / tsus64,
                "sustained_1024_in_flight_queries_per_call": nsus,
            })
            out["other_configs"].update({
                "single_query_L64_latency_us": lat * 1e6,
                "single_query_L64_qps": 1.0 / lat,
                "concurrent_1024_qps_at_L": 1024 / t1024,
                # the survey's protocol size (SIFT's query set): one launch of 10 000 queries
                "protocol_nq10000_qps_at_L": 10000 / t10k,
            })
            # per-call latency distribution of the single-query launch (what SearchResults reports per query:
            # mean / p90 / p99, diskann-benchmark-core/src/search/graph/knn.rs:300-330), device-resident buffers
            one = []
            for r in range(300):
                qptr = queries.data_ptr() + (r % 256) * args.dim * 4
                t_0 = time.perf_counter()
                lib.dann_search_batch_device(prov._h, C.c_void_p(qptr), 1, 64, W, k, C.c_void_p(d_ids.data_ptr()),
                                             C.c_void_p(d_dists.data_ptr()), C.c_void_p(d_stats.data_ptr()))
                one.append((time.perf_counter() - t_0) * 1e6)
            out["other_configs"]["single_query_L64_latency"] = pct(one[20:])
            # parity of the small-batch regimes (teams of wavefronts per query): 1024 queries as one batch, a single query
            timed_small(1024, chosen, 1)
            i_, d_ = out_small(1024)
            out["other_configs"]["concurrent_1024_identical_to_oracle"] = same_as_oracle(i_, d_, chosen, 1024)
            lib.dann_search_batch_device(prov._h, C.c_void_p(queries.data_ptr()), 1, 64, W, k, C.c_void_p(d_ids.data_ptr()),
                                         C.c_void_p(d_dists.data_ptr()), C.c_void_p(d_stats.data_ptr()))
            i_, d_ = out_small(1)
            out["other_configs"]["single_query_L64_identical_to_oracle"] = same_as_oracle(i_, d_, 64, 1)
            # the whole 100 000-query batch at L = 64 (BASELINE config 2's L on the throughput workload)
            prov.kernel_time_reset()
            t64 = timed_small(args.nq, 64, 5)
            ms64, n64 = prov.kernel_time(0)
            st64 = d_stats.cpu().numpy().view(np.uint32)
            alg64 = int(st64[:, 0].sum()) * row_bytes + int(st64[:, 1].sum()) * adj_bytes
            i_, d_ = out_small(2048)
            out["other_configs"]["batch_L64"] = {
                "queries": args.nq, "qps": args.nq / t64, "avg_kernel_ms": ms64 / max(n64, 1),
                "algorithmic_GBps": alg64 / (ms64 / max(n64, 1) * 1e-3) / 1e9,
                "frac_of_hbm_peak": alg64 / (ms64 / max(n64, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "ids_identical_to_oracle": same_as_oracle(i_, d_, 64, 2048)}
            # the drop-in call of a host that holds its queries in host memory: dann_search_batch with host pointers
            # (H2D, kernel, D2H; batches of 2 x 32768 queries and more go through the chunked two-stream pipeline)
            qh_host = qsets[0].cpu().numpy()
            h_ids = np.empty((args.nq, k), np.uint32)
            h_d = np.empty((args.nq, k), np.float32)

            def host_call():
                _ffi.check(lib.dann_search_batch(prov._h, qh_host.ctypes.data, args.nq, chosen, W, k, h_ids.ctypes.data,
                                                 h_d.ctypes.data, None), "dann_search_batch")
            # first sight of these buffers: the three lanes (copies through the pinned ring); from the second call on the
            # library page-locks buffers it has seen before for the length of the call and launches once, zero-copy
            prov.debug_set(host_pipeline=3)  # (lanes only: what a caller with fresh buffers on every call gets)
            host_call()
            t_0 = time.perf_counter()
            for _ in range(5):
                host_call()
            t_lanes = (time.perf_counter() - t_0) / 5
            prov.debug_set(host_pipeline=None)
            host_call()
            host_call()
            t_0 = time.perf_counter()
            for _ in range(5):
                host_call()
            th = (time.perf_counter() - t_0) / 5
            out["other_configs"]["host_pointer_search_batch"] = {
                "queries_per_call": args.nq, "qps_pcie_inclusive": args.nq / th, "ms_per_call": th * 1e3,
                "ids_identical_to_device_path": bool(np.array_equal(h_ids, evaluate.last_ids)),
                "fraction_of_device_resident_rate": (args.nq / th) / (qps / world),
                "fresh_buffers_every_call": {"qps_pcie_inclusive": args.nq / t_lanes, "ms_per_call": t_lanes * 1e3,
                                             "fraction_of_device_resident_rate": (args.nq / t_lanes) / (qps / world)},
                "note": "host (pageable) buffers in and out.  Buffers the index has been handed before (a serving loop reuses "
                        "them) are page-locked for the length of the call -- re-registering a range costs ~1 us on this "
                        "runtime -- and the batch is ONE zero-copy launch; `fresh_buffers_every_call`: three lanes (threads, "
                        "each with its own stream and pinned ring slot) take 16 384-query chunks round robin -- copy in, "
                        "kernel, copy out.  Never the reported `value`"}
            try:  # the same call on buffers the caller page-locked (hipHostMalloc): ONE launch, the kernel reads the queries
                  # from and writes the results to the caller's memory itself -- no copy, no chunks
                pq_ = torch.empty(qh_host.shape, dtype=torch.float32, pin_memory=True)
                pq_.numpy()[...] = qh_host
                pi_ = torch.empty((args.nq, k), dtype=torch.int32, pin_memory=True)
                pd_ = torch.empty((args.nq, k), dtype=torch.float32, pin_memory=True)

                def pinned_call():
                    _ffi.check(lib.dann_search_batch(prov._h, pq_.data_ptr(), args.nq, chosen, W, k, pi_.data_ptr(),
                                                     pd_.data_ptr(), None), "dann_search_batch")
                pinned_call()
                t_0 = time.perf_counter()
                for _ in range(5):
                    pinned_call()
                tp = (time.perf_counter() - t_0) / 5
                out["other_configs"]["host_pointer_search_batch"]["pinned_caller_buffers"] = {
                    "qps_pcie_inclusive": args.nq / tp, "ms_per_call": tp * 1e3,
                    "ids_identical_to_device_path": bool(np.array_equal(pi_.numpy().view(np.uint32), evaluate.last_ids)),
                    "fraction_of_device_resident_rate": (args.nq / tp) / (qps / world)}
            except Exception as e:  # noqa: BLE001
                out["other_configs"]["host_pointer_search_batch"]["pinned_caller_buffers"] = {"error": str(e)[:200]}
            # the reference's serving model on one shared index: 16 host threads, one query per call
            try:
                out["other_configs"]["concurrent_callers"] = callers_variant(prov, qh_host, chosen, k, same_as_oracle)
            except Exception as e:  # never lose the headline line over a secondary leg
                out["other_configs"]["concurrent_callers"] = {"error": str(e)[:300]}
            # the distance kernel on its own (ExpandBeam::expand_beam batched): 20 000 queries x 256
            # random row ids -> n_evals x 512 B of gathers; kernel time by HIP events (clock 1)
            gq, gl = 20000, 256
            rng = np.random.default_rng(7)
            gids = rng.integers(0, args.n, gq * gl, dtype=np.uint32)
            goff = (np.arange(gq + 1, dtype=np.uint64) * gl)
            qh = queries[:gq].cpu().numpy()
            prov.expand_beam_batch(qh, gids, goff)
            prov.kernel_time_reset()
            for _ in range(3):
                prov.expand_beam_batch(qh, gids, goff)
            gms, gn = prov.kernel_time(1)
            out["other_configs"]["distance_kernel"] = {
                "kernel": "expand_beam_kernel", "evals_per_launch": gq * gl, "avg_kernel_ms": gms / max(gn, 1),
                "algorithmic_GBps": gq * gl * row_bytes / (gms / max(gn, 1) * 1e-3) / 1e9,
                "frac_of_hbm_peak": gq * gl * row_bytes / (gms / max(gn, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS,
            }
            # achievable streaming bandwidth on this box (device-to-device copy, read + write bytes)
            buf = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
            dst = torch.empty_like(buf)
            dst.copy_(buf)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                dst.copy_(buf)
            e1.record()
            torch.cuda.synchronize()
            copy_gbs = 2 * buf.numel() * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del buf, dst
            out["roofline"]["measured_copy_GBps"] = copy_gbs
            rd = C.c_double(0.0)
            if lib.dann_debug_stream_read_gbps(local, 4 << 30, 10, C.byref(rd)) == 0:
                out["roofline"]["measured_stream_read_GBps"] = rd.value
            out["roofline"]["frac_of_measured_copy"] = achieved / copy_gbs
            if rd.value > 0:
                out["roofline"]["frac_of_measured_stream_read"] = achieved / rd.value
        # configs[2], int8 scalar-quantised variant: same data compressed to SQ-8 (128 B + 4 B rows),
        # index built on the GPU over the codes, recall measured against the exact f32 ground truth
        if not args.no_sq8 and not args.no_extras:
            try:
                out["other_configs"]["sq8"] = sq8_variant(args, torch, da, lib, _ffi, dev, local, base, queries, gt,
                                                          medoid, k, W, chosen, prov)
            except Exception as e:  # never lose the headline line over the secondary config
                out["other_configs"]["sq8"] = {"error": str(e)[:200]}
        # configs[2], u8 rows: the search kernel alone at L = 26 (this generator's recall point) and at SURVEY 8(a)'s
        # C-int8 sizing L = 64
        if not args.no_sq8 and not args.no_extras:
            try:
                r26, r64 = int_rows_variant(args, torch, da, lib, _ffi, dev, local, base, queries, medoid, "u8", [26, 64], k, W)
                out["other_configs"]["u8"] = {"L26": r26, "L64": r64}
            except Exception as e:
                out["other_configs"]["u8"] = {"error": str(e)[:200]}
        if not args.no_pq and not args.no_extras:
            try:
                out["other_configs"]["pq"] = pq_variant(args, torch, da, lib, _ffi, dev, local, base, queries, gt,
                                                        medoid, k, W, prov)
            except Exception as e:
                out["other_configs"]["pq"] = {"error": str(e)[:200]}
        # the survey's headline distribution (i.i.d. U(-1,1)): reported next to the SIFT-like headline
        if not args.no_extras and args.dist != "uniform":
            try:
                out["other_configs"]["uniform_U(-1,1)"] = uniform_variant(args, torch, da, lib, _ffi, dev, local, k, W)
            except Exception as e:
                out["other_configs"]["uniform_U(-1,1)"] = {"error": str(e)[:200]}
        # HBM traffic per launch from the committed PMC pass (rocprofv3 cannot run inside bench.py);
        # only reported when the profiled workload is this workload.
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
            wl = pm["workload"]
            if (wl["nq"], wl["L"], wl["beam_width"], wl["n"], wl["dim"]) == (args.nq, chosen, W, args.n, args.dim):
                out["roofline"]["traffic"] = pm["hbm_bytes_per_launch_corrected"]
                out["roofline"]["traffic_source"] = ("profiles/pmc_latest.json (FETCH_SIZE x2 + WRITE_SIZE, KiB); fabric-side "
                                                     "bytes = HBM + 256 MiB Infinity Cache hits")
        except (OSError, KeyError, ValueError):
            pass
        if oix_head is not None:  # rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(args, oix_head, qh_all, chosen, W, k, evaluate.last_ids)
            if not args.no_extras:
                out["other_configs"].update(cpu_small_regimes(oix_head, qh_all, chosen, W, k))
                try:  # the CPU distance kernels beside other_configs.distance_kernel (the GPU gather kernel)
                    out["other_configs"]["cpu_distance_kernels"] = cpu_distance_microbench(full=False)
                except Exception as e:  # noqa: BLE001
                    out["other_configs"]["cpu_distance_kernels"] = {"error": str(e)[:200]}
            del oix_head
        # the same kernel on a working set far beyond the 256 MiB Infinity Cache (the honest HBM fraction)
        if args.large != "none" and not args.no_extras and world == 1:
            del base, queries, gt
            prov.close()
            torch.cuda.empty_cache()
            specs = LARGE_DEFAULT if args.large == "auto" else args.large
            for i, spec in enumerate(specs.split(",")):
                key = "roofline_large" if i == 0 else f"roofline_large_d{spec.split(':')[1]}" + ("_f16" if spec.endswith(":f16") else "")
                try:
                    out[key] = large_variant(args, spec, torch, da, lib, _ffi, dev, local, k, W,
                                             out["roofline"].get("measured_stream_read_GBps"))
                except Exception as e:
                    out[key] = {"error": str(e)[:300]}
                torch.cuda.empty_cache()
        # north_star's int8 rows on the HBM side: the integer search kernel on 10 M x 128 u8 rows and SQ-8 codes
        if args.large != "none" and not args.no_extras and not args.no_sq8 and world == 1:
            for kind in ("u8", "sq8"):
                try:
                    out["roofline_large_" + kind] = large_int_variant(args, kind, torch, da, lib, _ffi, dev, local, k, W,
                                                                      out["roofline"].get("measured_stream_read_GBps"))
                except Exception as e:  # noqa: BLE001
                    out["roofline_large_" + kind] = {"error": str(e)[:300]}
                torch.cuda.empty_cache()
        for key in ("roofline_large_d768", "roofline_large_d768_f16"):  # north_star: MFMA utilisation of the build
            b = out.get(key, {}).get("build") if isinstance(out.get(key), dict) else None
            if b and b.get("used"):
                out.setdefault("mfma", {})[key.replace("roofline_large_", "build_1Mx")] = {
                    k2: b[k2] for k2 in ("kernel", "TFLOP_s", "peak", "frac", "share_of_prune_pairs", "build_seconds")}
        if isinstance(out.get("roofline_large"), dict) and "frac" in out["roofline_large"]:
            # the HBM-side fraction of the same kernel: the 6.4 GB index, 25 x the Infinity Cache
            out["roofline"]["hbm_side_frac"] = out["roofline_large"]["frac"]
            out["roofline"]["frac_hbm_side"] = out["roofline_large"]["frac"]  # (the same figure under the name the review asked for)
            out["roofline"]["hbm_side_workload"] = out["roofline_large"]["workload"]
            # what of that HBM provably served (first-touch bytes), and the distance kernel on rows read once per launch
            out["roofline"]["hbm_side"] = {
                "beam_search_first_touch": out["roofline_large"].get("hbm_side"),
                "distance_kernel_rows_read_once": out["roofline_large"].get("hbm_side_distance_kernel")}
        print(json.dumps(_strict(out)), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def family_of(before, after):
    """kernel families (include/dann_debug.h) that served the launches between two Provider.search_families() readings"""
    fams = [f for f in after if after[f][0] > before[f][0]]
    return "+".join(fams) if fams else "none"


def oracle_sample(prov, odt, metric, dim, n, R, start_rows, stored_rows_h, queries_h, L, W, k, gpu_ids, gpu_d, gpu_st,
                  family, nsample=256, **okw):
    """The first `nsample` rows of the TIMED launch's own output buffers (ids, distances, stats of its last repetition)
    against the CPU oracle on the same row bytes and the same graph -- no separate search is issued for the check, so
    the kernel that is checked is the kernel that was timed (`kernel_family`: the families that served the timed loop)."""
    import oracle
    oix = oracle.Index(odt, metric, dim, n, R, start_rows, **okw)
    oix.rows[:n, :oix.row_bytes] = stored_rows_h.view(np.uint8).reshape(n, -1)[:, :oix.row_bytes]
    oix.adj[:] = prov.download_graph()
    qh = queries_h[:nsample]
    ns = int(qh.shape[0])
    gi = np.ascontiguousarray(gpu_ids[:ns]).view(np.uint32)
    gd = np.ascontiguousarray(gpu_d[:ns]).view(np.uint32)
    gst = np.ascontiguousarray(gpu_st[:ns]).view(np.uint32)
    oi, od, oc, ost = oix.search_batch(qh, L, W, k, threads=min(16, os.cpu_count() or 1), fast=True)
    return {"queries": ns, "rows_of": "the timed launch's own output buffers", "kernel_family": family,
            "ids_identical_to_gpu": bool(np.array_equal(gi, oi)),
            "distances_cmps_hops_identical": bool(np.array_equal(gd, od.view(np.uint32)) and
                                                  np.array_equal(gst[:, 0], ost[:, 0]) and
                                                  np.array_equal(gst[:, 1], ost[:, 1]))}


def callers_variant(prov, qh, L, k, same_as_oracle):
    """N host threads x single-query calls on the shared index (diskann-benchmark-core/src/search/api.rs:409-425): through
    the launch path (every call its own kernel launch, calls of different threads side by side on the context pool) and
    through the resident server (dann_search_submit / dann_search_wait; depth = tickets outstanding per thread, 1 =
    strictly synchronous calls).  Native threads (dann_debug_concurrent_callers); latency = submit -> result in the
    caller's buffer, host clock."""
    res = {}
    for threads in (1, 16):
        nq = 1500 * threads
        prov.concurrent_callers(qh[:128 * threads], L, k, threads=threads, mode=0)
        sc0 = prov.small_call_stats()
        ids, d, lat, secs = prov.concurrent_callers(qh[:nq], L, k, threads=threads, mode=0)
        sc1 = prov.small_call_stats()
        res[f"launch_path_{threads}_threads"] = {"qps": nq / secs, "queries": nq, **pct(lat),
                                                # calls that arrive side by side share a launch (api.hip: small_call)
                                                "calls_per_launch": (sc1[1] - sc0[1]) / max(sc1[0] - sc0[0], 1),
                                                "ids_identical_to_oracle": same_as_oracle(ids, d, L, nq)}
    prov.server_start(L, k, workers=1024, ring=8192)
    try:
        for threads, depth in ((1, 1), (16, 1), (16, 64)):
            nq = min(qh.shape[0], 2000 if depth == 1 and threads == 1 else 20000 if depth == 1 else 100000)
            prov.concurrent_callers(qh[:2048], L, k, threads=threads, mode=1, depth=depth)
            ids, d, lat, secs = prov.concurrent_callers(qh[:nq], L, k, threads=threads, mode=1, depth=depth)
            res[f"server_{threads}_threads_depth_{depth}"] = {"qps": nq / secs, "queries": nq, "workers": 1024, **pct(lat),
                                                             "ids_identical_to_oracle": same_as_oracle(ids, d, L, nq)}
        sub, rel = prov.server_stats()
        res["server_tickets"], res["server_relaunches"] = sub, rel
    finally:
        prov.server_stop()
    return res


def sq8_variant(args, torch, da, lib, _ffi, dev, local, base, queries, gt, medoid, k, W, Lf32, full_prov):
    # ScalarQuantizationParameters::train (scalar/train.rs:33-52, standard_deviations = 2, the reference's default)
    # on a 131 072-row sample, through the library
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    sample = base[torch.randperm(args.n, generator=g, device=dev)[:min(args.n, 131072)]].cpu().numpy()
    shift, scale, _ = da.sq8_train(sample, 2.0, device=local)
    snorm = float(np.float32((shift ** 2).sum(dtype=np.float32)))
    codes = da.sq8_compress(base.cpu().numpy(), shift, scale, device=local)
    qcodes = da.sq8_compress(queries.cpu().numpy(), shift, scale, device=local)
    prov = da.Provider(da.SQ8, da.L2, args.dim, args.n, args.max_degree, codes[medoid:medoid + 1], device=local,
                       sq_scale=scale, sq_shift_norm_sq=snorm, row_stride=args.sq8_stride)
    prov.set_elements(0, codes)
    t0 = time.time()
    cfg = da.build_config(args.pruned_degree, args.max_degree, args.l_build, intra_batch_candidates=da.IBC_NONE)
    prov.build(cfg, 0, args.n, args.growth, args.max_batch)
    t_build = time.time() - t0
    dq = torch.from_numpy(qcodes).to(dev)
    d_ids = torch.empty((args.nq, k), dtype=torch.int32, device=dev)
    d_d = torch.empty((args.nq, k), dtype=torch.float32, device=dev)
    d_st = torch.empty((args.nq, 5), dtype=torch.int32, device=dev)

    def run(L):
        _ffi.check(lib.dann_search_batch_device(prov._h, C.c_void_p(dq.data_ptr()), args.nq, L, W, k,
                                                C.c_void_p(d_ids.data_ptr()), C.c_void_p(d_d.data_ptr()),
                                                C.c_void_p(d_st.data_ptr())), "dann_search_batch_device")
    res = {"row_bytes": args.dim + 4, "build_seconds": round(t_build, 2)}
    # (a) SQ-8 distances only (no full-precision data touched): recall saturates below the f32 index
    run(Lf32)
    res["no_rerank_L%d" % Lf32] = {"recall_at_10_vs_exact_f32": round(
        recall_at_k(d_ids.cpu().numpy().view(np.uint32), gt, k), 4)}
    # (b) the reference's deployment mode for quantised stores: search on the codes, then the Rerank
    # post-processor (full_precision.rs:348-397) re-scores the L candidates with the f32 rows
    d_out = torch.empty((args.nq, k), dtype=torch.int32, device=dev)
    d_outd = torch.empty((args.nq, k), dtype=torch.float32, device=dev)

    def run_rr(L):
        cand = torch.empty((args.nq, L), dtype=torch.int32, device=dev)
        cd = torch.empty((args.nq, L), dtype=torch.float32, device=dev)
        run_rr.last = (cand, cd)
        _ffi.check(lib.dann_search_batch_device(prov._h, C.c_void_p(dq.data_ptr()), args.nq, L, W, L,
                                                C.c_void_p(cand.data_ptr()), C.c_void_p(cd.data_ptr()),
                                                C.c_void_p(d_st.data_ptr())), "dann_search_batch_device")
        _ffi.check(lib.dann_rerank_batch_device(full_prov._h, C.c_void_p(queries.data_ptr()), args.nq,
                                                C.c_void_p(cand.data_ptr()), L, k, C.c_void_p(d_out.data_ptr()),
                                                C.c_void_p(d_outd.data_ptr())), "dann_rerank_batch_device")
    chosen, rec = None, 0.0
    for L in [10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30, 32, 36, 40, 48, 56, 64, 80, 96, 128]:
        run_rr(L)
        rec = recall_at_k(d_out.cpu().numpy().view(np.uint32), gt, k)
        chosen = L
        if rec >= args.target_recall:
            break
    st = d_st.cpu().numpy().view(np.uint32)
    for _ in range(2):
        run_rr(chosen)
    torch.cuda.synchronize()
    prov.kernel_time_reset()
    fam0 = prov.search_families()
    t0 = time.perf_counter()
    for _ in range(10):
        run_rr(chosen)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    family = family_of(fam0, prov.search_families())
    kms, kn = prov.kernel_time(0)
    search_ms = kms / max(kn, 1)
    alg_search = int(st[:, 0].sum()) * (args.dim + 4) + int(st[:, 1].sum()) * (args.max_degree + 1) * 4
    alg = alg_search + args.nq * chosen * args.dim * 4
    import oracle
    try:  # the candidate lists (k = L) the timed loop's last search wrote, against the oracle
        cand, cd = run_rr.last
        res["oracle_sample"] = oracle_sample(prov, oracle.SQ8, oracle.L2, args.dim, args.n, args.max_degree,
                                             codes[medoid:medoid + 1], codes, qcodes, chosen, W, chosen,
                                             cand[:256].cpu().numpy(), cd[:256].cpu().numpy(), d_st[:256].cpu().numpy(),
                                             family, sq_scale=scale, sq_shift_norm_sq=snorm)
    except Exception as e:  # noqa: BLE001
        res["oracle_sample"] = {"error": str(e)[:200]}
    res["search_kernel"] = {"kernel_family": family, "avg_kernel_ms": search_ms, "qps_search_only": args.nq / (search_ms * 1e-3),
                            "algorithmic_bytes_per_launch": alg_search,
                            "frac_of_hbm_peak": alg_search / (search_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    res["with_rerank"] = {"L": chosen, "recall_at_10_vs_exact_f32": round(rec, 4), "qps": args.nq / dt,
                          "mean_cmps": float(st[:, 0].mean()), "ms_per_100k_queries": dt * 1e3 * 1e5 / args.nq,
                          "algorithmic_bytes_per_query": alg / args.nq}
    return res


def pq_variant(args, torch, da, lib, _ffi, dev, local, base, queries, gt, medoid, k, W, full_prov):
    """PQ codes + the f32 index's graph, lookup-table beam search, Rerank on the f32 rows.  The codebook is
    trained with dann_pq_train (LightPQTrainingParameters::train: k-means++ seeding + 10 Lloyd iterations, the random
    draws from numpy generators standing in for the reference's per-chunk StdRng) on a 131 072-row sample; the rows are
    compressed with dann_pq_compress."""
    nch, dim = args.pq_chunks, args.dim
    bounds = np.linspace(0, dim, nch + 1).round().astype(np.uint32)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    sample = base[torch.randperm(args.n, generator=g, device=dev)[:min(args.n, 131072)]].cpu().numpy()
    gens = [np.random.default_rng(700 + c) for c in range(nch)]  # stand-ins for the per-chunk StdRng of the reference
    t_train = time.perf_counter()
    pivots_h = da.pq_train(sample, bounds, 256, 10, lambda c, n: int(gens[c].integers(0, n)),
                           lambda c, h: float(gens[c].random() * h), device=local)
    t_train = time.perf_counter() - t_train
    t_comp = time.perf_counter()
    codes_h = da.pq_compress(pivots_h, bounds, base.cpu().numpy(), device=local)
    t_comp = time.perf_counter() - t_comp
    prov = da.Provider(da.PQ, da.L2, dim, args.n, args.max_degree, codes_h[medoid:medoid + 1], device=local,
                       pq_pivots=pivots_h, pq_offsets=bounds)
    prov.set_elements(0, codes_h)
    prov.upload_graph(full_prov.download_graph())
    t_pack = None
    if not args.no_pq_pack:  # opt-in search layout: adjacency + the neighbours' code rows in one 64-byte-aligned row
        t_pack = time.perf_counter()
        prov.pq_pack_neighbors()
        t_pack = time.perf_counter() - t_pack
    d_st = torch.empty((args.nq, 5), dtype=torch.int32, device=dev)
    d_out = torch.empty((args.nq, k), dtype=torch.int32, device=dev)
    d_outd = torch.empty((args.nq, k), dtype=torch.float32, device=dev)

    def run_rr(L):
        cand = torch.empty((args.nq, L), dtype=torch.int32, device=dev)
        cd = torch.empty((args.nq, L), dtype=torch.float32, device=dev)
        run_rr.last = (cand, cd)
        _ffi.check(lib.dann_search_batch_device(prov._h, C.c_void_p(queries.data_ptr()), args.nq, L, W, L,
                                                C.c_void_p(cand.data_ptr()), C.c_void_p(cd.data_ptr()),
                                                C.c_void_p(d_st.data_ptr())), "dann_search_batch_device")
        _ffi.check(lib.dann_rerank_batch_device(full_prov._h, C.c_void_p(queries.data_ptr()), args.nq,
                                                C.c_void_p(cand.data_ptr()), L, k, C.c_void_p(d_out.data_ptr()),
                                                C.c_void_p(d_outd.data_ptr())), "dann_rerank_batch_device")
    chosen, rec = None, 0.0
    for L in ([args.L] if args.L else [16, 20, 24, 28, 32, 36, 40, 48, 56, 64, 80, 96, 128, 160, 192, 256]):
        run_rr(L)
        rec = recall_at_k(d_out.cpu().numpy().view(np.uint32), gt, k)
        chosen = L
        if rec >= args.target_recall:
            break
    st = d_st.cpu().numpy().view(np.uint32)
    for _ in range(2):
        run_rr(chosen)
    torch.cuda.synchronize()
    prov.kernel_time_reset()
    fam0 = prov.search_families()
    t0 = time.perf_counter()
    for _ in range(10):
        run_rr(chosen)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    family = family_of(fam0, prov.search_families())
    kms, kn = prov.kernel_time(0)
    search_ms = kms / max(kn, 1)
    alg_search = int(st[:, 0].sum()) * nch + int(st[:, 1].sum()) * (args.max_degree + 1) * 4
    alg = alg_search + args.nq * chosen * dim * 4
    # what bounds the lookup-table search: not HBM.  Per query the wave builds a 256 x chunks table in LDS from the
    # pivots (chunks x 256 x (dim / chunks) multiply-adds, pivots L2-resident) and then does one dependent LDS lookup
    # per chunk and candidate; the code rows are 16-byte gathers of which a 64-byte sector is fetched.
    lut_flop = 2.0 * 256 * dim * args.nq
    lds_lookups = int(st[:, 0].sum()) * nch
    import oracle
    try:
        cand, cd = run_rr.last  # the candidate lists (k = L) of the timed loop's last search
        osample = oracle_sample(prov, oracle.PQ, oracle.L2, dim, args.n, args.max_degree, codes_h[medoid:medoid + 1], codes_h,
                                queries.cpu().numpy(), chosen, W, chosen, cand[:256].cpu().numpy(), cd[:256].cpu().numpy(),
                                d_st[:256].cpu().numpy(), family, pq_pivots=pivots_h, pq_offsets=bounds)
    except Exception as e:  # noqa: BLE001
        osample = {"error": str(e)[:200]}
    f16_cmp = None
    if getattr(args, "pq768", False):
        # the same graph over the rows stored as f16 (Full<f16>, config 5's replica), searched at full precision: first L
        # of the sweep that reaches the recall the PQ + Rerank search reached
        p16 = da.Provider(da.F16, da.L2, dim, args.n, args.max_degree, base[medoid:medoid + 1].half().cpu().numpy(), device=local)
        for s0 in range(0, args.n, 1 << 18):
            p16.set_elements(s0, base[s0:s0 + (1 << 18)].half().cpu().numpy())
        p16.upload_graph(full_prov.download_graph())
        q16 = queries.half().contiguous()
        L16, rec16, st16, run16, _ = _sweep(torch, lib, _ffi, p16, q16, args.nq, k, W, gt, len(gt),
                                            [10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 32, 36, 40, 48, 56, 64, 80, 96, 128],
                                            min(rec, args.target_recall) if args.target_recall > 0 else 0.95)
        if L16:
            run16(L16)
            p16.kernel_time_reset()
            for _ in range(5):
                run16(L16)
            torch.cuda.synchronize()
            ms16, n16 = p16.kernel_time(0)
            f16_cmp = {"rows": "f16, 1536 B", "L": L16, "recall_at_10_vs_exact_f32": round(rec16, 4),
                       "mean_cmps": float(st16[:, 0].mean()), "avg_kernel_ms": ms16 / max(n16, 1),
                       "qps": args.nq / (ms16 / max(n16, 1) * 1e-3),
                       "bytes_per_point": dim * 2, "pq_bytes_per_point": nch}
        p16.close() if hasattr(p16, "close") else None
    traffic = None
    try:  # fabric traffic of the search kernel from the committed PMC pass (rocprofv3 cannot run inside bench.py)
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_pq_latest.json")))
        wl = pm["workload"]
        if (wl["nq"], wl["L"], wl["n"], wl["dim"], wl["chunks"]) == (args.nq, chosen, args.n, dim, nch):
            traffic = {"fabric_bytes_per_launch": pm["fabric_bytes_per_launch_corrected"],
                       "traffic_over_algorithmic": pm["fabric_bytes_per_launch_corrected"] / alg_search,
                       "l2_hit_rate": pm["l2_hit_rate"], "source": "profiles/pmc_pq_latest.json"}
    except (OSError, KeyError, ValueError):
        pass
    return {"oracle_sample": osample, "chunks": nch, "row_bytes": nch, "L": chosen, "recall_at_10_vs_exact_f32": round(rec, 4),
            "qps": args.nq / dt, "mean_cmps": float(st[:, 0].mean()), "mean_hops": float(st[:, 1].mean()),
            "search_kernel_traffic": traffic, "full_precision_f16_same_graph": f16_cmp,
            "algorithmic_bytes_per_query": alg / args.nq, "graph": "the f32 index's graph (full-precision build)",
            "search_kernel": {"kernel": "pq_search_kernel" if family == "pq_lut" else "beam_search_kernel<DT_PQ>",
                              "kernel_family": family, "avg_kernel_ms": search_ms,
                              "qps_search_only": args.nq / (search_ms * 1e-3),
                              "algorithmic_bytes_per_launch": alg_search,
                              "algorithmic_GBps": alg_search / (search_ms * 1e-3) / 1e9,
                              "frac_of_hbm_peak": alg_search / (search_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "bound": (f"vector-instruction issue / LDS crossbar (profiles/r05_pq_*): the query's {nch} x 256 f32 table "
                                        f"lives in {64 * ((nch + 15) // 16)} VGPRs per lane and is looked up with ds_bpermute_b32 (4 "
                                        "permutes + a bit-field select per chunk and candidate); "
                                        f"{16 if nch <= 16 else 8 if nch <= 48 else 4} queries per CU; one contiguous read "
                                        "per hop with the packed layout") if family == "pq_lut" else
                                       ("LDS / latency, not HBM: 16-byte code rows (a 64-byte sector each), a 16 KB lookup "
                                        "table per query in LDS, one dependent LDS lookup per chunk and candidate"),
                              "queries_per_cu": (16 if nch <= 16 else 8 if nch <= 48 else 4) if family == "pq_lut" else
                                                max(1, 160 // (10 + nch)),
                              "table_lookups_per_launch": lds_lookups,
                              "table_lookups_per_s": lds_lookups / (search_ms * 1e-3),
                              "lut_build_flop_per_launch": lut_flop},
            "rerank_share_of_time": max(0.0, 1.0 - search_ms * 1e-3 / dt),
            "packed_neighbor_codes": None if t_pack is None else {
                "seconds": round(t_pack, 4), "bytes": (args.n + 1) * ((((args.max_degree + 1) * 4 + 15) // 16 * 16 +
                                                                       (nch + 15) // 16 * 16 * args.max_degree + 63) // 64 * 64)},
            "train_seconds_kmeanspp_plus_10_lloyds_131072_rows": round(t_train, 3),
            "compress_seconds_incl_pcie": round(t_comp, 3)}


def _index_on_gpu(args, torch, da, dev, local, n, dim, dist, R, pruned, l_build, nq, max_batch, f16=False):
    base, queries = make_data(torch, dev, n, dim, nq, dist, 0xD15CA11, 0xD15CA12)
    mean = base.double().mean(0).float()
    medoid = int(torch.argmin(((base - mean[None, :]) ** 2).sum(1)).item())
    if f16:  # the reference's Full<f16> provider: rows and queries stored as f16 (ground truth stays on the f32 data)
        rows = base.half()
        start = rows[medoid:medoid + 1].cpu().numpy()
        prov = da.Provider(da.F16, da.L2, dim, n, R, start, device=local)
    else:
        rows = base
        start = base[medoid:medoid + 1].cpu().numpy()
        prov = da.Provider(da.F32, da.L2, dim, n, R, start, device=local)
    if args.visited_bits:  # experiment knob: explicit LDS visited-table size (never affects results)
        prov.set_visited_bits(args.visited_bits)
    for s0 in range(0, n, 1 << 21):
        prov.set_elements(s0, rows[s0:s0 + (1 << 21)].cpu().numpy())
    if f16:
        base = (base, rows)  # (f32 for the ground truth, f16 as stored)
        queries = (queries, queries.half().contiguous())
    t0 = time.time()
    cfg = da.build_config(pruned, R, l_build, intra_batch_candidates=da.IBC_NONE)
    prov.build(cfg, 0, n, args.growth, max_batch)
    torch.cuda.synchronize()
    return prov, base, queries, start, time.time() - t0


F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense (= the f32 vector rate)


F16_MFMA_PEAK_TFLOPS = 2516.6  # MI355X_MICROARCH.md: dense bf16 / fp16 peak (v_mfma_f32_32x32x16_f16 = 16 x the f32 core)


def build_mfma_info(prov, n, dim, row_bytes, t_build, f16=False):
    """The matrix-core share of an index build just finished on `prov` (north_star: "MFMA only for the dense ... case at
    index-build time ... MFMA utilisation against gfx950 peak"): flop from the library's work counter (Gram entries
    computed x dim x 2), time from HIP events around every gram_tiles_kernel launch on the build stream
    (dann_kernel_time, which = 5)."""
    c = [int(x) for x in prov.build_counters()]
    ms, launches = prov.kernel_time(5)
    if not launches:
        return {"kernel": "gram_tiles_kernel", "used": False,
                "note": "rows below 1 KiB keep the row kernels (measured faster there, DESIGN 3.4)"}
    flop = 2.0 * c[7] * dim
    tf = flop / (ms * 1e-3) / 1e12
    # f16 rows (round 6): v_mfma_f32_32x32x16_f16.  Its own peak is 16 x the f32 core's: at that rate the Gram of a list is
    # no longer matrix-pipe work at all -- the kernel is bound by filling its LDS slabs (rows x dim x 2 bytes once per list
    # from L2 / HBM) -- so the line carries both fractions: of the f16 peak (what the instruction could do) and of the f32
    # peak (what the same tiles cost until round 5)
    peak = F16_MFMA_PEAK_TFLOPS if f16 else F32_MFMA_PEAK_TFLOPS
    return {"kernel": "gram_tiles_f16_kernel (v_mfma_f32_32x32x16_f16)" if f16 else "gram_tiles_kernel (v_mfma_f32_32x32x2_f32)",
            "used": True, "launches": int(launches),
            "total_ms": ms, "flop": flop, "TFLOP_s": tf, "peak": peak, "frac": tf / peak,
            **({"frac_of_f32_matrix_peak": tf / F32_MFMA_PEAK_TFLOPS,
                "slab_fill_bytes": c[6] * dim * 2, "slab_fill_GBps": c[6] * dim * 2 / (ms * 1e-3) / 1e9} if f16 else {}),
            "share_of_prune_pairs": (c[8] - c[9]) / max(1, c[8] - c[9] + c[4]),
            "pairs_asked_by_the_sweeps": c[8], "of_those_re_evaluated_exactly": c[9], "row_kernel_pairs": c[4],
            "gram_rows": c[6], "gram_entries": c[7], "build_seconds": t_build, "points_per_s": n / t_build,
            "insert_search_comparisons": c[2], "insert_search_algorithmic_bytes": c[2] * row_bytes}


def _sweep(torch, lib, _ffi, prov, queries, nq, k, W, gt, ngt, sweep, target):
    """first L of `sweep` with recall@10 >= target over the first `ngt` queries; returns (L or None, recall, stats, ids)"""
    dev = queries.device
    d_ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
    d_d = torch.empty((nq, k), dtype=torch.float32, device=dev)
    d_st = torch.empty((nq, 5), dtype=torch.int32, device=dev)

    def run(L):
        _ffi.check(lib.dann_search_batch_device(prov._h, C.c_void_p(queries.data_ptr()), nq, L, W, k,
                                                C.c_void_p(d_ids.data_ptr()), C.c_void_p(d_d.data_ptr()),
                                                C.c_void_p(d_st.data_ptr())), "dann_search_batch_device")
    run.bufs = (d_ids, d_d, d_st)  # what the last run(L) wrote
    rec, hist = 0.0, []
    for L in sweep:
        run(L)
        rec = recall_at_k(d_ids[:ngt].cpu().numpy().view(np.uint32), gt, k)
        hist.append((L, round(rec, 4)))
        if rec >= target:
            return L, rec, d_st.cpu().numpy().view(np.uint32), run, hist
    return None, rec, d_st.cpu().numpy().view(np.uint32), run, hist


def uniform_variant(args, torch, da, lib, _ffi, dev, local, k, W):
    """SURVEY.md 8(d)'s headline distribution: N x 128 i.i.d. U(-1,1) (the reference's own test distribution), nq = the
    protocol's 10 000 queries.  An R = 32 graph does not reach recall 0.95 on it at any L <= 500 (no cluster
    structure, intrinsic dimension 128); the line says so with the numbers instead of omitting the distribution."""
    nq = 10000
    prov, base, queries, start, t_build = _index_on_gpu(args, torch, da, dev, local, args.n, args.dim, "uniform",
                                                        args.max_degree, args.pruned_degree, args.l_build, nq,
                                                        args.max_batch)
    gt = ground_truth(torch, base, queries, k)
    sweep = [10, 20, 32, 48, 64, 96, 128, 192, 256, 500]  # SURVEY.md 8d sweep (+500)
    L, rec, st, run, hist = _sweep(torch, lib, _ffi, prov, queries, nq, k, W, gt, nq, sweep, args.target_recall)
    res = {"data": f"{args.n}x{args.dim} f32 i.i.d. U(-1,1), {nq} queries", "build_seconds": round(t_build, 2),
           "recall_at_10_by_L": hist, "first_L_with_recall_0.95": L}
    Lq = L or 64
    run(Lq)
    st = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        run(Lq)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    res["qps_at_L%d" % Lq] = nq / dt
    prov.close()
    return res


def large_variant(args, spec, torch, da, lib, _ffi, dev, local, k, W, stream_read_gbps):
    """The beam-search kernel on an index whose working set is >> the 256 MiB Infinity Cache: by default 10 M x 128 f32
    (5.1 GB of rows + 1.3 GB of adjacency), the headline generator with the per-blob density kept (2 560 blobs).  Recall
    is measured on the first 10 000 queries of the 100 000-query batch, the kernel is timed with HIP events over the
    whole batch, and a sample of the queries is re-run through the CPU oracle on the same graph bytes."""
    import oracle
    f = spec.split(":")
    f16 = f[-1] == "f16"
    if f16:
        f = f[:-1]
    n, dim = int(f[0]), int(f[1])
    dist = ":".join(f[2:-3])
    R, pruned, l_build = int(f[-3]), int(f[-2]), int(f[-1])
    nq, ngt = args.nq, min(args.nq, 10000)
    max_batch = 65536 if n >= 4_000_000 and dim <= 256 else 16384
    prov, base, queries, start, t_build = _index_on_gpu(args, torch, da, dev, local, n, dim, dist, R, pruned, l_build, nq,
                                                        max_batch, f16=f16)
    stored = base
    if f16:
        (base, stored), (queries32, queries) = base, queries
    else:
        queries32 = queries
    esz, tname, odt = (2, "f16", oracle.F16) if f16 else (4, "f32", oracle.F32)
    mfma = build_mfma_info(prov, n, dim, dim * esz, t_build, f16=f16)
    sweep = [10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30, 32, 36, 40, 48, 56, 64, 80, 96, 128, 160, 192, 256]
    if args.L:  # profiling passes: fixed L, no ground truth
        gt = np.zeros((ngt, k), np.int64)
        L, rec, st, run, hist = _sweep(torch, lib, _ffi, prov, queries, nq, k, W, gt, ngt, [args.L], -1.0)
        rec, reached = float("nan"), False
    else:
        gt = ground_truth(torch, base, queries32[:ngt], k)
        L, rec, st, run, hist = _sweep(torch, lib, _ffi, prov, queries, nq, k, W, gt, ngt, sweep, args.target_recall)
        reached = L is not None
    L = L or sweep[-1]
    run(L)
    prov.kernel_time_reset()
    fam0 = prov.search_families()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        run(L)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    family = family_of(fam0, prov.search_families())
    ms, launches = prov.kernel_time(0)
    avg_ms = ms / max(launches, 1)
    row_bytes, adj_bytes = dim * esz, (R + 1) * 4
    alg = int(st[:, 0].sum()) * row_bytes + int(st[:, 1].sum()) * adj_bytes
    achieved = alg / (avg_ms * 1e-3) / 1e9
    res = {
        "workload": f"batched beam search over a {n}x{dim} {tname} index ({dist}; {n * row_bytes / 1e9:.1f} GB rows + "
                    f"{(n + 1) * adj_bytes / 1e9:.1f} GB adjacency resident in HBM), {nq} queries/launch, k=10, L={L}, "
                    f"beam_width={W}; Vamana R={R} (pruned {pruned}), l_build={l_build}, built on the GPU in "
                    f"{t_build:.1f} s",
        "recall_at_10": round(rec, 4), "recall_target_reached": reached, "recall_measured_on": f"first {ngt} queries",
        "L": L, "qps": nq / dt, "mean_cmps": float(st[:, 0].mean()), "mean_hops": float(st[:, 1].mean()),
        "kernel": "beam_search_kernel", "kernel_family": family, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": alg, "avg_kernel_ms": avg_ms,
        "working_set_bytes": n * row_bytes + (n + 1) * adj_bytes,
        "build": mfma,
    }
    if stream_read_gbps:
        res["measured_stream_read_GBps"] = stream_read_gbps
        res["frac_of_measured_stream_read"] = achieved / stream_read_gbps
        if achieved > stream_read_gbps:
            # more than this box streams out of HBM: the Infinity Cache serves part of it (100 000 clustered queries per
            # launch read every hot row many times) -- a fabric-side rate, not an HBM-side one
            res["bound"] = "fabric (HBM + Infinity Cache): above this box's stream-read probe"
    if args.L:
        prov.close()
        return res
    # parity at this scale: a sample of the batch through the CPU oracle on the same rows and graph
    ms_n = 256
    adj = prov.download_graph()
    oix = oracle.Index(odt, oracle.L2, dim, n, R, start)
    for s0 in range(0, n, 1 << 21):
        blk = stored[s0:s0 + (1 << 21)].cpu().numpy()
        oix.rows[s0:s0 + blk.shape[0], :] = blk.view(np.uint8).reshape(blk.shape[0], -1)
    oix.adj[:] = adj
    qh = queries[:ms_n].cpu().numpy()
    # the first rows of the TIMED launch's own output buffers (its last repetition), not a separate search
    t_ids, t_d, t_st = run.bufs
    gi = t_ids[:ms_n].cpu().numpy().view(np.uint32)
    gd = t_d[:ms_n].cpu().numpy().view(np.uint32)
    gst = t_st[:ms_n].cpu().numpy().view(np.uint32)
    oi, od, oc, ost = oix.search_batch(qh, L, W, k, threads=min(16, os.cpu_count() or 1), fast=True)
    res["oracle_sample"] = {"queries": ms_n, "rows_of": "the timed launch's own output buffers", "kernel_family": family,
                            "ids_identical_to_gpu": bool(np.array_equal(gi, oi)),
                            "distances_cmps_hops_identical": bool(
                                np.array_equal(gd, od.view(np.uint32)) and
                                np.array_equal(gst[:, 0], ost[:, 0]) and np.array_equal(gst[:, 1], ost[:, 1]))}
    # ---- what part of the launch's reads HBM must have served.  The fabric-side counters (FETCH_SIZE = TCC_EA0_RDREQ,
    # TCC_EA0_RDREQ_DRAM) count requests the L2 sends towards memory and cannot tell an Infinity-Cache hit from a DRAM
    # access, so the split is made from the algorithm: the rows a launch reads for the FIRST time must come from DRAM (the
    # 256 MiB cache cannot hold them from the launch before: the index is 25 x its size), re-reads may come from anywhere.
    # Distinct rows = the union over all queries of the neighbours of the nodes they expanded (every neighbour of an
    # expanded node is evaluated by that query once); distinct adjacency rows = the expanded nodes.
    try:
        rid, _, rn, _ = prov.search_record_queries(queries.cpu().numpy(), L)
        rid_t = torch.from_numpy(rid.view(np.int32)).to(dev)
        valid = torch.arange(rid.shape[1], device=dev)[None, :] < torch.from_numpy(rn.astype(np.int64)).to(dev)[:, None]
        expanded = torch.unique(rid_t[valid])
        del rid_t, valid
        adj_t = torch.from_numpy(adj.view(np.int32)).to(dev)
        rows_of = adj_t[expanded.long()]
        lens = rows_of[:, 0].clamp(max=R)
        nb = rows_of[:, 1:]
        keep = torch.arange(R, device=dev)[None, :] < lens[:, None]
        distinct_rows = int(torch.unique(nb[keep]).numel()) + 1  # (+ the start point)
        distinct_adj = int(expanded.numel())
        del adj_t, rows_of, nb, keep
        compulsory = distinct_rows * row_bytes + distinct_adj * adj_bytes
        res["hbm_side"] = {
            "distinct_rows_per_launch": distinct_rows, "distinct_adjacency_rows_per_launch": distinct_adj,
            "first_touch_bytes_per_launch": compulsory, "share_of_algorithmic_bytes": compulsory / alg,
            "dram_rate_at_least_GBps": compulsory / (avg_ms * 1e-3) / 1e9,
            # how likely is a RE-read to find its row still in the 256 MiB Infinity Cache?  The cache turns over in
            # 256 MiB / (fabric rate); a row is read again reads_per_row times per launch at times spread over the launch
            "mean_reads_per_distinct_row": int(st[:, 0].sum()) / max(distinct_rows, 1),
            "infinity_cache_turnover_us": (256 << 20) / (achieved * 1e9) * 1e6,
            "mean_interval_between_reads_of_a_row_us": avg_ms * 1e3 / max(int(st[:, 0].sum()) / max(distinct_rows, 1), 1e-9),
            "model_share_of_rereads_within_one_turnover": 1.0 - float(np.exp(-((256 << 20) / (achieved * 1e9) * 1e6) /
                                                                    (avg_ms * 1e3 / max(int(st[:, 0].sum()) / max(distinct_rows, 1), 1e-9)))),
            "note": "first-touch bytes must come from DRAM; the other reads of the launch (the same rows again, by other "
                    "queries) are served by L2 / Infinity Cache / DRAM in a mix no counter of this rocprofv3 separates "
                    "(TCC_EA0_RDREQ_DRAM counts requests towards the memory controller, Infinity-Cache hits included). The "
                    "model line: a re-read hits the Infinity Cache only if it falls within one cache turnover of the previous "
                    "read of that row (Poisson arrivals)"}
    except Exception as e:  # noqa: BLE001
        res["hbm_side"] = {"error": str(e)[:200]}
    # ---- and the distance kernel where nothing CAN be cached: ExpandBeam::expand_beam batched (expand_beam_kernel) over
    # row ids drawn WITHOUT repetition from the whole store -- every row is read once per launch
    try:
        gq = 20000
        gl = max(1, min(256, n // gq))
        gids = np.random.default_rng(7).permutation(n)[:gq * gl].astype(np.uint32)
        goff = np.arange(gq + 1, dtype=np.uint64) * gl
        qh_g = queries[:gq].cpu().numpy()
        prov.expand_beam_batch(qh_g, gids, goff)
        prov.kernel_time_reset()
        for _ in range(3):
            prov.expand_beam_batch(qh_g, gids, goff)
        gms, gn = prov.kernel_time(1)
        gbytes = gq * gl * row_bytes
        grate = gbytes / (gms / max(gn, 1) * 1e-3) / 1e9
        res["hbm_side_distance_kernel"] = {
            "kernel": "expand_beam_kernel", "rows_per_launch": gq * gl, "every_row_read_once": True,
            "bytes_per_launch": gbytes, "avg_kernel_ms": gms / max(gn, 1), "achieved_GBps": grate,
            "frac_of_hbm_peak": grate / HBM_PEAK_GBS,
            **({"frac_of_measured_stream_read": grate / stream_read_gbps} if stream_read_gbps else {})}
    except Exception as e:  # noqa: BLE001
        res["hbm_side_distance_kernel"] = {"error": str(e)[:200]}
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_large_latest.json")))
        if pm["workload"] == spec and pm["L"] == L and pm["nq"] == nq:
            res["traffic"] = pm["hbm_bytes_per_launch_corrected"]
            res["traffic_source"] = "profiles/pmc_large_latest.json (FETCH_SIZE x2 + WRITE_SIZE, KiB, separate passes)"
    except (OSError, KeyError, ValueError):
        pass
    if dim >= 512 and not f16 and not args.no_pq:
        # config 5's row shape through the PQ path: the same rows as 64-byte codes (chunks of dim / 64 columns), the same
        # graph, lookup-table search with the table in four register groups, Rerank on these f32 rows -- next to this leg's
        # full-precision number and the f16 leg's (`bench.py --only pq768` runs it alone, with other code lengths)
        try:
            import copy
            pa = copy.copy(args)
            pa.n, pa.dim, pa.max_degree, pa.pq_chunks, pa.L, pa.pq768 = n, dim, R, 64, 0, False
            mean = base.double().mean(0).float()
            med = int(torch.argmin(((base - mean[None, :]) ** 2).sum(1)).item())
            pq = pq_variant(pa, torch, da, lib, _ffi, dev, local, base, queries, gt, med, k, W, prov)
            res["pq64_plus_rerank_same_graph"] = {
                "chunks": 64, "bytes_per_point": 64, "L": pq["L"], "recall_at_10_vs_exact_f32": pq["recall_at_10_vs_exact_f32"],
                "qps": pq["qps"], "search_kernel": {kk: pq["search_kernel"][kk] for kk in
                                                    ("kernel", "kernel_family", "avg_kernel_ms", "qps_search_only", "queries_per_cu")},
                "mean_cmps": pq["mean_cmps"], "mean_hops": pq["mean_hops"], "oracle_sample": pq["oracle_sample"],
                "train_seconds": pq["train_seconds_kmeanspp_plus_10_lloyds_131072_rows"],
                "compress_seconds_incl_pcie": pq["compress_seconds_incl_pcie"],
                "full_precision_qps_this_leg": nq / dt,
                "reading": "a capacity format on this hardware: 48 x fewer bytes per point than these f32 rows for a "
                           "fraction of the queries per second at equal recall (DESIGN.md 3.8)"}
        except Exception as e:  # noqa: BLE001 -- never lose the leg over its side measurement
            res["pq64_plus_rerank_same_graph"] = {"error": str(e)[:300]}
    prov.close()
    return res


def _first_touch(torch, dev, prov, queries_h, L, R, adj, st, row_bytes, adj_bytes, alg, avg_ms, achieved):
    """What part of a launch's reads HBM must have served: the rows (and adjacency rows) a launch reads for the FIRST
    time cannot come from the 256 MiB Infinity Cache when the index is many times its size (large_variant explains the
    counters' blind spot).  Distinct rows = the union over all queries of the neighbours of the nodes they expanded."""
    rid, _, rn, _ = prov.search_record_queries(queries_h, L)
    rid_t = torch.from_numpy(rid.view(np.int32)).to(dev)
    valid = torch.arange(rid.shape[1], device=dev)[None, :] < torch.from_numpy(rn.astype(np.int64)).to(dev)[:, None]
    expanded = torch.unique(rid_t[valid])
    del rid_t, valid
    adj_t = torch.from_numpy(adj.view(np.int32)).to(dev)
    rows_of = adj_t[expanded.long()]
    lens = rows_of[:, 0].clamp(max=R)
    nb = rows_of[:, 1:]
    keep = torch.arange(R, device=dev)[None, :] < lens[:, None]
    distinct_rows = int(torch.unique(nb[keep]).numel()) + 1  # (+ the start point)
    distinct_adj = int(expanded.numel())
    del adj_t, rows_of, nb, keep
    compulsory = distinct_rows * row_bytes + distinct_adj * adj_bytes
    reads_per_row = int(st[:, 0].sum()) / max(distinct_rows, 1)
    turnover_us = (256 << 20) / (achieved * 1e9) * 1e6
    interval_us = avg_ms * 1e3 / max(reads_per_row, 1e-9)
    return {"distinct_rows_per_launch": distinct_rows, "distinct_adjacency_rows_per_launch": distinct_adj,
            "first_touch_bytes_per_launch": compulsory, "share_of_algorithmic_bytes": compulsory / alg,
            "dram_rate_at_least_GBps": compulsory / (avg_ms * 1e-3) / 1e9,
            "mean_reads_per_distinct_row": reads_per_row, "infinity_cache_turnover_us": turnover_us,
            "mean_interval_between_reads_of_a_row_us": interval_us,
            "model_share_of_rereads_within_one_turnover": 1.0 - float(np.exp(-turnover_us / interval_us))}


def large_int_variant(args, kind, torch, da, lib, _ffi, dev, local, k, W, stream_read_gbps):
    """BASELINE config 3's integer rows on a working set far beyond the 256 MiB Infinity Cache: 10 M x 128 u8 rows
    (1.28 GB + 1.32 GB adjacency) or SQ-8 codes (132-byte rows at a 256-byte stride: 2.56 GB), the headline generator at
    2 560 blobs, index built on the GPU over those rows.  The search kernel (two queries per wavefront) is timed at the
    first L whose recall@10 against the exact f32 neighbours reaches the target (no rerank) and at SURVEY 8(a)'s C-int8
    sizing L = 64; algorithmic bytes over kernel time are stated against the 8 TB/s peak AND this box's stream-read probe;
    the first rows of the timed launch's own output are checked against the CPU oracle; the distance kernel alone is run
    over row ids drawn without repetition (every row read once per launch: nothing can be cached)."""
    import oracle
    n, dim, R, pruned, l_build = 10_000_000, 128, 32, 28, 100
    if args.large_int_n:
        n = args.large_int_n
    dist = "sift_like:1:%d" % max(256, n // 3906)
    nq, ngt = args.nq, min(args.nq, 10000)
    base, queries = make_data(torch, dev, n, dim, nq, dist, 0xD15CA11, 0xD15CA12)
    mean = base.double().mean(0).float()
    medoid = int(torch.argmin(((base - mean[None, :]) ** 2).sum(1)).item())
    okw = {}
    if kind == "sq8":
        g = torch.Generator(device=dev)
        g.manual_seed(11)
        sample = base[torch.randperm(n, generator=g, device=dev)[:131072]].cpu().numpy()
        shift, scale, _ = da.sq8_train(sample, 2.0, device=local)
        snorm = float(np.float32((shift ** 2).sum(dtype=np.float32)))
        rows = np.empty((n, dim + 4), np.uint8)
        for s0 in range(0, n, 1 << 20):
            rows[s0:s0 + (1 << 20)] = da.sq8_compress(base[s0:s0 + (1 << 20)].cpu().numpy(), shift, scale, device=local)
        qrows = da.sq8_compress(queries.cpu().numpy(), shift, scale, device=local)
        okw = dict(sq_scale=scale, sq_shift_norm_sq=snorm)
        prov = da.Provider(da.SQ8, da.L2, dim, n, R, rows[medoid:medoid + 1], device=local, row_stride=args.sq8_stride, **okw)
        row_bytes, odt, stride = dim + 4, oracle.SQ8, args.sq8_stride
    else:
        lo, hi = float(base.min()), float(base.max())
        rows = np.empty((n, dim), np.uint8)
        for s0 in range(0, n, 1 << 21):
            rows[s0:s0 + (1 << 21)] = ((base[s0:s0 + (1 << 21)] - lo) * (255.0 / (hi - lo))).round().clamp(0, 255).to(torch.uint8).cpu().numpy()
        qrows = ((queries - lo) * (255.0 / (hi - lo))).round().clamp(0, 255).to(torch.uint8).cpu().numpy()
        prov = da.Provider(da.U8, da.L2, dim, n, R, rows[medoid:medoid + 1], device=local)
        row_bytes, odt, stride = dim, oracle.U8, dim
    for s0 in range(0, n, 1 << 21):
        prov.set_elements(s0, rows[s0:s0 + (1 << 21)])
    t0 = time.time()
    prov.build(da.build_config(pruned, R, l_build, intra_batch_candidates=da.IBC_NONE), 0, n, args.growth, 65536)
    torch.cuda.synchronize()
    t_build = time.time() - t0
    dq = torch.from_numpy(qrows).to(dev)
    sweep = [10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30, 32, 36, 40, 48, 56, 64, 80, 96]
    if args.L:  # profiling passes: fixed L, no ground truth
        Ls, rec_by_L = [args.L], {}
        _, _, _, run, _ = _sweep(torch, lib, _ffi, prov, dq, nq, k, W, np.zeros((ngt, k), np.int64), ngt, [args.L], -1.0)
    else:
        gt = ground_truth(torch, base, queries[:ngt], k)
        Lr, rec, _, run, hist = _sweep(torch, lib, _ffi, prov, dq, nq, k, W, gt, ngt, sweep, args.target_recall)
        rec_by_L = dict(hist)
        Ls = sorted({Lr or sweep[-1], 64})
    del base
    torch.cuda.empty_cache()
    adj_bytes = (R + 1) * 4
    res = {"workload": f"batched beam search over a {n}x{dim} {kind} index ({dist}; {n * stride / 1e9:.2f} GB of rows at a "
                       f"{stride}-byte stride + {(n + 1) * adj_bytes / 1e9:.2f} GB adjacency resident in HBM), {nq} "
                       f"queries/launch, k=10, beam_width={W}; Vamana R={R} (pruned {pruned}), l_build={l_build}, built "
                       f"on the GPU over the {kind} rows in {t_build:.1f} s",
           "rows": kind, "row_bytes": row_bytes, "row_stride": stride, "build_seconds": round(t_build, 2),
           "working_set_bytes": n * stride + (n + 1) * adj_bytes, "recall_at_10_vs_exact_f32_no_rerank_by_L": rec_by_L,
           "recall_measured_on": f"first {ngt} queries", "peak": HBM_PEAK_GBS, "unit": "GB/s",
           **({"measured_stream_read_GBps": stream_read_gbps} if stream_read_gbps else {})}
    t_ids, t_d, t_st = run.bufs
    adj = None
    for L in Ls:
        for _ in range(2):
            run(L)
        prov.kernel_time_reset()
        fam0 = prov.search_families()
        for _ in range(5):
            run(L)
        torch.cuda.synchronize()
        family = family_of(fam0, prov.search_families())
        ms, nl = prov.kernel_time(0)
        avg_ms = ms / max(nl, 1)
        st = t_st.cpu().numpy().view(np.uint32)
        alg = int(st[:, 0].sum()) * row_bytes + int(st[:, 1].sum()) * adj_bytes
        achieved = alg / (avg_ms * 1e-3) / 1e9
        leg = {"L": L, "kernel": "pair_search_kernel" if family == "pair" else "beam_search_kernel", "kernel_family": family,
               "recall_at_10_vs_exact_f32_no_rerank": rec_by_L.get(L), "mean_cmps": float(st[:, 0].mean()),
               "mean_hops": float(st[:, 1].mean()), "algorithmic_bytes_per_launch": alg, "avg_kernel_ms": avg_ms,
               "qps": nq / (avg_ms * 1e-3), "achieved": achieved, "frac": achieved / HBM_PEAK_GBS,
               "bound": "hbm"}
        if stream_read_gbps:
            leg["frac_of_measured_stream_read"] = achieved / stream_read_gbps
            if achieved > stream_read_gbps:
                leg["bound"] = "fabric (HBM + Infinity Cache): above this box's stream-read probe"
        if not args.L:
            try:
                leg["oracle_sample"] = oracle_sample(prov, odt, oracle.L2, dim, n, R, rows[medoid:medoid + 1], rows, qrows, L,
                                                     W, k, t_ids[:256].cpu().numpy(), t_d[:256].cpu().numpy(), st[:256],
                                                     family, **okw)
            except Exception as e:  # noqa: BLE001
                leg["oracle_sample"] = {"error": str(e)[:200]}
            try:
                if adj is None:
                    adj = prov.download_graph()
                leg["hbm_side"] = _first_touch(torch, dev, prov, qrows, L, R, adj, st, row_bytes, adj_bytes, alg, avg_ms,
                                               achieved)
            except Exception as e:  # noqa: BLE001
                leg["hbm_side"] = {"error": str(e)[:200]}
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", f"pmc_large_{kind}_latest.json")))
            if pm.get("n") == n and pm.get("L") == L and pm.get("nq") == nq:
                leg["traffic"] = pm["hbm_bytes_per_launch_corrected"]
                leg["traffic_over_algorithmic"] = pm["hbm_bytes_per_launch_corrected"] / alg
                leg["traffic_source"] = f"profiles/pmc_large_{kind}_latest.json (FETCH_SIZE x2 + WRITE_SIZE, KiB, separate passes)"
        except (OSError, KeyError, ValueError):
            pass
        res["L%d" % L] = leg
    if not args.L:
        # the distance kernel where nothing CAN be cached: row ids drawn WITHOUT repetition, every row read once per launch
        try:
            gq = 20000
            gl = max(1, min(256, n // gq))
            gids = np.random.default_rng(7).permutation(n)[:gq * gl].astype(np.uint32)
            goff = np.arange(gq + 1, dtype=np.uint64) * gl
            prov.expand_beam_batch(qrows[:gq], gids, goff)
            prov.kernel_time_reset()
            for _ in range(3):
                prov.expand_beam_batch(qrows[:gq], gids, goff)
            gms, gn = prov.kernel_time(1)
            gbytes = gq * gl * row_bytes
            grate = gbytes / (gms / max(gn, 1) * 1e-3) / 1e9
            res["hbm_side_distance_kernel"] = {
                "kernel": "expand_beam_kernel", "rows_per_launch": gq * gl, "every_row_read_once": True,
                "bytes_per_launch": gbytes, "avg_kernel_ms": gms / max(gn, 1), "achieved_GBps": grate,
                "frac_of_hbm_peak": grate / HBM_PEAK_GBS,
                # (128-byte rows at a 128- or 256-byte stride: HBM moves whole 128-byte lines, the SQ-8 row straddles two)
                **({"frac_of_measured_stream_read": grate / stream_read_gbps} if stream_read_gbps else {})}
        except Exception as e:  # noqa: BLE001
            res["hbm_side_distance_kernel"] = {"error": str(e)[:200]}
    prov.close()
    return res


class _DevView:
    """a device buffer of the library as a torch tensor (zero copy): __cuda_array_interface__ over the raw address"""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def build_large_variant(args, torch, da, lib, _ffi, dev, local):
    """BASELINE config 5's per-GPU shape on ONE MI355X: n x dim rows generated on the device (the benchmark mixture at the
    headline's per-blob density), Vamana build by dann_build (insert searches + RobustPrune on the matrix cores), search
    evaluation against an exact ground truth, and a replay of a query sample through the CPU oracle."""
    import oracle
    f = args.build_spec.split(":")
    n, dim, R, pruned, lb = (int(x) for x in f[:5])
    f16 = len(f) > 5 and f[5] == "f16"
    esz, tname, dt, odt = (2, "f16", da.F16, oracle.F16) if f16 else (4, "f32", da.F32, oracle.F32)
    nblobs = args.build_blobs or max(256, n // 3906)
    chunk = 1 << 20
    k, nq, ngt = 10, min(args.nq, 20000), 1000
    g0 = torch.Generator(device=dev)
    g0.manual_seed(0xD15CA11)
    hier = args.build_centres == "hier" or (args.build_centres == "auto" and nblobs > 4096)
    if hier:
        # Tens of thousands of i.i.d. centres in U(0,1)^768 are all nearly equidistant (concentration of measure): no
        # graph walk finds the query's blob among them, which is a property of that generator, not of embeddings -- text
        # embeddings cluster hierarchically.  256 topic centres as before; the blobs of a topic sit around it on a shared
        # 8-dimensional subspace with the spread of a blob's own points, so neighbouring blobs overlap and a walk moves
        # between them as it moves inside one.
        ntop = 256
        top = torch.rand((ntop, dim), generator=g0, device=dev, dtype=torch.float32)
        cbasis = torch.randn((8, dim), generator=g0, device=dev, dtype=torch.float32) / 4.0
        basis = torch.randn((16, dim), generator=g0, device=dev, dtype=torch.float32) / 4.0
        tlab = torch.randint(0, ntop, (nblobs,), generator=g0, device=dev)
        w = torch.randn((nblobs, 8), generator=g0, device=dev, dtype=torch.float32)
        centers = top[tlab] + 0.5 * (w @ cbasis)
    else:
        centers = torch.rand((nblobs, dim), generator=g0, device=dev, dtype=torch.float32)
        basis = torch.randn((16, dim), generator=g0, device=dev, dtype=torch.float32) / 4.0

    def draw(m, gen):
        lab = torch.randint(0, nblobs, (m,), generator=gen, device=dev)
        z = torch.randn((m, 16), generator=gen, device=dev, dtype=torch.float32)
        noise = torch.randn((m, dim), generator=gen, device=dev, dtype=torch.float32)
        return centers[lab] + 0.25 * (z @ basis) + 0.02 * noise

    def chunks():  # deterministic: chunk c depends on its own seed only, so every pass sees the same rows
        for c, s0 in enumerate(range(0, n, chunk)):
            g = torch.Generator(device=dev)
            g.manual_seed(0xC0FFEE00 + c)
            yield s0, draw(min(chunk, n - s0), g)
    t0 = time.time()
    gq = torch.Generator(device=dev)
    gq.manual_seed(0xD15CA12)
    queries = draw(nq, gq)
    # pass 1: f64 mean; pass 2: medoid (diskann-utils/src/sampling/medoid.rs:15-48)
    acc = torch.zeros(dim, dtype=torch.float64, device=dev)
    for s0, b in chunks():
        acc += b.double().sum(0)
    mean = (acc / n).float()
    best = (float("inf"), -1, None)
    for s0, b in chunks():
        d = ((b - mean[None, :]) ** 2).sum(1)
        i = int(torch.argmin(d).item())
        if float(d[i]) < best[0]:
            best = (float(d[i]), s0 + i, b[i:i + 1].clone())
    medoid = best[1]
    start = (best[2].half() if f16 else best[2]).cpu().numpy()
    prov = da.Provider(dt, da.L2, dim, n, R, start, device=local)
    # pass 3: rows into the index (device to device) + the exact ground truth of the first `ngt` queries in the same pass
    qd = queries[:ngt].double()
    best_d = torch.full((ngt, k), float("inf"), dtype=torch.float64, device=dev)
    best_i = torch.zeros((ngt, k), dtype=torch.int64, device=dev)
    for s0, b in chunks():
        rows = b.half().contiguous() if f16 else b.contiguous()
        torch.cuda.synchronize()
        prov.set_elements_device(s0, rows.data_ptr(), rows.shape[0])
        bn = (b * b).sum(1)
        d = bn[None, :] - 2.0 * (queries[:ngt] @ b.T)
        cand = torch.topk(d, 4 * k, dim=1, largest=False).indices
        del d
        diff = b[cand].double() - qd[:, None, :]
        dd = (diff * diff).sum(-1)
        alld = torch.cat([best_d, dd], 1)
        alli = torch.cat([best_i, cand + s0], 1)
        o = torch.argsort(alld, dim=1)[:, :k]
        best_d, best_i = torch.gather(alld, 1, o), torch.gather(alli, 1, o)
        del rows, b
    gt = best_i.cpu().numpy()
    torch.cuda.empty_cache()
    t_data = time.time() - t0
    log(f"[build768] {n}x{dim} {tname}: data + upload + ground truth {t_data:.1f}s, medoid {medoid}")
    # ---- the build ---------------------------------------------------------------------------------------------------
    cfg = da.build_config(pruned, R, lb, intra_batch_candidates=da.IBC_NONE)
    max_batch = 16384
    t1 = time.time()
    nb = prov.build(cfg, 0, n, args.growth, max_batch)
    torch.cuda.synchronize()
    t_build = time.time() - t1
    c = [int(x) for x in prov.build_counters()]
    row_b, adj_b = dim * esz, (R + 1) * 4
    log(f"[build768] build {t_build:.1f}s ({n / t_build:,.0f} pts/s), {nb} batches")
    res = {
        "workload": f"dann_build of a {n}x{dim} {tname} index (Vamana R={R}, pruned {pruned}, l_build={lb}, alpha 1.2, growth "
                    f"{args.growth}, max_batch {max_batch}) on one GPU: {n * row_b / 1e9:.1f} GB rows + {(n + 1) * adj_b / 1e9:.1f} GB "
                    f"adjacency resident in HBM; data: the benchmark mixture, {nblobs} blobs ("
                    + ("256 topics, blob centres on a shared 8-d subspace around their topic" if hier else "i.i.d. centres")
                    + "), generated on the device",
        "build_seconds": t_build, "points_per_second": n / t_build, "batches": nb, "data_seconds": t_data,
        "insert_search": {"cmps": c[2], "hops": c[3], "algorithmic_bytes": c[2] * row_b + c[3] * adj_b},
        "prune": {"row_kernel_pair_distances": c[4], "list_distances": c[5], "gram_rows": c[6], "gram_entries": c[7],
                  "mfma_flop": 2 * c[7] * dim, "pairs_asked_by_gram_sweeps": c[8], "of_those_exact_rechecks": c[9],
                  "backedge_prunes_mfma": c[0], "backedge_prunes_too_long_for_gram": c[1],
                  "mfma_share_of_all_prune_pair_distances": (c[8] - c[9]) / max(1, c[8] - c[9] + c[4])},
    }
    # ---- search evaluation ---------------------------------------------------------------------------------------------
    qs = queries.half().contiguous() if f16 else queries
    sweep = [10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30, 32, 36, 40, 48, 56, 64, 80, 96, 128, 160, 192, 256]
    L, rec, st, run, hist = _sweep(torch, lib, _ffi, prov, qs, nq, k, 1, gt, ngt, sweep, args.target_recall)
    reached = L is not None
    L = L or sweep[-1]
    run(L)
    prov.kernel_time_reset()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(5):
        run(L)
    torch.cuda.synchronize()
    dt_s = (time.perf_counter() - t2) / 5
    ms, nl = prov.kernel_time(0)
    alg = int(st[:, 0].sum()) * row_b + int(st[:, 1].sum()) * adj_b
    res["search"] = {"recall_at_10": round(rec, 4), "recall_target_reached": reached, "recall_by_L": hist,
                     "ground_truth": f"exact (shortlist of 40 per 2^20-row chunk re-ranked in f64), first {ngt} queries",
                     "L": L, "queries_per_launch": nq, "qps": nq / dt_s, "mean_cmps": float(st[:, 0].mean()),
                     "mean_hops": float(st[:, 1].mean()), "avg_kernel_ms": ms / max(nl, 1),
                     "algorithmic_GBps": alg / (ms / max(nl, 1) * 1e-3) / 1e9,
                     "frac_of_hbm_peak": alg / (ms / max(nl, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS}
    # ---- parity at this scale: a query sample replayed through the CPU oracle on the bytes the searches touch.
    # The GPU's VisitedSearchRecord names every node a search expanded; those nodes' adjacency rows name every row it
    # compared.  The oracle runs the reference algorithm on exactly that sub-graph (ids renumbered, adjacency of
    # non-expanded nodes empty): if it returns the GPU's ids, distances, cmps and hops it never left the sub-graph,
    # i.e. its run equals a run on the whole index.
    ms_n = 256
    qh = qs[:ms_n].cpu().numpy()
    gi, gd, gst = prov.search(da.Knn(L, 1), qh, k)
    rid, _, rn, _ = prov.search_record_queries(qh, L)
    p_rows, p_adj = prov.device_pointers()
    adj_t = torch.as_tensor(_DevView(p_adj, (n + 1, R + 1), "<i4"), device=dev)
    stride_e = prov.row_stride // esz
    rows_t = torch.as_tensor(_DevView(p_rows, (n + 1, stride_e), "<f2" if f16 else "<f4"), device=dev)
    expanded = np.unique(np.concatenate([rid[i, :rn[i]] for i in range(ms_n)] + [np.array([n], np.uint32)]))
    ex_t = torch.as_tensor(expanded.astype(np.int64), device=dev)
    ex_adj = adj_t[ex_t].cpu().numpy().view(np.uint32)                       # [len, ids...] of every expanded node
    lens = np.minimum(ex_adj[:, 0], R)
    nbrs = np.unique(np.concatenate([ex_adj[i, 1:1 + lens[i]] for i in range(len(expanded))]))
    touched = np.unique(np.concatenate([expanded, nbrs]))                    # sorted global ids; the start point n is last
    assert touched[-1] == n
    remap = {int(g): i for i, g in enumerate(touched[:-1])}
    m = len(touched) - 1
    sub_rows = rows_t[torch.as_tensor(touched.astype(np.int64), device=dev)][:, :dim].contiguous().cpu().numpy()
    oix = oracle.Index(odt, oracle.L2, dim, m, R, sub_rows[-1:])
    oix.set_rows(0, sub_rows[:-1])
    remap[n] = m  # the start point keeps the last slot
    lut = np.full(n + 1, 0xFFFFFFFF, np.uint32)
    lut[touched[:-1]] = np.arange(m, dtype=np.uint32)
    lut[n] = m
    for i, gnode in enumerate(expanded):
        row = ex_adj[i, 1:1 + lens[i]]
        oix.adj[int(lut[gnode]), 0] = lens[i]
        oix.adj[int(lut[gnode]), 1:1 + lens[i]] = lut[row]
    oi, od, oc, ost = oix.search_batch(qh, L, 1, k, threads=min(16, os.cpu_count() or 1), fast=True)
    back = np.concatenate([touched[:-1], np.array([n, 0xFFFFFFFF], np.uint32)]).astype(np.uint32)
    oi_g = np.where(oi == 0xFFFFFFFF, 0xFFFFFFFF, back[np.minimum(oi, m + 1)])
    res["oracle_replay"] = {"queries": ms_n, "rows_touched": int(m), "nodes_expanded": int(len(expanded)),
                            "ids_identical_to_gpu": bool(np.array_equal(gi, oi_g)),
                            "distances_cmps_hops_identical": bool(
                                np.array_equal(gd.view(np.uint32), od.view(np.uint32)) and
                                np.array_equal(gst["cmps"], ost[:, 0]) and np.array_equal(gst["hops"], ost[:, 1]))}
    prov.close()
    return res


def only_variant(args, torch, da, lib, _ffi, dev, local):
    """One secondary workload per process (so that a rocprofv3 pass sees only its kernel at full weight)."""
    k, W = 10, args.beam_width
    if args.only == "gather":
        # ExpandBeam::expand_beam batched on a store far beyond the Infinity Cache: 10 M x 128 f32 rows (5.1 GB),
        # 20 000 queries x 256 random row ids = 5.12 M evaluations x 512 B per launch
        n, dim = 10_000_000, args.dim
        base, queries = make_data(torch, dev, n, dim, 20000, "sift_like:1:2560", 0xD15CA11, 0xD15CA12)
        prov = da.Provider(da.F32, da.L2, dim, n, 4, base[:1].cpu().numpy(), device=local)
        for s0 in range(0, n, 1 << 21):
            prov.set_elements(s0, base[s0:s0 + (1 << 21)].cpu().numpy())
        gq, gl = 20000, 256
        gids = np.random.default_rng(7).integers(0, n, gq * gl, dtype=np.uint32)
        goff = np.arange(gq + 1, dtype=np.uint64) * gl
        qh = queries.cpu().numpy()
        prov.expand_beam_batch(qh, gids, goff)
        prov.kernel_time_reset()
        for _ in range(5):
            prov.expand_beam_batch(qh, gids, goff)
        ms, nl = prov.kernel_time(1)
        alg = gq * gl * dim * 4
        return {"kernel": "expand_beam_kernel", "workload": f"{gq} queries x {gl} random rows of a {n}x{dim} f32 store "
                f"({n * dim * 4 / 1e9:.1f} GB)", "evals_per_launch": gq * gl, "algorithmic_bytes_per_launch": alg,
                "avg_kernel_ms": ms / nl, "achieved_GBps": alg / (ms / nl * 1e-3) / 1e9,
                "frac_of_hbm_peak": alg / (ms / nl * 1e-3) / 1e9 / HBM_PEAK_GBS}
    # quantised / integer rows: the headline data as SQ-8 codes (dim + 4 bytes) or u8 rows (dim bytes)
    base, queries = make_data(torch, dev, args.n, args.dim, args.nq, args.dist, 0xD15CA11, 0xD15CA12)
    mean = base.double().mean(0).float()
    medoid = int(torch.argmin(((base - mean[None, :]) ** 2).sum(1)).item())
    if args.only == "pq":  # PQ codes walk the f32 index's graph: build that first
        full = da.Provider(da.F32, da.L2, args.dim, args.n, args.max_degree, base[medoid:medoid + 1].cpu().numpy(), device=local)
        full.set_elements(0, base.cpu().numpy())
        full.build(da.build_config(args.pruned_degree, args.max_degree, args.l_build, intra_batch_candidates=da.IBC_NONE),
                   0, args.n, args.growth, args.max_batch)
        gt = ground_truth(torch, base, queries[:10000] if getattr(args, "pq768", False) else queries, k)
        if args.L:  # profiling pass: fixed L
            args.target_recall = -1.0
        return pq_variant(args, torch, da, lib, _ffi, dev, local, base, queries, gt, medoid, k, W, full)
    return int_rows_variant(args, torch, da, lib, _ffi, dev, local, base, queries, medoid, args.only, [args.L or 26], k, W)[0]


def int_rows_variant(args, torch, da, lib, _ffi, dev, local, base, queries, medoid, kind, Ls, k, W):
    """The search kernel alone on 128-byte integer rows of the headline data -- u8 rows (min-max scaled) or SQ-8 codes --,
    index built on the GPU over those rows; one result per L of `Ls`.  Every result names the kernel family that served
    the timed launches and checks the first rows of the timed launch's own output against the CPU oracle."""
    import oracle
    if kind == "sq8":
        g = torch.Generator(device=dev)
        g.manual_seed(11)
        sample = base[torch.randperm(args.n, generator=g, device=dev)[:min(args.n, 131072)]].cpu().numpy()
        shift, scale, _ = da.sq8_train(sample, 2.0, device=local)
        snorm = float(np.float32((shift ** 2).sum(dtype=np.float32)))
        rows = da.sq8_compress(base.cpu().numpy(), shift, scale, device=local)
        qrows = da.sq8_compress(queries.cpu().numpy(), shift, scale, device=local)
        prov = da.Provider(da.SQ8, da.L2, args.dim, args.n, args.max_degree, rows[medoid:medoid + 1], device=local,
                           sq_scale=scale, sq_shift_norm_sq=snorm, row_stride=args.sq8_stride)
        row_bytes = args.dim + 4
    else:
        lo, hi = float(base.min()), float(base.max())
        rows = ((base - lo) * (255.0 / (hi - lo))).round().clamp(0, 255).to(torch.uint8).cpu().numpy()
        qrows = ((queries - lo) * (255.0 / (hi - lo))).round().clamp(0, 255).to(torch.uint8).cpu().numpy()
        prov = da.Provider(da.U8, da.L2, args.dim, args.n, args.max_degree, rows[medoid:medoid + 1], device=local)
        row_bytes = args.dim
    if args.visited_bits:  # experiment knob: explicit LDS visited-table size (never affects results)
        prov.set_visited_bits(args.visited_bits)
        prov.set_visited_format(16)  # (explicit sizes reach the pair kernel with 16-bit entries only)
    prov.set_elements(0, rows)
    prov.build(da.build_config(args.pruned_degree, args.max_degree, args.l_build, intra_batch_candidates=da.IBC_NONE),
               0, args.n, args.growth, args.max_batch)
    dq = torch.from_numpy(qrows).to(dev)
    d_ids = torch.empty((args.nq, k), dtype=torch.int32, device=dev)
    d_d = torch.empty((args.nq, k), dtype=torch.float32, device=dev)
    d_st = torch.empty((args.nq, 5), dtype=torch.int32, device=dev)
    gt = ground_truth(torch, base, queries[:10000], k)
    okw = dict(sq_scale=scale, sq_shift_norm_sq=snorm) if kind == "sq8" else {}
    out = []
    for L in Ls:
        def run():
            _ffi.check(lib.dann_search_batch_device(prov._h, C.c_void_p(dq.data_ptr()), args.nq, L, W, k,
                                                    C.c_void_p(d_ids.data_ptr()), C.c_void_p(d_d.data_ptr()),
                                                    C.c_void_p(d_st.data_ptr())), "dann_search_batch_device")
        for _ in range(3):
            run()
        prov.kernel_time_reset()
        fam0 = prov.search_families()
        for _ in range(10):
            run()
        family = family_of(fam0, prov.search_families())
        ms, nl = prov.kernel_time(0)
        st = d_st.cpu().numpy().view(np.uint32)
        alg = int(st[:, 0].sum()) * row_bytes + int(st[:, 1].sum()) * (args.max_degree + 1) * 4
        # recall of the quantised search alone against the exact f32 ground truth (no rerank), for orientation
        rec = recall_at_k(d_ids[:10000].cpu().numpy().view(np.uint32), gt, k)
        try:
            osample = oracle_sample(prov, oracle.SQ8 if kind == "sq8" else oracle.U8, oracle.L2, args.dim, args.n,
                                    args.max_degree, rows[medoid:medoid + 1], rows, qrows, L, W, k, d_ids[:256].cpu().numpy(),
                                    d_d[:256].cpu().numpy(), st[:256], family, **okw)
        except Exception as e:  # noqa: BLE001
            osample = {"error": str(e)[:200]}
        out.append({"oracle_sample": osample, "kernel_family": family,
                    "kernel": "pair_search_kernel" if family == "pair" else "beam_search_kernel", "rows": kind,
                    "row_bytes": row_bytes, "L": L, "nq": args.nq, "recall_at_10_vs_exact_f32_no_rerank": round(rec, 4),
                    "mean_cmps": float(st[:, 0].mean()), "mean_hops": float(st[:, 1].mean()),
                    "algorithmic_bytes_per_launch": alg, "avg_kernel_ms": ms / nl,
                    "achieved_GBps": alg / (ms / nl * 1e-3) / 1e9,
                    "frac_of_hbm_peak": alg / (ms / nl * 1e-3) / 1e9 / HBM_PEAK_GBS, "qps": args.nq / (ms / nl * 1e-3)})
    prov.close()
    return out


def cpu_distance_microbench(full=True):
    """The CPU side of SURVEY.md 8(d) / BASELINE.md 3: the reference's distance micro-benchmark
    (diskann-benchmark-simd/src/lib.rs:716-771, examples/simd.json: ONE query x 50 rows x 5000 loops, dims 100 / 128 /
    384 / 768) with this repository's AVX2 restatement of the V3 kernels, plus the cache-defeating variant that is the
    fair partner of the GPU gather kernel: rows of a table far beyond the caches (1 GiB; 512 MiB in the short form)
    visited in a random order, one thread and all host cores.  `full=False`: the two dims of the benchmark
    configurations, f32 only (about 15 s)."""
    import oracle
    cores, quota_note = host_cores()
    dims = (100, 128, 384, 768) if full else (128, 768)
    dts = (("f32", oracle.F32, 4), ("f16", oracle.F16, 2), ("u8", oracle.U8, 1)) if full else (("f32", oracle.F32, 4),)
    rows = []
    for name, dt, esz in dts:
        for dim in dims:
            r = oracle.bench_distance(dt, oracle.L2, dim, 50, 5000, False, 1)
            e = {"rows": name, "dim": dim, "metric": "L2", "simd_shape_ns_per_distance_T1": 1e9 / r}
            nrows = max(1 << 16, ((1 << 30) if full else (1 << 29)) // (dim * esz))
            for T in (1, cores):
                r = oracle.bench_distance(dt, oracle.L2, dim, nrows, 1, True, T)
                e[f"random_rows_Mdist_per_s_T{T}"] = r / 1e6
                e[f"random_rows_GBps_T{T}"] = r * dim * esz / 1e9
            e["random_rows_table_bytes"] = nrows * dim * esz
            rows.append(e)
    return {"shape": "diskann-benchmark-simd: 1 query x 50 rows x 5000 loops (L1-resident), and the same kernels over a "
                     "1 GiB table in random row order", "kernels": "oracle AVX2 + FMA (f32, f16: bitwise the V3 kernels), "
                     "auto-vectorised integer loop (u8)", "cores": cores, "note": quota_note.strip(), "results": rows}


def host_cores():
    """host cores this process may actually use: affinity mask and the cgroup CPU quota (the GPU boxes expose 256
    logical CPUs but cap the container at 16 CPUs' worth of time; more threads than that only measure a burst)"""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    note = ""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            lim = max(1, int(np.ceil(int(q) / int(per))))
            if lim < cores:
                cores = lim
                note = f" (cgroup cpu.max {q}/{per})"
    except (OSError, ValueError):
        pass
    return cores, note


def pct(lat_us):
    lat_us = np.asarray(lat_us, dtype=np.float64)
    return {"mean_us": float(lat_us.mean()), "p50_us": float(np.percentile(lat_us, 50)),
            "p90_us": float(np.percentile(lat_us, 90)), "p99_us": float(np.percentile(lat_us, 99))}


def cpu_baseline(args, oix, queries_h, L, W, k, gpu_ids):
    """The CPU restatement of the reference path (oracle, AVX2 kernels) on the same graph
    bytes, same queries, all host cores, static block partition of the queries
    (diskann-benchmark-core/src/search/api.rs:399-436)."""
    cores, quota_note = host_cores()
    nqc = min(args.cpu_queries, queries_h.shape[0])
    qs = queries_h[:nqc]
    oix.search_batch(qs[:256], L, W, k, threads=cores, fast=True)  # warm
    best = None
    ids = None
    for _ in range(3):
        t0 = time.perf_counter()
        ids, _, _, _ = oix.search_batch(qs, L, W, k, threads=cores, fast=True)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    same = bool(np.array_equal(ids, gpu_ids[:nqc]))
    return {
        "value": nqc / best,
        "unit": "queries/s",
        "cores": cores,
        "kind": "port",
        "sample": f"first {nqc} of the {queries_h.shape[0]} queries, same graph/L/beam, best of 3, {cores} threads"
                  f"{quota_note}",
        "ids_identical_to_gpu": same,
    }


def cpu_small_regimes(oix, queries_h, L, W, k):
    """The CPU path beside BASELINE configs 2 and 3: one thread, one query at a time at L = 64 (per-query latency as the
    reference reports it: mean / p90 / p99, search/graph/knn.rs:300-330), and 1024 queries over all host cores."""
    cores, quota_note = host_cores()
    n1 = min(2000, queries_h.shape[0])
    oix.search_batch(queries_h[:64], 64, 1, k, threads=1, fast=True)
    _, _, _, _, ns = oix.search_batch(queries_h[:n1], 64, 1, k, threads=1, fast=True, timing=True)
    single = {"threads": 1, "queries": n1, **pct(ns.astype(np.float64) / 1e3)}
    best = None
    for _ in range(5):
        t0 = time.perf_counter()
        oix.search_batch(queries_h[:1024], L, W, k, threads=cores, fast=True)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return {"cpu_single_query_L64": single,
            "cpu_1024_queries_at_L": {"qps": 1024 / best, "threads": cores, "note": f"best of 5{quota_note}"}}


if __name__ == "__main__":
    main()
