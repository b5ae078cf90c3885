"""Query-stream sharding across the GPUs of one node (replicated index, no data-path collective).

The multi-GPU build and search live behind the C ABI (csrc/sharded.hip: dann_comm_*, dann_build_sharded,
dann_search_sharded, dann_multi_*); this module binds them (`Comm`, `MultiProvider`, `build_sharded`) and keeps the
pure-Python statement of the exchange protocol that the CPU tests drive over gloo with a host-memory provider.

`partition` is the reference's task partition (diskann/src/utils/async_tools.rs:289-365,
used by diskann-benchmark-core/src/search/api.rs:410 to split the query set over tasks):
ranges are contiguous, disjoint, cover 0..nitems, and differ in length by at most one.
`search_sharded` runs one rank's slice through any search callable and assembles the full
result on every rank with all_gather (control plane only: k ids + k distances per query).
"""
import numpy as np


def partition(nitems, ntasks, task):
    if ntasks <= 0:
        raise ValueError("ntasks must be positive")
    if task >= ntasks or task < 0:
        raise ValueError(f"task id {task} must be less than the number of tasks {ntasks}")
    k, m = divmod(nitems, ntasks)
    if task >= m:
        start = m * (k + 1) + (task - m) * k
        return start, start + k
    start = task * (k + 1)
    return start, start + k + 1


def search_sharded(search_fn, queries, k, rank=0, world=1, group=None):
    """search_fn(queries_slice) -> (ids[nq_local, k] uint32, dists[nq_local, k] float32).

    Returns (ids[nq, k], dists[nq, k]) identical on every rank and identical to a single-rank
    run (queries are independent)."""
    queries = np.asarray(queries)
    nq = queries.shape[0]
    lo, hi = partition(nq, world, rank)
    ids, dists = search_fn(queries[lo:hi])
    ids = np.ascontiguousarray(ids, dtype=np.uint32).reshape(hi - lo, k)
    dists = np.ascontiguousarray(dists, dtype=np.float32).reshape(hi - lo, k)
    if world == 1:
        return ids, dists
    import torch
    import torch.distributed as dist
    # pad every shard to the longest one (lengths differ by at most 1)
    longest = partition(nq, world, 0)[1] - partition(nq, world, 0)[0]
    # RCCL ("nccl") gathers device tensors, gloo host tensors; the bits of ids and distances travel as int32
    tdev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    pad = torch.zeros((2, longest, k), dtype=torch.int32)
    pad[0, : hi - lo] = torch.from_numpy(ids.view(np.int32))
    pad[1, : hi - lo] = torch.from_numpy(dists.view(np.int32))
    pad = pad.reshape(-1).to(tdev)
    allp = torch.empty(world * pad.numel(), dtype=torch.int32, device=tdev)
    dist.all_gather_into_tensor(allp, pad, group=group)
    allp = allp.cpu().numpy().reshape(world, 2, longest, k)
    out_i = np.empty((nq, k), np.uint32)
    out_d = np.empty((nq, k), np.float32)
    for r in range(world):
        a, b = partition(nq, world, r)
        out_i[a:b] = allp[r, 0, : b - a].view(np.uint32)
        out_d[a:b] = allp[r, 1, : b - a].view(np.float32)
    return out_i, out_d


def batch_schedule(first, n, growth, max_batch):
    """The geometric batch schedule of dann_build: batch = clamp(ceil(inserted * growth), 1, max_batch)."""
    import math
    g = float(np.float32(growth))
    done = 0
    while done < n:
        b = int(math.ceil((first + done) * g))
        b = max(1, min(b, max_batch, n - done))
        yield first + done, b
        done += b


class Comm:
    """dann_comm: the collective the sharded build / search run on.

    Comm.local(devices)      one process, one host thread per device (in-process all-gather: hipMemcpyPeer)
    Comm.from_torch(group)   one process per GPU under torch.distributed: "nccl" -> an RCCL communicator created by the
                             library (the unique id travels through torch.distributed, the collectives are ncclAllGather
                             calls issued from C++ on device buffers); "gloo" -> a callback communicator whose all-gather
                             goes through host memory (CPU collective; tests with several ranks on one device)"""

    def __init__(self, handle, keep=None):
        self._h = handle
        self._keep = keep  # callback objects must outlive the communicator

    @staticmethod
    def local(devices):
        import ctypes as C
        from . import _ffi
        devs = (C.c_int32 * len(devices))(*devices)
        out = (C.c_void_p * len(devices))()
        _ffi.check(_ffi.lib().dann_comm_create_local(devs, len(devices), out), "dann_comm_create_local")
        return [Comm(C.c_void_p(h)) for h in out]

    @staticmethod
    def from_torch(group=None, device=-1):
        import ctypes as C
        import torch
        import torch.distributed as dist
        from . import _ffi
        lib = _ffi.lib()
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        h = C.c_void_p()
        if dist.get_backend(group) == "nccl":
            uid = (C.c_char * 128)()
            if rank == 0:
                _ffi.check(lib.dann_comm_rccl_unique_id(uid), "dann_comm_rccl_unique_id")
            t = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).clone().to(torch.device("cuda", torch.cuda.current_device()))
            dist.broadcast(t, 0, group=group)
            uid = (C.c_char * 128).from_buffer_copy(bytes(t.cpu().numpy().tobytes()))
            _ffi.check(lib.dann_comm_create_rccl(uid, rank, world, device, C.byref(h)), "dann_comm_create_rccl")
            return Comm(h)

        def all_gather(_ctx, d_send, d_recv, nbytes, _stream):  # device -> host -> gloo -> device
            try:
                send = np.empty(nbytes, np.uint8)
                _ffi.check(lib.dann_memcpy_device(device, send.ctypes.data, d_send, nbytes, 1), "dann_memcpy_device")
                recv = torch.empty(world * nbytes, dtype=torch.uint8)
                dist.all_gather_into_tensor(recv, torch.from_numpy(send), group=group)
                r = recv.numpy()
                _ffi.check(lib.dann_memcpy_device(device, d_recv, r.ctypes.data, world * nbytes, 0), "dann_memcpy_device")
                return 0
            except Exception:  # noqa: BLE001 -- nothing may unwind into C
                return -5
        cb = _ffi.COMM_ALL_GATHER_FN(all_gather)
        ops = _ffi.CommOps(None, rank, world, cb)
        _ffi.check(lib.dann_comm_create_callbacks(C.byref(ops), C.byref(h)), "dann_comm_create_callbacks")
        return Comm(h, keep=(cb, ops))

    def close(self):
        if self._h:
            from . import _ffi
            _ffi.lib().dann_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def build_sharded_native(provider, cfg, first, n, growth, max_batch, comm, stats=None):
    """dann_build_sharded: the whole batch loop, both all-gathers and the owner-partitioned commit in C++ (device
    buffers end to end).  Collective over `comm`; returns the number of batches."""
    import ctypes as C
    from . import _ffi
    st = (C.c_uint64 * 4)()
    nb = _ffi.check(_ffi.lib().dann_build_sharded(provider._h, comm._h, C.byref(cfg), first, n, growth, max_batch, st),
                    "dann_build_sharded")
    if stats is not None:
        stats["rounds"] = stats.get("rounds", 0) + st[0]
        stats["bytes_gathered"] = stats.get("bytes_gathered", 0) + st[1]
        stats["rows_rewritten"] = stats.get("rows_rewritten", 0) + st[2]
        stats["bytes_gathered_rows"] = stats.get("bytes_gathered_rows", 0) + st[3]
    return nb


def search_sharded_native(provider, comm, queries, l_value, beam_width, k):
    """dann_search_sharded: every rank passes the same query block and receives every result."""
    from . import _ffi
    q = np.ascontiguousarray(queries, dtype=provider.query_dtype).reshape(-1, provider.query_elems)
    ids = np.empty((q.shape[0], k), np.uint32)
    dists = np.empty((q.shape[0], k), np.float32)
    _ffi.check(_ffi.lib().dann_search_sharded(provider._h, comm._h, q.ctypes.data, q.shape[0], l_value, beam_width, k,
                                              ids.ctypes.data, dists.ctypes.data), "dann_search_sharded")
    return ids, dists


class MultiProvider:
    """dann_multi: one process driving several devices -- one replica per entry of `devices` (an ordinal may repeat)."""

    def __init__(self, dtype, metric, dim, capacity, max_degree, start_points, devices):
        import ctypes as C
        from . import _ffi
        from .provider import NP_DTYPE
        self.dtype, self.dim = dtype, int(dim)
        sp = np.ascontiguousarray(start_points, dtype=NP_DTYPE[dtype]).reshape(-1, self.dim)
        cfg = _ffi.Config(dtype, metric, self.dim, int(capacity), int(max_degree), sp.shape[0], 0, -1, 0.0, 0.0, 0, 0)
        devs = (C.c_int32 * len(devices))(*devices)
        h = C.c_void_p()
        _ffi.check(_ffi.lib().dann_multi_create(C.byref(cfg), sp.ctypes.data, sp.nbytes, devs, len(devices), C.byref(h)),
                   "dann_multi_create")
        self._h, self.ndev, self.max_degree, self.capacity, self.nstart = h, len(devices), int(max_degree), int(capacity), sp.shape[0]
        self._np = NP_DTYPE[dtype]

    def set_elements(self, first_slot, rows):
        from . import _ffi
        r = np.ascontiguousarray(rows, dtype=self._np).reshape(-1, self.dim)
        _ffi.check(_ffi.lib().dann_multi_set_elements(self._h, first_slot, r.shape[0], r.ctypes.data, r.nbytes),
                   "dann_multi_set_elements")

    def build(self, cfg, first, n, growth=0.02, max_batch=16384):
        import ctypes as C
        from . import _ffi
        st = (C.c_uint64 * 4)()
        nb = _ffi.check(_ffi.lib().dann_multi_build(self._h, C.byref(cfg), first, n, growth, max_batch, st), "dann_multi_build")
        return nb, {"rounds": st[0], "bytes_gathered": st[1], "rows_rewritten": st[2], "bytes_gathered_rows": st[3]}

    def search(self, params, queries, k=10):
        from . import _ffi
        from .provider import STATS_DTYPE
        q = np.ascontiguousarray(queries, dtype=self._np).reshape(-1, self.dim)
        ids = np.empty((q.shape[0], k), np.uint32)
        dists = np.empty((q.shape[0], k), np.float32)
        stats = np.zeros(q.shape[0], STATS_DTYPE)
        _ffi.check(_ffi.lib().dann_multi_search_batch(self._h, q.ctypes.data, q.shape[0], params.l_value, params.beam_width, k,
                                                      ids.ctypes.data, dists.ctypes.data, stats.ctypes.data),
                   "dann_multi_search_batch")
        return ids, dists, stats

    def download_graph(self, replica):
        import ctypes as C
        from . import _ffi
        lib = _ffi.lib()
        h = lib.dann_multi_replica(self._h, replica)
        rows = self.capacity + self.nstart
        adj = np.empty((rows, self.max_degree + 1), np.uint32)
        _ffi.check(lib.dann_download_graph(C.c_void_p(h), adj.ctypes.data, rows), "dann_download_graph")
        return adj

    def close(self):
        if self._h:
            from . import _ffi
            _ffi.lib().dann_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def build_sharded(provider, cfg, first, n, growth=0.02, max_batch=16384, rank=0, world=1, group=None, stats=None,
                  owner_prunes=True, comm=None):
    """Multi-GPU index build over identical replicas (one `provider` per rank, rows already stored).

    Every batch is one multi_insert (diskann/src/graph/index.rs:815-1030): each rank generates the
    candidates (insert-time search + RobustPrune) for its `partition` of the batch positions, the
    pending adjacency rows are all-gathered (RCCL over xGMI when the process group is "nccl": the
    only exchange step of the build), and every rank applies the same graph update, so the replicas
    stay byte-identical to a single-GPU dann_build.  The batch schedule is dann_build's, including its
    fallback: a batch whose bootstrap (index.rs:926-938) would not fit one prune pool leaves the graph
    untouched and is re-inserted as smaller batches -- the test runs in the commit phase on identical
    data, so every rank takes the same decision.  With `owner_prunes` (default) the expensive part of that
    update is partitioned too: a back-edge target whose list must be pruned is handled only by the rank that
    owns it (id % world), the rewritten rows are all-gathered (second, small exchange) and applied by the
    others -- the replicas still end every batch byte-identical.  Returns the number of batches."""
    import math

    import torch

    from ._ffi import DannError, EUNSUPPORTED
    # the HIP provider: the loop below exists in C++ behind the C ABI (dann_build_sharded) -- collectives on device
    # buffers issued by the library.  The Python statement of the protocol stays for the host-memory stand-in of the
    # CPU tests and for the un-partitioned commit (owner_prunes=False).
    if getattr(provider, "_h", None) is not None and provider.device >= 0 and owner_prunes and (comm is not None or world > 1):
        own = comm is None
        if own:
            comm = Comm.from_torch(group, provider.device)
        try:
            return build_sharded_native(provider, cfg, first, n, growth, max_batch, comm, stats)
        finally:
            if own:
                comm.close()
    # (a provider on device -1 is a host-memory stand-in: the CPU tests drive this exchange protocol over gloo)
    dev = torch.device("cuda", provider.device) if provider.device >= 0 else torch.device("cpu")

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    width = cfg.pruned_degree + 1
    g = float(np.float32(growth))
    done, batches, limit = 0, 0, max_batch
    while done < n:
        b = int(math.ceil((first + done) * g))
        b = max(1, min(b, limit))
        b = min(b, n - done)
        start = first + done
        slots = np.arange(start, start + b, dtype=np.uint32)
        lo, hi = partition(b, world, rank)
        longest = partition(b, world, 0)[1]
        mine = torch.zeros((max(longest, 1), width), dtype=torch.int32, device=dev)
        # the fill above runs on torch's stream, the library writes `mine` from the index's own stream
        sync()
        provider.insert_batch_candidates(cfg, slots, lo, hi, mine.data_ptr())
        if world > 1:
            import torch.distributed as dist
            backend = dist.get_backend(group)
            if backend == "gloo":  # CPU collective (tests: several ranks on one device)
                mine_h = mine.cpu().reshape(-1)
                gathered_h = torch.empty(world * mine_h.numel(), dtype=torch.int32)
                dist.all_gather_into_tensor(gathered_h, mine_h, group=group)
                gathered = gathered_h.to(dev).reshape(world, max(longest, 1), width)
            else:
                gathered = torch.empty(world * mine.numel(), dtype=torch.int32, device=dev)
                dist.all_gather_into_tensor(gathered, mine.reshape(-1), group=group)
                gathered = gathered.reshape(world, max(longest, 1), width)
            if stats is not None:  # the build's only exchange: pending adjacency rows of the batch
                stats["bytes_gathered"] = stats.get("bytes_gathered", 0) + world * max(longest, 1) * width * 4
                stats["rounds"] = stats.get("rounds", 0) + 1
            parts = []
            for r in range(world):
                a, z = partition(b, world, r)
                parts.append(gathered[r, : z - a])
            pending = torch.cat(parts).contiguous()
        else:
            pending = mine[:b].contiguous()
        sync()
        part = owner_prunes and world > 1
        try:
            if part:
                rw = provider.max_degree + 2
                cap = b * cfg.pruned_degree + 1  # one row per distinct back-edge target at most
                mine_rows = torch.zeros((cap, rw), dtype=torch.int32, device=dev)
                sync()
                cnt = provider.insert_batch_commit_part(cfg, slots, pending.data_ptr(), rank, world,
                                                        mine_rows.data_ptr(), cap)
            else:
                provider.insert_batch_commit(cfg, slots, pending.data_ptr())
        except DannError as e:
            if e.status == EUNSUPPORTED and b > 1:
                limit = max(1, b // 2)
                continue
            raise
        if part:  # second exchange: the rows each owner rewrote (counts first, then rows padded to the longest list)
            import torch.distributed as dist
            cpu = dist.get_backend(group) == "gloo"
            cdev = torch.device("cpu") if cpu else dev
            counts = torch.empty(world, dtype=torch.int32, device=cdev)
            dist.all_gather_into_tensor(counts, torch.tensor([cnt], dtype=torch.int32, device=cdev), group=group)
            counts = counts.cpu().tolist()
            longest_rows = max(counts)
            if longest_rows:
                send = mine_rows[:longest_rows].reshape(-1)
                send = send.cpu() if cpu else send.contiguous()
                got = torch.empty(world * send.numel(), dtype=torch.int32, device=cdev)
                dist.all_gather_into_tensor(got, send, group=group)
                got = got.to(dev).reshape(world, longest_rows, rw)
                sync()
                for r in range(world):
                    if r != rank and counts[r]:
                        rows_r = got[r, : counts[r]].contiguous()
                        sync()
                        provider.apply_neighbor_rows(rows_r.data_ptr(), counts[r])
                if stats is not None:
                    stats["bytes_gathered_rows"] = stats.get("bytes_gathered_rows", 0) + world * longest_rows * rw * 4
                    stats["rows_rewritten"] = stats.get("rows_rewritten", 0) + sum(counts)
        limit = min(max_batch, limit * 2)
        done += b
        batches += 1
    return batches


def rerank_sharded(shard, bounds, queries, cand_ids, k, rank=0, world=1, group=None, stats=None):
    """Rerank (diskann-providers full_precision.rs:348-397) when the full-precision rows are PARTITIONED:
    config 5's layout keeps a reduced-precision replica of every row on every GPU for the graph walk and
    the f32 rows only on their owner (`bounds[r] <= id < bounds[r + 1]` lives on rank r as local slot
    id - bounds[r] of `shard`).  Owner computes: every rank evaluates, with the bit-exact row kernel
    (ExpandBeam), the candidates it owns for the queries of all ranks; the distances travel back and each
    rank orders its own candidates exactly as the single-GPU Rerank does (distance, then candidate
    position).  Only all_gather collectives (RCCL or gloo); `queries` are this rank's f32 queries,
    `cand_ids` their candidate lists (global ids, 0xFFFFFFFF = padding).
    Returns (ids[nq, k] uint32, dists[nq, k] float32); stats["bytes_gathered"] counts the exchange."""
    q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, shard.dim)
    cand = np.ascontiguousarray(cand_ids, dtype=np.uint32).reshape(q.shape[0], -1)
    nq, L = cand.shape
    if world > 1:
        import torch
        import torch.distributed as dist
        tdev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")

        def gather(arr, pad_rows):  # all ranks' arrays, each padded to pad_rows rows
            a = np.zeros((pad_rows,) + arr.shape[1:], arr.dtype)
            a[: arr.shape[0]] = arr
            t = torch.from_numpy(a.view(np.int32).reshape(-1)).to(tdev)
            out = torch.empty(world * t.numel(), dtype=torch.int32, device=tdev)
            dist.all_gather_into_tensor(out, t, group=group)
            if stats is not None:
                stats["bytes_gathered"] = stats.get("bytes_gathered", 0) + out.numel() * 4
            return out.cpu().numpy().view(arr.dtype).reshape((world, pad_rows) + arr.shape[1:])
        counts = gather(np.array([[nq]], np.int32), 1).reshape(world)
        rows = int(counts.max())
        all_q = gather(q, rows)          # world x rows x dim
        all_c = gather(cand, rows)       # world x rows x L
    else:
        counts, rows = np.array([nq]), nq
        all_q, all_c = q[None], cand[None]
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    # the candidates this rank owns, as one ragged list per (rank, query)
    dists_mine = np.full((world, rows, L), np.nan, np.float32)
    for r in range(world):
        c = all_c[r, : counts[r]]
        own = (c >= lo) & (c < hi)
        lens = own.sum(1)
        if lens.sum() == 0:
            continue
        offsets = np.zeros(c.shape[0] + 1, np.uint64)
        np.cumsum(lens, out=offsets[1:])
        d = shard.expand_beam_batch(all_q[r, : counts[r]], (c[own] - lo).astype(np.uint32), offsets)
        block = dists_mine[r, : counts[r]]
        block[own] = d
    if world > 1:
        all_d = gather(dists_mine.reshape(world * rows, L), world * rows).reshape(world, world, rows, L)
        mine = all_d[:, rank, :nq]       # owner x nq x L : NaN where that owner does not hold the candidate
        owner = np.searchsorted(np.asarray(bounds[1:], dtype=np.uint64), cand.astype(np.uint64), side="right")
        owner = np.minimum(owner, world - 1)
        d = np.take_along_axis(mine, owner[None], 0)[0]
    else:
        d = dists_mine[0, :nq]
    # Rerank's order: valid candidates in list order, sorted by (distance with -0 == +0, position), first k
    out_i = np.full((nq, k), 0xFFFFFFFF, np.uint32)
    out_d = np.full((nq, k), np.inf, np.float32)
    valid = (cand != 0xFFFFFFFF) & (cand < np.uint64(bounds[-1]))
    for i in range(nq):
        ids = cand[i][valid[i]]
        dd = d[i][valid[i]] + np.float32(0.0)
        u = dd.view(np.uint32)
        keys = np.where(u & 0x80000000, ~u, u | 0x80000000).astype(np.uint64) << np.uint64(32) | np.arange(ids.size, dtype=np.uint64)
        order = np.argsort(keys, kind="stable")[:k]
        out_i[i, : order.size] = ids[order]
        out_d[i, : order.size] = dd[order]
    return out_i, out_d
