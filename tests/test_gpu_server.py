"""Concurrent callers on one shared index (the reference's model: N workers calling DiskANNIndex::search on a shared
&DiskANNIndex, diskann-benchmark-core/src/search/api.rs:399-436): the launch path on the context pool, and the resident
search server (dann_server_start / dann_search_submit / dann_search_wait).  Every result must equal the oracle's."""
import threading
import time

import numpy as np
import pytest

import oracle
from helpers import bits, make_pair, rand_vectors, random_graph

pytestmark = pytest.mark.gpu
da = pytest.importorskip("diskann_amd")


def _index(dtype, metric, n, dim, R, seed, nstart=1):
    rng = np.random.default_rng(seed)
    data = rand_vectors(rng, dtype, n, dim)
    adj = random_graph(rng, n, R, nstart=nstart)
    oix, gix = make_pair(dtype, metric, data, adj, data[:nstart], R)
    return rng, oix, gix


def test_single_query_calls_from_many_threads_overlap():
    """16 native threads x single-query dann_search_batch calls: same results as the oracle, and the calls overlap on
    the device (each on its own stream) instead of queueing behind a per-index lock."""
    rng, oix, gix = _index(oracle.F32, oracle.L2, 20000, 128, 32, 41)
    q = rand_vectors(rng, oracle.F32, 2048, 128)
    L, k = 64, 10
    oi, od, _, _ = oix.search_batch(q, L, 1, k)
    gix.concurrent_callers(q[:64], L, k, threads=4, mode=0)  # warm: contexts, calibration
    ids1, d1, lat1, t1 = gix.concurrent_callers(q, L, k, threads=1, mode=0)
    ids16, d16, lat16, t16 = gix.concurrent_callers(q, L, k, threads=16, mode=0)
    for ids, d in ((ids1, d1), (ids16, d16)):
        assert np.array_equal(ids, oi) and np.array_equal(bits(d), bits(od))
    # 16 callers must finish the same work in well under the serial time (the reference's workers do not serialise)
    assert t16 < 0.5 * t1, (t1, t16)


@pytest.mark.parametrize("odt", [oracle.F32, oracle.U8, oracle.F16, oracle.I8])
def test_small_host_calls_of_many_threads_share_launches(odt):
    """dann_search_batch calls of at most 16 queries go through the combiner (api.hip: small_call): queries and results
    in page-locked mapped staging, one launch per call -- and per *group* of calls when several threads call side by side.
    Twelve Python threads with call sizes 1 .. 16 and three different (L, k) between them (only equal parameters may
    share a launch): every call returns what the oracle returns for its queries, statistics included; the counters show
    that launches were shared; with the combiner off (host_pipeline = 0) the same calls give the same rows."""
    import ctypes as C
    dim = 128 if odt != oracle.F32 else 100
    rng, oix, gix = _index(odt, oracle.L2, 6000, dim, 24, 91)
    nq = 1500
    q = rand_vectors(rng, odt, nq, dim)
    params = [(20, 5), (33, 10), (64, 10)]
    want = {p: oix.search_batch(q, p[0], 1, p[1]) for p in params}
    lib = da._ffi.lib()
    # a middle-sized call first: it takes the copy path and allocates the contexts' page-locked block; the small calls
    # below must find that block mapped (or map a new one) rather than decline to the copy path for good
    mi, md, _ = gix.search(da.Knn(20), q[:40], 5)
    assert np.array_equal(mi, want[(20, 5)][0][:40])
    before = gix.small_call_stats()
    fam_before = gix.search_families()
    errs = []

    def caller(t, rounds):
        try:
            L, k = params[t % 3]
            oi, od, oc, ost = want[(L, k)]
            r = np.random.default_rng(1000 + t)
            for _ in range(rounds):
                n = int(r.integers(1, 17))
                s0 = int(r.integers(0, nq - n))
                qs = np.ascontiguousarray(q[s0:s0 + n])
                hi = np.zeros((n, k), np.uint32)
                hd = np.zeros((n, k), np.float32)
                hs = np.zeros(n, da.STATS_DTYPE)
                with_stats = bool(r.integers(0, 2))
                da._ffi.check(lib.dann_search_batch(gix._h, qs.ctypes.data_as(C.c_void_p), n, L, 1, k,
                                                    hi.ctypes.data_as(C.c_void_p), hd.ctypes.data_as(C.c_void_p),
                                                    hs.ctypes.data_as(C.c_void_p) if with_stats else None), "batch")
                assert np.array_equal(hi, oi[s0:s0 + n]) and np.array_equal(bits(hd), bits(od[s0:s0 + n])), (t, s0, n)
                if with_stats:
                    assert np.array_equal(hs["cmps"], ost[s0:s0 + n, 0]) and np.array_equal(hs["hops"], ost[s0:s0 + n, 1])
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    ths = [threading.Thread(target=caller, args=(t, 60)) for t in range(12)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs[:3]
    launches, calls = (a - b for a, b in zip(gix.small_call_stats(), before))
    assert calls == 12 * 60 and launches <= calls
    # the search launches of this phase were the combined ones: no call was declined to the copy path (a call that is
    # declined launches for itself; the slack is for re-runs of queries whose team gave them back)
    fam_after = gix.search_families()
    extra = sum(fam_after[f][0] - fam_before[f][0] for f in fam_after) - launches
    assert 0 <= extra < calls // 10, (extra, launches, calls)
    if odt == oracle.F32:
        # native threads in lockstep (no interpreter between the calls): launches are shared for certain
        before = gix.small_call_stats()
        oi, od, _, _ = want[(64, 10)]
        ids, d, _, _ = gix.concurrent_callers(q, 64, 10, threads=8, mode=0)
        assert np.array_equal(ids, oi) and np.array_equal(bits(d), bits(od))
        launches, calls = (a - b for a, b in zip(gix.small_call_stats(), before))
        assert calls == nq and launches < calls, (launches, calls)
    # the same calls without the combiner
    gix.debug_set(host_pipeline=0)
    before = gix.small_call_stats()
    caller(0, 20)
    caller(1, 20)
    assert not errs, errs[:3]
    assert gix.small_call_stats() == before
    gix.debug_set(host_pipeline=None)


@pytest.mark.parametrize("dtype,metric,dim,R,L,k", [
    (oracle.F32, oracle.L2, 128, 32, 64, 10),
    (oracle.F32, oracle.INNER_PRODUCT, 100, 24, 40, 5),
    (oracle.F16, oracle.L2, 128, 32, 26, 10),
    (oracle.U8, oracle.L2, 128, 32, 100, 20),
    (oracle.I8, oracle.COSINE, 64, 16, 30, 10),
])
def test_server_results_equal_the_oracle(dtype, metric, dim, R, L, k):
    """submit / wait from several Python threads, tickets waited for out of order, ring wrap-around (the ring is far
    smaller than the number of queries): ids, distances and stats identical to the oracle's single-query searches."""
    rng, oix, gix = _index(dtype, metric, 8000, dim, R, 77 + dim + L)
    q = rand_vectors(rng, dtype, 3000, dim)
    oi, od, oc, ost = oix.search_batch(q, L, 1, k)
    gix.server_start(L, k, workers=96, ring=256)
    try:
        got_i = np.zeros_like(oi)
        got_d = np.zeros_like(od)
        got_c = np.zeros(len(q), np.uint32)
        err = []

        def work(t, nthreads, depth):
            try:
                mine = list(range(t, len(q), nthreads))
                for s0 in range(0, len(mine), depth):
                    chunk = mine[s0:s0 + depth]
                    tickets = [(i, gix.submit(q[i])) for i in chunk]
                    for i, tk in reversed(tickets):      # out of submission order
                        ids, dists, st = gix.wait(tk)
                        got_i[i], got_d[i], got_c[i] = ids, dists, st["cmps"]
                        assert st["status"] == 0
            except Exception as e:  # noqa: BLE001
                err.append(e)
        th = [threading.Thread(target=work, args=(t, 6, 8)) for t in range(6)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not err, err
        assert np.array_equal(got_i, oi)
        assert np.array_equal(bits(got_d), bits(od))
        assert np.array_equal(got_c, ost[:, 0])
        sub, _ = gix.server_stats()
        assert sub == len(q)
    finally:
        gix.server_stop()


def test_server_survives_idle_exit_and_interleaves_with_batches():
    """The resident kernel leaves after idle_timeout_us without a submission and the next submit relaunches it; searches
    through the launch path keep working while the server is up; stop / start again on the same index."""
    rng, oix, gix = _index(oracle.F32, oracle.L2, 6000, 128, 32, 5)
    q = rand_vectors(rng, oracle.F32, 400, 128)
    L, k = 32, 10
    oi, od, _, _ = oix.search_batch(q, L, 1, k)
    gix.server_start(L, k, workers=64, ring=128, idle_timeout_us=20000)
    try:
        for rnd in range(3):
            ids, d, lat, _ = gix.concurrent_callers(q, L, k, threads=4, mode=1, depth=8)
            assert np.array_equal(ids, oi) and np.array_equal(bits(d), bits(od)), rnd
            bi, bd, _ = gix.search(da.Knn(L), q[:50], k)      # the launch path, server resident or not
            assert np.array_equal(bi, oi[:50])
            time.sleep(0.08)                                   # > idle timeout: the kernel has left by now
        _, rel = gix.server_stats()
        assert rel >= 2, rel
    finally:
        gix.server_stop()
    gix.server_start(L, k, workers=32)
    try:
        t = gix.submit(q[7])
        while not gix.poll(t):
            time.sleep(0.0005)
        ids, d, st = gix.wait(t)
        assert np.array_equal(ids, oi[7]) and np.array_equal(bits(d), bits(od[7]))
        with pytest.raises(da.DannError):
            gix.wait(t)                                        # a ticket is waited for exactly once
    finally:
        gix.server_stop()


def test_server_overflowing_query_falls_back_to_the_launch_path():
    """A tiny explicit visited table makes resident searches exhaust their scratch; dann_search_wait re-runs those
    queries through the launch path (which retries with larger tables): results still equal the oracle's."""
    rng, oix, gix = _index(oracle.F32, oracle.L2, 30000, 128, 32, 13)
    q = rand_vectors(rng, oracle.F32, 300, 128)
    L, k = 200, 10
    oi, od, _, _ = oix.search_batch(q, L, 1, k)
    gix.set_visited_bits(8)   # 256 entries
    gix.server_start(L, k, workers=48)
    try:
        ids, d, _, _ = gix.concurrent_callers(q, L, k, threads=3, mode=1, depth=4)
        assert np.array_equal(ids, oi) and np.array_equal(bits(d), bits(od))
    finally:
        gix.server_stop()
        gix.set_visited_bits(0)


def test_host_pointer_search_pipeline_equals_one_pass():
    """dann_search_batch with a batch large enough for the chunked H2D / kernel / D2H pipeline (16 384 queries per chunk,
    a helper thread staging chunk i + 1 and draining chunk i - 1 beside kernel i) returns exactly what the same queries
    return in small calls -- from pageable buffers (through the pinned ring) and from buffers the caller page-locked
    (read and written by the DMA directly), statistics included, ragged last chunk."""
    import ctypes as C
    import torch
    rng, oix, gix = _index(oracle.F32, oracle.L2, 5000, 64, 16, 3)
    nq = 70001
    q = rand_vectors(rng, oracle.F32, nq, 64)
    L, k = 20, 10
    ids, d, st = gix.search(da.Knn(L), q, k)
    for s0 in (0, 16384 - 100, 32768 - 100, 65536 - 50, nq - 300):
        si, sd, sst = gix.search(da.Knn(L), q[s0:s0 + 300], k)
        assert np.array_equal(ids[s0:s0 + 300], si) and np.array_equal(bits(d[s0:s0 + 300]), bits(sd))
        assert np.array_equal(st["cmps"][s0:s0 + 300], sst["cmps"])
    oi, od, _, _ = oix.search_batch(q[:500], L, 1, k)
    assert np.array_equal(ids[:500], oi)
    # page-locked caller buffers: queries, ids, distances and statistics
    lib = da._ffi.lib()
    pq = torch.empty((nq, 64), dtype=torch.float32, pin_memory=True)
    pq.numpy()[...] = q
    pi = torch.zeros((nq, k), dtype=torch.int32, pin_memory=True)
    pd = torch.zeros((nq, k), dtype=torch.float32, pin_memory=True)
    pst = torch.zeros((nq, da.STATS_DTYPE.itemsize // 4), dtype=torch.int32, pin_memory=True)
    for stats in (pst, None):
        pi.zero_()
        da._ffi.check(lib.dann_search_batch(gix._h, C.c_void_p(pq.data_ptr()), nq, L, 1, k, C.c_void_p(pi.data_ptr()),
                                            C.c_void_p(pd.data_ptr()), C.c_void_p(stats.data_ptr()) if stats is not None else None),
                      "dann_search_batch")
        assert np.array_equal(pi.numpy().view(np.uint32), ids) and np.array_equal(bits(pd.numpy()), bits(d))
    assert np.array_equal(pst.numpy().view(da.STATS_DTYPE).reshape(-1)["cmps"], st["cmps"])
    # mixed: pinned queries, pageable outputs -- the first call goes through the lanes, from the second on the library
    # page-locks the buffers it has seen before for the length of the call (one zero-copy launch); same results every time
    hi = np.zeros((nq, k), np.uint32)
    hd = np.zeros((nq, k), np.float32)
    for rep in range(4):
        hi[...] = 0
        da._ffi.check(lib.dann_search_batch(gix._h, C.c_void_p(pq.data_ptr()), nq, L, 1, k, hi.ctypes.data_as(C.c_void_p),
                                            hd.ctypes.data_as(C.c_void_p), None), "dann_search_batch")
        assert np.array_equal(hi, ids) and np.array_equal(bits(hd), bits(d)), rep
    # pageable everything, repeated (gix.search allocates fresh outputs per call: the lanes; here the same buffers again)
    hs = np.zeros(nq, da.STATS_DTYPE)
    for rep in range(4):
        hi[...] = 0
        da._ffi.check(lib.dann_search_batch(gix._h, q.ctypes.data_as(C.c_void_p), nq, L, 1, k, hi.ctypes.data_as(C.c_void_p),
                                            hd.ctypes.data_as(C.c_void_p), hs.ctypes.data_as(C.c_void_p)), "dann_search_batch")
        assert np.array_equal(hi, ids) and np.array_equal(bits(hd), bits(d)) and np.array_equal(hs["cmps"], st["cmps"]), rep
    # after the call nothing stays page-locked on the library's account: the buffers can be freed and reallocated
    del hi, hd, hs


def test_host_pointer_pipeline_under_concurrent_callers_and_other_row_types():
    """Eight threads call dann_search_batch on 40 000-query host batches at once: every call wants three lanes, the index
    has sixteen search contexts -- callers that find none free run with fewer lanes (CtxLease try_only) instead of waiting
    for one another; results equal the single-threaded call.  Then the zero-copy launch on page-locked buffers for u8, f16
    and i8 rows (the kernel reads its query from mapped host memory when it stages it), against the pageable call."""
    import ctypes as C
    import threading
    import torch
    rng, oix, gix = _index(oracle.F32, oracle.L2, 5000, 64, 16, 3)
    nq, L, k = 40000, 16, 5
    q = rand_vectors(rng, oracle.F32, nq, 64)
    want_i, want_d, _ = gix.search(da.Knn(L), q, k)
    lib = da._ffi.lib()
    outs, errs = [None] * 8, []

    def caller(t):
        try:
            hi = np.zeros((nq, k), np.uint32)
            hd = np.zeros((nq, k), np.float32)
            for _ in range(3):
                da._ffi.check(lib.dann_search_batch(gix._h, q.ctypes.data_as(C.c_void_p), nq, L, 1, k,
                                                    hi.ctypes.data_as(C.c_void_p), hd.ctypes.data_as(C.c_void_p), None), "batch")
            outs[t] = (hi, hd)
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    ths = [threading.Thread(target=caller, args=(t,)) for t in range(8)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    for hi, hd in outs:
        assert np.array_equal(hi, want_i) and np.array_equal(bits(hd), bits(want_d))
    for odt in (oracle.U8, oracle.F16, oracle.I8):
        rng2, oix2, gix2 = _index(odt, oracle.L2, 4000, 128, 32, 1)
        qq = rand_vectors(rng2, odt, 33000, 128)
        wi, wd, wst = gix2.search(da.Knn(20), qq, 10)
        nb = qq.nbytes
        pq = torch.empty(nb, dtype=torch.uint8, pin_memory=True)
        pq.numpy()[...] = qq.reshape(-1).view(np.uint8)
        pi = torch.zeros((33000, 10), dtype=torch.int32, pin_memory=True)
        pd = torch.zeros((33000, 10), dtype=torch.float32, pin_memory=True)
        da._ffi.check(lib.dann_search_batch(gix2._h, C.c_void_p(pq.data_ptr()), 33000, 20, 1, 10, C.c_void_p(pi.data_ptr()),
                                            C.c_void_p(pd.data_ptr()), None), "batch")
        assert np.array_equal(pi.numpy().view(np.uint32), wi) and np.array_equal(bits(pd.numpy()), bits(wd)), odt
    oi, od, _, _ = oix2.search_batch(qq[:200], 20, 1, 10)
    assert np.array_equal(wi[:200], oi)


def test_mutations_are_refused_while_tickets_are_outstanding():
    """dann.h: mutations of the index are refused with DANN_EBUSY while tickets are outstanding; once every ticket has
    been collected they go through (the server stays up), and the results after the mutation follow the new rows."""
    rng, oix, gix = _index(oracle.F32, oracle.L2, 4000, 128, 32, 23)
    q = rand_vectors(rng, oracle.F32, 64, 128)
    L, k = 32, 10
    gix.server_start(L, k, workers=32)
    try:
        tickets = [gix.submit(q[i]) for i in range(8)]
        new_row = rand_vectors(rng, oracle.F32, 1, 128)
        for call in (lambda: gix.set_elements(5, new_row),
                     lambda: gix.set_neighbors(5, np.array([1, 2, 3], np.uint32)),
                     lambda: gix.upload_graph(oix.adj)):
            with pytest.raises(da.DannError) as e:
                call()
            assert e.value.status == da._ffi.EBUSY
        oi, od, _, _ = oix.search_batch(q, L, 1, k)
        for i, t in enumerate(tickets):
            ids, d, st = gix.wait(t)
            assert np.array_equal(ids, oi[i]) and np.array_equal(bits(d), bits(od[i]))
        gix.set_elements(5, new_row)          # nothing outstanding any more: accepted, server still resident
        oix.set_rows(5, new_row)
        oi, od, _, _ = oix.search_batch(q, L, 1, k)
        ids, d, _, _ = gix.concurrent_callers(q, L, k, threads=2, mode=1, depth=4)
        assert np.array_equal(ids, oi) and np.array_equal(bits(d), bits(od))
    finally:
        gix.server_stop()


def test_stop_while_callers_are_inside_wait_and_submit():
    """dann_server_stop racing with threads inside submit / wait / poll: the calls return an error (the server is gone)
    or a correct result, nothing crashes or hangs, and a new server starts on the same index afterwards."""
    rng, oix, gix = _index(oracle.F32, oracle.L2, 6000, 128, 32, 29)
    q = rand_vectors(rng, oracle.F32, 512, 128)
    L, k = 48, 10
    oi, od, _, _ = oix.search_batch(q, L, 1, k)
    for rnd in range(3):
        gix.server_start(L, k, workers=32, ring=64)
        stop_now = threading.Event()
        bad = []

        def caller(t):
            i = t
            while not stop_now.is_set() or i < 64:
                try:
                    tk = gix.submit(q[i % len(q)])
                    while not gix.poll(tk):
                        pass
                    ids, d, st = gix.wait(tk)
                    if not (np.array_equal(ids, oi[i % len(q)]) and np.array_equal(bits(d), bits(od[i % len(q)]))):
                        bad.append(i)
                except da.DannError as e:   # the server went away under the call
                    if e.status not in (da._ffi.EINVAL,):
                        bad.append((i, e.status))
                    return
                i += 8
        th = [threading.Thread(target=caller, args=(t,)) for t in range(8)]
        [t.start() for t in th]
        time.sleep(0.02 * (rnd + 1))
        stop_now.set()
        gix.server_stop()
        [t.join(timeout=60) for t in th]
        assert not any(t.is_alive() for t in th)
        assert not bad, bad[:5]
    gix.server_start(L, k, workers=16)
    try:
        ids, d, st = gix.wait(gix.submit(q[3]))
        assert np.array_equal(ids, oi[3])
    finally:
        gix.server_stop()


def test_poll_alone_brings_an_exited_kernel_back():
    """A caller that only polls (dann.h allows looping on poll before wait) must not spin for ever when the resident
    kernel left on its idle timeout right after the submit: poll relaunches it."""
    rng, oix, gix = _index(oracle.F32, oracle.L2, 3000, 128, 32, 31)
    q = rand_vectors(rng, oracle.F32, 40, 128)
    L, k = 24, 10
    oi, od, _, _ = oix.search_batch(q, L, 1, k)
    gix.server_start(L, k, workers=16, ring=64, idle_timeout_us=2000)
    try:
        for i in range(40):
            time.sleep(0.004)               # > idle timeout: the kernel has usually left when the submit arrives
            tk = gix.submit(q[i])
            t0 = time.time()
            while not gix.poll(tk):
                assert time.time() - t0 < 20, "poll never saw the result"
            ids, d, st = gix.wait(tk)
            assert np.array_equal(ids, oi[i]) and np.array_equal(bits(d), bits(od[i]))
        _, rel = gix.server_stats()
        assert rel >= 5, rel
    finally:
        gix.server_stop()


def test_resident_kernel_leaves_under_load_and_results_stay_exact():
    """The resident kernel leaves not only when it is idle but also after max_resident_us of residence under a steady
    stream of tickets (a hipFree elsewhere in the process is a device-wide synchronisation and would otherwise wait for
    the callers to pause).  With the bound lowered to 1 ms, thousands of tickets from eight native threads cross dozens
    of drain / relaunch cycles: every result is still the oracle's, no ticket is lost, and another index can be created
    and destroyed (hipMalloc / hipFree) by a second thread while the server is kept busy."""
    rng, oix, gix = _index(oracle.F32, oracle.L2, 6000, 128, 32, 37)
    gix.debug_set(server_max_resident_us=1000)   # (read by server_start)
    q = rand_vectors(rng, oracle.F32, 6000, 128)
    L, k = 32, 10
    oi, od, _, _ = oix.search_batch(q, L, 1, k)
    gix.server_start(L, k, workers=256, ring=1024)
    try:
        done = threading.Event()
        churn = []

        def other_index():  # device-wide synchronisations from another thread, while the callers keep the server busy
            r2 = np.random.default_rng(5)
            while not done.is_set():
                t0 = time.time()
                o = da.Provider(oracle.F32, oracle.L2, 64, 2000, 8, r2.random((1, 64), dtype=np.float32))
                o.set_elements(0, r2.random((2000, 64), dtype=np.float32))
                o.close()
                churn.append(time.time() - t0)
        th = threading.Thread(target=other_index)
        th.start()
        try:
            for _ in range(5):
                ids, d, lat, _ = gix.concurrent_callers(q, L, k, threads=8, mode=1, depth=16)
                assert np.array_equal(ids, oi) and np.array_equal(bits(d), bits(od))
        finally:
            done.set()
            th.join(timeout=60)
        assert not th.is_alive()
        sub, rel = gix.server_stats()
        assert sub >= 5 * len(q) and rel >= 3, (sub, rel)    # the kernel did leave and come back under load
        assert churn and max(churn) < 5.0, max(churn)         # and nobody waited for the callers to pause
    finally:
        gix.server_stop()
