"""GPU build path (RobustPrune, multi_insert) against the CPU oracle: identical adjacency."""
import math

import numpy as np
import pytest

import oracle
from helpers import teams_on, bits, make_pair, rand_vectors, random_graph

pytestmark = pytest.mark.gpu
da = pytest.importorskip("diskann_amd")


def _cfgs(pruned, maxdeg, l_build, **kw):
    return oracle.build_config(pruned, maxdeg, l_build, **kw), da.build_config(pruned, maxdeg, l_build, **kw)


@pytest.mark.parametrize("dtype,metric", [(oracle.F32, oracle.L2), (oracle.F16, oracle.L2),
                                          (oracle.F32, oracle.INNER_PRODUCT), (oracle.U8, oracle.L2),
                                          (oracle.F32, oracle.COSINE)])
def test_prune_batch_matches_oracle(dtype, metric):
    rng = np.random.default_rng(21)
    n, dim, R = 1500, 40, 12
    data = rand_vectors(rng, dtype, n, dim)
    adj = random_graph(rng, n, 16)
    oix, gix = make_pair(dtype, metric, data, adj, data[:1], 16)
    ocfg, gcfg = _cfgs(R, 16, 50)
    locs = rng.choice(n, 24, replace=False).astype(np.uint32)
    pools, dists, off = [], [], [0]
    for i, loc in enumerate(locs):
        m = [0, 1, 5, 70, 200, 333][i % 6]
        ids = rng.choice(n, m, replace=False).astype(np.uint32)
        if m > 3:
            ids[2] = loc  # the point itself is masked out (index.rs:2607-2613)
        d = np.array([oracle.distance(dtype, metric, data[loc], data[j]) for j in ids], np.float32)
        pools.append(ids)
        dists.append(d)
        off.append(off[-1] + m)
    pid = np.concatenate(pools) if pools else np.zeros(0, np.uint32)
    pdd = np.concatenate(dists) if dists else np.zeros(0, np.float32)
    for sat in (False, True):
        got = gix.prune_batch(gcfg, locs, pid, pdd, np.array(off, np.uint64), force_saturate=sat)
        for i, loc in enumerate(locs):
            want, _ = oix.prune_pool(ocfg, int(loc), pools[i], dists[i], force_saturate=sat)
            assert got[i, 0] == want.size, (i, sat)
            assert np.array_equal(got[i, 1:1 + want.size], want), (i, sat)


@pytest.mark.parametrize("dtype,metric", [(oracle.F32, oracle.L2), (oracle.F16, oracle.L2), (oracle.F32, oracle.INNER_PRODUCT),
                                          (oracle.I8, oracle.L2), (oracle.F32, oracle.COSINE)])
def test_prune_with_duplicate_rows_ties_and_long_pools(dtype, metric):
    """pools in which many rows are copies of each other: pair distances of exactly zero (update_occlude_factor's f32::MAX
    branch, config/mod.rs:86-88), equal pool distances (the sort's tie rule) and runs of candidates that are rejected or
    selected together -- what the in-flight sweep of prune_sorted_pool has to order exactly like the sequential visit.
    Pools of up to 900 entries (several worklist passes, max_occlusion cuts the list), both saturation settings."""
    rng = np.random.default_rng(77)
    n, dim, R, maxdeg = 2400, 24, 20, 24
    data = rand_vectors(rng, dtype, n, dim)
    data[1::3] = data[0:-1:3][: data[1::3].shape[0]]        # every third row repeats its predecessor
    data[300:340] = data[300]                               # a block of forty identical rows
    adj = random_graph(rng, n, maxdeg)
    oix, gix = make_pair(dtype, metric, data, adj, data[:1], maxdeg)
    for alpha in (1.0, 1.2):
        ocfg, gcfg = _cfgs(R, maxdeg, 50, alpha=alpha)
        locs = rng.choice(n, 40, replace=False).astype(np.uint32)
        locs[0] = 301
        pools, dists, off = [], [], [0]
        for i, loc in enumerate(locs):
            m = [3, 40, 64, 65, 129, 257, 600, 900][i % 8]
            ids = rng.choice(n, m, replace=False).astype(np.uint32)
            if i % 4 == 0 and m >= 40:
                ids[:30] = np.arange(300, 330, dtype=np.uint32)   # thirty copies of one row at the front of the list
                ids = np.unique(ids)[: m]
                rng.shuffle(ids)
                m = ids.size
            d = np.array([oracle.distance(dtype, metric, data[loc], data[j]) for j in ids], np.float32)
            pools.append(ids)
            dists.append(d)
            off.append(off[-1] + m)
        pid, pdd = np.concatenate(pools), np.concatenate(dists)
        for sat in (False, True):
            got = gix.prune_batch(gcfg, locs, pid, pdd, np.array(off, np.uint64), force_saturate=sat)
            for i, loc in enumerate(locs):
                want, _ = oix.prune_pool(ocfg, int(loc), pools[i], dists[i], force_saturate=sat)
                assert got[i, 0] == want.size, (i, sat, alpha)
                assert np.array_equal(got[i, 1:1 + want.size], want), (i, sat, alpha)


BUILD_CASES = [
    (oracle.F32, oracle.L2, 32, oracle.IBC_NONE),
    (oracle.F32, oracle.L2, 32, 4),
    (oracle.F32, oracle.L2, 32, oracle.IBC_ALL),
    (oracle.F16, oracle.L2, 24, oracle.IBC_NONE),
    (oracle.F32, oracle.INNER_PRODUCT, 16, oracle.IBC_NONE),
    (oracle.U8, oracle.L2, 16, 4),
]


@pytest.mark.parametrize("dtype,metric,dim,ibc", BUILD_CASES)
def test_insert_batch_matches_oracle_multi_insert(dtype, metric, dim, ibc):
    """Same batches through dann_insert_batch and the oracle's multi_insert: the whole
    adjacency buffer must be identical after every batch (bootstrap included)."""
    rng = np.random.default_rng(100 + dim)
    n, R, maxdeg, lb = 700, 8, 10, 24
    data = rand_vectors(rng, dtype, n, dim)
    start = data.mean(axis=0, keepdims=True).astype(oracle.NP_DTYPE[dtype])
    adj = np.zeros((n + 1, maxdeg + 1), np.uint32)
    oix, gix = make_pair(dtype, metric, data, adj, start, maxdeg)
    ocfg, gcfg = _cfgs(R, maxdeg, lb, intra_batch_candidates=ibc)
    sizes = [1, 1, 2, 3, 5, 8, 16, 30, 64, 120, 450]
    s = 0
    for b in sizes:
        e = min(s + b, n)
        slots = np.arange(s, e, dtype=np.uint32)
        oix.multi_insert(ocfg, slots)
        gix.insert_batch(gcfg, slots)
        got = gix.download_graph()
        lens = oix.adj[:, 0]
        assert np.array_equal(got[:, 0], lens), (b, np.nonzero(got[:, 0] != lens)[0][:5])
        mask = np.arange(maxdeg)[None, :] < lens[:, None]
        assert np.array_equal(got[:, 1:][mask], oix.adj[:, 1:][mask]), b
        s = e
    assert s == n


@pytest.mark.parametrize("dtype,metric", [(oracle.F32, oracle.L2), (oracle.U8, oracle.L2)])
def test_small_batches_of_128d_rows_are_searched_by_teams(dtype, metric):
    """the batches of a build's geometric phase are smaller than the chip: their insert-time searches (VisitedSearchRecord:
    every popped node recorded, search/record.rs:86-93) go to the team-of-four kernel where one exists (128-element rows).
    Same records, hence the same adjacency as the oracle's multi_insert after every batch."""
    rng = np.random.default_rng(128)
    n, dim, R, maxdeg, lb = 2200, 128, 12, 16, 60
    data = rand_vectors(rng, dtype, n, dim)
    start = data[:1].copy()
    adj = np.zeros((n + 1, maxdeg + 1), np.uint32)
    oix, gix = make_pair(dtype, metric, data, adj, start, maxdeg)
    ocfg, gcfg = _cfgs(R, maxdeg, lb, intra_batch_candidates=oracle.IBC_NONE)
    s = 0
    for b in (1, 2, 5, 40, 150, 600, 1024, 378):
        slots = np.arange(s, s + b, dtype=np.uint32)
        oix.multi_insert(ocfg, slots)
        _, fam = gix.last_family(lambda: gix.insert_batch(gcfg, slots))
        if teams_on():  # (the whole-suite 16-bit-table mode switches the teams off: same records from one wave per query)
            assert "team" in fam and fam <= {"team", "one_wave"}, (b, fam)   # (one_wave: re-runs of a search that outgrew its table)
        got = gix.download_graph()
        lens = oix.adj[:, 0]
        assert np.array_equal(got[:, 0], lens), (b, np.nonzero(got[:, 0] != lens)[0][:5])
        mask = np.arange(maxdeg)[None, :] < lens[:, None]
        assert np.array_equal(got[:, 1:][mask], oix.adj[:, 1:][mask]), b
        s += b
    assert s == n


def test_dann_build_schedule_and_recall():
    rng = np.random.default_rng(8)
    n, dim, R, maxdeg, lb = 3000, 32, 16, 20, 40
    # clustered data so that recall is meaningful
    centers = rng.standard_normal((20, dim)).astype(np.float32)
    data = (centers[rng.integers(0, 20, n)] + 0.3 * rng.standard_normal((n, dim))).astype(np.float32)
    mid, _ = oracle.medoid_f32(data)
    adj = np.zeros((n + 1, maxdeg + 1), np.uint32)
    oix, gix = make_pair(oracle.F32, oracle.L2, data, adj, data[mid:mid + 1], maxdeg)
    ocfg, gcfg = _cfgs(R, maxdeg, lb, intra_batch_candidates=oracle.IBC_NONE)
    growth, max_batch = 0.05, 256
    nb = gix.build(gcfg, 0, n, growth, max_batch)
    done, batches = 0, 0
    while done < n:
        b = max(1, min(int(math.ceil(done * np.float64(np.float32(growth)))), max_batch, n - done))
        oix.multi_insert(ocfg, np.arange(done, done + b, dtype=np.uint32))
        done += b
        batches += 1
    assert nb == batches
    got = gix.download_graph()
    lens = oix.adj[:, 0]
    assert np.array_equal(got[:, 0], lens)
    mask = np.arange(maxdeg)[None, :] < lens[:, None]
    assert np.array_equal(got[:, 1:][mask], oix.adj[:, 1:][mask])
    # recall of the GPU-built graph
    q = (centers[rng.integers(0, 20, 200)] + 0.3 * rng.standard_normal((200, dim))).astype(np.float32)
    d2 = ((q[:, None, :] - data[None, :, :]) ** 2).sum(-1)
    gt = np.argsort(d2, axis=1)[:, :10]
    ids, _, _ = gix.search(da.Knn(64), q, 10)
    recall = np.mean([len(set(ids[i]) & set(gt[i])) / 10 for i in range(200)])
    assert recall > 0.9, recall


def test_two_phase_build_equals_single_gpu_build():
    """Multi-GPU build logic on one GPU: two replicas play rank 0 / rank 1; each generates the candidates
    of its partition of every batch, the pending rows are concatenated (the all-gather) and both commit.
    Both replicas must equal a plain dann_build (and therefore the oracle)."""
    import torch
    from diskann_amd.sharding import batch_schedule, build_sharded, partition
    rng = np.random.default_rng(77)
    n, dim, R, maxdeg, lb = 1500, 24, 8, 10, 24
    data = rand_vectors(rng, oracle.F32, n, dim)
    start = data.mean(0, keepdims=True).astype(np.float32)
    gcfg = da.build_config(R, maxdeg, lb, intra_batch_candidates=4)
    provs = [da.Provider(da.F32, da.L2, dim, n, maxdeg, start) for _ in range(3)]
    for p in provs:
        p.set_elements(0, data)
    ref, reps = provs[0], provs[1:]
    growth, max_batch = 0.1, 200
    nb = ref.build(gcfg, 0, n, growth, max_batch)
    want = ref.download_graph()
    w = R + 1
    batches = 0
    for s0, b in batch_schedule(0, n, growth, max_batch):
        slots = np.arange(s0, s0 + b, dtype=np.uint32)
        parts = []
        for r, p in enumerate(reps):
            lo, hi = partition(b, 2, r)
            buf = torch.zeros((max(hi - lo, 1), w), dtype=torch.int32, device="cuda")
            p.insert_batch_candidates(gcfg, slots, lo, hi, buf.data_ptr())
            parts.append(buf[: hi - lo])
        pending = torch.cat(parts).contiguous()
        torch.cuda.synchronize()
        for p in reps:
            p.insert_batch_commit(gcfg, slots, pending.data_ptr())
        batches += 1
    assert batches == nb
    for p in reps:
        assert np.array_equal(p.download_graph(), want)
    # and the world-size-1 driver is the same thing
    solo = da.Provider(da.F32, da.L2, dim, n, maxdeg, start)
    solo.set_elements(0, data)
    assert build_sharded(solo, gcfg, 0, n, growth, max_batch) == nb
    assert np.array_equal(solo.download_graph(), want)


def test_build_parity_at_scale():
    """20 000 points through dann_build (batches up to 2048) vs the oracle's multi_insert with the same
    schedule: the whole adjacency buffer must be identical."""
    from diskann_amd.sharding import batch_schedule
    rng = np.random.default_rng(2024)
    n, dim, R, maxdeg, lb = 20000, 32, 16, 20, 48
    centers = rng.standard_normal((32, dim)).astype(np.float32)
    data = (centers[rng.integers(0, 32, n)] + 0.4 * rng.standard_normal((n, dim))).astype(np.float32)
    mid, _ = oracle.medoid_f32(data)
    adj = np.zeros((n + 1, maxdeg + 1), np.uint32)
    oix, gix = make_pair(oracle.F32, oracle.L2, data, adj, data[mid:mid + 1], maxdeg)
    ocfg, gcfg = _cfgs(R, maxdeg, lb, intra_batch_candidates=oracle.IBC_NONE)
    growth, max_batch = 0.02, 2048
    nb = gix.build(gcfg, 0, n, growth, max_batch)
    k = 0
    for s0, b in batch_schedule(0, n, growth, max_batch):
        oix.multi_insert(ocfg, np.arange(s0, s0 + b, dtype=np.uint32))
        k += 1
    assert k == nb
    got = gix.download_graph()
    lens = oix.adj[:, 0]
    assert np.array_equal(got[:, 0], lens)
    mask = np.arange(maxdeg)[None, :] < lens[:, None]
    assert np.array_equal(got[:, 1:][mask], oix.adj[:, 1:][mask])


def test_provider_smoke_on_gpu():
    """The reference's in-source provider test (diskann-inmem/src/provider.rs:1080-1256): 5x5 grid,
    degree 6, pruned degree 4, l_build 10, points inserted one by one with external id 10*i+1; the
    search from [0,0] returns (1, 0.0), the two distance-1 points, then (61, 2.0)."""
    from gridutil import grid_data, grid_start_point
    data = grid_data(2, 5)
    p = da.Provider(da.F32, da.L2, 2, 25, 6, grid_start_point(2, 5))
    p.set_elements(0, data)
    p.set_external_ids(0, [10 * i + 1 for i in range(25)])
    cfg = da.build_config(4, 6, 10)
    oix = oracle.Index(oracle.F32, oracle.L2, 2, 25, 6, grid_start_point(2, 5))
    ocfg = oracle.build_config(4, 6, 10)
    for i in range(25):                    # DiskANNIndex::insert == multi_insert of one point
        p.insert_batch(cfg, [i])
        oix.set_row(i, data[i])
        oix.insert(ocfg, i)
        assert np.array_equal(p.download_graph(), oix.adj), i
    ids, d, st = p.search(da.Knn(10), np.zeros((1, 2), np.float32), 10)
    ext = p.to_external(ids[0])
    assert (int(ext[0]), float(d[0, 0])) == (1, 0.0)
    assert sorted(int(e) for e in ext[1:3]) == [11, 51] and d[0, 1] == 1.0 and d[0, 2] == 1.0
    assert (int(ext[3]), float(d[0, 3])) == (61, 2.0)
    assert p.to_external(np.array([25, 0xFFFFFFFF], np.uint32)).tolist() == [2**64 - 1, 2**64 - 1]
    with pytest.raises(da.DannError):      # wrong-length vector leaves the provider untouched (:1226-1233)
        p.set_element(3, np.zeros(3, np.float32))


def test_single_insert_equals_batch_of_one():
    rng = np.random.default_rng(91)
    n, dim, R, maxdeg, lb = 400, 12, 6, 8, 20
    data = rand_vectors(rng, oracle.F32, n, dim)
    adj = np.zeros((n + 1, maxdeg + 1), np.uint32)
    oix, gix = make_pair(oracle.F32, oracle.L2, data, adj, data[:1], maxdeg)
    ocfg, gcfg = _cfgs(R, maxdeg, lb)
    for i in range(n):
        oix.insert(ocfg, i)
        gix.insert_batch(gcfg, [i])
    assert np.array_equal(gix.download_graph()[:, 0], oix.adj[:, 0])
    lens = oix.adj[:, 0]
    mask = np.arange(maxdeg)[None, :] < lens[:, None]
    assert np.array_equal(gix.download_graph()[:, 1:][mask], oix.adj[:, 1:][mask])


def test_build_splits_batches_that_need_an_oversized_bootstrap():
    """An aggressive schedule (growth 1.0) makes multi_insert's bootstrap test fire for batches larger than one prune
    pool; dann_build re-inserts those points in smaller batches (the graph is untouched when the test fires) instead of
    failing, and the result is a normal graph."""
    rng = np.random.default_rng(29)
    n, dim, R = 14000, 16, 24
    centers = rng.random((32, dim)).astype(np.float32)
    data = (centers[rng.integers(0, 32, n)] + 0.05 * rng.standard_normal((n, dim))).astype(np.float32)
    p = da.Provider(da.F32, da.L2, dim, n, R, data[:1].copy())
    p.set_elements(0, data)
    cfg = da.build_config(20, R, 48, intra_batch_candidates=da.IBC_NONE)
    nb = p.build(cfg, 0, n, 1.0, 1 << 20)
    assert nb > 14                       # more batches than the pure doubling schedule
    adj = p.download_graph()
    assert adj[:n, 0].min() >= 1 and adj[:n, 0].mean() > 10
    q = data[rng.choice(n, 200, replace=False)] + 0.01
    ids, d, st = p.search(da.Knn(40), q, 1)
    truth = ((q[:, None, :] - data[None]) ** 2).sum(-1).argmin(1)
    assert (ids[:, 0] == truth).mean() > 0.9
    with pytest.raises(da.DannError):    # a single oversized batch still reports the limit
        p2 = da.Provider(da.F32, da.L2, dim, n, R, data[:1].copy())
        p2.set_elements(0, data)
        p2.insert_batch(cfg, np.arange(8000, dtype=np.uint32))


# ---- MFMA path of the prunes (BUILD_MFMA_BACKEDGE / BUILD_MFMA_POOL; default for rows of 1 KiB and more) ------------------
@pytest.mark.parametrize("metric,dim,R,maxdeg", [
    (oracle.L2, 128, 24, 32),                  # pg = 64
    (oracle.L2, 100, 56, 64),                  # K tail (100 = 3 x 32 + 4), pg = 96
    (oracle.INNER_PRODUCT, 96, 28, 32),        # PruneKind::Occluding
    (oracle.COSINE_NORMALIZED, 64, 20, 24),    # 1 - <x, y> on unit vectors
])
def test_mfma_backedge_build_identical_to_oracle(metric, dim, R, maxdeg):
    """dann_build with the MFMA back-edge and pool prunes == the oracle's multi_insert, adjacency byte for byte (tie-free
    data); the same again with the error interval widened 10^6 x, which drives every comparison through the exact
    re-check."""
    from diskann_amd.sharding import batch_schedule
    rng = np.random.default_rng(1000 + dim)
    n, lb = 6000, 48
    centers = rng.random((24, dim)).astype(np.float32)
    data = (centers[rng.integers(0, 24, n)] + 0.15 * rng.standard_normal((n, dim))).astype(np.float32)
    if metric == oracle.COSINE_NORMALIZED:
        data /= np.linalg.norm(data, axis=1, keepdims=True)
    start = data.mean(0, keepdims=True).astype(np.float32)
    ocfg, gcfg = _cfgs(R, maxdeg, lb, intra_batch_candidates=oracle.IBC_NONE)
    growth, max_batch = 0.1, 1024
    oix = oracle.Index(oracle.F32, metric, dim, n, maxdeg, start)
    oix.set_rows(0, data)
    for s0, b in batch_schedule(0, n, growth, max_batch):
        oix.multi_insert(ocfg, np.arange(s0, s0 + b, dtype=np.uint32))
    for escale in (None, 1e6):
        gix = da.Provider(oracle.F32, metric, dim, n, maxdeg, start)
        gix.debug_set(gram_escale=escale)   # 1e6: every decision falls into the widened interval -> the exact path
        gix.set_elements(0, data)
        gix.set_build_options(da.BUILD_MFMA_BACKEDGE | da.BUILD_MFMA_POOL)
        gix.build(gcfg, 0, n, growth, max_batch)
        assert np.array_equal(gix.download_graph(), oix.adj), (metric, dim, escale)
        cnt = gix.build_counters()
        mfma, lazy = int(cnt[0]), int(cnt[1])
        assert mfma > 1000 and mfma > 20 * lazy, (mfma, lazy)   # the matrix-core path did the work
        assert cnt[6] >= mfma * (maxdeg + 1) and cnt[7] > 0 and cnt[2] > 0 and cnt[3] > 0
    # and the lazy path (no option) gives the same graph
    plain = da.Provider(oracle.F32, metric, dim, n, maxdeg, start)
    plain.set_elements(0, data)
    plain.build(gcfg, 0, n, growth, max_batch)
    assert np.array_equal(plain.download_graph(), oix.adj)
    pc = plain.build_counters()
    assert pc[0] == 0 and pc[1] == 0 and pc[6] == 0 and pc[4] > 0 and pc[5] > 0   # lazy path: row-kernel pairs only


def test_mfma_pool_prune_with_intra_batch_candidates_and_big_pools():
    """Pool prune on the matrix cores alone (no MFMA back-edges), with intra-batch candidates (the extras get their exact
    distances before the sort) and l_build large enough that pools exceed the 128 x 96 Gram block: pairs outside the
    block take the exact row kernel, the adjacency still equals the oracle's."""
    from diskann_amd.sharding import batch_schedule
    rng = np.random.default_rng(4321)
    n, dim, R, maxdeg, lb = 5000, 48, 20, 24, 150
    centers = rng.random((12, dim)).astype(np.float32)
    data = (centers[rng.integers(0, 12, n)] + 0.2 * rng.standard_normal((n, dim))).astype(np.float32)
    start = data.mean(0, keepdims=True).astype(np.float32)
    ocfg, gcfg = _cfgs(R, maxdeg, lb, intra_batch_candidates=6)
    growth, max_batch = 0.2, 700
    oix = oracle.Index(oracle.F32, oracle.L2, dim, n, maxdeg, start)
    oix.set_rows(0, data)
    for s0, b in batch_schedule(0, n, growth, max_batch):
        oix.multi_insert(ocfg, np.arange(s0, s0 + b, dtype=np.uint32))
    gix = da.Provider(oracle.F32, oracle.L2, dim, n, maxdeg, start)
    gix.set_elements(0, data)
    gix.set_build_options(da.BUILD_MFMA_POOL)
    gix.build(gcfg, 0, n, growth, max_batch)
    assert np.array_equal(gix.download_graph(), oix.adj)
    cnt = gix.build_counters()
    assert cnt[0] == 0 and cnt[6] > n * 20 and cnt[7] > 0     # Gram rows came from the pool prunes only


@pytest.mark.parametrize("dtype", [oracle.F32, oracle.F16])
def test_gram_tiles_equal_one_fmaf_chain_per_entry(dtype):
    """gram_tiles_kernel (the Gram of the three-kernel MFMA pool prune): v_mfma_f32_32x32x2_f32 with the accumulator
    running through the whole row is one k-ordered f32 FMA chain per entry -- bit-identical to orc_gram_chain on the
    lower triangle of the block (the sweep only asks for j < i), for sizes that exercise 1..8 row blocks, partial
    blocks, the partial K slab and rows narrower than a slab; f16 rows widened exactly (DANN_F16 | 0x100: the round-3
    form) likewise.  Norms: f64 sums."""
    import ctypes as C
    rng = np.random.default_rng(55)
    lib = da._ffi.lib()
    npdt = np.float32 if dtype == oracle.F32 else np.float16
    for n, dim, mg in ((1, 8, 32), (7, 33, 32), (33, 100, 64), (70, 768, 96), (96, 96, 96), (130, 260, 96), (200, 128, 96),
                       (256, 64, 96), (160, 1536, 96)):
        rows = (rng.standard_normal((n, dim)) * rng.uniform(0.1, 8.0, (n, 1))).astype(npdt)
        got = np.empty((n, mg), np.float32)
        nrm = np.empty(n, np.float32)
        da._ffi.check(lib.dann_debug_gram_tiles(-1, dtype | (0x100 if dtype == oracle.F16 else 0), rows.ctypes.data_as(C.c_void_p),
                                                n, dim, mg, got.ctypes.data_as(C.c_void_p), nrm.ctypes.data_as(C.c_void_p)),
                      "dann_debug_gram_tiles")
        wide = rows.astype(np.float32)
        want = oracle.gram_chain(wide)
        for i in range(n):
            m = min(i + 1, mg)  # columns j <= i inside the block
            assert np.array_equal(bits(got[i, :m]), bits(want[i, :m])), (n, dim, i)
        exact = (wide.astype(np.float64) ** 2).sum(1)
        assert np.all(np.abs(nrm.astype(np.float64) - exact) <= np.spacing(exact.astype(np.float32)).astype(np.float64)), (n, dim)


def test_gram_tiles_of_f16_rows_on_the_f16_matrix_core_stay_inside_the_error_interval():
    """gram_tiles_f16_kernel (round 6, the builds' default for f16 rows): v_mfma_f32_32x32x16_f16 -- exact products, f32
    sums in the hardware's own order.  Not the reference's chain bit for bit, and it need not be: the sweep trusts a Gram
    entry only through E = c1 (|x|^2 + |y|^2) + c2 |d'| with c1 = 1.05 (K + 4) 2^-24 (K = dim rounded up to 32).  Every
    entry of the lower triangle lies within c1 / 2 (|x|^2 + |y|^2) of the exact product -- half the interval is left for
    the chain's own error --, is exact on integer-valued rows, and the norms are the f64 sums."""
    import ctypes as C
    rng = np.random.default_rng(56)
    lib = da._ffi.lib()
    for n, dim, mg in ((1, 8, 32), (7, 33, 32), (33, 100, 64), (70, 768, 96), (96, 96, 96), (130, 260, 96), (200, 128, 96),
                       (256, 64, 96), (160, 1536, 96)):
        for ints in (False, True):
            rows = (rng.integers(-4, 5, (n, dim)) if ints else
                    rng.standard_normal((n, dim)) * rng.uniform(0.1, 8.0, (n, 1))).astype(np.float16)
            got = np.empty((n, mg), np.float32)
            nrm = np.empty(n, np.float32)
            da._ffi.check(lib.dann_debug_gram_tiles(-1, oracle.F16, rows.ctypes.data_as(C.c_void_p), n, dim, mg,
                                                    got.ctypes.data_as(C.c_void_p), nrm.ctypes.data_as(C.c_void_p)),
                          "dann_debug_gram_tiles")
            wide = rows.astype(np.float64)
            exact = wide @ wide.T
            sq = (wide ** 2).sum(1)
            c1 = 1.05 * (((dim + 31) // 32 * 32) + 4) * 2.0 ** -24
            for i in range(n):
                m = min(i + 1, mg)
                err = np.abs(got[i, :m].astype(np.float64) - exact[i, :m])
                if ints:
                    assert np.all(err == 0), (n, dim, i)
                else:
                    assert np.all(err <= 0.5 * c1 * (sq[i] + sq[:m])), (n, dim, i, err.max())
            assert np.all(np.abs(nrm.astype(np.float64) - sq) <= np.spacing(sq.astype(np.float32)).astype(np.float64)), (n, dim)


@pytest.mark.parametrize("dtype,metric,dim", [(oracle.F16, oracle.L2, 96), (oracle.F16, oracle.INNER_PRODUCT, 100),
                                              (oracle.F32, oracle.L2, 260)])
def test_mfma_pool_prune_f16_and_default_policy_identical_to_oracle(dtype, metric, dim):
    """The three-kernel pool prune on f16 rows (widened exactly while the Gram slabs are filled; exact re-checks by the
    reference's f16 x f16 pair kernel) and as the default policy for rows of 1 KiB and more (dim 260 f32): adjacency ==
    the oracle's multi_insert, also with every decision forced through the exact path, and most pair distances the
    lazy scan asks for are answered from the Gram."""
    from diskann_amd.sharding import batch_schedule
    rng = np.random.default_rng(3000 + dim)
    n, R, maxdeg, lb = 5000, 24, 32, 64
    npdt = np.float32 if dtype == oracle.F32 else np.float16
    centers = rng.random((16, dim)).astype(np.float32)
    data = (centers[rng.integers(0, 16, n)] + 0.15 * rng.standard_normal((n, dim))).astype(npdt)
    start = data.astype(np.float32).mean(0, keepdims=True).astype(npdt)
    ocfg, gcfg = _cfgs(R, maxdeg, lb, intra_batch_candidates=oracle.IBC_NONE)
    growth, max_batch = 0.1, 1024
    oix = oracle.Index(dtype, metric, dim, n, maxdeg, start)
    oix.set_rows(0, data)
    for s0, b in batch_schedule(0, n, growth, max_batch):
        oix.multi_insert(ocfg, np.arange(s0, s0 + b, dtype=np.uint32))
    for escale in (None, 1e6):
        gix = da.Provider(dtype, metric, dim, n, maxdeg, start)
        gix.debug_set(gram_escale=escale)
        gix.set_elements(0, data)
        if dim * data.itemsize < 1024:
            gix.set_build_options(da.BUILD_MFMA_POOL)   # small rows: opt in; rows >= 1 KiB take the path by default
        gix.build(gcfg, 0, n, growth, max_batch)
        assert np.array_equal(gix.download_graph(), oix.adj), (dtype, metric, dim, escale)
        cnt = gix.build_counters()
        assert cnt[6] > n * 20 and cnt[7] > 0 and cnt[8] > 0
        if not escale:  # the Gram answered nearly everything the lazy scan asked for
            assert cnt[9] < 0.1 * cnt[8], (int(cnt[9]), int(cnt[8]))


@pytest.mark.parametrize("flags", [0, "row_only"])
def test_large_rows_take_the_split_back_edge_path(flags):
    """Rows of 1 KiB and more: back-edges go through scan/append + worklist prunes -- with the MFMA Gram by default, with
    the row kernel under BUILD_ROW_KERNEL_ONLY (also the route of f16 / integer / cosine rows of that size).  Both equal
    the oracle; a hub that receives hundreds of back-edges in one batch goes through the long-list launch."""
    from diskann_amd.sharding import batch_schedule
    rng = np.random.default_rng(260)
    n, dim, R, maxdeg, lb = 3000, 260, 12, 16, 40
    centers = rng.random((10, dim)).astype(np.float32)
    data = (centers[rng.integers(0, 10, n)] + 0.1 * rng.standard_normal((n, dim))).astype(np.float32)
    start = data.mean(0, keepdims=True).astype(np.float32)
    ocfg, gcfg = _cfgs(R, maxdeg, lb, intra_batch_candidates=oracle.IBC_NONE)
    growth, max_batch = 0.5, 1500      # big early batches: the start point's neighbourhood collects many back-edges at once
    oix = oracle.Index(oracle.F32, oracle.L2, dim, n, maxdeg, start)
    oix.set_rows(0, data)
    for s0, b in batch_schedule(0, n, growth, max_batch):
        oix.multi_insert(ocfg, np.arange(s0, s0 + b, dtype=np.uint32))
    gix = da.Provider(oracle.F32, oracle.L2, dim, n, maxdeg, start)
    gix.set_elements(0, data)
    if flags:
        gix.set_build_options(da.BUILD_ROW_KERNEL_ONLY)
    gix.build(gcfg, 0, n, growth, max_batch)
    assert np.array_equal(gix.download_graph(), oix.adj)
    cnt = gix.build_counters()
    assert (cnt[0] == 0) == bool(flags)


@pytest.mark.parametrize("max_backedges", [1, 2, None])
def test_dann_insert_is_the_references_single_insert(max_backedges):
    """DiskANNIndex::insert (index.rs:226-341): back-edges go to the first max_backedges of the new neighbours only
    (:324-327).  dann_insert == the oracle's insert for a small max_backedges and for the default (pruned_degree), where it
    also equals dann_insert_batch of one point; more than pruned_degree is refused (config/mod.rs:308-311)."""
    rng = np.random.default_rng(92)
    n, dim, R, maxdeg, lb = 300, 12, 6, 8, 20
    data = rand_vectors(rng, oracle.F32, n, dim)
    adj = np.zeros((n + 1, maxdeg + 1), np.uint32)
    oix, gix = make_pair(oracle.F32, oracle.L2, data, adj, data[:1], maxdeg)
    ocfg, gcfg = _cfgs(R, maxdeg, lb, max_backedges=max_backedges)
    for i in range(n):
        oix.insert(ocfg, i)
        gix.insert(gcfg, i)
    g = gix.download_graph()
    assert np.array_equal(g[:, 0], oix.adj[:, 0])
    mask = np.arange(maxdeg)[None, :] < oix.adj[:, :1]
    assert np.array_equal(g[:, 1:][mask], oix.adj[:, 1:][mask])
    if max_backedges == 1:   # fewer back-edges than multi_insert sends: the graphs differ
        _, gb = make_pair(oracle.F32, oracle.L2, data, adj, data[:1], maxdeg)
        for i in range(n):
            gb.insert_batch(gcfg, [i])
        assert not np.array_equal(gb.download_graph()[:, 0], g[:, 0])
    with pytest.raises(da.DannError):
        gix.insert(da.build_config(R, maxdeg, lb, max_backedges=R + 1), 0)
