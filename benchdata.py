"""Synthetic data, ground truth and graph diagnostics shared by bench.py and the scripts under scratch/
(measurement plumbing on torch; nothing here is product code)."""
import numpy as np


def make_data(torch, dev, n, dim, nq, dist, seed, qseed, labels=False):
    """SIFT-shaped synthetic f32 vectors (no dataset exists on the box).

    sift_like        256 Gaussian blobs around U(0,1)^dim centres, within-blob variation on a shared
                     16-dimensional random subspace (sigma 0.25) plus isotropic noise (sigma 0.02): low intrinsic
                     dimension like SIFT descriptors, blobs well separated (centre distance^2 ~ dim/6).
    sift_like:<s>[:<blobs>]   the same with the centres scaled by <s> (s = 0: one 16-d manifold) and <blobs>
                     blobs.  The per-blob density decides how hard the set is (3 900 points per blob at 1 M / 256):
                     10 M points keep that density with 2 560 blobs, while 10 M points in 256 blobs (39 000 per blob,
                     nearest-neighbour shell 0.6-0.77 wide against thousands of points within 1.0) are a different,
                     much harder problem for an R = 32 graph -- see DESIGN.md, "the 10 M plateau".
    uniform          i.i.d. U(-1, 1), the reference's own test distribution (diskann-inmem/src/layers/full.rs:528-532).
    Base vectors depend on `seed` only (every rank holds the same index); queries on `qseed`."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    scale, nblobs = 1.0, 256
    if dist.startswith("sift_like:"):
        parts = dist.split(":")
        scale = float(parts[1])
        if len(parts) > 2:
            nblobs = int(parts[2])
    if dist != "uniform":
        centers = torch.rand((nblobs, dim), generator=g, device=dev, dtype=torch.float32) * scale
        basis = torch.randn((16, dim), generator=g, device=dev, dtype=torch.float32) / 4.0

    def draw(m, gen):
        if dist == "uniform":
            return torch.rand((m, dim), generator=gen, device=dev, dtype=torch.float32) * 2 - 1, None
        lab = torch.randint(0, nblobs, (m,), generator=gen, device=dev)
        out = torch.empty((m, dim), device=dev, dtype=torch.float32)
        for s in range(0, m, 1 << 20):  # chunks bound the temporaries at 10 M+ rows
            e = min(m, s + (1 << 20))
            z = torch.randn((e - s, 16), generator=gen, device=dev, dtype=torch.float32)
            noise = torch.randn((e - s, dim), generator=gen, device=dev, dtype=torch.float32)
            out[s:e] = centers[lab[s:e]] + 0.25 * (z @ basis) + 0.02 * noise
        return out, lab

    base, blab = draw(n, g)
    gq = torch.Generator(device=dev)
    gq.manual_seed(qseed)
    queries, _ = draw(nq, gq)
    if labels:
        return base.contiguous(), queries.contiguous(), blab
    return base.contiguous(), queries.contiguous()


def ground_truth(torch, base, queries, k, chunk=None):
    """exact top-k by brute force: f32 GEMM shortlist of 4k, re-ranked in f64."""
    n = base.shape[0]
    bn = (base.double() ** 2).sum(1).float()
    if chunk is None:
        # the score matrix of one chunk stays below 2^31 elements: torch's brute force (GEMM + topk) returned wrong
        # shortlists for ~12 % of the queries on a 2000 x 10 M matrix (32-bit indexing), which read as a recall
        # plateau of 0.857 at every L on the 10 M-point index in round 1 ("the 10 M plateau")
        chunk = max(1, min(2048, (2 ** 31 - 1) // max(n, 1)))
    out = []
    for s in range(0, queries.shape[0], chunk):
        q = queries[s:s + chunk]
        d = bn[None, :] - 2.0 * (q @ base.T)
        cand = torch.topk(d, 4 * k, dim=1, largest=False).indices
        del d
        diff = base[cand].double() - q.double()[:, None, :]
        dd = (diff * diff).sum(-1)
        order = torch.argsort(dd, dim=1)[:, :k]
        out.append(torch.gather(cand, 1, order))
    return torch.cat(out).cpu().numpy()


def ground_truth_f64(torch, base, queries, k, rows=1 << 20):
    """exact top-k with every distance in f64 (no shortlist): the check of ground_truth() on a small query sample."""
    qd = queries.double()
    best_d = torch.full((queries.shape[0], k), float("inf"), dtype=torch.float64, device=base.device)
    best_i = torch.zeros((queries.shape[0], k), dtype=torch.int64, device=base.device)
    for s in range(0, base.shape[0], rows):
        b = base[s:s + rows].double()
        d = (qd * qd).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2.0 * (qd @ b.T)
        dd, ii = torch.topk(d, k, dim=1, largest=False)
        # exact re-evaluation of the chunk's shortlist (the expansion above cancels)
        diff = base[s:s + rows][ii].double() - qd[:, None, :]
        dd = (diff * diff).sum(-1)
        alld = torch.cat([best_d, dd], 1)
        alli = torch.cat([best_i, ii + s], 1)
        o = torch.argsort(alld, dim=1)[:, :k]
        best_d, best_i = torch.gather(alld, 1, o), torch.gather(alli, 1, o)
    return best_i.cpu().numpy()


def recall_at_k(ids, gt, k):
    # k-recall@k (diskann-benchmark-core/src/recall.rs:146-240), tie-free data
    hit = 0
    for a, b in zip(ids, gt):
        hit += len(set(a[:k].tolist()) & set(b[:k].tolist()))
    return hit / (len(gt) * k)


def reachable_from(torch, adj, starts, dev):
    """breadth-first reachability over an adjacency buffer in the Neighbors layout (rows [len, ids...]);
    returns a bool tensor over the slots and the number of BFS levels."""
    a = torch.as_tensor(adj.astype(np.int32, copy=False)).to(dev)
    nslots, w = a.shape
    col = torch.arange(w - 1, device=dev, dtype=torch.int32)[None, :]
    seen = torch.zeros(nslots, dtype=torch.bool, device=dev)
    frontier = torch.as_tensor(np.asarray(starts, dtype=np.int64), device=dev)
    seen[frontier] = True
    levels = 0
    while frontier.numel():
        levels += 1
        nxt = []
        for s in range(0, frontier.numel(), 1 << 22):
            rows = a[frontier[s:s + (1 << 22)]]
            valid = col < rows[:, :1].clamp(max=w - 1)
            ids = rows[:, 1:][valid].long()
            ids = ids[(ids >= 0) & (ids < nslots)]
            ids = ids[~seen[ids]]
            nxt.append(torch.unique(ids))
        cand = torch.unique(torch.cat(nxt)) if nxt else frontier[:0]
        cand = cand[~seen[cand]]
        seen[cand] = True
        frontier = cand
    return seen, levels
