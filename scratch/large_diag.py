"""Larger-than-Infinity-Cache diagnosis: build an n-point index on the GPU, then
  * reachability of every slot from the start point (BFS over the built graph), per blob;
  * recall@10 over an L sweep, and recall counted only over the reachable ground-truth ids;
  * the shortlist ground truth checked against an all-f64 pass on a 1 000-query sample.
usage: python scratch/large_diag.py --n 10000000 --dist sift_like [--dim 128 --R 32 --pruned 28 --l-build 100]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--nq", type=int, default=10000)
    ap.add_argument("--dist", default="sift_like")
    ap.add_argument("--R", type=int, default=32)
    ap.add_argument("--pruned", type=int, default=28)
    ap.add_argument("--l-build", type=int, default=100)
    ap.add_argument("--growth", type=float, default=0.05)
    ap.add_argument("--max-batch", type=int, default=65536)
    ap.add_argument("--ibc", default="none")
    ap.add_argument("--Ls", default="16,26,40,64,128,256")
    ap.add_argument("--out", default="")
    ap.add_argument("--oracle-sample", type=int, default=0, help="check this many queries against the CPU oracle on the same graph")
    ap.add_argument("--oracle-L", type=int, default=64)
    args = ap.parse_args()
    import torch
    import diskann_amd as da
    from diskann_amd import _ffi
    from benchdata import make_data, ground_truth, ground_truth_f64, recall_at_k, reachable_from
    dev = torch.device("cuda", 0)
    t0 = time.time()
    base, queries, lab = make_data(torch, dev, args.n, args.dim, args.nq, args.dist, 0xD15CA11, 0xD15CA12, labels=True)
    mean = base.double().mean(0).float()
    medoid = int(torch.argmin(((base - mean[None, :]) ** 2).sum(1)).item())
    start = base[medoid:medoid + 1].cpu().numpy()
    prov = da.Provider(da.F32, da.L2, args.dim, args.n, args.R, start, device=0)
    for s in range(0, args.n, 1 << 21):
        prov.set_elements(s, base[s:s + (1 << 21)].cpu().numpy())
    t1 = time.time()
    ibc = {"none": da.IBC_NONE, "all": da.IBC_ALL}.get(args.ibc, None)
    cfg = da.build_config(args.pruned, args.R, args.l_build, intra_batch_candidates=int(args.ibc) if ibc is None else ibc)
    nb = prov.build(cfg, 0, args.n, args.growth, args.max_batch)
    t_build = time.time() - t1
    print(f"[diag] {args.dist} n={args.n} dim={args.dim}: data {t1 - t0:.1f}s build {t_build:.1f}s ({nb} batches)", flush=True)
    res = {"dist": args.dist, "n": args.n, "dim": args.dim, "build_seconds": t_build, "batches": nb, "medoid": medoid}
    # ---- reachability -------------------------------------------------------------------------------
    adj = prov.download_graph()
    seen, levels = reachable_from(torch, adj, [args.n], dev)
    reach = seen[:args.n]
    unreachable = int((~reach).sum().item())
    res["unreachable_points"] = unreachable
    res["bfs_levels"] = levels
    deg = torch.as_tensor(adj[:args.n, 0].astype(np.int64))
    res["mean_degree"] = float(deg.float().mean())
    if lab is not None:
        nblobs = int(lab.max().item()) + 1
        per_blob_total = torch.bincount(lab, minlength=nblobs)
        per_blob_unreach = torch.bincount(lab[~reach], minlength=nblobs)
        frac = (per_blob_unreach.float() / per_blob_total.clamp(min=1).float()).cpu().numpy()
        res["blobs_fully_unreachable"] = int((frac > 0.999).sum())
        res["blobs_partly_unreachable"] = int(((frac > 0.001) & (frac <= 0.999)).sum())
        res["blob_of_medoid"] = int(lab[medoid].item())
        # inter-blob edges: how many edges leave their blob
        a = torch.as_tensor(adj[:args.n].astype(np.int32)).to(dev)
        src_lab = lab[:, None].expand(-1, args.R)
        valid = torch.arange(args.R, device=dev)[None, :] < a[:, :1]
        dst = a[:, 1:].long().clamp(max=args.n - 1)
        cross = (lab[dst] != src_lab) & valid & (a[:, 1:] < args.n)
        res["cross_blob_edge_fraction"] = float(cross.sum().item() / max(valid.sum().item(), 1))
        del a, dst, cross, valid
    print(f"[diag] unreachable {unreachable} of {args.n} ({unreachable / args.n:.4f}), bfs levels {levels}, "
          f"blobs fully unreachable {res.get('blobs_fully_unreachable')}, partly {res.get('blobs_partly_unreachable')}, "
          f"cross-blob edges {res.get('cross_blob_edge_fraction')}", flush=True)
    # ---- ground truth + its check ---------------------------------------------------------------------
    k = 10
    gt = ground_truth(torch, base, queries, k)
    gt64 = ground_truth_f64(torch, base, queries[:1000], k)
    res["gt_shortlist_equals_f64_on_1000"] = float((np.sort(gt[:1000], 1) == np.sort(gt64, 1)).all(1).mean())
    gt_reach = reach.cpu().numpy()[gt]
    res["gt_ids_reachable_fraction"] = float(gt_reach.mean())
    print(f"[diag] gt check (shortlist == all-f64 on 1000 queries): {res['gt_shortlist_equals_f64_on_1000']:.4f}; "
          f"ground-truth ids reachable: {res['gt_ids_reachable_fraction']:.4f}", flush=True)
    # ---- recall sweep -----------------------------------------------------------------------------------
    lib = _ffi.lib()
    d_ids = torch.empty((args.nq, k), dtype=torch.int32, device=dev)
    d_d = torch.empty((args.nq, k), dtype=torch.float32, device=dev)
    d_st = torch.empty((args.nq, 5), dtype=torch.int32, device=dev)
    res["sweep"] = []
    gt64_set = [set(r.tolist()) for r in gt64]
    for L in [int(x) for x in args.Ls.split(",")]:
        for rep in range(2):
            prov.kernel_time_reset()
            _ffi.check(lib.dann_search_batch_device(prov._h, C.c_void_p(queries.data_ptr()), args.nq, L, 1, k,
                                                    C.c_void_p(d_ids.data_ptr()), C.c_void_p(d_d.data_ptr()),
                                                    C.c_void_p(d_st.data_ptr())), "search")
        ms, _ = prov.kernel_time(0)
        ids = d_ids.cpu().numpy().view(np.uint32)
        st = d_st.cpu().numpy().view(np.uint32)
        rec = recall_at_k(ids, gt, k)
        rec64 = sum(len(set(a.tolist()) & b) for a, b in zip(ids[:1000], gt64_set)) / (1000 * k)
        alg = int(st[:, 0].sum()) * args.dim * 4 + int(st[:, 1].sum()) * (args.R + 1) * 4
        row = {"L": L, "recall": rec, "recall_vs_f64_gt_first_1000": rec64, "cmps": float(st[:, 0].mean()), "hops": float(st[:, 1].mean()), "kernel_ms": ms,
               "alg_GBps": alg / (ms * 1e-3) / 1e9}
        res["sweep"].append(row)
        print(f"[diag] L={L} recall@10={rec:.4f} (vs all-f64 gt on 1000: {rec64:.4f}) cmps={row['cmps']:.0f} hops={row['hops']:.0f} kernel={ms:.3f} ms "
              f"alg={row['alg_GBps']:.0f} GB/s", flush=True)
    if args.oracle_sample:
        import oracle
        m, L = args.oracle_sample, args.oracle_L
        oix = oracle.Index(oracle.F32, oracle.L2, args.dim, args.n, args.R, start)
        base_h = base.cpu().numpy()
        oix.rows[:args.n, :] = base_h.view(np.uint8).reshape(args.n, -1)
        oix.adj[:] = adj
        qh = queries[:m].cpu().numpy()
        gi, gd, gst = prov.search(da.Knn(L, 1), qh, k)
        oi, od, oc, ost = oix.search_batch(qh, L, 1, k, threads=16, fast=True)
        same = bool(np.array_equal(gi, oi) and np.array_equal(gd.view(np.uint32), od.view(np.uint32)) and
                    np.array_equal(gst["cmps"], ost[:, 0]) and np.array_equal(gst["hops"], ost[:, 1]))
        res["oracle_sample"] = {"queries": m, "L": L, "ids_dists_cmps_hops_identical": same}
        print(f"[diag] oracle on the same graph, {m} queries at L={L}: identical = {same}", flush=True)
    print(json.dumps(res), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
