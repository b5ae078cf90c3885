"""What dann_set_prune_tie_order(DANN_TIE_RUST) costs and changes (round 5).  Two builds each of (a) 200 k x 128 f32 rows
of the bench generator (continuous: no ties) and (b) the same rows scaled to u8 (integer distances: tied pools), under
DANN_TIE_POSITION and DANN_TIE_RUST: build seconds, rows of the graph that differ, recall@10 at L = 32 of both.
usage: python scratch/tie_order_cost.py [n] [f32|u8|both] [position|rust|both] > gpurun_out/<tag>_tie_cost.json
(one dtype and one order per process under rocprofv3: scratch/r05_tie_trace.sh)"""
import json
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diskann_amd as da
from benchdata import ground_truth, make_data, recall_at_k

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
only_dt = sys.argv[2] if len(sys.argv) > 2 else "both"
only_order = sys.argv[3] if len(sys.argv) > 3 else "both"
dev = torch.device("cuda", 0)
base, queries = make_data(torch, dev, n, 128, 2000, "sift_like", 0xD15CA11, 0xD15CA12)
lo, hi = float(base.min()), float(base.max())
out = {"n": n, "dim": 128, "R": 32, "l_build": 100}
for name, dt, tobytes in (("f32", da.F32, lambda t: t.cpu().numpy()),
                          ("u8", da.U8, lambda t: ((t - lo) / (hi - lo) * 255.0).round().clamp(0, 255).to(torch.uint8).cpu().numpy())):
    if only_dt not in ("both", name):
        continue
    rows, qs = tobytes(base), tobytes(queries)
    mean = rows.astype(np.float64).mean(0)
    start = np.round(mean).astype(rows.dtype)[None, :] if rows.dtype == np.uint8 else mean.astype(np.float32)[None, :]  # not a copy of a row
    gt = ground_truth(torch, torch.from_numpy(rows.astype(np.float32)).to(dev), torch.from_numpy(qs.astype(np.float32)).to(dev), 10)
    graphs, res = [], {}
    for order, oname in ((da.TIE_POSITION, "position"), (da.TIE_RUST, "rust")):
        if only_order not in ("both", oname):
            continue
        p = da.Provider(dt, da.L2, 128, n, 32, start)
        p.set_elements(0, rows)
        p.set_prune_tie_order(order)
        torch.cuda.synchronize()
        t0 = time.time()
        p.build(da.build_config(28, 32, 100, intra_batch_candidates=da.IBC_NONE), 0, n, 0.02, 16384)
        torch.cuda.synchronize()
        secs = time.time() - t0
        ids, _, st = p.search(da.Knn(32, 1), qs, 10)
        g = p.download_graph()
        g[:, 1:][np.arange(32)[None, :] >= g[:, :1]] = 0
        graphs.append(g)
        res[oname] = {"build_seconds": round(secs, 3), "recall_at_10_L32": round(float(recall_at_k(ids, gt, 10)), 4),
                      "mean_degree": round(float(g[:n, 0].mean()), 3)}
        p.close()
    if len(graphs) == 2:
        res["graph_rows_that_differ"] = int((graphs[0] != graphs[1]).any(1).sum())
        res["slowdown"] = round(res["rust"]["build_seconds"] / res["position"]["build_seconds"], 2)
    out[name] = res
print(json.dumps(out))
