"""Concurrent-caller numbers on the headline index (1 M x 128 f32): N host threads x single-query calls through the
launch path (mode 0) and through the resident server (mode 1, several pipeline depths); latency percentiles.
usage: python scratch/server_lab.py [n] [L]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import diskann_amd as da
from benchdata import make_data

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 26
dim, R, k = 128, 32, 10
dev = torch.device("cuda", 0)
base, q = make_data(torch, dev, n, dim, 200000, "sift_like", 0xD15CA11, 0xD15CA12)
mean = base.double().mean(0).float()
medoid = int(torch.argmin(((base - mean[None, :]) ** 2).sum(1)).item())
p = da.Provider(da.F32, da.L2, dim, n, R, base[medoid:medoid + 1].cpu().numpy())
p.set_elements(0, base.cpu().numpy())
t0 = time.time()
p.build(da.build_config(28, R, 100, intra_batch_candidates=da.IBC_NONE), 0, n, 0.05, 16384)
print(f"build {time.time() - t0:.2f}s", flush=True)
qh = q.cpu().numpy()
ref_ids, ref_d, _ = p.search(da.Knn(L), qh[:20000], k)
out = {"n": n, "L": L}


def pct(lat):
    return {"mean_us": float(lat.mean()), "p50_us": float(np.percentile(lat, 50)), "p90_us": float(np.percentile(lat, 90)),
            "p99_us": float(np.percentile(lat, 99))}


for threads in (1, 16):
    nq = 2000 * threads
    p.concurrent_callers(qh[:256], L, k, threads=threads, mode=0)
    ids, d, lat, secs = p.concurrent_callers(qh[:nq], L, k, threads=threads, mode=0)
    out[f"launch_path_{threads}_threads"] = {"qps": nq / secs, "identical": bool(np.array_equal(ids[:20000], ref_ids[:nq])), **pct(lat)}
    print(json.dumps({f"launch_path_{threads}_threads": out[f"launch_path_{threads}_threads"]}), flush=True)
for workers in (1024, 2048):
    p.server_start(L, k, workers=workers, ring=8192)
    try:
        for threads, depth in ((1, 1), (16, 1), (16, 8), (16, 64), (16, 128)):
            nq = min(200000, max(4000, 20000 * min(depth, 8) * threads // 16))
            p.concurrent_callers(qh[:2000], L, k, threads=threads, mode=1, depth=depth)
            ids, d, lat, secs = p.concurrent_callers(qh[:nq], L, k, threads=threads, mode=1, depth=depth)
            m = min(nq, 20000)
            key = f"server_w{workers}_t{threads}_d{depth}"
            out[key] = {"qps": nq / secs, "identical": bool(np.array_equal(ids[:m], ref_ids[:m])), "queries": nq, **pct(lat)}
            print(json.dumps({key: out[key]}), flush=True)
        out[f"server_w{workers}_stats"] = p.server_stats()
    finally:
        p.server_stop()
print(json.dumps(out), flush=True)
