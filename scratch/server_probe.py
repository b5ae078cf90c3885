"""Server throughput probe: threads x depth grid."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import diskann_amd as da
from benchdata import make_data
n, L, dim, R, k = 1000000, 26, 128, 32, 10
dev = torch.device("cuda", 0)
base, q = make_data(torch, dev, n, dim, 200000, "sift_like", 0xD15CA11, 0xD15CA12)
mean = base.double().mean(0).float()
medoid = int(torch.argmin(((base - mean[None, :]) ** 2).sum(1)).item())
p = da.Provider(da.F32, da.L2, dim, n, R, base[medoid:medoid + 1].cpu().numpy())
p.set_elements(0, base.cpu().numpy())
p.build(da.build_config(28, R, 100, intra_batch_candidates=da.IBC_NONE), 0, n, 0.05, 16384)
qh = q.cpu().numpy()
ring = int(os.environ.get("RING", "8192"))
p.server_start(L, k, workers=1024, ring=ring)
try:
    for threads, depth in ((1, 64), (1, 256), (4, 64), (16, 8), (16, 64), (16, 128)):
        nq = 60000
        p.concurrent_callers(qh[:2000], L, k, threads=threads, mode=1, depth=depth)
        ids, d, lat, secs = p.concurrent_callers(qh[:nq], L, k, threads=threads, mode=1, depth=depth)
        print(f"ring {ring} threads {threads} depth {depth}: {nq / secs / 1e6:.3f} M QPS, mean {lat.mean():.0f} us", flush=True)
finally:
    p.server_stop()
