"""small host-pointer calls through the combiner: 1 / 4 / 16 / 32 native threads x single-query dann_search_batch calls
(L = 26 and 64), against the same calls with the combiner off (host_pipeline = 0)"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import diskann_amd as da
from bench import make_data

dev = torch.device("cuda", 0)
n, dim, R = 1000000, 128, 32
base, queries = make_data(torch, dev, n, dim, 100000, "sift_like", 0xD15CA11, 0xD15CA12)
mean = base.double().mean(0).float()
medoid = int(torch.argmin(((base - mean[None, :]) ** 2).sum(1)).item())
prov = da.Provider(da.F32, da.L2, dim, n, R, base[medoid:medoid + 1].cpu().numpy(), device=0)
prov.set_elements(0, base.cpu().numpy())
prov.build(da.build_config(28, R, 100, intra_batch_candidates=da.IBC_NONE), 0, n, 0.05, 16384)
qh = queries.cpu().numpy()
for mode in (1,):
    prov.debug_set(host_pipeline=mode)
    for L in (26, 64):
        for threads in (1, 16, 24, 32, 64):
            nq = 1500 * threads
            prov.concurrent_callers(qh[:128 * threads], L, 10, threads=threads, mode=0)
            s0 = prov.small_call_stats()
            ids, d, lat, secs = prov.concurrent_callers(qh[:nq], L, 10, threads=threads, mode=0)
            s1 = prov.small_call_stats()
            print(f"combiner {'on ' if mode else 'off'} L={L} threads={threads:2d}: {nq / secs:10,.0f} QPS  mean {lat.mean():7.1f} us "
                  f"p50 {np.percentile(lat, 50):7.1f} p99 {np.percentile(lat, 99):7.1f}  calls/launch "
                  f"{(s1[1] - s0[1]) / max(s1[0] - s0[0], 1):.2f}  p99.9 {np.percentile(lat, 99.9):.0f} max {lat.max():.0f} over-1ms {int((lat > 1000).sum())}", flush=True)

