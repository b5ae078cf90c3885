"""latency-regime launches on u8 rows (BASELINE config 3's variant): 1 / 16 / 256 / 1024 queries per launch at L = 26 and 64,
teams (default below 1025 queries) against two queries per wavefront (forced) and one wavefront per query"""
import ctypes as C
import sys
import time
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
lo, hi = float(base.min()), float(base.max())
rows = ((base - lo) * (255.0 / (hi - lo))).round().clamp(0, 255).to(torch.uint8).cpu().numpy()
qrows = ((queries - lo) * (255.0 / (hi - lo))).round().clamp(0, 255).to(torch.uint8)
prov = da.Provider(da.U8, da.L2, dim, n, R, rows[medoid:medoid + 1], device=0)
prov.set_elements(0, rows)
prov.build(da.build_config(28, R, 100, intra_batch_candidates=da.IBC_NONE), 0, n, 0.05, 16384)
prov.debug_set(time_small_launches=1)
lib = da._ffi.lib()
k = 10
d_ids = torch.empty((4096, k), dtype=torch.int32, device=dev)
d_d = torch.empty((4096, k), dtype=torch.float32, device=dev)
d_st = torch.empty((4096, 5), dtype=torch.int32, device=dev)


def run(nq, L, off):
    qptr = qrows.data_ptr() + off * dim
    da._ffi.check(lib.dann_search_batch_device(prov._h, C.c_void_p(qptr), nq, L, 1, k, C.c_void_p(d_ids.data_ptr()),
                                               C.c_void_p(d_d.data_ptr()), C.c_void_p(d_st.data_ptr())), "search")


for label, sw in (("teams (default)", {}), ("two queries per wavefront", dict(pair_min_queries=1, team_max_queries=0)),
                  ("one wavefront per query", dict(tune_off=4 | 16))):
    prov.debug_set(pair_min_queries=None, team_max_queries=None, tune_off=None)
    prov.debug_set(**sw)
    for L in (26, 64):
        for nq in (1, 16, 256, 1024):
            for r in range(3):
                run(nq, L, r * nq)
            torch.cuda.synchronize()
            prov.kernel_time_reset()
            f0 = prov.search_families()
            reps = 100
            t0 = time.perf_counter()
            for r in range(reps):
                run(nq, L, (r * nq) % (100000 - nq))
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / reps
            ms, launches = prov.kernel_time(0)
            f1 = prov.search_families()
            fam = "+".join(f for f in f1 if f1[f][0] > f0[f][0])
            print(f"u8 {label:26s} nq={nq:5d} L={L}: wall {wall * 1e6:7.1f} us kernel {ms / max(launches, 1) * 1e3:7.1f} us "
                  f"{nq / wall / 1e6:6.3f} M QPS [{fam}]", flush=True)
