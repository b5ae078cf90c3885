"""one launch of 1024 queries (BASELINE config 3), f32 1 M x 128, L = 26: teams of four wavefronts against one wavefront per
query, with the per-query hop counts of the batch (the launch lasts as long as its slowest query)"""
import sys, time
import ctypes as C
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
prov.debug_set(time_small_launches=1)  # (kernel microseconds of single-query launches are printed below)
lib = da._ffi.lib()
k = 10
d_ids = torch.empty((100000, k), dtype=torch.int32, device=dev)
d_d = torch.empty((100000, k), dtype=torch.float32, device=dev)
d_st = torch.empty((100000, 5), dtype=torch.int32, device=dev)


def run(nq, L, off):
    qptr = queries.data_ptr() + off * dim * 4
    da._ffi.check(lib.dann_search_batch_device(prov._h, C.c_void_p(qptr), nq, L, 1, k, C.c_void_p(d_ids.data_ptr()),
                                               C.c_void_p(d_d.data_ptr()), C.c_void_p(d_st.data_ptr())), "search")


for L in (26, 64):
    for nq in (256, 512, 1024, 2048, 4096):
        for label, sw in (("teams", dict(team_max_queries=1 << 20)), ("one wave", dict(team_max_queries=0)),
                          ("one wave, row prefetch", dict(team_max_queries=0, tune_on=1))):
            prov.debug_set(team_max_queries=None, tune_on=None)
            prov.debug_set(**sw)
            for r in range(3):
                run(nq, L, r * nq)
            torch.cuda.synchronize()
            prov.kernel_time_reset()
            fam0 = prov.search_families()
            t0 = time.perf_counter()
            reps = 50
            for r in range(reps):
                run(nq, L, (r * nq) % (100000 - nq))
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / reps
            ms, launches = prov.kernel_time(0)
            fam1 = prov.search_families()
            fam = "+".join(f for f in fam1 if fam1[f][0] > fam0[f][0])
            st = d_st[:nq].cpu().numpy().view(np.uint32)
            hops = st[:, 1]
            print(f"L={L} nq={nq:5d} {label:24s} [{fam:9s}]: wall {wall * 1e6:7.1f} us kernel {ms / launches * 1e3:7.1f} us  "
                  f"{nq / wall / 1e6:5.2f} M QPS | hops mean {hops.mean():.1f} max {hops.max()}", flush=True)
