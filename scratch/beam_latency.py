"""Single-query latency vs beam width at L=64 (1 M x 128 index): wall per query, hops, recall@10."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, diskann_amd as da
from diskann_amd import _ffi
from benchdata import make_data, ground_truth, recall_at_k
lib = _ffi.lib(); dev = torch.device("cuda", 0)
n, dim = 1_000_000, 128
base, queries = make_data(torch, dev, n, dim, 2000, "sift_like", 0xD15CA11, 0xD15CA12)
mean = base.double().mean(0).float(); medoid = int(torch.argmin(((base - mean[None, :]) ** 2).sum(1)).item())
prov = da.Provider(da.F32, da.L2, dim, n, 32, base[medoid:medoid + 1].cpu().numpy()); prov.set_elements(0, base.cpu().numpy())
prov.build(da.build_config(28, 32, 100, intra_batch_candidates=da.IBC_NONE), 0, n, 0.05, 16384)
gt = ground_truth(torch, base, queries, 10)
k = 10
d_ids = torch.empty((2000, k), dtype=torch.int32, device=dev); d_d = torch.empty((2000, k), dtype=torch.float32, device=dev)
d_st = torch.empty((2000, 5), dtype=torch.int32, device=dev)
for L in (64, 26):
    for W in (1, 2, 4, 8):
        _ffi.check(lib.dann_search_batch_device(prov._h, C.c_void_p(queries.data_ptr()), 2000, L, W, k, C.c_void_p(d_ids.data_ptr()), C.c_void_p(d_d.data_ptr()), C.c_void_p(d_st.data_ptr())), "s")
        rec = recall_at_k(d_ids.cpu().numpy().view(np.uint32), gt, k); st = d_st.cpu().numpy().view(np.uint32)
        for r in range(20):
            lib.dann_search_batch_device(prov._h, C.c_void_p(queries.data_ptr() + r * dim * 4), 1, L, W, k, C.c_void_p(d_ids.data_ptr()), C.c_void_p(d_d.data_ptr()), C.c_void_p(d_st.data_ptr()))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for r in range(300):
            lib.dann_search_batch_device(prov._h, C.c_void_p(queries.data_ptr() + (r % 1000) * dim * 4), 1, L, W, k, C.c_void_p(d_ids.data_ptr()), C.c_void_p(d_d.data_ptr()), C.c_void_p(d_st.data_ptr()))
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300
        print(f"L={L} W={W}: single-query {dt * 1e6:.1f} us  recall@10 {rec:.4f}  hops {st[:, 1].mean():.1f} cmps {st[:, 0].mean():.0f}", flush=True)
