"""A/B of the row prefetch (DANN_TUNE_OFF bit 1) in team launches of 64 .. 1024 queries."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import diskann_amd as da
from diskann_amd import _ffi
from benchdata import make_data
n, dim, R, k = 1000000, 128, 32, 10
dev = torch.device("cuda", 0)
base, q = make_data(torch, dev, n, dim, 100000, "sift_like", 0xD15CA11, 0xD15CA12)
mean = base.double().mean(0).float()
medoid = int(torch.argmin(((base - mean[None, :]) ** 2).sum(1)).item())
p = da.Provider(da.F32, da.L2, dim, n, R, base[medoid:medoid + 1].cpu().numpy())
p.set_elements(0, base.cpu().numpy())
p.build(da.build_config(28, R, 100, intra_batch_candidates=da.IBC_NONE), 0, n, 0.05, 16384)
lib = _ffi.lib()
d_ids = torch.empty((100000, k), dtype=torch.int32, device=dev)
d_d = torch.empty((100000, k), dtype=torch.float32, device=dev)
d_st = torch.empty((100000, 5), dtype=torch.int32, device=dev)
def timed(nq, L, reps):
    def call(r):
        qptr = q.data_ptr() + (r % max(1, min(64, 100000 // nq))) * nq * dim * 4
        lib.dann_search_batch_device(p._h, C.c_void_p(qptr), nq, L, 1, k, C.c_void_p(d_ids.data_ptr()), C.c_void_p(d_d.data_ptr()), C.c_void_p(d_st.data_ptr()))
    for r in range(3): call(r)
    torch.cuda.synchronize(); p.kernel_time_reset()
    for r in range(reps): call(r)
    torch.cuda.synchronize()
    ms, launches = p.kernel_time(0)
    return ms / max(launches, 1) * 1e3
for rnd in range(2):
    for tune in ("0", "1", "4", "5"):
        os.environ["DANN_TUNE_OFF"] = tune
        print(f"tune_off {tune}: " + "  ".join(f"{nq}x{L}: {timed(nq, L, 60):.1f} us" for nq, L in ((64, 26), (256, 26), (512, 26), (1024, 26), (1024, 64), (2048, 26), (4096, 26))), flush=True)
