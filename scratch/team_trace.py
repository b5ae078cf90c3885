"""One-point run for a kernel trace of the team kernel: 400 launches of one query at L = 64 (and 400 at L = 26) on the
headline index; prints the HIP-event average next to which the rocprofv3 trace of the same process is read."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
d_ids = torch.empty((1024, k), dtype=torch.int32, device=dev)
d_d = torch.empty((1024, k), dtype=torch.float32, device=dev)
d_st = torch.empty((1024, 5), dtype=torch.int32, device=dev)
for nq, L in ((1, 64), (1, 26), (256, 26)):
    def call(r):
        qptr = q.data_ptr() + (r % 64) * nq * dim * 4
        lib.dann_search_batch_device(p._h, C.c_void_p(qptr), nq, L, 1, k, C.c_void_p(d_ids.data_ptr()), C.c_void_p(d_d.data_ptr()), C.c_void_p(d_st.data_ptr()))
    for r in range(3): call(r)
    torch.cuda.synchronize(); p.kernel_time_reset()
    for r in range(400): call(r)
    torch.cuda.synchronize()
    ms, launches = p.kernel_time(0)
    print(f"team kernel, {nq} x L={L}: {ms / launches * 1e3:.1f} us by HIP events over {launches} launches", flush=True)
