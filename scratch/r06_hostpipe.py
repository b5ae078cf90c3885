"""dann_search_batch with host pointers, 100 000 queries: pageable / page-locked caller buffers x lanes x chunk size"""
import time, ctypes as C
import numpy as np, torch
import diskann_amd as da
from diskann_amd import _ffi
from benchdata import make_data
dev = torch.device("cuda:0")
n, dim, nq, L, k = 1_000_000, 128, 100_000, 26, 10
base, queries = make_data(torch, dev, n, dim, nq, "sift_like", 0xD15CA11, 0xD15CA12)
mean = base.double().mean(0).float()
medoid = int(torch.argmin(((base - mean[None, :]) ** 2).sum(1)).item())
prov = da.Provider(da.F32, da.L2, dim, n, 32, base[medoid:medoid + 1].cpu().numpy())
prov.set_elements(0, base.cpu().numpy())
prov.build(da.build_config(28, 32, 100, intra_batch_candidates=da.IBC_NONE), 0, n, 0.05, 16384)
lib = _ffi.lib()
qh = queries.cpu().numpy()
hi, hd = np.empty((nq, k), np.uint32), np.empty((nq, k), np.float32)
pq = torch.empty(qh.shape, dtype=torch.float32, pin_memory=True); pq.numpy()[...] = qh
pi = torch.empty((nq, k), dtype=torch.int32, pin_memory=True); pd = torch.empty((nq, k), dtype=torch.float32, pin_memory=True)
d_q = queries; d_i = torch.empty((nq, k), dtype=torch.int32, device=dev); d_d = torch.empty((nq, k), dtype=torch.float32, device=dev); d_s = torch.empty((nq, 5), dtype=torch.int32, device=dev)
def devcall():
    _ffi.check(lib.dann_search_batch_device(prov._h, C.c_void_p(d_q.data_ptr()), nq, L, 1, k, C.c_void_p(d_i.data_ptr()), C.c_void_p(d_d.data_ptr()), C.c_void_p(d_s.data_ptr())), "dev")
def pageable():
    _ffi.check(lib.dann_search_batch(prov._h, qh.ctypes.data, nq, L, 1, k, hi.ctypes.data, hd.ctypes.data, None), "h")
def pinned():
    _ffi.check(lib.dann_search_batch(prov._h, pq.data_ptr(), nq, L, 1, k, pi.data_ptr(), pd.data_ptr(), None), "p")
def t(f, reps=8):
    f(); f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print(f"device-resident call: {t(devcall):.3f} ms", flush=True)
for lanes in (1, 2, 3, 4, 6):
    for chunk in (8192, 16384, 32768):
        prov.debug_set(host_pipeline=lanes if lanes > 1 else 2, host_chunk=chunk)
        if lanes == 1: prov.debug_set(host_pipeline=2, host_chunk=chunk)
        a, b = t(pageable), t(pinned)
        print(f"lanes {lanes if lanes > 1 else 2} chunk {chunk}: pageable {a:.3f} ms ({nq / a / 1e3:.2f} M QPS)  pinned {b:.3f} ms ({nq / b / 1e3:.2f} M QPS)", flush=True)
assert np.array_equal(hi, pi.numpy().view(np.uint32))
