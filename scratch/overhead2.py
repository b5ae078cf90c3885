import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import diskann_amd as da
from diskann_amd import _ffi
rng = np.random.default_rng(0)
n, dim, R = 100000, 128, 32
data = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
adj = np.zeros((n + 1, R + 1), np.uint32); adj[:, 0] = R; adj[:, 1:] = rng.integers(0, n, (n + 1, R), dtype=np.uint32)
p = da.Provider(da.F32, da.L2, dim, n, R, data[:1]); p.set_elements(0, data); p.upload_graph(adj)
lib=_ffi.lib()
nq=10000
q = torch.rand((nq, dim), device='cuda')*2-1
ids = torch.empty((nq,10),dtype=torch.int32,device='cuda'); d=torch.empty((nq,10),device='cuda'); st=torch.empty((nq,4),dtype=torch.int32,device='cuda')
args=(p._h, C.c_void_p(q.data_ptr()), nq, 32, 1, 10, C.c_void_p(ids.data_ptr()), C.c_void_p(d.data_ptr()), C.c_void_p(st.data_ptr()))
for _ in range(5): lib.dann_search_batch_device(*args)
for i in range(6):
    t0=time.perf_counter(); lib.dann_search_batch_device(*args); t1=time.perf_counter()
    lib.dann_layer_bytes(0,128); t2=time.perf_counter()
    print(f"call {1e6*(t1-t0):.1f} us  trivial {1e6*(t2-t1):.1f} us", file=sys.stderr)
t0=time.perf_counter()
for _ in range(50): lib.dann_search_batch_device(*args)
print("loop avg us", (time.perf_counter()-t0)/50*1e6, file=sys.stderr)
