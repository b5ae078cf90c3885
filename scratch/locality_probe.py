#!/usr/bin/env python3
"""Does the ORDER in which a batch of searches is handed to the kernel matter on a 768-d index?  (The build's insert
searches of one batch are independent: they may run in any order.)  Same queries, same kernel; three orders:
random, sorted by blob label (the best any locality key could do on this generator), sorted by nearest of 64 pivots.
Usage: python scratch/locality_probe.py [n] [dim] [L]"""
import ctypes as C, sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diskann_amd as da
from diskann_amd import _ffi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
L = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dev = torch.device("cuda:0")
lib = _ffi.lib()
nblobs = max(256, n // 3906)
g0 = torch.Generator(device=dev); g0.manual_seed(0xD15CA11)
centers = torch.rand((nblobs, dim), generator=g0, device=dev)
basis = torch.randn((16, dim), generator=g0, device=dev) / 4.0


def draw(m, gen):
    lab = torch.randint(0, nblobs, (m,), generator=gen, device=dev)
    z = torch.randn((m, 16), generator=gen, device=dev)
    noise = torch.randn((m, dim), generator=gen, device=dev)
    return centers[lab] + 0.25 * (z @ basis) + 0.02 * noise, lab


g = torch.Generator(device=dev); g.manual_seed(1)
base, _ = draw(n, g)
start = base.mean(0, keepdim=True).cpu().numpy()
prov = da.Provider(da.F32, da.L2, dim, n, 64, start, device=0)
torch.cuda.synchronize()
prov.set_elements_device(0, base.data_ptr(), n)
t = time.time()
prov.build(da.build_config(56, 64, 128, intra_batch_candidates=da.IBC_NONE), 0, n, 2.0, 16384)
torch.cuda.synchronize()
print(f"build {time.time() - t:.2f}s", flush=True)
nq, k = 16384, 10
gq = torch.Generator(device=dev); gq.manual_seed(2)
q, lab = draw(nq, gq)
piv = base[torch.arange(64, device=dev) * (n // 64)]
near = torch.cdist(q, piv).argmin(1)
proj = q @ torch.randn(dim, device=dev)
orders = {"random": torch.arange(nq, device=dev), "by_blob": torch.argsort(lab), "by_pivot64": torch.argsort(near),
          "by_projection": torch.argsort(proj)}
d_ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
d_d = torch.empty((nq, k), dtype=torch.float32, device=dev)
d_st = torch.empty((nq, 5), dtype=torch.int32, device=dev)
for name, o in orders.items():
    qq = q[o].contiguous()

    def run():
        _ffi.check(lib.dann_search_batch_device(prov._h, C.c_void_p(qq.data_ptr()), nq, L, 1, k, C.c_void_p(d_ids.data_ptr()),
                                                C.c_void_p(d_d.data_ptr()), C.c_void_p(d_st.data_ptr())), "search")
    run(); run()
    prov.kernel_time_reset()
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    ms, nl = prov.kernel_time(0)
    st = d_st.cpu().numpy().view(np.uint32)
    alg = int(st[:, 0].sum()) * dim * 4 + int(st[:, 1].sum()) * 65 * 4
    print(f"{name:14s} {ms / nl:8.3f} ms  {alg / (ms / nl * 1e-3) / 1e9:8.0f} GB/s algorithmic  cmps {st[:, 0].mean():.0f}", flush=True)
