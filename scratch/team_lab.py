"""Latency regime A/B on the headline index: teams of wavefronts per query (default) vs one wave per query
(DANN_TUNE_OFF=4).  usage: python scratch/team_lab.py [n]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import diskann_amd as da
from diskann_amd import _ffi
from benchdata import make_data

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dim, R, k = 128, 32, 10
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
        lib.dann_search_batch_device(p._h, C.c_void_p(qptr), nq, L, 1, k, C.c_void_p(d_ids.data_ptr()), C.c_void_p(d_d.data_ptr()),
                                     C.c_void_p(d_st.data_ptr()))
    for r in range(3):
        call(r)
    torch.cuda.synchronize()
    p.kernel_time_reset()
    t0 = time.perf_counter()
    for r in range(reps):
        call(r)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    ms, nl = p.kernel_time(0)
    return wall, ms / max(nl, 1)


out = {}
for label, env in (("team_spec", None), ("team_nospec", "8"), ("one_wave", "4"), ("team_spec", None), ("team_nospec", "8"), ("one_wave", "4")):
    if env is None:
        os.environ.pop("DANN_TUNE_OFF", None)
    else:
        os.environ["DANN_TUNE_OFF"] = env
    for nq, L, reps in ((1, 64, 300), (1, 26, 300), (16, 64, 200), (256, 26, 100), (1024, 26, 100), (1024, 64, 50), (2048, 26, 50)):
        wall, kms = timed(nq, L, reps)
        key = f"{label}_nq{nq}_L{L}"
        e = out.setdefault(key, [])
        e.append({"wall_us": wall * 1e6, "kernel_us": kms * 1e3, "qps": nq / wall})
        print(key, json.dumps(e[-1]), flush=True)
os.environ.pop("DANN_TUNE_OFF", None)
# teams for larger batches too? (threshold study)
for lim in (512, 1024, 2048, 4096):
    pass
print(json.dumps(out), flush=True)
