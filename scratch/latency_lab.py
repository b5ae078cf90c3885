"""Latency-regime lab (BASELINE configs 2 and 3): single query at L=64, 1024 and 10 000 concurrent queries, and the
100 000-query batch as the regression guard, for each DANN_TUNE_OFF setting; with --prof the -DDANN_PHASE_CYCLES
library (diskann_amd/libdann_prof.so, built by scratch/build_prof.sh) prints the per-hop cycle breakdown.
usage: python scratch/latency_lab.py [--prof] [--n 1000000]"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--prof", action="store_true")
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--tunes", default="3,2,1,0")
ap.add_argument("--points", default="", help="comma list nq:L[:reps] run instead of the standard set")
args = ap.parse_args()
import diskann_amd._ffi as ffi
if args.prof:
    ffi.LIB_PATH = os.environ.get("DANN_PROF_LIB") or os.path.join(ROOT, "diskann_amd", "libdann_prof.so")
import torch
import diskann_amd as da
from benchdata import make_data
lib = ffi.lib()
if args.prof:
    lib.dann_debug_phase_cycles.argtypes = [C.c_void_p, C.c_int]
dev = torch.device("cuda", 0)
n, dim, R = args.n, 128, 32
base, queries = make_data(torch, dev, n, dim, 100000, "sift_like", 0xD15CA11, 0xD15CA12)
mean = base.double().mean(0).float()
medoid = int(torch.argmin(((base - mean[None, :]) ** 2).sum(1)).item())
prov = da.Provider(da.F32, da.L2, dim, n, R, base[medoid:medoid + 1].cpu().numpy(), device=0)
prov.set_elements(0, base.cpu().numpy())
prov.build(da.build_config(28, R, 100, intra_batch_candidates=da.IBC_NONE), 0, n, 0.05, 16384)
prov.debug_set(time_small_launches=1)  # (kernel microseconds of single-query launches are printed below)
k = 10
d_ids = torch.empty((100000, k), dtype=torch.int32, device=dev)
d_d = torch.empty((100000, k), dtype=torch.float32, device=dev)
d_st = torch.empty((100000, 5), dtype=torch.int32, device=dev)


def run(nq, L, off=0):
    qptr = queries.data_ptr() + off * dim * 4
    ffi.check(lib.dann_search_batch_device(prov._h, C.c_void_p(qptr), nq, L, 1, k, C.c_void_p(d_ids.data_ptr()),
                                           C.c_void_p(d_d.data_ptr()), C.c_void_p(d_st.data_ptr())), "search")


def timed(nq, L, reps):
    for r in range(3):
        run(nq, L, (r * nq) % max(1, 100000 - nq))
    torch.cuda.synchronize()
    prov.kernel_time_reset()
    if args.prof:
        lib.dann_debug_phase_cycles(None, 1)
    t0 = time.perf_counter()
    hops = cmps = 0
    for r in range(reps):
        run(nq, L, (r * nq) % max(1, 100000 - nq))
        if args.prof:
            st = d_st[:nq].cpu().numpy().view(np.uint32)
            hops += int(st[:, 1].sum())
            cmps += int(st[:, 0].sum())
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    ms, launches = prov.kernel_time(0)
    line = f"nq={nq:6d} L={L:3d}: wall {wall * 1e6:9.1f} us  kernel {ms / max(launches, 1) * 1e3:9.1f} us  QPS {nq / wall:12,.0f}"
    if args.prof:
        buf = (C.c_ulonglong * 16)()
        lib.dann_debug_phase_cycles(buf, 0)
        v = [buf[i] / max(hops, 1) for i in range(16)]
        line += (f"\n      hops/q {hops / reps / nq:.1f} cmps/q {cmps / reps / nq:.0f} | cycles/hop: pop {v[0]:.0f} expand {v[1]:.0f} "
                 f"(visited loop {v[7]:.0f}) gather {v[2]:.0f} merge {v[3]:.0f} (ranks {v[11]:.0f}) total {v[4]:.0f} | "
                 f"pf hit {buf[5] / max(buf[5] + buf[6], 1):.2f} | survivors/merge {buf[8] / max(buf[10], 1):.1f} "
                 f"slow merges {buf[9] / max(buf[10], 1):.3f} | team: control wave's decision with the visited wave's candidates {buf[12] / max(buf[5], 1):.0f} "
                 f"(x{buf[5] / max(buf[5] + buf[6], 1):.2f}) otherwise {buf[13] / max(buf[6], 1):.0f} (waited for the pop x{buf[7] / max(buf[5] + buf[6], 1):.2f}) "
                 f"its loads {v[1]:.0f} start {v[2]:.0f} barrier wait {v[15]:.0f} | queue wave's barrier wait {v[14]:.0f}")
        if os.environ.get("DANN_PROF_FINE"):  # scratch build with the finer timers of the control wave's short path
            fp = max(buf[5], 1)
            line += f"\n      per hop: first gather wave go -> distances written {buf[10] / max(hops, 1):.0f}, its barrier wait {buf[8] / max(hops, 1):.0f}, visited wave's barrier wait {buf[9] / max(hops, 1):.0f}"
    print(line, flush=True)


for tune in [int(x) for x in args.tunes.split(",")]:
    prov.debug_set(tune_off=tune)
    print(f"---- DANN_TUNE_OFF={tune} (1: no row prefetch, 2: no latency-mode table sizing)", flush=True)
    if args.points:
        for pt in args.points.split(","):
            f = [int(x) for x in pt.split(":")]
            timed(f[0], f[1], f[2] if len(f) > 2 else 100)
        continue
    timed(1, 64, 300)
    timed(1, 26, 300)
    timed(1024, 26, 100)
    timed(1024, 64, 50)
    timed(10000, 26, 30)
    if not args.prof:
        timed(100000, 26, 10)
        timed(100000, 64, 5)
    for cap in (1024, 256):  # sustained rate with `cap` queries in flight (persistent waves, dann_set_max_concurrency)
        prov.set_max_concurrency(cap)
        print(f"  max_concurrency {cap}:", end=" ")
        timed(20 * cap, 26, 20)
        print(f"  max_concurrency {cap}:", end=" ")
        timed(20 * cap, 64, 10)
    prov.set_max_concurrency(0)
