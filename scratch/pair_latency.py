"""u8 rows, L = 26: kernel time per launch of the pair kernel, the team kernel and one wave per query at small batch
sizes -- how short is a hop whose bookkeeping is branch-free, for one wavefront alone?
usage: python scratch/pair_latency.py"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    mode = sys.argv[1]
    import torch
    import diskann_amd as da
    import diskann_amd._ffi as ffi
    from benchdata import make_data
    lib = ffi.lib()
    dev = torch.device("cuda", 0)
    n, dim, R = 1_000_000, 128, 32
    base, queries = make_data(torch, dev, n, dim, 100000, "sift_like", 0xD15CA11, 0xD15CA12)
    mean = base.double().mean(0).float()
    medoid = int(torch.argmin(((base - mean[None, :]) ** 2).sum(1)).item())
    lo, hi = float(base.min()), float(base.max())
    rows = ((base - lo) * (255.0 / (hi - lo))).round().clamp(0, 255).to(torch.uint8)
    qrows = ((queries - lo) * (255.0 / (hi - lo))).round().clamp(0, 255).to(torch.uint8).contiguous()
    prov = da.Provider(da.U8, da.L2, dim, n, R, rows[medoid:medoid + 1].cpu().numpy(), device=0)
    prov.set_elements(0, rows.cpu().numpy())
    prov.build(da.build_config(28, R, 100, intra_batch_candidates=da.IBC_NONE), 0, n, 0.05, 16384)
    k = 10
    d_ids = torch.empty((100000, k), dtype=torch.int32, device=dev)
    d_d = torch.empty((100000, k), dtype=torch.float32, device=dev)
    d_st = torch.empty((100000, 5), dtype=torch.int32, device=dev)
    out = []
    for nq in [int(x) for x in os.environ.get("PAIR_LAT_NQ", "1,2,16,256,1024,2048,4096,100000").split(",")]:
        for L in (26,):
            def run(off):
                ffi.check(lib.dann_search_batch_device(prov._h, C.c_void_p(qrows.data_ptr() + off * dim), nq, L, 1, k,
                                                       C.c_void_p(d_ids.data_ptr()), C.c_void_p(d_d.data_ptr()),
                                                       C.c_void_p(d_st.data_ptr())), "search")
            for r in range(3):
                run((r * nq) % max(1, 100000 - nq))
            prov.kernel_time_reset()
            reps = 200 if nq <= 16 else 50 if nq <= 4096 else 10
            hops = 0
            for r in range(reps):
                run((r * nq) % max(1, 100000 - nq))
                if nq <= 2:
                    hops += int(d_st[:nq].cpu().numpy().view(np.uint32)[:, 1].max())
            ms, nl = prov.kernel_time(0)
            out.append((nq, L, round(ms / nl * 1e3, 1), round(nq / (ms / nl * 1e-3) / 1e6, 2), round(hops / reps, 1) if hops else None))
    print(mode, out)
else:
    for mode, env in (("pair", {"DANN_PAIR_MIN_QUERIES": "1"}), ("team/one-wave (default below 1025)", {"DANN_TUNE_OFF": "16"}),
                      ("one wave per query", {"DANN_TUNE_OFF": "20"})):
        e = dict(os.environ, **env)
        r = subprocess.run([sys.executable, __file__, mode], env=e, capture_output=True, text=True)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:])
