"""per-hop timeline of ONE query through the team kernel (scratch build with the TR() timestamps of /tmp/profx: every
wave's arrival at / release from the hop's barrier, the control wave's "go" and its words for the visited wave): who
arrives last, hop by hop.  DANN_PROF_LIB = the instrumented library."""
import ctypes as C
import os
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
os.environ["DANN_LIB_PATH"] = os.environ["DANN_PROF_LIB"]
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
lib = da._ffi.lib()
lib.dann_debug_phase_cycles.argtypes = [C.c_void_p, C.c_int]
k = 10
d_ids = torch.empty((16, k), dtype=torch.int32, device=dev)
d_d = torch.empty((16, k), dtype=torch.float32, device=dev)
d_st = torch.empty((16, 5), dtype=torch.int32, device=dev)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 64
NW = 16 + 256 * 8 * 2
names = ["queue", "control", "visited", "gather0", "gather1"]
last_count = {nm: 0 for nm in names}
rows = []
for qi in range(40):
    qptr = queries.data_ptr() + qi * dim * 4
    for rep in range(2):  # second run of the same query: warm
        lib.dann_debug_phase_cycles(None, 1)
        da._ffi.check(lib.dann_search_batch_device(prov._h, C.c_void_p(qptr), 1, L, 1, k, C.c_void_p(d_ids.data_ptr()),
                                                   C.c_void_p(d_d.data_ptr()), C.c_void_p(d_st.data_ptr())), "search")
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * NW)()
    lib.dann_debug_phase_cycles(buf, 2)
    t = np.frombuffer(buf, dtype=np.uint64)[16:].reshape(256, 8, 2).astype(np.int64)
    hops = int(d_st[0, 1].item())
    for h in range(2, hops):  # hop h: released from barrier h-1 at t[h-1][1][1]
        t0 = t[h - 1, 1, 1]  # control wave's release from the previous barrier
        if t0 == 0 or t[h, 1, 0] == 0:
            continue
        arr = {nm: t[h, w, 0] - t0 for w, nm in enumerate(names)}
        rel = max(t[h, w, 1] for w in range(5)) - t0
        slow = int(t[h, 6, 0])
        go = t[h, 5, 0] - t0
        spec = t[h, 5, 1] - t0
        vspec = t[h, 6, 1] - t0
        merged = t[h - 1, 7, 0] - t0  # the queue wave's merge of the previous hop's distances done
        lastw = max(arr, key=arr.get)
        rows.append((slow, go, spec, vspec, merged, arr["queue"], arr["control"], arr["visited"], arr["gather0"], arr["gather1"], rel, names.index(lastw)))
rows = np.array(rows)
for slow in (0, 1):
    r = rows[rows[:, 0] == slow]
    if len(r) == 0:
        continue
    print(f"{'prepared hops (the visited wave had the candidates)' if slow == 0 else 'other hops (the control wave expands)'}: {len(r)} hops; cycles after the previous barrier's release, mean (median)")
    for j, nm in ((1, "control: go"), (2, "control: words for the visited wave"), (3, "visited wave has seen them"), (4, "queue: merge done"),
                  (5, "queue at the barrier"), (6, "control at the barrier"), (7, "visited at the barrier"), (8, "gather 0 at the barrier"),
                  (9, "gather 1 at the barrier"), (10, "barrier released")):
        print(f"    {nm:38s} {r[:, j].mean():7.0f} ({np.median(r[:, j]):6.0f})")
    who, cnt = np.unique(r[:, 11], return_counts=True)
    print("    last at the barrier: " + ", ".join(f"{names[int(w)]} {c / len(r):.2f}" for w, c in zip(who, cnt)))
