"""Phase split of the GPU index build (HIP-event clocks of the library: beam search with record, pool prune, back-edge
prune; the rest is sorting / host orchestration).  usage: python scratch/build_phases.py [n dim R pruned l_build max_batch]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import diskann_amd as da
from benchdata import make_data

mfma = "--mfma" in sys.argv
metric = da.INNER_PRODUCT if "--ip" in sys.argv else da.L2
a = [int(x) for x in sys.argv[1:] if not x.startswith("--")]
n, dim, R, pruned, lb, mb = (a + [1000000, 128, 32, 28, 100, 16384][len(a):])[:6]
dev = torch.device('cuda', 0)
base, q = make_data(torch, dev, n, dim, 1000, 'sift_like', 0xD15CA11, 0xD15CA12)
mean = base.double().mean(0).float()
medoid = int(torch.argmin(((base - mean[None, :]) ** 2).sum(1)).item())
f16 = "--f16" in sys.argv
rows = base.half() if f16 else base
p = da.Provider(da.F16 if f16 else da.F32, metric, dim, n, R, rows[medoid:medoid + 1].cpu().numpy())
if "--rowonly" in sys.argv:
    p.set_build_options(da.BUILD_ROW_KERNEL_ONLY)
if mfma:
    p.set_build_options(da.BUILD_MFMA_BACKEDGE | (0 if "--nopool" in sys.argv else da.BUILD_MFMA_POOL))
for s0 in range(0, n, 1 << 20):
    p.set_elements(s0, rows[s0:s0 + (1 << 20)].cpu().numpy())
p.kernel_time_reset()
torch.cuda.synchronize()
t = time.time()
nb = p.build(da.build_config(pruned, R, lb, intra_batch_candidates=da.IBC_NONE), 0, n, 0.05, mb)
torch.cuda.synchronize()
dt = time.time() - t
ks = [p.kernel_time(i) for i in range(4)]
import json
c = [int(x) for x in p.build_counters()]
row_b, adj_b = dim * (2 if f16 else 4), (R + 1) * 4
print(json.dumps({"n": n, "dim": dim, "R": R, "pruned": pruned, "l_build": lb, "max_batch": mb, "mfma": mfma, "metric": int(metric),
                  "build_seconds": dt, "batches": nb,
                  "search": {"cmps": c[2], "hops": c[3], "algorithmic_bytes": c[2] * row_b + c[3] * adj_b},
                  "prune_row_kernel": {"pair_distances": c[4], "list_distances": c[5],
                                       "algorithmic_bytes": (2 * c[4] + c[5]) * row_b},
                  "mfma": {"prunes": c[0], "too_long_for_gram": c[1], "gram_rows": c[6], "gram_entries": c[7],
                           "flop": 2 * c[7] * dim, "row_bytes_read": c[6] * row_b,
                           "pairs_asked_by_gram_sweeps": c[8], "of_those_exact_rechecks": c[9],
                           "mfma_share_of_all_prune_pair_distances": (c[8] - c[9]) / max(1, c[8] - c[9] + c[4])},
                  "f16": f16, "env": {k: os.environ[k] for k in os.environ if k.startswith("DANN_")}}), flush=True)
print(f"mfma={mfma} metric={metric} ", end="")
print(f"n={n} dim={dim} R={R}/{pruned} l_build={lb} max_batch={mb}: build {dt:.3f}s ({n / dt:,.0f} pts/s) batches {nb}; "
      f"search {ks[0][0]:.0f} ms ({ks[0][1]}), prune {ks[2][0]:.0f} ms ({ks[2][1]}), backedge {ks[3][0]:.0f} ms ({ks[3][1]}); "
      f"other {dt * 1e3 - sum(k[0] for k in ks):.0f} ms", flush=True)
tms, tl = p.kernel_time(5)
print(f"gram_tiles {tms:.1f} ms over {tl} launches: {2 * c[7] * dim / max(tms, 1e-9) / 1e9:.1f} TFLOP/s", flush=True)
