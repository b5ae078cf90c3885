"""time the PQ trainer's two halves on a 131 072 x 128 sample (bench's shape): k-means++ and 10 Lloyd iterations"""
import time, sys
import numpy as np
import diskann_amd as da
n, dim, nch = 131072, 128, 16
rng = np.random.default_rng(5)
cent = rng.normal(size=(256, dim)).astype(np.float32)
x = (cent[rng.integers(0, 256, n)] + 0.3 * rng.normal(size=(n, dim))).astype(np.float32)
off = np.linspace(0, dim, nch + 1).round().astype(np.uint32)
for rep in range(2):
    gens = [np.random.default_rng(700 + c) for c in range(nch)]
    t = time.perf_counter()
    cen, sel = da.pq_kmeanspp(x, off, 256, lambda c, m: int(gens[c].integers(0, m)), lambda c, h: float(gens[c].random() * h))
    t1 = time.perf_counter() - t
    t = time.perf_counter()
    piv, asg, res = da.pq_lloyds(x, off, cen, 10)
    t2 = time.perf_counter() - t
    print(f"rep {rep}: kmeans++ {t1:.3f} s, 10 lloyds {t2:.3f} s, selected {sel.min()}..{sel.max()}, residual sum {float(res.sum()):.6g}", flush=True)
import hashlib
print("pivots sha", hashlib.sha256(piv.tobytes()).hexdigest()[:16], "seeds sha", hashlib.sha256(cen.tobytes()).hexdigest()[:16])
