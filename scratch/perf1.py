import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import diskann_amd as da
rng = np.random.default_rng(0)
n, dim, R = 1_000_000, 128, 32
data = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
adj = np.zeros((n + 1, R + 1), np.uint32)
adj[:, 0] = R
adj[:, 1:] = rng.integers(0, n, (n + 1, R), dtype=np.uint32)
p = da.Provider(da.F32, da.L2, dim, n, R, data[:1])
p.set_elements(0, data); p.upload_graph(adj)
nq = 10000
q = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
for L in (32, 64, 128):
    for bits in (0, 11, 12, 13):
        p.set_visited_bits(bits)
        try:
            p.search(da.Knn(L), q[:64], 10)
        except Exception as e:
            print("L", L, "bits", bits, "ERR", str(e)[:80]); continue
        p.kernel_time_reset()
        for _ in range(3):
            ids, d, st = p.search(da.Knn(L), q, 10)
        ms, k = p.kernel_time(0)
        ms /= k
        cm, hp = st["cmps"].mean(), st["hops"].mean()
        byts = st["cmps"].sum() * 512.0 + st["hops"].sum() * 132.0
        print(f"L={L} bits={bits} kernel {ms:.3f} ms  QPS {nq/ms*1e3:,.0f}  cmps {cm:.0f} hops {hp:.0f}  {byts/ms/1e6:.1f} GB/s")
# pure gather
p.set_visited_bits(0)
for per in (256, 2048):
    nqq = 4096
    ids = rng.integers(0, n, nqq * per, dtype=np.uint32)
    off = (np.arange(nqq + 1) * per).astype(np.uint64)
    p.expand_beam_batch(q[:nqq], ids, off)
    p.kernel_time_reset()
    for _ in range(3):
        p.expand_beam_batch(q[:nqq], ids, off)
    ms, k = p.kernel_time(1); ms /= k
    print(f"gather {nqq}x{per}: {ms:.3f} ms  {ids.size*512/ms/1e6:.1f} GB/s")
