import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import diskann_amd as da
rng = np.random.default_rng(0)
n, R = 1_000_000, 32
for dtype, dim, npd in ((da.F32,128,np.float32),(da.F16,128,np.float16),(da.F16,256,np.float16),(da.U8,128,np.uint8),(da.U8,512,np.uint8)):
    data = (rng.random((n, dim))*100).astype(npd)
    p = da.Provider(dtype, da.L2, dim, n, R, data[:1]); p.set_elements(0, data)
    nqq, per = 4096, 2048
    q = data[:nqq]
    ids = rng.integers(0, n, nqq * per, dtype=np.uint32)
    off = (np.arange(nqq + 1) * per).astype(np.uint64)
    p.expand_beam_batch(q, ids, off); p.kernel_time_reset()
    for _ in range(3): p.expand_beam_batch(q, ids, off)
    ms, k = p.kernel_time(1); ms /= k
    rb = data.shape[1]*data.itemsize
    print(f"gather dtype={dtype} dim={dim} row={rb}B: {ms:.3f} ms  {ids.size*rb/ms/1e6:.0f} GB/s  {ids.size/ms/1e6:.2f} Grows/s", flush=True)
    p.close()
