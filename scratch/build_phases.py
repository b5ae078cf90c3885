import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import diskann_amd as da
sys.argv=['x']; import bench
n, dim = 1000000, 128
dev=torch.device('cuda',0)
base, q = bench.make_data(torch, dev, n, dim, 1000, 'sift_like', 0xD15CA11, 0xD15CA12)
b=base.cpu().numpy()
mean = base.double().mean(0).float(); medoid=int(torch.argmin(((base-mean[None,:])**2).sum(1)).item())
for mb in (16384,):
    p=da.Provider(da.F32,da.L2,dim,n,32,b[medoid:medoid+1]); p.set_elements(0,b)
    p.kernel_time_reset(); torch.cuda.synchronize(); t=time.time()
    nb=p.build(da.build_config(28,32,100,intra_batch_candidates=da.IBC_NONE),0,n,0.05,mb)
    torch.cuda.synchronize(); dt=time.time()-t
    ks=[p.kernel_time(i) for i in range(4)]
    print(f"max_batch {mb}: build {dt:.3f}s batches {nb}; search {ks[0][0]:.0f} ms ({ks[0][1]}), gather {ks[1][0]:.0f}, prune {ks[2][0]:.0f} ms ({ks[2][1]}), backedge {ks[3][0]:.0f} ms ({ks[3][1]}); other {dt*1e3-sum(k[0] for k in ks):.0f} ms", flush=True)
    del p
