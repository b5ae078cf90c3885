import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import diskann_amd as da
sys.argv=['x']; import bench
n, dim, nq = 1000000, 128, 100000
dev=torch.device('cuda',0)
base, q = bench.make_data(torch, dev, n, dim, nq, 'sift_like', 0xD15CA11, 0xD15CA12)
b=base.cpu().numpy(); qq=q.cpu().numpy()
mean = base.double().mean(0).float(); medoid=int(torch.argmin(((base-mean[None,:])**2).sum(1)).item())
p=da.Provider(da.F32,da.L2,dim,n,32,b[medoid:medoid+1]); p.set_elements(0,b)
p.build(da.build_config(28,32,100,intra_batch_candidates=da.IBC_NONE),0,n,0.02,16384)
for L in (10,16,26,40,64,100,160,250):
    p.set_visited_bits(0)
    r=[]
    for it in range(4):
        p.kernel_time_reset()
        p.search(da.Knn(L), qq, 10)
        ms,launches=p.kernel_time(0); rms,rq=p.kernel_time(4)
        r.append((round(ms/launches,3), rq))
    print('L',L,'auto: call1(prior), call2.. (calibrated):',r,flush=True)
