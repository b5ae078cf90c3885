import sys, os
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
for L in (26,64,100):
    ids,_,st=p.search(da.Knn(L), qq, 10)
    c=st['cmps']
    est=32*(0.55*(L+1)+12)+1
    print(L, 'mean',c.mean(),'p50',np.quantile(c,.5),'p99',np.quantile(c,.99),'p99.9',np.quantile(c,.999),'max',c.max(),'est',est,'cap(1.25)',1.25*est, 'frac>cap', (c>1.25*est).mean(), 'frac>1.5est', (c>1.5*est).mean())
