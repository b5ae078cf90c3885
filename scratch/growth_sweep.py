import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import diskann_amd as da
sys.argv=['x']; import bench
n, dim, nq = 1000000, 128, 20000
dev=torch.device('cuda',0)
base, q = bench.make_data(torch, dev, n, dim, nq, 'sift_like', 0xD15CA11, 0xD15CA12)
b=base.cpu().numpy(); qq=q.cpu().numpy()
gt = bench.ground_truth(torch, base, q, 10)
mean = base.double().mean(0).float(); medoid=int(torch.argmin(((base-mean[None,:])**2).sum(1)).item())
for growth, mb in ((0.02,16384),(0.05,16384),(0.1,16384),(0.1,65536),(0.2,65536),(0.5,131072)):
    p=da.Provider(da.F32,da.L2,dim,n,32,b[medoid:medoid+1]); p.set_elements(0,b)
    torch.cuda.synchronize(); t=time.time()
    nb=p.build(da.build_config(28,32,100,intra_batch_candidates=da.IBC_NONE),0,n,growth,mb)
    torch.cuda.synchronize(); dt=time.time()-t
    out=[]
    for L in (20,26,32,48):
        ids,_,st=p.search(da.Knn(L),qq,10)
        out.append((L, round(bench.recall_at_k(ids,gt,10),4), int(st['cmps'].mean())))
    adj=p.download_graph(); deg=adj[:n,0].mean()
    print(f"growth {growth} max_batch {mb}: build {dt:.3f}s batches {nb} mean degree {deg:.1f} recall/cmps {out}", flush=True)
    del p
