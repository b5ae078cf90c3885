import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import diskann_amd as da
sys.argv=['x']; import bench
n, dim, nq = 1000000, 128, 100000
dev=torch.device('cuda',0)
base, q = bench.make_data(torch, dev, n, dim, nq, 'sift_like', 0xD15CA11, 0xD15CA12)
b=base.cpu().numpy(); qq=q.cpu().numpy()
gt = bench.ground_truth(torch, base, q, 10)
mean = base.double().mean(0).float(); medoid=int(torch.argmin(((base-mean[None,:])**2).sum(1)).item())
p=da.Provider(da.F32,da.L2,dim,n,32,b[medoid:medoid+1]); p.set_elements(0,b)
p.build(da.build_config(28,32,100,intra_batch_candidates=da.IBC_NONE),0,n,0.05,16384)
def t(nqs, L, W, reps):
    p.search(da.Knn(L,W),qq[:nqs],10); p.search(da.Knn(L,W),qq[:nqs],10)
    p.kernel_time_reset()
    for _ in range(reps): ids,_,st=p.search(da.Knn(L,W),qq[:nqs],10)
    ms,c=p.kernel_time(0); return ms/c, ids, st
for W in (1,2,4,8):
    chosen=None
    for L in (10,12,14,16,18,20,22,24,26,28,30,32,36,40,48,64):
        ids,_,st=p.search(da.Knn(L,W),qq,10)
        r=bench.recall_at_k(ids,gt,10)
        if r>=0.95: chosen=L; break
    ms,ids,st=t(nq,chosen,W,3)
    ms1k,_,_=t(1024,chosen,W,20)
    ms1,_,_=t(1,chosen,W,50)
    ms64,ids64,_=t(1,64,W,50)
    print(f"W={W}: L={chosen} recall {r:.4f} cmps {st['cmps'].mean():.0f} hops {st['hops'].mean():.0f}: 100k: {ms:.3f} ms ({nq/ms*1e3/1e6:.2f}M QPS)  1024: {ms1k*1e3:.0f} us ({1024/ms1k*1e3/1e6:.2f}M QPS)  single@L: {ms1*1e3:.0f} us  single@L=64: {ms64*1e3:.0f} us", flush=True)
