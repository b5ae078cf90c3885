import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import diskann_amd as da, oracle
sys.argv=['x']; import bench
n, dim, nq = 100000, 128, 2000
dev=torch.device('cuda',0)
base, q = bench.make_data(torch, dev, n, dim, nq, 'sift_like', 1, 2)
gt = bench.ground_truth(torch, base, q, 10)
b=base.cpu().numpy(); qq=q.cpu().numpy()
mean = base.double().mean(0).float(); medoid=int(torch.argmin(((base-mean[None,:])**2).sum(1)).item())
def recalls(p, tag):
    out=[]
    for L in (10,20,32,64,128):
        ids,_,st=p.search(da.Knn(L), qq, 10)
        out.append((L, round(bench.recall_at_k(ids, gt, 10),4), int(st['cmps'].mean())))
    print(tag, out, flush=True)
for growth,maxb,ibc,tag in ((0.02,16384,da.IBC_NONE,'gpu batch g=.02'),(0.005,1024,da.IBC_NONE,'gpu batch g=.005 mb=1024'),(0.02,16384,32,'gpu batch ibc=32')):
    p=da.Provider(da.F32,da.L2,dim,n,32,b[medoid:medoid+1]); p.set_elements(0,b)
    t=time.time(); p.build(da.build_config(28,32,100,intra_batch_candidates=ibc),0,n,growth,maxb); 
    recalls(p, f"{tag} ({time.time()-t:.1f}s)")
# oracle sequential single insert (reference benchmark's build mode)
oix=oracle.Index(oracle.F32,oracle.L2,dim,n,32,b[medoid:medoid+1]); oix.set_rows(0,b)
cfg=oracle.build_config(28,32,100)
t=time.time()
for i in range(n): oix.insert(cfg,i)
print("oracle single-insert build %.1fs"%(time.time()-t), flush=True)
p=da.Provider(da.F32,da.L2,dim,n,32,b[medoid:medoid+1]); p.set_elements(0,b); p.upload_graph(oix.adj)
recalls(p,'oracle single-insert graph')
