import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import diskann_amd as da
rng=np.random.default_rng(0)
n,dim=200000,128
g=torch.Generator(device='cuda'); g.manual_seed(1)
centers=torch.rand((256,dim),generator=g,device='cuda'); basis=torch.randn((16,dim),generator=g,device='cuda')/4
def draw(m):
    lab=torch.randint(0,256,(m,),generator=g,device='cuda'); z=torch.randn((m,16),generator=g,device='cuda'); e=torch.randn((m,dim),generator=g,device='cuda')
    return (centers[lab]+0.25*(z@basis)+0.02*e).cpu().numpy()
b=draw(n); q=draw(4096)
p=da.Provider(da.F32,da.L2,dim,n,32,b[:1]); p.set_elements(0,b)
p.build(da.build_config(28,32,64,intra_batch_candidates=da.IBC_NONE),0,n,0.05,16384)
for nq in (1,2,4,8,16,32,64,128,256,1024,4096):
    p.search(da.Knn(64),q[:nq],10)
    reps=100
    p.kernel_time_reset()
    t=time.perf_counter()
    for _ in range(reps): p.search(da.Knn(64),q[:nq],10)
    dt=(time.perf_counter()-t)/reps
    ms,cnt=p.kernel_time(0); rms,rq=p.kernel_time(4)
    print(f"host API nq={nq}: {dt*1e6:.0f} us per call, kernel {ms/cnt*1e3:.0f} us avg, retry queries/call {rq/reps:.2f}", flush=True)
