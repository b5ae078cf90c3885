import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import diskann_amd as da
def run(dtype, dim, n=200000, nq=50000, L=32):
    g=torch.Generator(device='cuda'); g.manual_seed(1)
    centers=torch.rand((256,dim),generator=g,device='cuda'); basis=torch.randn((16,dim),generator=g,device='cuda')/4
    def draw(m):
        lab=torch.randint(0,256,(m,),generator=g,device='cuda'); z=torch.randn((m,16),generator=g,device='cuda'); e=torch.randn((m,dim),generator=g,device='cuda')
        return (centers[lab]+0.25*(z@basis)+0.02*e)
    base=draw(n); q=draw(nq)
    if dtype==da.F16: conv=lambda t: t.half().cpu().numpy()
    elif dtype==da.U8: conv=lambda t: (t*160+40).clamp(0,255).round().to(torch.uint8).cpu().numpy()
    elif dtype==da.I8: conv=lambda t: (t*100-50).clamp(-128,127).round().to(torch.int8).cpu().numpy()
    else: conv=lambda t: t.cpu().numpy()
    b=conv(base); qq=conv(q)
    p=da.Provider(dtype,da.L2,dim,n,32,b[:1]); p.set_elements(0,b)
    t0=time.time(); p.build(da.build_config(28,32,64,intra_batch_candidates=da.IBC_NONE),0,n,0.02,16384); tb=time.time()-t0
    p.search(da.Knn(L),qq[:1000],10); p.kernel_time_reset()
    for _ in range(3): ids,d,st=p.search(da.Knn(L),qq,10)
    ms,k=p.kernel_time(0); ms/=k
    esz={da.F32:4,da.F16:2,da.U8:1,da.I8:1}[dtype]
    byts=st['cmps'].sum()*dim*esz+st['hops'].sum()*132
    print(f"dtype={dtype} dim={dim}: build {tb:.2f}s  search {ms:.3f} ms  {nq/ms*1e3:,.0f} QPS  {byts/ms/1e6:.0f} GB/s  cmps {st['cmps'].mean():.0f}", flush=True)
for dt,dim in ((da.F32,128),(da.F32,100),(da.F32,384),(da.F32,768),(da.F16,128),(da.F16,768),(da.U8,128),(da.I8,100)):
    run(dt,dim)
