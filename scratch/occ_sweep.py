import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import diskann_amd as da
def run(dtype, dim, n=200000, nq=50000, Ls=(26,64)):
    g=torch.Generator(device='cuda'); g.manual_seed(1)
    centers=torch.rand((256,dim),generator=g,device='cuda'); basis=torch.randn((16,dim),generator=g,device='cuda')/4
    def draw(m):
        lab=torch.randint(0,256,(m,),generator=g,device='cuda'); z=torch.randn((m,16),generator=g,device='cuda'); e=torch.randn((m,dim),generator=g,device='cuda')
        return (centers[lab]+0.25*(z@basis)+0.02*e)
    base=draw(n); q=draw(nq)
    if dtype==da.F16: conv=lambda t: t.half().cpu().numpy()
    elif dtype==da.U8: conv=lambda t: (t*160+40).clamp(0,255).round().to(torch.uint8).cpu().numpy()
    else: conv=lambda t: t.cpu().numpy()
    b=conv(base); qq=conv(q)
    p=da.Provider(dtype,da.L2,dim,n,32,b[:1]); p.set_elements(0,b)
    p.build(da.build_config(28,32,64,intra_batch_candidates=da.IBC_NONE),0,n,0.02,16384)
    for L in Ls:
        res=[]
        for ent in [0]+list(range(1024,2432,64)):
            p.set_visited_bits(ent)
            p.search(da.Knn(L),qq,10); p.search(da.Knn(L),qq,10); p.kernel_time_reset()
            for _ in range(3): ids,d,st=p.search(da.Knn(L),qq,10)
            ms,k=p.kernel_time(0); res.append((ent, round(ms/k,3)))
        print(f"dtype={dtype} dim={dim} L={L} cmps {st['cmps'].mean():.0f} p90 {np.quantile(st['cmps'],.9):.0f}:", res, flush=True)
run(da.F16,128,Ls=(26,32,40)); run(da.U8,128,Ls=(32,40))
