import sys, os, subprocess, ctypes as C
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R)
src=os.path.join(R,'diskann_amd','csrc'); out='/tmp/libdann_prof.so'
cmd=['/opt/rocm/bin/hipcc','--offload-arch=gfx950','-O3','-std=c++17','-fPIC','-ffp-contract=off','-fno-gpu-flush-denormals-to-zero','-DDANN_PHASE_CYCLES','-shared','-o',out]+[os.path.join(src,f) for f in ('api.hip','search_kernels.hip','search_f32.hip','search_f16.hip','search_u8.hip','search_i8.hip','search_sq8.hip','search_pq.hip','distance_kernels.hip','build_kernels.hip','pq_kernels.hip')]
subprocess.check_call(cmd)
import diskann_amd._ffi as ffi
ffi.LIB_PATH=out
import numpy as np, torch, diskann_amd as da
lib=ffi.lib(); lib.dann_debug_phase_cycles.argtypes=[C.c_void_p,C.c_int]
n,dim,R_=200000,128,32
g=torch.Generator(device='cuda'); g.manual_seed(1)
centers=torch.rand((256,dim),generator=g,device='cuda'); basis=torch.randn((16,dim),generator=g,device='cuda')/4
def draw(m):
    lab=torch.randint(0,256,(m,),generator=g,device='cuda'); z=torch.randn((m,16),generator=g,device='cuda'); e=torch.randn((m,dim),generator=g,device='cuda')
    return (centers[lab]+0.25*(z@basis)+0.02*e).contiguous()
base=draw(n); q=draw(50000)
p32=da.Provider(da.F32,da.L2,dim,n,R_,base[:1].cpu().numpy()); p32.set_elements(0,base.cpu().numpy())
p32.build(da.build_config(28,32,64,intra_batch_candidates=da.IBC_NONE),0,n,0.02,16384)
adj=p32.download_graph()
b16=base.half().cpu().numpy(); q16=q.half().cpu().numpy()
p16=da.Provider(da.F16,da.L2,dim,n,R_,b16[:1]); p16.set_elements(0,b16); p16.upload_graph(adj)
for name,p,qq in (("f32",p32,q.cpu().numpy()),("f16",p16,q16)):
    for nq in (64,50000):
        p.search(da.Knn(32),qq[:nq],10); lib.dann_debug_phase_cycles(None,1); p.kernel_time_reset()
        ids,d,st=p.search(da.Knn(32),qq[:nq],10)
        buf=(C.c_ulonglong*8)(); lib.dann_debug_phase_cycles(buf,0)
        hops=st['hops'].sum(); v=[buf[i]/hops for i in range(5)]; ms,k=p.kernel_time(0)
        print(f"{name} nq={nq} kernel {ms/k:.3f} ms hops/q={hops/nq:.0f} cmps/q={st['cmps'].mean():.0f} cyc/hop: pop {v[0]:.0f} adj+hash {v[1]:.0f} gather {v[2]:.0f} merge {v[3]:.0f} total {v[4]:.0f}")
