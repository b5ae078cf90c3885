import sys, os, subprocess, ctypes as C
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R)
# build a profiling variant of the library
src=os.path.join(R,'diskann_amd','csrc'); out='/tmp/libdann_prof.so'
cmd=['/opt/rocm/bin/hipcc','--offload-arch=gfx950','-O3','-std=c++17','-fPIC','-ffp-contract=off','-fno-gpu-flush-denormals-to-zero','-DDANN_PHASE_CYCLES','-shared','-o',out]+[os.path.join(src,f) for f in ('api.hip','search_kernels.hip','distance_kernels.hip','build_kernels.hip')]
subprocess.check_call(cmd)
import diskann_amd._ffi as ffi
ffi.LIB_PATH=out
import numpy as np, torch, diskann_amd as da
lib=ffi.lib()
rng=np.random.default_rng(0)
n,dim,R_=200000,128,32
g=torch.Generator(device='cuda'); g.manual_seed(1)
centers=torch.rand((256,dim),generator=g,device='cuda'); basis=torch.randn((16,dim),generator=g,device='cuda')/4
def draw(m):
    lab=torch.randint(0,256,(m,),generator=g,device='cuda'); z=torch.randn((m,16),generator=g,device='cuda'); e=torch.randn((m,dim),generator=g,device='cuda')
    return (centers[lab]+0.25*(z@basis)+0.02*e).contiguous()
base=draw(n); q=draw(10000)
p=da.Provider(da.F32,da.L2,dim,n,R_,base[:1].cpu().numpy()); p.set_elements(0,base.cpu().numpy())
p.build(da.build_config(28,32,100,intra_batch_candidates=da.IBC_NONE),0,n,0.02,16384)
lib.dann_debug_phase_cycles.argtypes=[C.c_void_p,C.c_int]
for nq in (64,10000):
    for L in (32,64):
        qq=q[:nq].cpu().numpy()
        p.search(da.Knn(L),qq,10)
        lib.dann_debug_phase_cycles(None,1)
        ids,d,st=p.search(da.Knn(L),qq,10)
        buf=(C.c_ulonglong*8)(); lib.dann_debug_phase_cycles(buf,0)
        hops=st['hops'].sum(); v=[buf[i]/hops for i in range(5)]
        ms,k=p.kernel_time(0)
        print(f"prefetch hits {buf[5]} misses {buf[6]}")
        print(f"nq={nq} L={L} hops/q={hops/nq:.0f} cmps/q={st['cmps'].mean():.0f} cycles/hop: pop {v[0]:.0f} adj+hash {v[1]:.0f} gather {v[2]:.0f} merge {v[3]:.0f} total {v[4]:.0f}")

for nq in (10000, 20000, 50000, 100000):
    qq=draw(nq).cpu().numpy()
    p.search(da.Knn(32),qq,10); p.kernel_time_reset()
    for _ in range(3): ids,d,st=p.search(da.Knn(32),qq,10)
    ms,k=p.kernel_time(0); ms/=k
    print(f"nq={nq} L=32 kernel {ms:.3f} ms  {nq/ms*1e3:,.0f} QPS  {(st['cmps'].sum()*512+st['hops'].sum()*132)/ms/1e6:.0f} GB/s")
