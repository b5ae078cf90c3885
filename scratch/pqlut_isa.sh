#!/bin/bash
# compile search_pqlut.hip alone: resource usage + per-block instruction statistics of one instantiation
cd /root/repo/diskann_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero -Wno-unused-function"
/opt/rocm/bin/hipcc $F -Rpass-analysis=kernel-resource-usage --cuda-device-only -S search_pqlut.hip -o /tmp/pqlut.s 2> /tmp/pqlut_res.txt
grep -E " error" /tmp/pqlut_res.txt | head
python /root/repo/scratch/kres.py /tmp/pqlut_res.txt
python - "$@" <<'PY'
import re,sys
s=open('/tmp/pqlut.s').read()
name=sys.argv[1] if len(sys.argv)>1 else '_ZN4dann12_GLOBAL__N_116pq_search_kernelILi0ELi2ELb1EEEvNS_10SearchArgsE'
i=s.index(name+':'); j=s.index('.Lfunc_end',i)
body=s[i:j].split('\n')
cur='entry'; stats={}; order=['entry']
for l in body:
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m: cur=m.group(1); order.append(cur)
    st=stats.setdefault(cur,{'n':0,'scr':0,'bperm':0,'vmem':0,'ds':0,'salu':0})
    t=l.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    st['n']+=1
    if 'scratch_' in t: st['scr']+=1
    if 'ds_bpermute' in t: st['bperm']+=1
    if t.startswith('global_') or t.startswith('buffer_'): st['vmem']+=1
    if t.startswith('ds_'): st['ds']+=1
    if t.startswith('s_'): st['salu']+=1
tot=sum(st['n'] for st in stats.values()); print('total',tot,'scratch ops',sum(st['scr'] for st in stats.values()))
for b in order:
    st=stats[b]
    if st['scr'] or st['bperm'] or st['n']>40: print(b,st)
PY
