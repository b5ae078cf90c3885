"""summarise `hipcc -Rpass-analysis=kernel-resource-usage` output: kernel, VGPRs, AGPRs, scratch, occupancy, SGPRs, LDS"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
blocks = re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
names = [b.split()[0] for b in blocks]
dem = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.strip().split('\n')
for n, b in zip(dem, blocks):
    g = lambda k: (re.search(k + r': (\d+)', b) or [0, -1])[1]
    n = re.sub(r'void dann::\(anonymous namespace\)::|\(dann::SearchArgs\)', '', n)
    sc, oc = g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]')
    print(f"{n[:80]:80s} VGPR {g('VGPRs')} AGPR {g('AGPRs')} scratch {sc} occ {oc} SGPR {g('TotalSGPRs')}")
