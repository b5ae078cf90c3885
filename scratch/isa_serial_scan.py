#!/usr/bin/env python3
"""Scan hipcc -S output for wide global loads that are waited on immediately (s_waitcnt vmcnt(0) within two
instructions): the signature of a load compiled under a per-lane branch, which leaves one request in flight.
Usage: isa_serial_scan.py file.s [...]   (prints kernels with >= 2 such loads in a row)"""
import re, sys
for fn in sys.argv[1:]:
    name = None; k = []
    kernels = []
    for line in open(fn):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            name = m.group(1); k = []
            continue
        if name is not None:
            k.append(line)
            if line.strip().startswith('s_endpgm'):
                kernels.append((name, k)); name = None
    for name, k in kernels:
        ins = [l.strip() for l in k if l.startswith('\t') and not l.strip().startswith((';', '.'))]
        runs = []; run = 0; loads = 0
        i = 0
        last_ser = -100
        for i, l in enumerate(ins):
            if re.match(r'(global|buffer|flat)_load_(dword|ushort|ubyte|short|sbyte)', l):
                loads += 1
                if any(x.startswith('s_waitcnt vmcnt(0)') for x in ins[i+1:i+3]):
                    if i - last_ser < 14: run += 1
                    else:
                        if run >= 3: runs.append(run)
                        run = 1
                    last_ser = i
        if run >= 3: runs.append(run)
        if runs:
            short = re.sub(r'^_ZN4dann12_GLOBAL__N_1\d+', '', name)[:90]
            print(f"{fn.split('/')[-1]:22s} {short:90s} wide_loads={loads:4d} serial_runs={runs}")
