#!/usr/bin/env python3
"""Static instruction counts per basic block of one kernel in build/asm/*.s (make -C raytracingweekend.jl_amd/csrc asm).
usage: python tools/asm_blocks.py <kernel name substring> [min instructions per block to list]"""
import re, sys
s = open('/root/repo/build/asm/rtw_launch-hip-amdgcn-amd-amdhsa-gfx950.s').read()
pat = sys.argv[1]; thr = int(sys.argv[2]) if len(sys.argv) > 2 else 25
m = [x for x in re.finditer(r'^(_Z\S*):\s*(;.*)?$', s, re.M) if pat in x.group(1)][0]
print(m.group(1))
start = m.end(); end = s.index('.Lfunc_end', start)
blocks = []; cur = {'name': 'entry', 'valu': 0, 'salu': 0, 'lds': 0, 'vmem': 0, 'mfma': 0, 'notes': {}}; blocks.append(cur)
for ln in s[start:end].split('\n'):
    t = ln.strip()
    mm = re.match(r'^(\.LBB\d+_\d+):', t)
    if mm:
        cur = {'name': mm.group(1), 'valu': 0, 'salu': 0, 'lds': 0, 'vmem': 0, 'mfma': 0, 'notes': {}}; blocks.append(cur); continue
    if not t or t.startswith(';') or t.startswith('.'): continue
    op = t.split()[0]
    if op.startswith('v_mfma'): cur['mfma'] += 1
    elif op.startswith('v_'):
        cur['valu'] += 1
        for key in ('v_sqrt', 'v_rcp', 'v_div_scale', 'v_rsq', 'v_alignbit', 'v_readlane', 'v_writelane', 'v_cvt_f64', 'v_mul_f64', 'v_fma_f64', 'v_permlane', 'v_mbcnt'):
            if op.startswith(key): cur['notes'][key] = cur['notes'].get(key, 0) + 1
    elif op.startswith('s_'):
        cur['salu'] += 1
        if op.startswith(('s_cbranch', 's_branch')): cur['notes'][t.replace('\t', ' ')] = 1
    elif op.startswith('ds_'): cur['lds'] += 1
    elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): cur['vmem'] += 1
print('total static VALU', sum(b['valu'] for b in blocks), 'SALU', sum(b['salu'] for b in blocks), 'blocks', len(blocks))
for b in blocks:
    if b['valu'] + b['lds'] + b['mfma'] + b['salu'] >= thr:
        print(f"{b['name']:12s} valu {b['valu']:4d} salu {b['salu']:4d} lds {b['lds']:3d} vmem {b['vmem']:2d} mfma {b['mfma']:2d} {b['notes']}")
