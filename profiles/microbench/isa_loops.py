#!/usr/bin/env python3
"""Diagnostic: instruction mix of every loop of one kernel in the device assembly (back edges found by label order).
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iquadrotorilqr_amd/csrc -S --cuda-device-only -o capi.s quadrotorilqr_amd/csrc/ilqr_capi.hip
  python profiles/microbench/isa_loops.py _ZN5qilqr11k_rollout16IdEE      (run where capi.s is)"""
import re,sys,collections
name=sys.argv[1]
L=open('capi.s').read().split('\n')
start=[i for i,l in enumerate(L) if l.startswith(name) and l.rstrip().split(';')[0].strip().endswith(':')][0]
end=[i for i in range(start,len(L)) if L[i].startswith('.Lfunc_end')][0]
body=L[start:end]
labels={}
for i,l in enumerate(body):
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m: labels[m.group(1)]=i
def isinst(l):
    l=l.strip()
    return l and not l.startswith(('.',';')) and not l.endswith(':') and not re.match(r'^\.LBB',l)
loops=[]
for i,l in enumerate(body):
    m=re.match(r'\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)',l) or re.match(r'\s+s_branch\s+(\.LBB\d+_\d+)',l)
    if m and m.group(1) in labels and labels[m.group(1)]<i:
        loops.append((labels[m.group(1)],i))
for a,b in loops:
    ins=[x.strip().split()[0] for x in body[a:b+1] if isinst(x)]
    if len(ins)<40: continue
    c=collections.Counter()
    for x in ins:
        if x.startswith('v_') and 'f64' in x: c['v_f64']+=1
        elif x.startswith('v_mov_b32_dpp') or 'dpp' in x: c['dpp']+=1
        elif x.startswith('v_'): c['v_other']+=1
        elif x.startswith('s_waitcnt'): c['waitcnt']+=1
        elif x.startswith('s_nop'): c['nop']+=1
        elif x.startswith('s_'): c['s_']+=1
        elif x.startswith('ds_'): c['ds']+=1
        elif x.startswith(('global_','flat_','buffer_','scratch_')): c[x.split('_')[0]+'_'+x.split('_')[1]]+=1
        else: c[x]+=1
    # dpp detection by operand
    dpp=sum(1 for x in body[a:b+1] if isinst(x) and ('row_newbcast' in x or 'quad_perm' in x))
    print(f"loop lines {a}-{b}: {len(ins)} instr, dpp-operand {dpp}", dict(c))
