#!/usr/bin/env python3
"""Offline check of the dispatcher against dispatch_check.py outputs: for every measured line, which family the CURRENT library plans (qqq_w4a8_plan, no GPU needed)
and how far that family's measured time is from the best measured one.  usage: python tools/dispatch_regret.py profiles/r04_dispatch_check_final*.txt"""
import os
sys_path_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import re, sys
sys.path.insert(0, sys_path_root)
from qqq_amd import _lib
files=sys.argv[1:]
fam={1:'stream',2:'tiled',3:'column',4:'panel',5:'wide'}
n=0; bad=0; worst=0
for f in files:
  for l in open(f):
    m=re.match(r'N=\s*(\d+) K=\s*(\d+) (\w+)\s+M=\s*(\d+)\s+auto\(k(\d),ks(\d+).*?\)\s+([\d.]+) \|(.*)',l)
    if not m: continue
    N,K,mode,M=int(m.group(1)),int(m.group(2)),m.group(3),int(m.group(4))
    rest=m.group(8).split('<--')[0].split()
    res={rest[i]:float(rest[i+1]) for i in range(0,len(rest)-1,2)}
    res={k:v for k,v in res.items() if v==v}
    p=_lib.plan(M,N,K,128 if mode=='g128' else -1,16)
    f_=fam[p['kernel']]
    if f_=='panel' and p['bm']==256: f_='panel256x2' if p['pw']==2 else 'panel256'
    if f_=='wide':
        f_={(16,256,1):'wide',(16,256,2):'w16x2',(8,256,1):'w8',(16,128,1):'w128',(16,128,2):'w128x2'}.get((p['mt'],p['bm'],p['ksplit']),'wide')
        if p['glds']==2 and 'walk' in res: f_='walk'
    best=min(res.values()); got=res.get(f_)
    if got is None: continue
    n+=1; reg=got/best-1; worst=max(worst,reg)
    old=(int(m.group(5)),int(m.group(6)))
    if reg>0.03:
        bad+=1
        print(f"N={N:5d} K={K:5d} {mode:4s} M={M:4d} was k{old[0]},ks{old[1]} now k{p['kernel']},ks{p['ksplit']},bm{p['bm']} -> {f_} {got:.1f} best {best:.1f} regret {100*reg:.0f}%")
print(n,'points',bad,'above 3%, worst',round(100*worst,1),'%')
