import sys,re,subprocess
cur=None; rows={}
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m: cur=m.group(1); rows[cur]={}; continue
    m=re.search(r'remark:\s+([A-Za-z \[\]/]+): (\d+)',line)
    if m and cur: rows[cur][m.group(1).strip()]=int(m.group(2))
    if 'error' in line or 'warning' in line: print(line.rstrip())
for k,v in rows.items():
    name=subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.strip().split('(')[0]
    name=name.replace('void ','')
    print(f"{name[:64]:64s} V={v.get('VGPRs')} A={v.get('AGPRs')} spill={v.get('VGPRs Spill')} occ={v.get('Occupancy [waves/SIMD]')} lds={v.get('LDS Size [bytes/block]')} sgpr={v.get('TotalSGPRs')}")
