#!/bin/bash
# VGPR / SGPR / occupancy of every kernel of one source file (default sgk_step.hip), optionally filtered by a substring;
# extra compiler flags through EXTRA (e.g. EXTRA=-DSGK_STREAM_MIN_WAVES=5)
cd "$(dirname "$0")/../safe-grid-agents_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $EXTRA -Rpass-analysis=kernel-resource-usage -c "${1:-sgk_step.hip}" -o /dev/null 2>&1 | python3 -c "
import sys,re,subprocess
flt=sys.argv[1] if len(sys.argv)>1 else ''
cur=None;rows={}
for l in sys.stdin:
    if 'error' in l: print(l, end='')
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); rows[cur]={}
    for k in ('TotalSGPRs','VGPRs:','Occupancy','ScratchSize','LDS Size','SGPRs Spill'):
        if k in l and cur: rows[cur][k]=l.split(':')[-1].split('[')[0].strip()
print('kernel'.ljust(78),'VGPR SGPR occ scratch sgpr-spill')
for k,v in rows.items():
    name=subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.strip()[:76]
    if flt in name: print(name.ljust(78),v.get('VGPRs:'),v.get('TotalSGPRs'),v.get('Occupancy'),v.get('ScratchSize'),v.get('SGPRs Spill'))
" "$2"
