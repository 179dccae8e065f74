#!/bin/bash
# Lists the kernels whose code reads the AQL dispatch packet (.amdhsa_user_sgpr_dispatch_ptr 1): on MI355X that is a scalar load
# from host-visible queue memory and cost a flat 13-25 us per launch where it happened (a local array indexed at run time,
# promoted to LDS). Expected output: nothing.
cd "$(dirname "$0")/../safe-grid-agents_amd/csrc"
T=$(mktemp -d)
for f in sgk_step sgk_tabq sgk_policy sgk_learn sgk_api sgk_comm; do
  (cd $T && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -save-temps -c $OLDPWD/$f.hip -o /dev/null 2>/dev/null
   python3 - $f <<'PY'
import re, sys, subprocess
f = sys.argv[1]
s = open(f + '-hip-amdgcn-amd-amdhsa-gfx950.s').read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    d = re.search(r'\.amdhsa_user_sgpr_dispatch_ptr (\d)', m.group(2))
    if d and d.group(1) == '1':
        print(f, subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()[:110])
PY
  )
done
rm -rf $T
