#!/bin/bash
# round 3: tabq_learn_kernel's line waste. What does the fabric fetch for the scattered 32-byte row read --
# 32 / 64 / 128-byte requests -- and does a non-temporal load (build 1) or a non-temporal load + store (build 2) change it?
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp SGK_NO_BUILD=1
O=gpurun_out/r03h; mkdir -p $O
L=$PWD/safe-grid-agents_amd/lib
{
for lib in libsgk.so libsgk_nt1.so libsgk_nt2.so; do
  echo "== $lib"
  export SGK_LIB_PATH=$L/$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 tools/prof_tabq_stepwise.py 262144 calls > $O/tr.log 2>&1
  for f in $(find $O/tr -name "*kernel_stats.csv"); do grep -E "tabq_learn|tabq_act|Name" $f | cut -d, -f1-4 | cut -c1-110; done; rm -rf $O/tr
  for ctr in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_READ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum TCC_HIT_sum"; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/p -- python3 tools/prof_tabq_stepwise.py 262144 calls > $O/p.log 2>&1
    python tools/pmc_summary.py $O/p | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if 'tabq_learn' in k: print('   learn', {c: round(x['avg_per_dispatch']) for c,x in v.items()})
"
    rm -rf $O/p
  done
done
} > $O/exp_tabq_learn_gather.log 2>&1
cat $O/exp_tabq_learn_gather.log
