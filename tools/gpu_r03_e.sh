#!/bin/bash
# round 3, fifth GPU call: the trajectory ring's rate IS address translation (call four: the L2 translation cache is busy 48 % of
# a 100-slice pass and 1 % of a 32-slice pass). Which store pattern / which allocation keeps the translation caches hitting?
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03e; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/exp_write_patterns.hip -o /tmp/wp_probe > $O/wp_build.log 2>&1
WP_LIST=1 /tmp/wp_probe > $O/variants.txt
P1="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_LFIFO_NO_RES_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"
for alloc in malloc vmm; do
  { WP_ALLOC=$alloc timeout 600 /tmp/wp_probe "base slice sc1"; WP_ALLOC=$alloc timeout 600 /tmp/wp_probe "slice ring size"; WP_ALLOC=$alloc timeout 600 /tmp/wp_probe "tile-major sc1";
    WP_ALLOC=$alloc timeout 600 /tmp/wp_probe "tile-major burst sc1"; WP_ALLOC=$alloc timeout 600 /tmp/wp_probe "XCD-contiguous"; WP_ALLOC=$alloc timeout 600 /tmp/wp_probe "tile-major M x burst"; } 2>&1 | grep -v "^fill" > $O/timing_$alloc.log
done
cat $O/timing_vmm.log | head -40
sel() { grep "$1" $O/variants.txt | head -1 | sed 's/^\[ *\([0-9]*\)\].*/\1/'; }
{
for alloc in malloc vmm; do
for pat in "base slice sc1 " "slice ring size M=1 B=1 lay=0 aux=16 wg=8 dly=0 ring=32 " "slice ring size fine M=1 B=1 lay=0 aux=16 wg=8 dly=0 ring=40 " "slice ring size fine M=1 B=1 lay=0 aux=16 wg=8 dly=0 ring=64 " "tile-major sc1 " "tile-major burst sc1 M=1 B=10 " "XCD-contiguous slice M=1 B=1 lay=0 aux=16 wg=8 dly=0 ring=100 " "XCD-contiguous tile-major M=1 B=1 lay=1 aux=16 wg=8 dly=0 ring=100 " "tile-major M x burst sc1 M=4 B=10 " "slice M tiles/wave M=4 "; do
  idx=$(sel "$pat")
  [ -z "$idx" ] && { echo "no variant for: $pat"; continue; }
  WP_ALLOC=$alloc WP_INDEX=$idx rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $O/p_tmp -- /tmp/wp_probe > $O/p_tmp.log 2>&1
  echo "== $alloc [$idx] $pat: $(grep '^\[' $O/p_tmp.log | head -1)"
  python tools/pmc_summary.py $O/p_tmp | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if 'wr<' in k: print('   ', {c: round(x['avg_per_dispatch']) for c,x in v.items()})
"
  rm -rf $O/p_tmp
done
done
} > $O/translation_counters.log 2>&1
cat $O/translation_counters.log
