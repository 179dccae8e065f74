#!/bin/bash
# round 3, fourth GPU call: counters that separate the explanations of the trajectory ring's write rate (translation misses vs
# DRAM credits vs too many write requests in flight), on the pure store probe and on the streamed rollout kernel; and the
# streamed kernel morphed towards the probe (no transition arithmetic / records as 16-byte write-through stores)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03d; mkdir -p $O
export SGK_NO_BUILD=1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/exp_write_patterns.hip -o /tmp/wp_probe > $O/wp_build.log 2>&1
{ timeout 300 /tmp/wp_probe "base slice sc1"; timeout 300 /tmp/wp_probe "dword"; } 2>&1 | grep -v "^fill" > $O/write_patterns_dword_records.log
cat $O/write_patterns_dword_records.log
L=$PWD/safe-grid-agents_amd/lib
{
for lib in libsgk.so libsgk_rec16.so libsgk_nocompute.so; do
  SGK_LIB_PATH=$L/$lib timeout 300 python tools/exp_stream_grid.py --rings 32,100 2>&1 | grep "n="
done
SGK_LIB_PATH=$L/libsgk_rec16.so timeout 300 python tools/exp_stream_grid.py --rings 32,100 --layout tile 2>&1 | grep "n=" | sed "s/^/tile-major /"
} > $O/stream_morphs.log 2>&1
cat $O/stream_morphs.log
P1="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_LFIFO_NO_RES_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"
P2="TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_sum GRBM_EA_BUSY GRBM_TC_BUSY"
P3="TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_64B_sum TCC_TAG_STALL_sum TCC_BUSY_sum"
P4="TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_LFIFO_FULL_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum"
pass() { # tag, counters, command...
  tag=$1; ctr=$2; shift 2
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/p_$tag -- "$@" > $O/p_$tag.log 2>&1
  python tools/pmc_summary.py $O/p_$tag > $O/pmc_$tag.json
  rm -rf $O/p_$tag
}
i=0
for ctr in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  pass probe_slice_ring100_$i "$ctr" /tmp/wp_probe "base slice sc1"
  WP_RING=32 pass probe_slice_ring32_$i "$ctr" /tmp/wp_probe "slice ring size"
  WP_RING=100 pass probe_xcd_ring100_$i "$ctr" /tmp/wp_probe "XCD-contiguous slice"
  pass kernel_ring100_$i "$ctr" python3 tools/pmc_run.py BoatRace-v0 compact 1048576 ring 100
  pass kernel_ring32_$i "$ctr" python3 tools/pmc_run.py BoatRace-v0 compact 1048576 ring 32
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03d/pmc_*.json")):
    d = json.load(open(f))
    for k, v in d.items():
        if "wr<" in k or "rollout_random" in k:
            print(f.split("/")[-1][4:-5].ljust(28), k[:40].ljust(40), {c: round(x["avg_per_dispatch"]) for c, x in v.items()})
PY
