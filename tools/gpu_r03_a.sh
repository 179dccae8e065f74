#!/bin/bash
# round 3, first GPU call: (1) the write-pattern probe, (2) ring-size sweep of the streamed rollout (timings + WRITE_SIZE),
# (3) SQ counters of the outputs-once rollout kernel and of the streamed kernels
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03a; mkdir -p $O
export SGK_NO_BUILD=1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/exp_write_patterns.hip -o /tmp/wp_probe > $O/wp_build.log 2>&1
timeout 900 /tmp/wp_probe > $O/write_patterns_boat.log 2>&1
tail -5 $O/write_patterns_boat.log
timeout 600 /tmp/wp_probe "" 3072 > $O/write_patterns_island.log 2>&1
timeout 900 python tools/exp_ring_size_sweep.py > $O/ring_size_sweep.log 2>&1
tail -3 $O/ring_size_sweep.log
for s in 1 2 4 8 16 32 64 100; do
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_ring$s -- python3 tools/pmc_run.py BoatRace-v0 compact 1048576 ring $s > $O/pmc_ring$s.log 2>&1
  python tools/pmc_summary.py $O/pmc_ring$s > $O/pmc_ring${s}_WRITE_SIZE_summary.json
  rm -rf $O/pmc_ring$s
done
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"
SQ2="SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_WAVE_CYCLES"
for w in "BoatRace-v0 fused" "TomatoWatering-v0 fused" "TomatoWatering-v0 stream" "BoatRace-v0 stream" "BoatRace-v0 ring"; do
  set -- $w
  tag=$(echo $1 | tr 'A-Z' 'a-z' | sed 's/-v0//')_$2
  rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/sq_$tag -- python3 tools/pmc_run.py $1 compact 1048576 $2 > $O/sq_$tag.log 2>&1
  python tools/pmc_summary.py $O/sq_$tag > $O/pmc_sq_rollout_$tag.json
  rm -rf $O/sq_$tag
  rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $O/sq2_$tag -- python3 tools/pmc_run.py $1 compact 1048576 $2 > $O/sq2_$tag.log 2>&1
  python tools/pmc_summary.py $O/sq2_$tag > $O/pmc_sq2_rollout_$tag.json
  rm -rf $O/sq2_$tag
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_fused -- python3 tools/pmc_run.py BoatRace-v0 compact 1048576 fused > $O/trace_fused.log 2>&1
for f in $(find $O/trace_fused -name "*kernel_stats.csv"); do cp $f $O/fused_kernel_stats.csv; done
rm -rf $O/trace_fused
ls $O
