#!/bin/bash
# round 3: GPU suite after the loop restructure + issue peaks + outputs-once kernel timing and SQ counters, all ten envs
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp SGK_NO_BUILD=1
O=gpurun_out/r03g; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/exp_issue_peak.hip -o /tmp/issue_peak && /tmp/issue_peak > $O/issue_peak.log 2>&1; cat $O/issue_peak.log
timeout 900 python tools/bench_stream.py --envs BoatRace-v0,IslandNavigation-v0,SideEffectsSokoban-v0,TomatoWatering-v0,FriendFoe-v0 --sizes 1048576 --k 1000 --modes fused --reps 5 2>&1 | grep "n=" > $O/fused_all.log; cat $O/fused_all.log
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES"
for e in BoatRace-v0 TomatoWatering-v0; do
  tag=$(echo $e | tr 'A-Z' 'a-z' | sed 's/-v0//')
  rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/sq_$tag -- python3 tools/pmc_run.py $e compact 1048576 fused > $O/sq_$tag.log 2>&1
  python tools/pmc_summary.py $O/sq_$tag > $O/pmc_sq_rollout_${tag}_fused.json; rm -rf $O/sq_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_$tag -- python3 tools/pmc_run.py $e compact 1048576 fused > $O/tr_$tag.log 2>&1
  for f in $(find $O/tr_$tag -name "*kernel_stats.csv"); do cp $f $O/fused_kernel_stats_$tag.csv; done; rm -rf $O/tr_$tag
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03g/pmc_sq_rollout_*_fused.json")):
    d=json.load(open(f))
    for k,v in d.items():
        if "rollout_random" in k: print(f.split("/")[-1], {c: round(x["avg_per_dispatch"]) for c,x in v.items()})
PY
head -3 $O/fused_kernel_stats_*.csv
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>$O/bench.err; tail -c 3000 $O/bench.log; tail -5 $O/bench.err
