#!/bin/bash
# The FETCH_SIZE / WRITE_SIZE passes behind profiles/traffic.json alone (gpu_final.sh <tag> pmc runs them after everything else):
# bash tools/gpu_traffic.sh <round tag>. 24 profiler runs, ~15 minutes; summaries land in gpurun_out/<tag>_final/ -- copy them to
# profiles/<tag>/ and run `python tools/make_traffic_json.py profiles/<tag> 1048576 524288 262144 131072` there.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp SGK_NO_BUILD=1
T=${1:-r05}; O=gpurun_out/${T}_final; mkdir -p $O
for n in 1048576 524288 262144 131072; do for mode in ring stream launch; do for ctr in FETCH_SIZE WRITE_SIZE; do
  tag=$([ $n = 1048576 ] && echo "" || echo "_n$n")
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_tmp -- python3 tools/pmc_run.py BoatRace-v0 compact $n $mode > $O/pmc_${mode}${tag}_$ctr.log 2>&1
  python tools/pmc_summary.py $O/pmc_tmp > $O/pmc_${mode}${tag}_${ctr}_summary.json
  rm -rf $O/pmc_tmp
done; done; done
python tools/make_traffic_json.py $O 1048576 524288 262144 131072
