#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
rm -rf gpurun_out/prof_r01 gpurun_out/pmc_*
export SGK_NO_BUILD=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01 -- python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_prof.log 2>&1
tail -1 gpurun_out/bench_prof.log
for f in $(find gpurun_out/prof_r01 -name "*kernel_stats.csv"); do head -6 $f; done
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc_$ctr -- python3 tools/pmc_run.py BoatRace-v0 compact 1048576 > gpurun_out/pmc_$ctr.log 2>&1
  tail -1 gpurun_out/pmc_$ctr.log
  python tools/pmc_summary.py gpurun_out/pmc_$ctr > gpurun_out/pmc_${ctr}_summary.json; cat gpurun_out/pmc_${ctr}_summary.json
  find gpurun_out/pmc_$ctr -name "*.csv" -size +2M -delete
done
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_plain.log 2>&1; tail -1 gpurun_out/bench_plain.log
