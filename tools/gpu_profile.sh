#!/bin/bash
# rocprofv3 kernel-trace of the bench + layout / size sweep. Logs under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
for layout in compact pitched; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/$layout -- python3 bench.py --steps 500 --warmup 100 --no-cpu-baseline --layout $layout > gpurun_out/prof_bench_$layout.log 2>&1
  echo "rc=$?" >> gpurun_out/prof_bench_$layout.log
  tail -2 gpurun_out/prof_bench_$layout.log
done
find gpurun_out/prof -name "*stats*" | head
for f in $(find gpurun_out/prof -name "*kernel_stats.csv"); do echo "== $f"; head -12 $f; done
python tools/sweep.py > gpurun_out/sweep.log 2>&1; tail -40 gpurun_out/sweep.log
