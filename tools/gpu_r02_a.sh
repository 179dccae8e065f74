#!/bin/bash
# round 2, first GPU call: new BASELINE-size parity tests, the whole GPU suite, the bench at the driver's flags and at the
# long flags, and the rocprof kernel trace of the long run. Logs under gpurun_out/r02a/.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r02a; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
timeout 1500 python -m pytest tests/test_gpu_baseline_sizes.py -m gpu -q --timeout=900 -x > $O/pytest_sizes.log 2>&1; echo "rc=$?" >> $O/pytest_sizes.log; tail -15 $O/pytest_sizes.log
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 --deselect tests/test_gpu_baseline_sizes.py > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -8 $O/pytest_gpu.log
for i in 1 2 3; do timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20_$i.log 2>&1; tail -1 $O/bench_20_$i.log | cut -c1-1200; done
timeout 600 python bench.py --gpus 1 --steps 2000 --warmup 200 > $O/bench_2000.log 2>&1; tail -1 $O/bench_2000.log | cut -c1-2400
export SGK_NO_BUILD=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 2000 --warmup 200 --no-cpu-baseline > $O/bench_prof.log 2>&1
tail -1 $O/bench_prof.log | cut -c1-600
for f in $(find $O/prof -name "*kernel_stats.csv"); do head -8 $f; cp $f $O/bench_kernel_stats.csv; done
find $O/prof -name "*.csv" -size +1M -delete
