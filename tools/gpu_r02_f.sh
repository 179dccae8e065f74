#!/bin/bash
# round 2, sixth GPU call: kernel breakdown of the graph-replayed tabular-Q drop-in sequence; bench at the driver's flags after
# the pinned-metrics allocation moved to create; C example test
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r02f; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -1 $O/smoke.log
timeout 1200 python -m pytest tests/test_c_abi_example.py tests/test_gpu_parity.py -m gpu -q --timeout=900 -x > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log | cut -c1-300
for i in 1 2 3; do SGK_BENCH_TRACE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20_$i.log 2> $O/bench_20_$i.err; tail -1 $O/bench_20_$i.log | cut -c1-200; grep "bench trace" $O/bench_20_$i.err | head -1; done
export SGK_NO_BUILD=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tabq -- python3 tools/prof_tabq_stepwise.py 262144 graph > $O/prof_tabq.log 2>&1
for f in $(find $O/prof_tabq -name "*kernel_stats.csv"); do head -8 $f | cut -c1-200; cp $f $O/tabq_learn_steps_kernel_stats.csv; done
find $O/prof_tabq -name "*.csv" -size +1M -delete
