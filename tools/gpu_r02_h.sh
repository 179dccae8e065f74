#!/bin/bash
# round 2: GPU suite after the tabular-Q row hand-off / reset tile skipping / TransitionBoatRace; tabq breakdown again
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r02h; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -1 $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -8 $O/pytest_gpu.log | cut -c1-300
timeout 900 python tools/bench_configs.py > $O/configs.log 2>&1; grep '"config": 3' $O/configs.log | cut -c1-330
export SGK_NO_BUILD=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tabq -- python3 tools/prof_tabq_stepwise.py 262144 graph > $O/prof_tabq.log 2>&1
for f in $(find $O/prof_tabq -name "*kernel_stats.csv"); do head -6 $f | cut -c1-200; cp $f $O/tabq_learn_steps_kernel_stats.csv; done
find $O/prof_tabq -name "*.csv" -size +1M -delete
