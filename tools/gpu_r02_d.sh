#!/bin/bash
# round 2, fourth GPU call: GPU suite, the bench with the streamed path as primary (driver flags and long flags), rocprof
# kernel trace, PMC traffic of the stream kernel (own buffers and trajectory ring), write-only bandwidth probe, grid sweeps.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r02d; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -12 $O/pytest_gpu.log | cut -c1-300
for i in 1 2; do SGK_BENCH_TRACE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20_$i.log 2> $O/bench_20_$i.err; tail -1 $O/bench_20_$i.log | cut -c1-400; grep "bench trace" $O/bench_20_$i.err; done
timeout 900 python bench.py --gpus 1 --steps 2000 --warmup 200 > $O/bench_2000.log 2>&1; tail -1 $O/bench_2000.log
timeout 300 python tools/write_bw_probe.py > $O/write_bw.log 2>&1; cat $O/write_bw.log
for g in 1024 1536 2048 3072 4096; do echo "SGK_STREAM_GRID=$g"; SGK_STREAM_GRID=$g timeout 300 python tools/bench_stream.py --envs BoatRace-v0 --sizes 131072,1048576 --ring 100 2>&1 | grep n=; done > $O/stream_grid_sweep.log 2>&1; cat $O/stream_grid_sweep.log
for g in 1024 1536 2048 3072; do echo "SGK_MAX_GRID=$g"; SGK_MAX_GRID=$g timeout 300 python tools/bench_stream.py --envs BoatRace-v0 --sizes 65536,1048576 2>&1 | grep n= | cut -c1-60; done > $O/step_grid_sweep.log 2>&1; cat $O/step_grid_sweep.log
export SGK_NO_BUILD=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 2000 --warmup 200 --no-cpu-baseline > $O/bench_prof.log 2>&1
for f in $(find $O/prof -name "*kernel_stats.csv"); do head -8 $f; cp $f $O/bench_kernel_stats.csv; done
find $O/prof -name "*.csv" -size +1M -delete
for mode in stream ring launch; do for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_${mode}_$ctr -- python3 tools/pmc_run.py BoatRace-v0 compact 1048576 $mode > $O/pmc_${mode}_$ctr.log 2>&1
  python tools/pmc_summary.py $O/pmc_${mode}_$ctr > $O/pmc_${mode}_${ctr}_summary.json; grep -A3 "rollout_random_kernel\|step_kernel\|reset_kernel" $O/pmc_${mode}_${ctr}_summary.json | grep -v "^--" | tr -d '\n '; echo
  find $O/pmc_${mode}_$ctr -name "*.csv" -size +1M -delete
done; done
