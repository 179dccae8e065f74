#!/bin/bash
# round 2: the measurements the docs quote, in one GPU call. Everything lands in gpurun_out/r02final/ (copied to profiles/r02/).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r02final; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -1 $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -8 $O/pytest_gpu.log | grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl" | cut -c1-300
for i in 1 2 3; do SGK_BENCH_TRACE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags_$i.log 2> $O/bench_driver_flags_$i.err; tail -1 $O/bench_driver_flags_$i.log | cut -c1-200; grep "bench trace" $O/bench_driver_flags_$i.err | head -1; done
timeout 900 python bench.py --gpus 1 --steps 200 --warmup 20 > $O/bench_200.log 2>&1; tail -1 $O/bench_200.log | cut -c1-300
# round 1's step definition (one lockstep step per bench step) at the driver's flags: latency-shaped, kept for the record
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --lockstep-per-step 1 --no-cpu-baseline > $O/bench_driver_flags_one_lockstep_per_step.log 2>&1; tail -1 $O/bench_driver_flags_one_lockstep_per_step.log | cut -c1-200
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --path launch --no-cpu-baseline > $O/bench_path_launch.log 2>&1; tail -1 $O/bench_path_launch.log | cut -c1-300
# the multi-rank control flow end to end on the one GPU of the box: two ranks on GPU 0, gloo collectives (RCCL needs two GPUs)
SGK_BENCH_BACKEND=gloo SGK_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --no-fused > $O/bench_2rank_one_gpu_gloo.log 2>&1; tail -1 $O/bench_2rank_one_gpu_gloo.log | cut -c1-400
timeout 900 python tools/bench_configs.py > $O/configs_1_to_5.log 2>&1
timeout 900 python tools/bench_stream.py --envs BoatRace-v0,IslandNavigation-v0,SideEffectsSokoban-v0,DistributionalShift-v0,WhiskyGold-v0,AbsentSupervisor-v0,SafeInterruptibility-v0,ConveyorBelt-v0,TomatoWatering-v0,FriendFoe-v0 --ring 100 > $O/stream_all_envs.log 2>&1
export SGK_NO_BUILD=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-fused > $O/bench_under_rocprof.log 2>&1
for f in $(find $O/prof -name "*kernel_stats.csv"); do head -6 $f | cut -c1-200; cp $f $O/bench_kernel_stats.csv; done
find $O/prof -name "*.csv" -size +1M -delete
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tabq -- python3 tools/prof_tabq_stepwise.py 262144 graph > $O/prof_tabq.log 2>&1
for f in $(find $O/prof_tabq -name "*kernel_stats.csv"); do cp $f $O/tabq_learn_steps_kernel_stats.csv; done
find $O/prof_tabq -name "*.csv" -size +1M -delete
for mode in stream ring launch; do for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_${mode}_$ctr -- python3 tools/pmc_run.py BoatRace-v0 compact 1048576 $mode > $O/pmc_${mode}_$ctr.log 2>&1
  python tools/pmc_summary.py $O/pmc_${mode}_$ctr > $O/pmc_${mode}_${ctr}_summary.json
  find $O/pmc_${mode}_$ctr -name "*.csv" -size +1M -delete
done; done
# the per-GPU shares of the 1 M batch at 2 / 4 / 8 GPUs, so that roofline.traffic is filled at every N
for n in 524288 262144 131072; do for mode in stream launch; do for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_${mode}_n${n}_$ctr -- python3 tools/pmc_run.py BoatRace-v0 compact $n $mode > $O/pmc_${mode}_n${n}_$ctr.log 2>&1
  python tools/pmc_summary.py $O/pmc_${mode}_n${n}_$ctr > $O/pmc_${mode}_n${n}_${ctr}_summary.json
  find $O/pmc_${mode}_n${n}_$ctr -name "*.csv" -size +1M -delete
done; done; done
python tools/make_traffic_json.py $O 1048576 524288 262144 131072
