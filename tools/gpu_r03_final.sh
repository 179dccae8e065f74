#!/bin/bash
# round 3: the measurements the docs quote, in one GPU call. Everything lands in gpurun_out/r03final/ (copied to profiles/r03/).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03final; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -1 $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -8 $O/pytest_gpu.log | grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl" | cut -c1-300
for i in 1 2 3; do SGK_BENCH_TRACE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags_$i.log 2> $O/bench_driver_flags_$i.err; tail -1 $O/bench_driver_flags_$i.log | cut -c1-200; grep "bench trace" $O/bench_driver_flags_$i.err | head -1; done
timeout 900 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_200.log 2>&1; tail -1 $O/bench_200.log | cut -c1-300
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --path own --no-cpu-baseline > $O/bench_path_own.log 2>&1; tail -1 $O/bench_path_own.log | cut -c1-300
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --path launch --no-cpu-baseline > $O/bench_path_launch.log 2>&1; tail -1 $O/bench_path_launch.log | cut -c1-300
# the multi-rank control flow end to end on the one GPU of the box: two ranks on GPU 0, gloo collectives (RCCL needs two GPUs)
SGK_BENCH_BACKEND=gloo SGK_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --no-fused > $O/bench_2rank_one_gpu_gloo.log 2>&1; tail -1 $O/bench_2rank_one_gpu_gloo.log | cut -c1-400
SGK_BENCH_BACKEND=gloo SGK_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus 8 --steps 5 --warmup 2 --no-fused --no-weak-line --no-secondary > $O/bench_8rank_one_gpu_gloo.log 2>&1; tail -1 $O/bench_8rank_one_gpu_gloo.log | cut -c1-400
timeout 900 python tools/bench_configs.py > $O/configs_1_to_5.log 2>&1
timeout 900 python tools/bench_stream.py --envs BoatRace-v0,IslandNavigation-v0,SideEffectsSokoban-v0,DistributionalShift-v0,WhiskyGold-v0,AbsentSupervisor-v0,SafeInterruptibility-v0,ConveyorBelt-v0,TomatoWatering-v0,FriendFoe-v0 --ring 100 > $O/stream_all_envs.log 2>&1
timeout 600 python tools/bench_single_env.py > $O/single_env.log 2>&1; grep -v amdgpu $O/single_env.log
timeout 600 python tools/exp_ring_size_sweep.py > $O/ring_size_sweep.log 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/exp_write_patterns.hip -o /tmp/wp_probe > /dev/null 2>&1
{ /tmp/wp_probe "base slice"; /tmp/wp_probe "slice ring size"; /tmp/wp_probe "tile-major sc1"; /tmp/wp_probe "tile-major burst sc1"; /tmp/wp_probe "slice M tiles"; } 2>&1 | grep -v "^fill" > $O/write_patterns_same_box.log
python tools/write_bw_probe.py > $O/write_only_bandwidth_probe.log 2>&1
export SGK_NO_BUILD=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-fused > $O/bench_under_rocprof.log 2>&1
for f in $(find $O/prof -name "*kernel_stats.csv"); do head -6 $f | cut -c1-200; cp $f $O/bench_kernel_stats.csv; done
find $O/prof -name "*.csv" -size +1M -delete
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tabq -- python3 tools/prof_tabq_stepwise.py 262144 graph > $O/prof_tabq.log 2>&1
for f in $(find $O/prof_tabq -name "*kernel_stats.csv"); do cp $f $O/tabq_learn_steps_kernel_stats.csv; done
find $O/prof_tabq -name "*.csv" -size +1M -delete
for mode in ring stream launch; do for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_${mode}_$ctr -- python3 tools/pmc_run.py BoatRace-v0 compact 1048576 $mode > $O/pmc_${mode}_$ctr.log 2>&1
  python tools/pmc_summary.py $O/pmc_${mode}_$ctr > $O/pmc_${mode}_${ctr}_summary.json
  rm -rf $O/pmc_${mode}_$ctr
done; done
# the per-GPU shares of the 1 M batch at 2 / 4 / 8 GPUs, so that roofline.traffic is filled at every N
for n in 524288 262144 131072; do for mode in ring stream launch; do for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_${mode}_n${n}_$ctr -- python3 tools/pmc_run.py BoatRace-v0 compact $n $mode > $O/pmc_${mode}_n${n}_$ctr.log 2>&1
  python tools/pmc_summary.py $O/pmc_${mode}_n${n}_$ctr > $O/pmc_${mode}_n${n}_${ctr}_summary.json
  rm -rf $O/pmc_${mode}_n${n}_$ctr
done; done; done
python tools/make_traffic_json.py $O 1048576 524288 262144 131072
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES"
for e in BoatRace-v0 TomatoWatering-v0 IslandNavigation-v0 SideEffectsSokoban-v0; do
  tag=$(echo $e | tr 'A-Z' 'a-z' | sed 's/-v0//')
  rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/sq_$tag -- python3 tools/pmc_run.py $e compact 1048576 fused > $O/sq_$tag.log 2>&1
  python tools/pmc_summary.py $O/sq_$tag > $O/pmc_sq_rollout_${tag}_fused.json; rm -rf $O/sq_$tag
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/exp_issue_peak.hip -o /tmp/issue_peak && /tmp/issue_peak > $O/issue_peak.log 2>&1
rm -f $O/*.err.tmp; ls $O | wc -l
