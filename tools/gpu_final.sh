#!/bin/bash
# Everything the docs quote, in one GPU call: bash tools/gpu_final.sh <round tag, e.g. r06> [pmc]. Logs land in
# gpurun_out/<tag>_final/ (copy what is to be judged into profiles/<tag>/). With `pmc` also the FETCH_SIZE / WRITE_SIZE passes of the
# three bench kernels that profiles/traffic.json is made from (24 profiler runs: ~15 minutes).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
T=${1:-r06}; O=gpurun_out/${T}_final; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -1 $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -8 $O/pytest_gpu.log | grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl" | cut -c1-300
# the driver's bench command, three processes
for i in 1 2 3; do SGK_BENCH_TRACE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags_$i.log 2> $O/bench_driver_flags_$i.err; tail -1 $O/bench_driver_flags_$i.log | cut -c1-200; done
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --path own --no-cpu-baseline > $O/bench_path_own.log 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --path launch --no-cpu-baseline > $O/bench_path_launch.log 2>&1
# the multi-rank control flow end to end on the one GPU of the box: ranks share GPU 0, gloo collectives (RCCL needs one GPU per rank)
SGK_BENCH_BACKEND=gloo SGK_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --no-fused > $O/bench_2rank_one_gpu_gloo.log 2>&1; tail -1 $O/bench_2rank_one_gpu_gloo.log | cut -c1-300
SGK_BENCH_BACKEND=gloo SGK_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus 8 --steps 5 --warmup 2 --no-fused --no-weak-line --no-secondary > $O/bench_8rank_one_gpu_gloo.log 2>&1; tail -1 $O/bench_8rank_one_gpu_gloo.log | cut -c1-300
# BASELINE configs 2-4 (roofline + cpu_baseline each), their counters; config 3's kernel under the SQ counters
bash tools/gpu_configs.sh ${T}_final > $O/gpu_configs.out 2>&1; tail -12 $O/gpu_configs.out | cut -c1-300
bash tools/gpu_pmc_tabq.sh ${T}_final final > $O/gpu_pmc_tabq.out 2>&1; tail -4 $O/gpu_pmc_tabq.out | cut -c1-300
timeout 900 python tools/bench_configs.py > $O/configs_1_to_5.log 2>&1
timeout 900 python tools/bench_stream.py --envs BoatRace-v0 --sizes 1024,4096,16384,65536,131072,262144,524288,1048576 --ring 100 > $O/batch_sweep_1k_to_1m.log 2>&1
timeout 900 python tools/bench_stream.py --envs BoatRace-v0,IslandNavigation-v0,SideEffectsSokoban-v0,DistributionalShift-v0,WhiskyGold-v0,AbsentSupervisor-v0,SafeInterruptibility-v0,ConveyorBelt-v0,TomatoWatering-v0,FriendFoe-v0 --ring 100 > $O/stream_all_envs.log 2>&1
timeout 600 python tools/bench_single_env.py > $O/single_env.log 2>&1
SGK_STEP_SERVER=0 timeout 600 python tools/bench_single_env.py > $O/single_env_launch_per_step.log 2>&1
timeout 600 python tools/bench_policy_rollout.py > $O/policy_rollout_env_counts.log 2>&1
# round 6: what a caller of the per-step tabular-Q API pays (one launch per step / four launches / from Python, by stream mode), the DeepQ
# learner's device time and its barrier-to-barrier timeline, and the step-facing tests without asserts (python -O)
timeout 600 python tools/exp_tabq_dropin.py 2>&1 | grep -v amdgpu.ids > $O/tabq_dropin.log; tail -6 $O/tabq_dropin.log | cut -c1-250
timeout 300 python tools/exp_dqn_learner_ab.py 2>&1 | grep -v amdgpu.ids > $O/dqn_learner.log; cat $O/dqn_learner.log | cut -c1-200
timeout 600 bash tools/gpu_dqn_timeline.sh $O/dqn_timeline.log > /dev/null 2>&1; tail -4 $O/dqn_timeline.log | cut -c1-200
timeout 900 python -O -m pytest tests -m gpu -q -k step > $O/pytest_gpu_O_step.log 2>&1; tail -1 $O/pytest_gpu_O_step.log
# round 6: the conv body's kernel (sgk_convq_sample / sgk_convq_act) per level and channel count, ppo-cnn's gather through it and through
# the torch module, and the kernel under the kernel trace + SQ counters (issue / MFMA busy / LDS)
timeout 600 python tools/exp_convq.py 2>&1 | grep -v amdgpu.ids > $O/convq_levels.log; cat $O/convq_levels.log | cut -c1-250
timeout 600 python tools/bench_ppo.py cnn 2>&1 | grep -v amdgpu.ids > $O/ppo_cnn_gather.log; cat $O/ppo_cnn_gather.log | cut -c1-250
timeout 900 bash tools/gpu_pmc_convq.sh > $O/convq_pmc.out 2>&1; cp gpurun_out/pmc_convq_summary.json $O/convq_pmc_summary.json 2>/dev/null
for f in $(find gpurun_out/convq_trace -name "*kernel_stats.csv" 2>/dev/null); do cp $f $O/convq_kernel_stats.csv; head -3 $f | cut -c1-200; done
export SGK_NO_BUILD=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-fused --sustain-seconds 0 > $O/bench_under_rocprof.log 2>&1
for f in $(find $O/prof -name "*kernel_stats.csv"); do head -6 $f | cut -c1-200; cp $f $O/bench_kernel_stats.csv; done
rm -rf $O/prof
# config 2's kernel (one step_kernel launch per lockstep step) and config 4's learner (dqn_sgd_kernel) under the kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof2 -- python3 bench.py --config 2 --no-cpu-baseline > $O/config2_under_rocprof.log 2>&1
for f in $(find $O/prof2 -name "*kernel_stats.csv"); do head -4 $f | cut -c1-200; cp $f $O/config2_kernel_stats.csv; done
rm -rf $O/prof2
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof4 -- python3 tools/prof_deepq_learn.py > $O/dqn_learn_under_rocprof.log 2>&1
for f in $(find $O/prof4 -name "*kernel_stats.csv"); do head -12 $f | cut -c1-200; cp $f $O/dqn_learn_kernel_stats.csv; done
rm -rf $O/prof4
if [ "$2" = "pmc" ]; then
  for n in 1048576 524288 262144 131072; do for mode in ring stream launch; do for ctr in FETCH_SIZE WRITE_SIZE; do
    tag=$([ $n = 1048576 ] && echo "" || echo "_n$n")
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_tmp -- python3 tools/pmc_run.py BoatRace-v0 compact $n $mode > $O/pmc_${mode}${tag}_$ctr.log 2>&1
    python tools/pmc_summary.py $O/pmc_tmp > $O/pmc_${mode}${tag}_${ctr}_summary.json
    rm -rf $O/pmc_tmp
  done; done; done
  python tools/make_traffic_json.py $O 1048576 524288 262144 131072
fi
ls $O | wc -l
