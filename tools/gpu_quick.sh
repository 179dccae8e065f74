#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ppo.py tests/test_gpu_deepq.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -3
export TMPDIR=/tmp
rm -rf gpurun_out/prof_ppo gpurun_out/prof_dq
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ppo -o ppo --output-format csv -- python3 tools/prof_ppo_learn.py > gpurun_out/prof_ppo.log 2>&1
f=$(find gpurun_out/prof_ppo -name "*kernel_stats.csv" | head -1); grep "ppo_epochs" "$f" | cut -c1-140
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_dq -o dq --output-format csv -- python3 tools/prof_deepq_learn.py > gpurun_out/prof_dq.log 2>&1
f=$(find gpurun_out/prof_dq -name "*kernel_stats.csv" | head -1); grep "dqn_sgd" "$f" | cut -c1-140
find gpurun_out/prof_ppo gpurun_out/prof_dq -name "*kernel_trace.csv" -delete
