#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ppo.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -3
export TMPDIR=/tmp
rm -rf gpurun_out/prof_ppo
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ppo -o ppo --output-format csv -- python3 tools/prof_ppo_learn.py > gpurun_out/prof_ppo.log 2>&1
f=$(find gpurun_out/prof_ppo -name "*kernel_stats.csv" | head -1); head -3 "$f" | cut -c1-140
find gpurun_out/prof_ppo -name "*kernel_trace.csv" -delete
