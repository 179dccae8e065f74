#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python tools/sweep.py 2>&1 | grep -v amdgpu.ids > gpurun_out/sweep.log; wc -l gpurun_out/sweep.log
timeout 900 python tools/bench_configs.py > gpurun_out/configs.log 2>&1; tail -10 gpurun_out/configs.log | cut -c1-260
