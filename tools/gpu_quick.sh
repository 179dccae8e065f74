#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ppo.py tests/test_gpu_parity.py -q -m gpu -k "ppo or gather" -x > gpurun_out/pytest_ppo.log 2>&1
echo "rc=$?"; tail -25 gpurun_out/pytest_ppo.log
timeout 900 python tools/bench_ppo.py 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_ppo.log
cat gpurun_out/bench_ppo.log | tail -12
