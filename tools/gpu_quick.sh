#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ppo.py tests/test_gpu_parity.py -q -m gpu -k "ppo or gather" --durations=6 > gpurun_out/pytest_ppo.log 2>&1
echo "rc=$?"; tail -12 gpurun_out/pytest_ppo.log
