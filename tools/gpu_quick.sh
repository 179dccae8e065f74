#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ppo.py -q -m gpu > gpurun_out/pytest_ppo.log 2>&1
echo "rc=$?"; tail -40 gpurun_out/pytest_ppo.log
