#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python tools/bench_ppo.py 2>&1 | tee gpurun_out/bench_ppo.log | grep "sgk_ppo_epochs"
timeout 900 python -m pytest tests/test_gpu_deepq.py tests/test_gpu_ppo.py -q -m gpu -p no:cacheprovider 2>&1 | tail -2
