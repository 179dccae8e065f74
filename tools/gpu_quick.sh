#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 1200 python -m pytest tests/test_gpu_deepq.py tests/test_gpu_ppo.py tests/test_gpu_parity.py -q -m gpu -x -k "fused or eval or ppo or deepq" 2>&1 | tail -5
timeout 900 python tools/bench_configs.py 2>&1 | grep '"config": 4' | cut -c1-250
