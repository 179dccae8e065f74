#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 600 python -m pytest tests/test_gpu_deepq.py -q -m gpu -x 2>&1 | tail -2
timeout 900 python tools/bench_configs.py 2>&1 | grep '"config": 4' | grep learning | cut -c1-250
