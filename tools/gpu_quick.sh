#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 1200 python -m pytest tests/test_gpu_deepq.py tests/test_gpu_ppo.py -q -m gpu -x 2>&1 | tail -4
