#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 1200 python -m pytest tests/test_gpu_deepq.py -q -m gpu -x -k "cli" 2>&1 | tail -12
