#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "single_env" 2>&1 | tail -5
