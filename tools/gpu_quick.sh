#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "sharded" -v 2>&1 | tail -6
