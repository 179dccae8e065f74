#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
timeout 900 python tools/sweep.py 1024 65536 1048576 2>&1 | grep compact | cut -c1-130
