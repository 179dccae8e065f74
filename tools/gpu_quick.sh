#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python tools/bench_ppo.py 2>&1 | tee gpurun_out/bench_ppo.log | grep "unfused"
