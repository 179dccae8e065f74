#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python tools/sweep.py WhiskyGold-v0 2>&1 | tee gpurun_out/sweep_whisky.log | tail -12
timeout 300 python bench.py --env WhiskyGold-v0 --steps 500 --warmup 100 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-900
