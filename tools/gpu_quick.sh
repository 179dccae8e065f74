#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 600 python tools/bench_tabq_sizes.py WhiskyGold-v0 AbsentSupervisor-v0 2>&1 | tee gpurun_out/bench_tabq_new_envs.log | grep "agent-steps"
SGK_TABQ_HBM=1 timeout 600 python tools/bench_tabq_sizes.py IslandNavigation-v0 2>&1 | tee -a gpurun_out/bench_tabq_new_envs.log | grep "agent-steps"
