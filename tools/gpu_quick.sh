#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 600 python tools/exp_flags.py IslandNavigation-v0 2>&1 | tee gpurun_out/exp_flags.log | grep "us/step"
timeout 600 python tools/exp_flags.py SideEffectsSokoban-v0 2>&1 | tee -a gpurun_out/exp_flags.log | grep "us/step"
