#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 900 python tools/bench_configs.py > gpurun_out/configs.log 2>&1; grep '"config": 4' gpurun_out/configs.log | cut -c1-250
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -2
