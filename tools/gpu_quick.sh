#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for v in "" 1; do
  echo "SGK_TABQ_HBM=$v"
  if [ -n "$v" ]; then export SGK_TABQ_HBM=1; fi
  timeout 900 python tools/bench_configs.py 2>&1 | grep '"config": 3' | head -1 | cut -c1-230
  timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "tabq_fused" 2>&1 | tail -1
done
