#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for f in 0 1; do
  echo "SGK_FUSED_ADAM=$f"
  SGK_FUSED_ADAM=$f timeout 900 python -m pytest tests/test_gpu_deepq.py tests/test_gpu_ppo.py -q -m gpu -x 2>&1 | tail -2
  SGK_FUSED_ADAM=$f timeout 600 python tools/bench_ppo.py 2>&1 | grep "n=32768 body=mlp fused=True" | head -1
  SGK_FUSED_ADAM=$f timeout 600 python tools/bench_configs.py 2>&1 | grep "with learning" | cut -c1-200
done
