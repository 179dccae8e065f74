#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "single_env" 2>&1 | tail -5
timeout 900 python tools/sweep.py AbsentSupervisor-v0 2>&1 | tee gpurun_out/sweep_super.log | tail -12
timeout 300 python bench.py --env AbsentSupervisor-v0 --steps 500 --warmup 100 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['parity_sample_bit_exact'], d['parity_sample_envs'])"
