#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export SGK_BENCH_SIZES=49152,65536,81920,98304,131072,196608
timeout 600 python tools/bench_tabq_sizes.py DistributionalShift-v0 2>&1 | sed 's/^lds/auto/' | tee -a gpurun_out/bench_tabq_auto.log | grep "agent-steps"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "tabq" --durations=3 2>&1 | tail -8
