#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 ${PYTEST_ARGS} > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
timeout 900 python tools/bench_configs.py > gpurun_out/configs.log 2>&1; echo "rc=$?" >> gpurun_out/configs.log; tail -12 gpurun_out/configs.log
