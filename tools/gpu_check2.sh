#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
tail -2 gpurun_out/smoke.log
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 ${PYTEST_ARGS} > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
python tools/sweep.py > gpurun_out/sweep.log 2>&1; tail -30 gpurun_out/sweep.log
