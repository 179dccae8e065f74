#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
( time python bench.py ) > gpurun_out/bench_default.log 2>&1; tail -5 gpurun_out/bench_default.log
