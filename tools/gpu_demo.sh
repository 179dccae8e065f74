#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
python tools/demo_train.py 65536 2>&1 | grep -v amdgpu.ids | tee gpurun_out/demo_train.log
