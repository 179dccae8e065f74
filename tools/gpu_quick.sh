#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 900 python tools/bench_tabq_sizes.py 2>&1 | grep -v amdgpu.ids > /tmp/lds.log
SGK_TABQ_HBM=1 timeout 900 python tools/bench_tabq_sizes.py 2>&1 | grep -v amdgpu.ids > /tmp/hbm.log
paste -d'\n' /tmp/lds.log /tmp/hbm.log
