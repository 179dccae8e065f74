#!/bin/bash
# round 2: after removing the sub-dword stores: GPU suite, per-kernel tabular-Q times, configs
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r02j; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -1 $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -8 $O/pytest_gpu.log | grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl" | cut -c1-300
python tools/prof_tabq_kernels.py 262144 2>&1 | grep -v amdgpu.ids | tee $O/tabq_kernels_262144.log
python tools/prof_tabq_kernels.py 65536 2>&1 | grep -v amdgpu.ids | tee $O/tabq_kernels_65536.log
timeout 900 python tools/bench_configs.py > $O/configs.log 2>&1; cat $O/configs.log | cut -c1-330
