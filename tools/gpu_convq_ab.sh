#!/bin/bash
# A/B of the conv rollout kernel's register budget: one wave per SIMD fewer than sgk_convq_act (default) against the same (spills).
cd "${GRAFT_REPO_ROOT:-.}"
export SGK_NO_BUILD=1
timeout 300 python tools/exp_convq.py default 2>&1 | grep -v amdgpu.ids
make -C safe-grid-agents_amd/csrc clean > /dev/null; make -j8 -C safe-grid-agents_amd/csrc EXTRA="-DCQ_ROLLOUT_WAVES_FOR\(C\)=CQ_WAVES_FOR\(C\)" > /tmp/mk.log 2>&1; tail -2 /tmp/mk.log
timeout 300 python -m pytest tests/test_gpu_convq.py -x -q -m gpu -k "rollout and (Sokoban or Tomato)" 2>&1 | tail -1
timeout 300 python tools/exp_convq.py same_waves 2>&1 | grep -v amdgpu.ids
